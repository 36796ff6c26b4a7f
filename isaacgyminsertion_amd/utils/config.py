"""Plain-YAML / dict configuration with the access semantics the reference trainers rely on
(OmegaConf is not a dependency): ``cfg.a.b``, ``cfg['a']``, ``cfg.get('a', default)``.

``default_config`` returns the RESOLVED hot-path values of the reference's Hydra tree
(isaacgyminsertion/cfg/config.yaml, cfg/task/FactoryTaskInsertionTactile.yaml:40-126,
cfg/train/FactoryTaskInsertionTactilePPOv2.yaml:1-70); interpolations such as
``num_actors: ${...task.env.numEnvs}`` are resolved here by construction.
"""
import copy

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return to_attr({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return [to_attr(v) for v in d]
    return d


def merge(base, override):
    out = copy.deepcopy(base)
    for k, v in (override or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = v
    return to_attr(out)


def default_config(num_envs=4096, horizon_length=32, rl_device="cuda:0", multi_gpu=False, **ppo_overrides):
    ppo = {
        "output_name": "debug", "multi_gpu": multi_gpu, "normalize_input": True, "normalize_value": True,
        "value_bootstrap": True, "shared_parameters": False, "num_actors": num_envs,
        "normalize_advantage": True, "gamma": 0.99, "tau": 0.95, "learning_rate": 2.5e-4,
        "kl_threshold": 0.02, "horizon_length": horizon_length, "mini_epochs": 8, "minibatch_size": 24,
        "clip_value": True, "critic_coef": 4, "entropy_coef": 0.0, "e_clip": 0.2, "bounds_loss_coef": 1e-4,
        "truncate_grads": True, "grad_norm": 1, "save_best_after": 1000000, "save_frequency": 100,
        "max_agent_steps": 1500000000, "priv_info": True, "priv_info_dim": 64, "compute_contact_gt": False,
        "only_contact": False, "num_points": 400,
        # student modality switches (cfg/train/...PPOv2.yaml:54-59; scripts/train_s2.sh set them)
        "obs_info": True, "tactile_info": False, "img_info": False, "seg_info": False, "pcl_info": False,
    }
    ppo.update(ppo_overrides)
    cfg = {
        "seed": 42, "rl_device": rl_device, "sim_device": rl_device, "multi_gpu": multi_gpu, "test": False,
        "offline_training": False, "offline_training_w_env": False, "checkpoint": "",
        "restore_train": False, "restore_student": False, "phase": 2, "task_name": "FactoryTaskInsertionTactile",
        "headless": True,
        "task": {
            "name": "FactoryTaskInsertionTactile",
            "env": {"numEnvs": num_envs, "numObservations": 15, "numObsHist": 1, "numStates": 64,
                    "numActions": 6, "numObsStudent": 15, "numObsStudentHist": 1, "compute_contact_gt": False,
                    "record_video_every": 10 ** 9, "num_points": 400, "num_points_socket": 400,
                    "num_points_goal": 400, "include_plug_pcl": True, "merge_socket_pcl": True,
                    "merge_goal_pcl": False, "include_all_pcl": False,
                    "tactile_history_len": 1, "img_history_len": 1},
            "rl": {"max_episode_length": 512},
            "data_logger": {"collect_data": False},
            "tactile": {"encoder": {"width": 64, "height": 64, "num_channels": 1}, "crop_roi": True},
        },
        # cfg/offline_train/offline_config.yaml:5-103 (hot-path values)
        "offline_train": {
            "only_bc": True, "from_offline": False, "multi_gpu": False, "gpu_ids": [0],
            "tactile_type": "gray", "tactile_width": 32, "tactile_height": 64, "tactile_crop_w": 0,
            "tactile_crop_h": 0,
            "model": {"model_type": "tact", "use_tactile": False, "use_img": False, "use_seg": False,
                      "use_lin": True, "use_pcl": False, "linear": {"input_size": 15},
                      "transformer": {"sequence_length": 1, "num_layers": 2, "num_heads": 2, "dim_factor": 4,
                                      "output_size": 8, "lin_encoding_size": 32, "tactile_encoding_size": 32,
                                      "img_encoding_size": 32, "seg_encoding_size": 32, "load_tact": False}},
            "img_type": "depth", "img_width": 54, "img_height": 96, "img_crop_w": 0, "img_crop_h": 0,
            "tactile_patch_size": 16, "tactile_gaussian_noise": 0.001, "tactile_masking_prob": 0.0,
            "tactile_color_jitter": False, "seed": 0, "data_folder": "", "output_dir": "outputs/offline",
            # supervised learning (offline_config.yaml:28-83)
            "train": {"latent_scale": 1.0, "action_scale": 1.0, "action_regularization": False, "epochs": 100, "train_batch_size": 64,
                      "val_batch_size": 64, "learning_rate": 1e-4, "train_test_split": 0.98,
                      "scheduler": "cosine", "warmup": False, "warmup_epochs": 4,
                      "print_every": 1000, "eval_every": 1000, "test_every": 2000,
                      "only_test": False, "only_validate": False,
                      "obs_keys": ["eef_pos", "action", "latent", "obs_hist", "noisy_socket_pos", "socket_pos",
                                   "hand_joints", "plug_hand_quat", "plug_hand_pos", "plug_pos_error",
                                   "plug_quat_error"],
                      "normalize_obs_keys": ["eef_pos", "noisy_socket_pos", "action", "plug_hand_quat",
                                             "plug_hand_pos", "socket_pos"],
                      "load_stats": False, "normalize_file": "", "load_checkpoint": False,
                      "student_ckpt_path": ""},
            "wandb": {"wandb_enabled": False, "wandb_project_name": "tactile_insertion"},
        },
        # cfg/deploy/FactoryTaskInsertionTactileDeploy.yaml:17-92 (what the deployment players read)
        "deploy": {
            "data_logger": {"collect_data": False, "total_trajectories": 5},
            "rl": {"max_episode_length": 500, "pos_action_scale": [0.01, 0.01, 0.005],
                   "rot_action_scale": [0.02, 0.02, 0.1]},
            "ppo": {"priv_info": True, "extrin_adapt": False, "tactile_info": True, "pcl_info": True,
                    "obs_info": True, "seg_info": False, "img_info": False, "ft_info": False,
                    "student_obs_input_shape": 15},
        },
        "train": {
            "algo": "PPO", "load_path": "",
            "network": {"mlp": {"units": [512, 256, 128]}, "priv_mlp": {"units": [256, 128, 8]},
                        "contact_mlp": {"units": [128, 64, 8]}},
            "ppo": ppo,
        },
    }
    return to_attr(cfg)


def load_config(path, **overrides):
    with open(path) as f:
        user = yaml.safe_load(f) or {}
    return merge(merge(default_config(), user), overrides)
