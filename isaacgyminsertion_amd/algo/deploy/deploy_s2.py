"""Stage-2 (visuotactile student) deployment player with the reference's names (algo/deploy/deploy_s2.py:54-238,
883-1050): ``HardwarePlayer(full_config)``, ``restore(fn)``, ``restore_student(fn, from_offline, phase)``,
``set_eval()``, ``set_student_eval()``, ``process_obs(obs)``, ``deploy()``.

``deploy.ppo.{obs,tactile,img,seg,pcl}_info`` select the student's modalities and ``only_bc`` is forced on
(deploy_s2.py:139-152): the student emits the action and the stage-1 agent is never evaluated (its load is
commented out in the reference, deploy_s2.py:169-175), so none is built.  Checkpoints: ``stage2_nn/*_stud.pth``
(keys student / stud_obs_mean_std / pcl_mean_std) or the offline pair ``model_last.pt`` + ``normalization.pkl``.
One control tick = process_obs (mask, eval-mode normalisers) -> ``Student.predict`` -> clamp, all HIP kernels."""
import copy
import pickle

import torch

from ..models.running_mean_std import RunningMeanStd
from ..models.transformer.runner import Runner as Student


class HardwarePlayer:
    def __init__(self, full_config, robot=None):
        self.num_envs = 1
        full_config = copy.deepcopy(full_config)
        self.full_config = full_config
        self.deploy_config = full_config.deploy
        self.train_config = full_config.offline_train
        self.device = full_config["rl_device"]
        env = full_config.task.env
        self.num_observations = env.numObservations
        self.num_obs_stud = self.train_config.model.linear.input_size
        self.num_actions = env.numActions
        self.priv_info = False
        flags = self.deploy_config.ppo
        self.obs_info, self.tactile_info = flags.obs_info, flags.tactile_info
        self.img_info, self.seg_info, self.pcl_info = flags.img_info, flags.seg_info, flags.pcl_info
        m = full_config.offline_train.model
        full_config.offline_train.only_bc = True
        m.use_tactile, m.use_seg, m.use_lin = self.tactile_info, self.seg_info, self.obs_info
        m.use_img, m.use_pcl = self.img_info, self.pcl_info
        self.stud_obs_mean_std = RunningMeanStd((self.num_obs_stud,)).to(self.device)
        self.pcl_mean_std = RunningMeanStd((3,)).to(self.device)
        self.student = Student(full_config)
        self.stats = None
        self.env = robot
        self.episode_length = torch.zeros((1, 1), device=self.device, dtype=torch.float)
        self.set_student_eval()

    # -- checkpoints ------------------------------------------------------------------------------------
    def restore(self, fn):
        """deploy_s2.py:167-183: ``fn`` names the stage-1 checkpoint; the student sits next to it."""
        stud_fn = fn.replace('stage1_nn/last.pth', 'stage2_nn/last_stud.pth')
        self.restore_student(stud_fn, from_offline=False, phase=1)
        self.set_eval()
        self.set_student_eval()

    def restore_student(self, fn, from_offline=False, phase=1):
        """deploy_s2.py:185-217"""
        if from_offline:
            if phase == 2:
                self.student.model.load_state_dict(torch.load(fn, map_location=self.device)['student'])
            else:
                self.student.model.load_state_dict(
                    torch.load(self.train_config.train.student_ckpt_path, map_location=self.device))
            with open(self.train_config.train.normalize_file, "rb") as f:
                stats = pickle.load(f)
            self.stats = {kind: {k: torch.as_tensor(v, dtype=torch.float32, device=self.device)
                                 for k, v in stats[kind].items()} for kind in ('mean', 'std')}
            self.train_config.from_offline = True
            return
        checkpoint = torch.load(fn, map_location=self.device)
        self.stud_obs_mean_std.load_state_dict(checkpoint['stud_obs_mean_std'])
        self.pcl_mean_std.load_state_dict(checkpoint['pcl_mean_std'])
        self.student.model.load_state_dict(checkpoint['student'])
        self.stats = None
        self.train_config.from_offline = False

    def set_eval(self):
        pass                                       # no stage-1 agent in this player (see the module docstring)

    def set_student_eval(self):
        """deploy_s2.py:229-238"""
        self.student.model.eval()
        self.stud_obs_mean_std.eval()
        self.pcl_mean_std.eval()

    # -- one control tick ---------------------------------------------------------------------------------
    def process_obs(self, obs, obj_id=2, socket_id=3, distinct=True, display=False):
        """deploy_s2.py:883-928.  The reference's offline-statistics branch is unreachable there
        (``assert NotImplementedError`` is a no-op and what follows it runs); here it is the live path for
        ``restore_student(from_offline=True)`` and matches ExtrinsicAdapt.process_obs (ext_adapt.py:411-417)."""
        # one entry per modality the student consumes; a modality that is switched off is handed on as None
        wanted = (('student_obs', self.obs_info), ('tactile', self.tactile_info), ('img', self.img_info),
                  ('seg', self.seg_info), ('pcl', self.pcl_info))
        out = {name: (obs[name] if on else None) for name, on in wanted}
        if out['seg'] is not None:
            # keep the plug and the socket: their ids (distinct) or one foreground plane; the depth image sees the same mask
            keep = torch.logical_or(out['seg'] == obj_id, out['seg'] == socket_id).to(torch.float32)
            out['seg'] = out['seg'] * keep if distinct else keep
            if out['img'] is not None:
                out['img'] = out['img'] * keep
        if out['pcl'] is not None:
            clouds = out['pcl'].shape[0]
            out['pcl'] = self.pcl_mean_std(out['pcl'].reshape(-1, 3)).reshape(clouds, -1, 3)
        so = out['student_obs']
        if so is not None:
            offline = bool(self.train_config.from_offline)
            if offline and self.stats is None:
                raise RuntimeError("from_offline=True needs restore_student(..., from_offline=True) first")
            if offline:
                m, sd = self.stats['mean'], self.stats['std']
                out['student_obs'] = torch.cat([(so[:, :9] - m['eef_pos_rot6d']) / sd['eef_pos_rot6d'],
                                                (so[:, 9:12] - m['socket_pos'][:3]) / sd['socket_pos'][:3],
                                                so[:, 12:]], dim=-1)
            else:
                out['student_obs'] = self.stud_obs_mean_std(so)
        return out

    @torch.no_grad()
    def policy_step(self, obs_dict):
        """deploy_s2.py:975-990 with only_bc: (action clamped to [-1, 1], raw student output)."""
        obs_dict = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in obs_dict.items()}
        latent, _ = self.student.predict(self.process_obs(obs_dict), requires_grad=False)
        return torch.clamp(latent, -1.0, 1.0), latent

    def deploy(self, num_episodes=None):
        """deploy_s2.py:930-1050 with the robot behind RobotIO; returns the number of control ticks."""
        if self.env is None:
            raise RuntimeError("HardwarePlayer.deploy needs a RobotIO (robot=...): the ROS / MoveIt side of the "
                               "reference's player is not part of this package")
        if num_episodes is None:
            num_episodes = self.deploy_config.data_logger.total_trajectories
        ticks, cur_episode = 0, 0
        while cur_episode < num_episodes:
            self.episode_length.zero_()
            while True:
                action, _ = self.policy_step(self.env.observe())
                self.env.apply(action)
                self.episode_length += 1
                ticks += 1
                if self.env.done():
                    cur_episode += 1
                    break
            self.env.reset()
        return ticks
