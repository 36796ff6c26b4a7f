"""Online training / evaluation entry point with the reference's control flow (isaacgyminsertion/train.py:45-141):

    python -m isaacgyminsertion_amd.train [--config my.yaml] [--env pkg.module:make_env] \\
        train.algo=PPO task.env.numEnvs=4096 train.ppo.horizon_length=32
    python -m isaacgyminsertion_amd.train train.algo=ExtrinsicAdapt restore_train=True \\
        train.load_path=outputs/.../stage1_nn/last.pth train.ppo.tactile_info=True
    python -m isaacgyminsertion_amd.train test=True train.load_path=.../stage1_nn/last.pth
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m isaacgyminsertion_amd.train train.ppo.multi_gpu=True

rank / device / seed selection, ``offline_training`` -> ``Runner.run()``, environment construction, the
``outputs/<date>/<time>`` run directory with the resolved config written into it, ``agent = PPO | ExtrinsicAdapt``,
``restore_test + test`` or ``restore_train + train``.  The reference builds an IsaacGym task here (closed-source,
CUDA-only); ``--env`` names a factory ``make_env(cfg) -> env`` honouring the VecTask contract, and without it
the seeded ``SyntheticInsertionEnv`` with the modalities the config switches on stands in.  Hydra is replaced by a
plain YAML file merged over the built-in defaults plus ``a.b.c=value`` overrides."""
import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # read when HSA initialises: before any GPU call (dmabuf IPC only)
import argparse
import importlib
import os
from datetime import datetime

import yaml

from .utils.config import default_config, load_config, merge


def _override(cfg, dotted, value):
    node = cur = {}
    keys = dotted.split('.')
    for k in keys[:-1]:
        cur[k] = {}
        cur = cur[k]
    cur[keys[-1]] = yaml.safe_load(value)
    return merge(cfg, node)


def _plain(d):
    if isinstance(d, dict):
        return {k: _plain(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [_plain(v) for v in d]
    return d


def build_config(config=None, overrides=()):
    cfg = load_config(config) if config else default_config()
    for o in overrides:
        k, _, v = o.partition('=')
        cfg = _override(cfg, k, v)
    # interpolations the reference resolves through Hydra (cfg/train/...PPOv2.yaml:17, cfg/config.yaml)
    cfg.train.ppo.num_actors = cfg.task.env.numEnvs
    cfg.train.ppo.multi_gpu = bool(cfg.train.ppo.multi_gpu or cfg.get('multi_gpu', False))
    return cfg


def make_synthetic_env(cfg):
    """The stand-in for ``isaacgym_task_map[cfg.task_name](...)`` (train.py:98-106)."""
    from .envs.synthetic import SyntheticInsertionEnv
    env, ppo, off = cfg.task.env, cfg.train.ppo, cfg.offline_train
    pts = (env.num_points + env.num_points_socket) if ppo.pcl_info else 0
    return SyntheticInsertionEnv(
        num_envs=env.numEnvs, obs_dim=env.numObservations * env.numObsHist, priv_dim=ppo.priv_info_dim,
        act_dim=env.numActions, device=cfg.rl_device, seed=1234 + int(cfg.seed),
        max_episode_length=cfg.task.rl.max_episode_length,
        tactile_hw=(off.tactile_width, off.tactile_height) if ppo.tactile_info else None, pcl_points=pts,
        img_hw=(off.img_width, off.img_height) if (ppo.img_info or ppo.seg_info) else None)


def run(cfg, env_factory=None):
    from .algo.models.transformer.utils import set_seed
    if cfg.checkpoint:
        cfg.checkpoint = os.path.abspath(cfg.checkpoint)
    if cfg.train.ppo.multi_gpu:                       # train.py:58-64: one process per GPU, seed offset by rank
        import torch
        rank = int(os.getenv("LOCAL_RANK", "0"))
        index = rank if os.environ.get("IGI_DIST_BACKEND", "nccl") == "nccl" \
            else rank % max(torch.cuda.device_count(), 1)          # single-GPU test mode, see utils/dist.py
        cfg.sim_device = cfg.rl_device = f"cuda:{index}"
        cfg.seed = cfg.seed + rank
    else:
        rank = -1
    set_seed(cfg.seed)
    if cfg.offline_training:                          # train.py:86-93
        from .algo.models.transformer.runner import Runner
        runner = Runner(cfg, agent=None)
        runner.run()
        return runner
    envs = (env_factory or make_synthetic_env)(cfg)
    now = datetime.now()
    output_dif = os.path.join(cfg.get('output_root', 'outputs'), now.strftime("%m-%d-%y"), now.strftime("%H-%M-%S"))
    os.makedirs(output_dif, exist_ok=True)
    from .algo.ppo.frozen_ppo import PPO
    from .algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    algos = {'PPO': PPO, 'ExtrinsicAdapt': ExtrinsicAdapt}
    if cfg.train.algo not in algos:
        raise ValueError(f"train.algo must be one of {sorted(algos)}, got {cfg.train.algo!r}")
    agent = algos[cfg.train.algo](envs, output_dif, full_config=cfg)
    if cfg.test:                                      # train.py:113-128
        assert cfg.train.load_path, "test=True needs train.load_path"
        agent.restore_test(cfg.train.load_path)
        agent.set_eval()
        if not cfg.offline_training_w_env:
            num_success, total_trials = agent.test()
            print(f"Success rate: {num_success / max(total_trials, 1)}")
            agent.last_test = (num_success, total_trials)
        else:                                         # train.py:123-128: offline student, online teacher pass
            from .algo.models.transformer.runner import Runner
            runner = Runner(cfg, agent, action_regularization=cfg.offline_train.train.get('action_regularization',
                                                                                          False))
            runner.run()
            agent.offline_runner = runner
    else:
        if rank <= 0:
            with open(os.path.join(output_dif, f"config_{now.strftime('%m%d%H')}.yaml"), "w") as f:
                yaml.safe_dump(_plain(cfg), f)
        if cfg.restore_train:
            agent.restore_train(cfg.train.load_path, cfg.restore_student, cfg.phase)
        agent.train()
    return agent


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', default=None, help='YAML file with the reference config keys')
    ap.add_argument('--env', default=None, help='pkg.module:factory returning an env for the resolved config')
    ap.add_argument('overrides', nargs='*', help='a.b.c=value')
    args = ap.parse_args(argv)
    factory = None
    if args.env:
        mod, _, fn = args.env.partition(':')
        factory = getattr(importlib.import_module(mod), fn)
    return run(build_config(args.config, args.overrides), factory)


if __name__ == "__main__":
    main()
