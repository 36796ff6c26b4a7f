"""Import harness for the *reference* implementation (osheraz/IsaacGymInsertion).

TEST INFRASTRUCTURE ONLY.  This module is used by ``make_golden_*.py`` to import
the reference's Python modules from ``/root/reference`` on CPU, in the build
container, so that golden input/output vectors can be generated from the
reference's own arithmetic.  Nothing here (and nothing under /root/reference)
travels to the GPU box; the committed ``*.npz`` fixtures do.

The reference imports a number of third-party packages that the arithmetic on
the PPO / student path never touches (gym, cv2, tensorboardX, isaacgym, ...).
They are absent from this image, so we register empty stand-in modules in
``sys.modules`` *for import only* -- none of the stubbed symbols is executed on
the code paths the goldens exercise.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("IGI_REFERENCE_ROOT", "/root/reference")


class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


class _Anything(metaclass=_AnyMeta):
    """Attribute sink: any attribute access / call returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = sys.modules.get(f"{self.__name__}.{name}")
        if sub is not None:
            return sub
        return _Anything


_STUBS = [
    "gym", "gym.spaces", "cv2", "deepdish", "imageio", "termcolor", "tensorboardX",
    "isaacgym", "isaacgym.gymapi", "isaacgym.gymtorch", "isaacgym.torch_utils",
    "efficientnet_pytorch", "wandb", "warmup_scheduler", "hydra", "hydra.utils", "hydra.core",
    "hydra.core.hydra_config", "hydra.core.global_hydra",
    "omegaconf", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
    "torchvision.models", "pytorch3d", "pytorch3d.transforms", "open3d", "trimesh",
    "isaacgyminsertion.tasks", "isaacgyminsertion.tasks.factory_tactile",
    "isaacgyminsertion.tasks.factory_tactile.factory_utils",
]


def install():
    """Put /root/reference on sys.path and register the import-only stubs."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(
            f"reference tree not found at {REFERENCE_ROOT}: golden generation only runs "
            "in the build container")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        m = _StubModule(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
    # termcolor.cprint is called on restore paths: make it a plain print
    sys.modules["termcolor"].cprint = lambda *a, **k: None


class AttrDict(dict):
    """dict with attribute access (what the reference expects from OmegaConf)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, list):
        return [to_attr(v) for v in d]
    return d


def teacher_config(num_envs, horizon, mini_epochs, units=(512, 256, 128), priv_units=(256, 128, 8),
                   obs_dim=15, priv_dim=64, act_dim=6, multi_gpu=False):
    """Resolved hot-path config values as the attribute dict
    the reference trainer reads (cfg/train/FactoryTaskInsertionTactilePPOv2.yaml)."""
    return to_attr({
        "rl_device": "cpu",
        "offline_training": False,
        "offline_training_w_env": False,
        "test": False,
        "task": {"env": {"numActions": act_dim, "numObservations": obs_dim, "numObsHist": 1,
                         "record_video_every": 10 ** 9, "numStates": priv_dim,
                         "compute_contact_gt": False}},
        "train": {
            "network": {"mlp": {"units": list(units)}, "priv_mlp": {"units": list(priv_units)},
                        "contact_mlp": {"units": [128, 64, 8]}},
            "ppo": {
                "multi_gpu": multi_gpu, "num_actors": num_envs, "priv_info": True,
                "priv_info_dim": priv_dim, "compute_contact_gt": False, "only_contact": False,
                "num_points": 8, "shared_parameters": False, "learning_rate": 2.5e-4,
                "e_clip": 0.2, "clip_value": True, "entropy_coef": 0.0, "critic_coef": 4,
                "bounds_loss_coef": 1e-4, "gamma": 0.99, "tau": 0.95, "truncate_grads": True,
                "grad_norm": 1, "value_bootstrap": True, "normalize_advantage": True,
                "normalize_input": True, "normalize_value": True, "horizon_length": horizon,
                "mini_epochs": mini_epochs, "kl_threshold": 0.02, "save_frequency": 100,
                "save_best_after": 10 ** 9, "max_agent_steps": 10 ** 12,
            },
        },
    })
