"""Helpers shared by the parity tests: load the golden fixtures captured from the reference."""
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ROLLOUT_KEYS = ["obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus",
                "sigmas", "last_values"]


def load_teacher(case):
    z = np.load(os.path.join(GOLDEN, f"teacher_{case}.npz"))
    g = {k: z[k] for k in z.files}
    num_envs, horizon, mini_epochs, n_updates = [int(x) for x in g["meta"]]
    meta = dict(num_envs=num_envs, horizon=horizon, mini_epochs=mini_epochs, n_updates=n_updates,
                units=[int(x) for x in g["units"]], priv_units=[int(x) for x in g["priv_units"]])
    init = OrderedDict((k[len("init/"):], torch.from_numpy(v)) for k, v in g.items()
                       if k.startswith("init/"))
    return g, meta, init


def rollout(g, u):
    return {k: torch.from_numpy(g[f"u{u}/in/{k}"]) for k in ROLLOUT_KEYS}


def load_teacher_dp():
    z = np.load(os.path.join(GOLDEN, "teacher_dp2.npz"))
    g = {k: z[k] for k in z.files}
    num_envs, horizon, mini_epochs, _ = [int(x) for x in g["meta"]]
    meta = dict(num_envs=num_envs, horizon=horizon, mini_epochs=mini_epochs,
                units=[int(x) for x in g["units"]], priv_units=[int(x) for x in g["priv_units"]])
    init = OrderedDict((k[len("init/"):], torch.from_numpy(v)) for k, v in g.items() if k.startswith("init/"))
    return g, meta, init


def rollout_dp(g, rank):
    return {k: torch.from_numpy(g[f"r{rank}/in/{k}"]) for k in ROLLOUT_KEYS}


def load_golden(name):
    """np.load of a fixture under tests/golden/."""
    return np.load(os.path.join(GOLDEN, name))
