"""Offline supervised path golden vectors from the REFERENCE implementation (build container only).

Three cases, all produced by the reference's own code on CPU:
  * ``cfg1``  -- BASELINE configs[0]: the reference ``Runner.train`` / ``Runner.validate``
    (runner.py:194-372) run unmodified over 256 proprio-only frames (stud_obs ~ N(0,1), action = tanh(N(0,1)),
    batch 64, AdamW(1e-4, wd 1e-6) created exactly as run_train does, runner.py:481), three passes in a fixed
    batch order; records per-step losses, validation losses and the final state_dict.
  * ``data``  -- the on-disk trajectory format: synthetic ``*.npz`` trajectories are written to a temp
    folder, then the reference ``DataNormalizer`` and ``TactileDataset`` (data.py:129-440) produce the
    normalisation statistics and every dataset item.  The trajectories themselves are stored in the fixture.
  * ``logger`` -- the reference ``DataLoggerSim.update`` (experience.py:422-462) fed a seeded sequence of steps;
    records every trajectory it hands to its writer.

pytorch3d is absent from the image (and unpinned by the reference): ``pytorch3d.transforms`` is given the
two functions the path calls, restated from their published definitions (matrix_to_rotation_6d = first two
rows; quaternion_to_matrix for real-part-first quaternions).  ``log_output`` (matplotlib figures) is a no-op.

    python tests/golden/make_golden_offline.py  ->  tests/golden/offline.npz
"""
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()


def _matrix_to_rotation_6d(m):
    return m[..., :2, :].clone().reshape(m.size()[:-2] + (6,))


def _quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


pt = sys.modules["pytorch3d.transforms"]
pt.matrix_to_rotation_6d = _matrix_to_rotation_6d
pt.quaternion_to_matrix = _quaternion_to_matrix
pt.rotation_6d_to_matrix = lambda x: (_ for _ in ()).throw(NotImplementedError())
pt.matrix_to_quaternion = lambda x: (_ for _ in ()).throw(NotImplementedError())
sys.modules["pytorch3d"].transforms = pt

import algo.models.transformer.runner as ref_runner_mod  # noqa: E402  (reference)
from algo.models.transformer.data import DataNormalizer, TactileDataset  # noqa: E402  (reference)
from algo.ppo.experience import DataLoggerSim  # noqa: E402  (reference)
from make_golden_student import student_config, _sizes_only_transforms  # noqa: E402

RefRunner = ref_runner_mod.Runner
RefRunner._init_transforms = _sizes_only_transforms
ref_runner_mod.log_output = lambda *a, **k: None

OBS_KEYS = ["eef_pos", "action", "latent", "obs_hist", "noisy_socket_pos", "socket_pos", "hand_joints",
            "plug_hand_quat", "plug_hand_pos", "plug_pos_error", "plug_quat_error"]
NORM_KEYS = ["eef_pos", "noisy_socket_pos", "action", "plug_hand_quat", "plug_hand_pos", "socket_pos"]


def offline_cfg(lin_size):
    cfg = student_config(8, 4, 2, False, False)
    cfg.offline_train.model.linear.input_size = lin_size
    cfg.offline_train.train.update(rh.to_attr({
        "obs_keys": OBS_KEYS, "normalize_obs_keys": NORM_KEYS, "load_stats": False, "normalize_file": "",
        "print_every": 1, "eval_every": 10 ** 9}))
    cfg.offline_train["wandb"] = rh.to_attr({"wandb_enabled": False})
    return cfg


def case_cfg1(out):
    torch.manual_seed(0)
    cfg = offline_cfg(15)
    orig_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self
    try:
        runner = RefRunner(cfg, agent=None)
    finally:
        torch.nn.Module.to = orig_to
    runner.device = "cpu"
    model = runner.model
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():                                   # O(1)-scale weights (Appendix A15)
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight, generator=g)
                m.bias.uniform_(-0.1, 0.1, generator=g)
    for k, v in model.state_dict().items():
        out[f"cfg1/init/{k}"] = v.numpy().copy()
    g = torch.Generator().manual_seed(0)
    stud_obs = torch.randn(256, 1, 15, generator=g)
    action = torch.tanh(torch.randn(256, 1, 6, generator=g))
    latent = torch.randn(256, 1, 8, generator=g)
    vobs = torch.randn(64, 1, 15, generator=g)
    vact = torch.tanh(torch.randn(64, 1, 6, generator=g)) * 1.3       # some targets outside the clamp range
    out["cfg1/stud_obs"], out["cfg1/action"], out["cfg1/latent"] = stud_obs.numpy(), action.numpy(), latent.numpy()
    out["cfg1/val_obs"], out["cfg1/val_action"] = vobs.numpy(), vact.numpy()
    z = torch.zeros(64, 1)
    dl = [(z, z, z, stud_obs[i:i + 64], z, torch.zeros(64, 1, 15), latent[i:i + 64], action[i:i + 64])
          for i in range(0, 256, 64)]
    val_dl = [(z, z, z, vobs, z, torch.zeros(64, 1, 15), torch.zeros(64, 1, 8), vact)]
    runner.optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6)   # runner.py:481
    runner.loss_fn_mean = torch.nn.MSELoss(reduction='mean')                                # runner.py:580
    runner.fig, runner.ax1 = rh._Anything(), rh._Anything()
    runner.train_loss, runner.val_loss = [], []
    runner.save_folder = tempfile.mkdtemp()
    vals = []
    for epoch in range(3):
        vals.append(runner.validate(val_dl))
        vals.append(runner.train(dl, val_dl, runner.save_folder, print_every=1, eval_every=10 ** 9))
    out["cfg1/train_loss"] = np.array(runner.train_loss, dtype=np.float64)
    out["cfg1/val_loss"] = np.array(vals, dtype=np.float64)
    for k, v in model.state_dict().items():
        out[f"cfg1/final/{k}"] = v.numpy().copy()
    print("cfg1 params", sum(p.numel() for p in model.parameters()), "train", out["cfg1/train_loss"], "val", vals)


def synth_trajectory(rng, T, end):
    from scipy.spatial.transform import Rotation
    d = {}
    rot = Rotation.from_rotvec(rng.normal(size=(T, 3)) * 0.4).as_matrix().reshape(T, 9)
    d["eef_pos"] = np.concatenate([rng.normal(size=(T, 3)) * 0.1 + [0.5, 0.0, 0.2], rot], 1).astype(np.float32)
    d["socket_pos"] = np.concatenate([rng.normal(size=(T, 3)) * 0.02 + [0.5, 0.1, 0.0],
                                      np.tile(np.eye(3).reshape(1, 9), (T, 1))], 1).astype(np.float32)
    d["noisy_socket_pos"] = (d["socket_pos"] + rng.normal(size=(T, 12)) * 0.002).astype(np.float32)
    d["action"] = np.tanh(rng.normal(size=(T, 6))).astype(np.float32)
    d["latent"] = rng.normal(size=(T, 8)).astype(np.float32)
    d["obs_hist"] = rng.normal(size=(T, 15)).astype(np.float32)
    d["hand_joints"] = rng.normal(size=(T, 6)).astype(np.float32)
    q = Rotation.from_rotvec(rng.normal(size=(T, 3)) * 0.3).as_quat()
    d["plug_hand_quat"] = q.astype(np.float32)
    d["plug_hand_pos"] = (rng.normal(size=(T, 3)) * 0.01).astype(np.float32)
    d["plug_pos_error"] = (rng.normal(size=(T, 3)) * 0.01).astype(np.float32)
    d["plug_quat_error"] = rng.normal(size=(T, 4)).astype(np.float32)
    done = np.zeros(T, dtype=bool)
    if end is not None:
        done[end] = True
    d["done"] = done
    for k in list(d):
        if k != "done":
            d[k][(end if end is not None else T - 1) + 1:] = 0.0      # the logger leaves zeros after the episode
    return d


def case_data(out):
    rng = np.random.default_rng(7)
    root = tempfile.mkdtemp()
    files, ends = [], [30, 22, 39, None, 3]
    for i, end in enumerate(ends):
        d = synth_trajectory(rng, 40, end)
        folder = os.path.join(root, "w0", f"traj{i}", "obs")
        os.makedirs(folder)
        f = os.path.join(folder, "obs.npz")
        np.savez(f, **d)
        files.append(f)
        for k, v in d.items():
            out[f"data/traj{i}/{k}"] = v
    out["data/ends"] = np.array([-1 if e is None else e for e in ends], dtype=np.int64)
    cfg = offline_cfg(18).offline_train
    random.seed(3)
    norm = DataNormalizer(cfg, list(files), root)
    norm.run()
    out["data/kept"] = np.array([files.index(f) for f in norm.file_list], dtype=np.int64)
    out["data/deleted"] = np.array([int(not os.path.exists(f)) for f in files], dtype=np.int64)
    for kind in ("mean", "std"):
        for k, v in norm.stats[kind].items():
            out[f"data/stats/{kind}/{k}"] = np.asarray(v)
    for L in (1, 4):
        ds = TactileDataset(traj_files=norm.file_list, sequence_length=L, stats=norm.stats, include_img=False,
                            include_seg=False, include_lin=True, include_tactile=False, obs_keys=OBS_KEYS)
        out[f"data/L{L}/indices"] = np.array(ds.indices_per_trajectory, dtype=np.int64)
        items = [ds[i] for i in range(len(ds))]
        for j, name in ((3, "lin_input"), (4, "obj_pos_rpy"), (5, "obs_hist"), (6, "latent"), (7, "action")):
            out[f"data/L{L}/{name}"] = torch.stack([it[j] for it in items]).numpy()
        print("data L", L, "items", len(ds))


def case_logger(out):
    N, T = 3, 6
    lg = DataLoggerSim(N, T, "cpu", tempfile.mkdtemp(), 10 ** 6, False, a_shape=2, b_shape=3)
    saved = []
    lg.pbar = rh._Anything()
    lg.total_trajectories = 10 ** 6          # (only set by the constructor when it also forks its writers)
    lg._save_batch_trajectories = lambda data: saved.append({k: v.numpy().copy() for k, v in data.items()})
    g = torch.Generator().manual_seed(5)
    dones = torch.zeros(8, N, dtype=torch.bool)
    dones[2, 0] = dones[4, 1] = dones[5, 2] = dones[5, 0] = dones[7, 1] = True
    a_seq = torch.randn(8, N, 2, generator=g)
    b_seq = torch.randn(8, N, 3, generator=g)
    for t in range(8):
        lg.update(save_trajectory=True, a=a_seq[t], b=b_seq[t] if t % 3 else None, done=dones[t])
    out["logger/a_seq"], out["logger/b_seq"], out["logger/dones"] = a_seq.numpy(), b_seq.numpy(), dones.numpy()
    out["logger/count"] = np.array([len(saved)], dtype=np.int64)
    for i, s in enumerate(saved):
        for k, v in s.items():
            out[f"logger/traj{i}/{k}"] = v
    print("logger trajectories", len(saved))


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    case_cfg1(out)
    case_data(out)
    case_logger(out)
    np.savez_compressed(os.path.join(HERE, "offline.npz"), **out)
    print("wrote", os.path.join(HERE, "offline.npz"), len(out), "arrays")
