"""Rollout storage with the reference's interface (algo/ppo/experience.py:39-46, 148-263):
``ExperienceBuffer`` -- ``update_data / computer_return / prepare_training / __len__ / __getitem__ /
update_mu_sigma``, ``storage_dict`` and ``data_dict`` keys unchanged.

Layout differs on purpose (MI355X-first): the arena stays time-major ``(T, N, ...)`` in HBM and is never
transposed; GAE, advantage normalisation and value normalisation are one native pass
(igi_teacher_prepare).  ``data_dict`` / ``__getitem__`` materialise the reference's env-major
``(N*T, ...)`` tensors lazily, for inspection and parity tests only -- the minibatch loop gathers
straight from the arena inside the fused kernels.  The dead ``contacts`` traffic
is not stored per step unless ``compute_contact_gt`` is used; the key still exists.
``DataLoggerSim`` (experience.py:352-490) writes the per-trajectory ``*.npz`` files the offline loader reads.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def transform_op(arr):
    """swap and then flatten axes 0 and 1 (experience.py:39-46)"""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


class _EnvMajorView:
    """Mapping that produces the reference's ``data_dict`` entries on access."""

    _PREPARED = {"returns": "returns_n", "values": "values_n", "advantages": "advantages", "mus": "mus_w",
                 "sigmas": "sigmas_w"}

    def __init__(self, buf):
        self._b = buf
        self._override = {}

    def keys(self):
        return list(self._b.storage_dict.keys()) + ["advantages"]

    def __contains__(self, k):
        return k in self.keys()

    def __getitem__(self, k):
        if k in self._override:
            return self._override[k]
        b = self._b
        if k in self._PREPARED:
            return transform_op(getattr(b.engine, self._PREPARED[k]))
        return transform_op(b.storage_dict[k])

    def __setitem__(self, k, v):  # frozen_ppo.py:724-725 assigns normalised values/returns back
        self._override[k] = v

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class StudentBuffer(Dataset):
    """Distillation rollout storage with the reference's interface (experience.py:49-145): keys
    ``n_obs n_priv_info rewards teacher_actions student_actions latent_gt`` (+ ``n_tactile n_img n_seg n_pcl
    n_student_obs`` when present), ``update_data / prepare_training / __len__ / __getitem__``.

    The arena stays time-major in HBM: ``__getitem__`` gathers minibatch rows straight from it
    (sample id b = n*T + t -> element [t, n]) instead of first transposing every tensor -- for the
    tactile arena (2048 envs x 32 x 3 x 2048 floats = 1.6 GB) that removes a full copy per update."""

    def __init__(self, num_envs, horizon_length, batch_size, minibatch_size, obs_dim, act_dim, priv_dim,
                 student_dims, device):
        self.device = torch.device(device)
        self.num_envs = num_envs
        self.transitions_per_env = horizon_length
        self.priv_info_dim = priv_dim
        self.data_dict = None
        self.obs_dim, self.act_dim, self.priv_dim = obs_dim, act_dim, priv_dim
        self.tactile_info = student_dims.get('tactile') is not None
        self.img_info = student_dims.get('img') is not None
        self.seg_info = student_dims.get('seg') is not None
        self.student_obs_info = student_dims.get('student_obs') is not None
        self.pcl_info = student_dims.get('pcl') is not None
        T, N, f32 = horizon_length, num_envs, dict(dtype=torch.float32, device=self.device)
        self.storage_dict = {
            'n_obs': torch.zeros((T, N, obs_dim), **f32),
            'n_priv_info': torch.zeros((T, N, priv_dim), **f32),
            'rewards': torch.zeros((T, N, 1), **f32),
            'teacher_actions': torch.zeros((T, N, act_dim), **f32),
            'student_actions': torch.zeros((T, N, act_dim), **f32),
            'latent_gt': torch.zeros((T, N, 8), **f32),
        }
        if self.tactile_info:
            self.storage_dict['n_tactile'] = torch.zeros((T, N, *student_dims['tactile']), **f32)
        if self.img_info:
            self.storage_dict['n_img'] = torch.zeros((T, N, *student_dims['img']), **f32)
        if self.seg_info:
            self.storage_dict['n_seg'] = torch.zeros((T, N, *student_dims['seg']), **f32)
        if self.pcl_info:
            self.storage_dict['n_pcl'] = torch.zeros((T, N, *student_dims['pcl']), **f32)
        if self.student_obs_info:
            self.storage_dict['n_student_obs'] = torch.zeros((T, N, student_dims['student_obs']), **f32)
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size
        self.length = self.batch_size // self.minibatch_size
        self.indices = torch.randperm(self.batch_size, requires_grad=False, device=self.device)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        self.last_range = (start, end)
        # sample id b = n*T + t lives at flat row t*N + n of the time-major arena; index_select moves whole
        # rows with wide loads (the fancy-index kernel it replaces ran at 2 TB/s on the 24 KB tactile rows).  The
        # permutation is drawn once per buffer (the reference draws it in __init__ and never again: SURVEY A4), so its arena rows are computed once as well --
        # per minibatch that was four index kernels -- and again only if someone replaces or edits ``indices``
        T, N = self.transitions_per_env, self.num_envs
        ind = self.indices
        if getattr(self, "_flat_src", None) is not ind or self._flat_ver != ind._version:
            # validated ONCE per permutation (one host read-back, not per step): igi_gather_rows answers an out-of-range
            # row with a NaN row and success, where the reference's fancy indexing (experience.py:117-139) fails loudly --
            # a corrupted or replaced ``indices`` must not train on NaN minibatches silently
            if ind.numel() and (ind.dtype != torch.int64 or int(ind.min()) < 0 or int(ind.max()) >= T * N):
                raise IndexError(f"StudentBuffer.indices: int64 values in [0, {T * N}) expected "
                                 f"(dtype {ind.dtype}, min {int(ind.min())}, max {int(ind.max())})")
            self._flat_rows = (ind % T) * N + ind // T
            self._flat_src, self._flat_ver = ind, ind._version
        flat = self._flat_rows[start:end]
        return _LazyMinibatch(self.storage_dict, flat, T * N)

    def update_data(self, name, index, val):
        self.storage_dict[name][index, :] = val

    def prepare_training(self):
        """experience.py:141-145: nothing to copy here; ``data_dict`` entries are produced on access."""
        self.data_dict = _LazyEnvMajor(self.storage_dict)
        return self.data_dict


class _LazyMinibatch:
    """The minibatch dict of ``StudentBuffer.__getitem__`` (experience.py:117-139 gathers every key) with each key's rows
    gathered on first access: a behaviour-cloning step reads four of the eleven keys (the others -- n_obs, n_priv_info,
    rewards, latent_gt, student_actions ... -- cost seven index kernels per optimizer step for nothing).  Same values,
    same key set, dict interface."""

    def __init__(self, storage, flat_rows, rows_total):
        self._s, self._flat, self._rt, self._c = storage, flat_rows, rows_total, {}

    def __getitem__(self, k):
        if k not in self._c:
            v = self._s[k]
            self._c[k] = v.reshape(self._rt, -1).index_select(0, self._flat).reshape(self._flat.numel(), *v.shape[2:])
        return self._c[k]

    def get(self, k, default=None):
        return self[k] if k in self._s else default

    def prefetch(self, keys):
        """Gather the rows of several keys in ONE launch (torch.ops.mi355ppo.gather_rows; a key by itself is one
        index_select launch each).  Same values; keys that are absent or already gathered are skipped."""
        ks = [k for k in keys if k in self._s and k not in self._c]
        ks = [k for k in ks if self._s[k].is_cuda and self._s[k].dtype is torch.float32 and self._s[k].numel() > 0][:8]
        if len(ks) < 2:
            return
        outs = torch.ops.mi355ppo.gather_rows([self._s[k].reshape(self._rt, -1) for k in ks], self._flat)
        for k, o in zip(ks, outs):
            self._c[k] = o.reshape(self._flat.numel(), *self._s[k].shape[2:])

    def __contains__(self, k):
        return k in self._s

    def __iter__(self):
        return iter(self._s)

    def __len__(self):
        return len(self._s)

    def keys(self):
        return self._s.keys()

    def values(self):
        return [self[k] for k in self._s]

    def items(self):
        return [(k, self[k]) for k in self._s]


class _LazyEnvMajor:
    def __init__(self, storage):
        self._s = storage

    def keys(self):
        return self._s.keys()

    def __contains__(self, k):
        return k in self._s

    def __getitem__(self, k):
        return transform_op(self._s[k])

    def items(self):
        return [(k, self[k]) for k in self._s]


class ExperienceBuffer(Dataset):
    def __init__(self, num_envs, horizon_length, batch_size, minibatch_size, obs_dim, act_dim, priv_dim, pts_dim,
                 vt_poilcy, device, engine=None):
        if vt_poilcy:
            raise NotImplementedError("vt_policy is a dead branch in the reference (frozen_ppo.py:139)")
        self.device = torch.device(device)
        self.num_envs = num_envs
        self.transitions_per_env = horizon_length
        self.priv_info_dim = priv_dim
        self.data_dict = None
        self.obs_dim, self.act_dim, self.priv_dim, self.pts_dim = obs_dim, act_dim, priv_dim, pts_dim
        self.vt_policy = vt_poilcy
        T, N, f32 = horizon_length, num_envs, dict(dtype=torch.float32, device=self.device)
        self.storage_dict = {
            'obses': torch.zeros((T, N, obs_dim), **f32),
            'priv_info': torch.zeros((T, N, priv_dim), **f32),
            # (T,N,pts) zeros in the reference even when unused; kept as a broadcast view (no HBM cost)
            'contacts': torch.zeros((1, 1, pts_dim), **f32).expand(T, N, pts_dim),
            'rewards': torch.zeros((T, N, 1), **f32),
            'values': torch.zeros((T, N, 1), **f32),
            'neglogpacs': torch.zeros((T, N), **f32),
            'dones': torch.zeros((T, N), dtype=torch.uint8, device=self.device),
            'actions': torch.zeros((T, N, act_dim), **f32),
            'mus': torch.zeros((T, N, act_dim), **f32),
            'sigmas': torch.zeros((T, N, act_dim), **f32),
            'returns': torch.zeros((T, N, 1), **f32),
        }
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size
        self.length = self.batch_size // self.minibatch_size
        self.indices = torch.randperm(self.batch_size, requires_grad=False, device=self.device)  # drawn once
        self.last_values = torch.zeros((N, 1), **f32)
        self.gamma, self.tau = 0.99, 0.95
        self.engine = engine
        self.last_range = (0, minibatch_size)

    def attach(self, engine):
        self.engine = engine

    def _own_engine(self):
        if self.engine is None:  # stand-alone use: a tiny network is enough for the prepare pass
            from ...teacher_native import TeacherEngine
            self.engine = TeacherEngine(self.num_envs, self.transitions_per_env, max(1, self.length), units=(8,),
                                        priv_units=(8,), obs_dim=self.obs_dim, priv_dim=self.priv_dim,
                                        act_dim=self.act_dim, device=self.device, perm=self.indices,
                                        normalize_value=False)
        return self.engine

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        """experience.py:207-226 (tuple order unchanged); torch indexing, off the hot path."""
        start, end = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        self.last_range = (start, end)
        b = self.indices[start:end]
        d = self.data_dict
        return tuple(d[k][b] for k in ('values', 'neglogpacs', 'advantages', 'mus', 'sigmas', 'returns',
                                       'actions', 'obses', 'priv_info', 'contacts'))

    def update_mu_sigma(self, mu, sigma):
        """experience.py:228-233 (the fused loss kernel does this itself during training)."""
        b = self.indices[self.last_range[0]:self.last_range[1]]
        T = self.transitions_per_env
        eng = self._own_engine()
        eng.mus_w[b % T, b // T] = mu
        eng.sigmas_w[b % T, b // T] = sigma

    def update_data(self, name, index, val):
        if name == 'contacts':
            return
        self.storage_dict[name][index, :] = val

    def computer_return(self, last_values, gamma, tau):
        """experience.py:242-255: recorded here, executed fused with prepare_training."""
        self.last_values.copy_(last_values.reshape(self.num_envs, 1))
        self.gamma, self.tau = gamma, tau

    def prepare_training(self, value_mean_std=None):
        """experience.py:257-263 (+ frozen_ppo.py:717-725 when a value normaliser is given):
        GAE, advantage normalisation and value/return normalisation in one native pass."""
        eng = self._own_engine()
        eng.cfg.gamma, eng.cfg.tau = float(self.gamma), float(self.tau)
        eng.hp["normalize_value"] = value_mean_std is not None
        ro = dict(self.storage_dict)
        ro.pop('contacts')
        ro.pop('returns')
        ro['last_values'] = self.last_values
        if not getattr(eng, "_workspace_tuned", False):
            # once, before the first update: keep the workspace allocation the update runs fastest on (up to 2 %; every state
            # tensor is put back -- TeacherEngine.tune_workspace; IGI_WS_TRIALS=1 turns it off)
            eng._workspace_tuned = True
            eng.set_rollout(ro)
            eng.tune_workspace()
        eng.prepare(ro)
        self.storage_dict['returns'] = eng.returns_raw   # what computer_return writes (experience.py:255)
        self.data_dict = _EnvMajorView(self)
        return self.data_dict


class DataLoggerSim:
    """Per-trajectory ``*.npz`` writer with the reference's interface (experience.py:352-490):
    ``DataLoggerSim(num_envs, episode_length, device, dir_path, total_trajectories, save_trajectory,
    <key>_shape=...)``, ``update(save_trajectory=True, done=..., <key>=tensor)``, ``get_data()``,
    ``reset()``.  Every finished episode becomes one file ``<dir>/<writer>/<stamp>.npz`` holding the
    episode's ``(episode_length, dim)`` float32 arrays (zero after the last step) plus ``done``
    (episode_length,) bool -- the format ``TactileDataset`` / ``DataNormalizer`` read.

    Differences (behaviour-preserving): buffers stay on the device and all episodes that finish in a step
    leave in ONE gather + device-to-host copy; compression runs on writer threads (the reference forks
    eight processes); file names carry a counter, because the reference's
    second-resolution time stamps overwrite episodes that finish within the same second; reaching
    ``total_trajectories`` drains the writers and sets ``finished`` instead of calling ``exit()``."""

    num_workers = 8

    def __init__(self, num_envs, episode_length, device, dir_path, total_trajectories, save_trajectory, **kwargs):
        import queue
        import threading
        self.num_envs, self.device = num_envs, device
        self.transitions_per_env = episode_length
        os.makedirs(dir_path, exist_ok=True)
        self.dir = dir_path
        self.data_shapes = {k[:-len("_shape")]: v for k, v in kwargs.items() if k.endswith("_shape")}
        self.trajectory_ctr = 0
        self.total_trajectories = total_trajectories
        self.finished = False
        self._init_buffers()
        self.q_s, self.workers = [], []
        if save_trajectory:
            self.q_s = [queue.Queue(maxsize=episode_length) for _ in range(self.num_workers)]
            self.workers = [threading.Thread(target=self.worker, args=(q, i), daemon=True)
                            for i, q in enumerate(self.q_s)]
            for w in self.workers:
                w.start()

    def _init_buffers(self):
        self.log_data = {}
        for key, shape in self.data_shapes.items():
            if shape is None:
                continue
            tail = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
            self.log_data[key] = torch.zeros((self.num_envs, self.transitions_per_env) + tail, dtype=torch.float32,
                                             device=self.device)
        self.done = torch.zeros((self.num_envs, self.transitions_per_env), dtype=torch.bool, device=self.device)
        self.env_step_counter = torch.zeros((self.num_envs, 1), dtype=torch.long, device=self.device)
        self.env_ids = torch.arange(self.num_envs, dtype=torch.long, device=self.device).unsqueeze(-1)

    def _reset_buffers(self, env_ids):
        ids = env_ids.reshape(-1)
        for buf in self.log_data.values():
            buf[ids] = 0.
        self.done[ids] = False
        self.env_step_counter[ids] = 0

    def _save_batch_trajectories(self, data):
        self.q_s[self.trajectory_ctr % len(self.q_s)].put(data)

    def update(self, save_trajectory=True, **kwargs):
        step = self.env_step_counter
        for key, value in kwargs.items():
            if key == "done":
                continue
            if value is None:
                value = torch.zeros((self.num_envs, self.data_shapes[key]), dtype=torch.float32, device=self.device)
            self.log_data[key][self.env_ids, step] = value.to(torch.float32).unsqueeze(1)
        done = kwargs.get('done', None)
        if done is None:
            done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        done = done.to(torch.bool)
        self.done[self.env_ids, step] = done.unsqueeze(1)
        self.env_step_counter += 1
        ids = done.nonzero().reshape(-1)
        if ids.numel() == 0:
            return
        if save_trajectory and self.q_s and not self.finished:
            host = {k: v.index_select(0, ids).cpu().numpy() for k, v in self.log_data.items()}
            host_done = self.done.index_select(0, ids).cpu().numpy()
            for j in range(ids.numel()):
                item = {k: v[j] for k, v in host.items()}
                item['done'] = host_done[j]
                self._save_batch_trajectories(item)
                self.trajectory_ctr += 1
        self._reset_buffers(ids)
        if save_trajectory and self.q_s and self.trajectory_ctr >= self.total_trajectories and not self.finished:
            self._shutdown_workers()
            print('Data collection finished!')

    def worker(self, q, q_idx):
        import time
        path = os.path.join(self.dir, f'{q_idx}')
        os.makedirs(path, exist_ok=True)
        n = 0
        while True:
            item = q.get()
            try:
                if item is None:
                    return
                stamp = time.strftime("%Y-%m-%d_%H-%M-%S")
                np.savez_compressed(os.path.join(path, f'{stamp}_{n:06d}.npz'), **item)
                n += 1
            finally:
                q.task_done()

    def _shutdown_workers(self):
        for q in self.q_s:
            q.put(None)
        for w in self.workers:
            w.join()
        self.finished = True

    def flush(self):
        """Block until every queued trajectory is on disk."""
        for q in self.q_s:
            q.join()

    def get_data(self):
        return self.log_data

    def reset(self):
        self._reset_buffers(self.env_ids)
