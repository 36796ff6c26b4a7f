"""Rollout storage with the reference's interface (algo/ppo/experience.py:39-46, 148-263):
``ExperienceBuffer`` -- ``update_data / computer_return / prepare_training / __len__ / __getitem__ /
update_mu_sigma``, ``storage_dict`` and ``data_dict`` keys unchanged.

Layout differs on purpose (MI355X-first): the arena stays time-major ``(T, N, ...)`` in HBM and is never
transposed; GAE, advantage normalisation and value normalisation are one native pass
(igi_teacher_prepare).  ``data_dict`` / ``__getitem__`` materialise the reference's env-major
``(N*T, ...)`` tensors lazily, for inspection and parity tests only -- the minibatch loop gathers
straight from the arena inside the fused kernels.  The dead ``contacts`` traffic
is not stored per step unless ``compute_contact_gt`` is used; the key still exists.
"""
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
    ``n_obs n_priv_info rewards teacher_actions student_actions latent_gt`` (+ ``n_tactile n_pcl
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
        if student_dims.get('img') is not None or student_dims.get('seg') is not None:
            raise NotImplementedError("depth / segmentation student inputs are the next scope row (SURVEY 8f-3)")
        self.tactile_info = student_dims.get('tactile') is not None
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
        b = self.indices[start:end]
        t, n = b % self.transitions_per_env, b // self.transitions_per_env
        return {k: v[t, n] for k, v in self.storage_dict.items()}

    def update_data(self, name, index, val):
        self.storage_dict[name][index, :] = val

    def prepare_training(self):
        """experience.py:141-145: nothing to copy here; ``data_dict`` entries are produced on access."""
        self.data_dict = _LazyEnvMajor(self.storage_dict)
        return self.data_dict


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
        eng.prepare(ro)
        self.storage_dict['returns'] = eng.returns_raw   # what computer_return writes (experience.py:255)
        self.data_dict = _EnvMajorView(self)
        return self.data_dict
