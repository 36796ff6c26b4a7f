"""Offline trajectory data with the reference's interface (algo/models/transformer/data.py:17-440):
``RotationTransformer``, ``get_last_sequence``, ``DataNormalizer`` (writes / reads ``normalization.pkl``),
``TactileDataset``; plus ``ResidentLoader``, the MI355X-side batch source.

On-disk contract: one ``*.npz`` per trajectory under
``<data_folder>/*/*/obs/`` holding ``(T, dim)`` float arrays keyed as the sim logger names them
(experience.py:640-675) and ``done`` (T,) whose LAST non-zero index ends the trajectory; tactile frames
live next to it as ``<...>/tactile/tactile_{i}.npz['tactile']`` with shape (fingers, C, W, H).

Design differences from the reference (behaviour-preserving):
  * a trajectory file is opened ONCE (the reference ``np.load``s it for every sample it serves,
    data.py:399) and its normalised proprioception is computed for all frames in one vectorised pass;
  * ``ResidentLoader`` stacks every sub-sequence into device tensors once (288 GB of HBM holds millions
    of 24 KB tactile frames) and serves shuffled minibatches by an index gather on the GPU, replacing 16
    DataLoader worker processes + pinned copies (runner.py:524-551).
pytorch3d (reference dependency, absent here and unpinned in the reference's setup.py) supplies
``matrix_to_rotation_6d`` / ``quaternion_to_matrix``; they are restated from their published definitions
(Zhou et al. 2019 6-D representation = first two ROWS of the matrix; real-part-first quaternions).
"""
import os
import pickle
import random
from pathlib import Path

import numpy as np
import torch
from scipy.spatial.transform import Rotation
from torch.utils.data import Dataset


# ---------------------------------------------------------------------------------------------
# rotation representations (pytorch3d.transforms semantics)
# ---------------------------------------------------------------------------------------------
def quaternion_to_matrix(q):
    """(…,4) real-part-first quaternions -> (…,3,3)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(m):
    """(…,3,3) -> (…,4) real-part-first, non-negative real part."""
    flat = m.reshape(-1, 3, 3).double().numpy() if isinstance(m, torch.Tensor) else np.asarray(m).reshape(-1, 3, 3)
    xyzw = Rotation.from_matrix(flat).as_quat()
    wxyz = np.concatenate([xyzw[:, 3:], xyzw[:, :3]], axis=1)
    wxyz = np.where(wxyz[:, :1] < 0, -wxyz, wxyz)
    out = torch.from_numpy(wxyz).reshape(tuple(m.shape[:-2]) + (4,))
    return out.to(m.dtype) if isinstance(m, torch.Tensor) else out


def matrix_to_rotation_6d(m):
    """(…,3,3) -> (…,6): the first two rows, flattened."""
    return m[..., :2, :].clone().reshape(m.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


_TO_MATRIX = {'quaternion': quaternion_to_matrix, 'rotation_6d': rotation_6d_to_matrix}
_FROM_MATRIX = {'quaternion': matrix_to_quaternion, 'rotation_6d': matrix_to_rotation_6d}


class RotationTransformer:
    """data.py:17-98: converts between rotation representations through the matrix form; accepts
    numpy arrays or tensors.  'quaternion', 'rotation_6d' and 'matrix' are built (the only ones the
    training path uses, data.py:137-138, 289)."""
    valid_reps = ['axis_angle', 'euler_angles', 'quaternion', 'rotation_6d', 'matrix']

    def __init__(self, from_rep='quaternion', to_rep='rotation_6d', from_convention=None, to_convention=None):
        assert from_rep != to_rep
        assert from_rep in self.valid_reps and to_rep in self.valid_reps
        for rep in (from_rep, to_rep):
            if rep not in ('quaternion', 'rotation_6d', 'matrix'):
                raise NotImplementedError(f"rotation representation {rep!r} is not on the training path")
        self.forward_funcs, self.inverse_funcs = [], []
        if from_rep != 'matrix':
            self.forward_funcs.append(_TO_MATRIX[from_rep])
            self.inverse_funcs.append(_FROM_MATRIX[from_rep])
        if to_rep != 'matrix':
            self.forward_funcs.append(_FROM_MATRIX[to_rep])
            self.inverse_funcs.append(_TO_MATRIX[to_rep])
        self.inverse_funcs = self.inverse_funcs[::-1]

    @staticmethod
    def _apply_funcs(x, funcs):
        is_np = isinstance(x, np.ndarray)
        y = torch.from_numpy(x) if is_np else x
        for f in funcs:
            y = f(y)
        return y.numpy() if is_np else y

    def forward(self, x):
        return self._apply_funcs(x, self.forward_funcs)

    def inverse(self, x):
        return self._apply_funcs(x, self.inverse_funcs)


def get_last_sequence(input_tensor, progress_buf, sequence_length):
    """data.py:100-126: (E, T, …) -> (E, sequence_length, …): the most recent ``sequence_length`` entries
    up to ``progress_buf[e]``, front-padded with 1e-6 while the episode is still shorter than that."""
    E = input_tensor.shape[0]
    out = torch.full((E, sequence_length) + tuple(input_tensor.shape[2:]), 1e-6, dtype=torch.float32,
                     device=input_tensor.device)
    prog = torch.as_tensor(progress_buf, device=input_tensor.device).reshape(-1).long()
    ar = torch.arange(sequence_length, device=input_tensor.device)
    short = prog < sequence_length
    n_valid = torch.where(short, prog + 1, torch.full_like(prog, sequence_length))       # entries copied
    first_src = torch.where(short, torch.zeros_like(prog), prog - sequence_length)         # source start
    dst_pad = sequence_length - n_valid
    src = first_src[:, None] + ar[None, :] - dst_pad[:, None]
    valid = ar[None, :] >= dst_pad[:, None]
    src = src.clamp(0, input_tensor.shape[1] - 1)
    gathered = input_tensor[torch.arange(E, device=input_tensor.device)[:, None], src].to(torch.float32)
    mask = valid.reshape(valid.shape + (1,) * (input_tensor.dim() - 2))
    return torch.where(mask, gathered, out)


# ---------------------------------------------------------------------------------------------
# normalisation statistics
# ---------------------------------------------------------------------------------------------
def _trajectory_end(done):
    nz = np.asarray(done).nonzero()[0]
    return int(nz[-1]) if len(nz) else None


class DataNormalizer:
    """data.py:129-270.  ``stats = {"mean": {...}, "std": {...}}`` over the concatenated, done-trimmed
    trajectories; derived entries for poses (position differences, Euler angles, 6-D rotations) exactly as
    the reference names them.  Like the reference, trajectories without a ``done`` flag (or unreadable
    files) are dropped from the list AND deleted from disk unless ``delete_failed=False``."""

    def __init__(self, cfg, file_list, save_path=None, delete_failed=True):
        self.cfg = cfg
        self.normalize_obs_keys = self.cfg.train.normalize_obs_keys
        self.normalization_path = (self.cfg.train.normalize_file if self.cfg.train.load_stats
                                   else save_path + '/normalization.pkl')
        self.stats = {"mean": {}, "std": {}}
        self.file_list = file_list
        self.delete_failed = delete_failed
        self.remove_failed_trajectories()
        self.rot_tf = RotationTransformer(from_rep='matrix', to_rep='rotation_6d')
        self.rot_tf_from_quat = RotationTransformer()

    def ensure_directory_exists(self, path):
        Path(path).parent.absolute().mkdir(parents=True, exist_ok=True)

    def remove_failed_trajectories(self):
        kept = []
        for f in self.file_list:
            ok = False
            try:
                with np.load(f) as d:
                    ok = _trajectory_end(d['done']) is not None
            except KeyboardInterrupt:
                raise
            except Exception as e:                      # unreadable file
                print(f"Error processing {f}: {e}")
            if ok:
                kept.append(f)
            elif self.delete_failed and os.path.exists(f):
                os.remove(f)
        self.file_list = kept

    def load_or_create_normalization_file(self):
        if self.cfg.train.load_stats and os.path.exists(self.normalization_path):
            with open(self.normalization_path, 'rb') as f:
                self.stats = pickle.load(f)
            print('Loaded stats file from: ', self.normalization_path)
        else:
            self.create_normalization_file()

    def create_normalization_file(self):
        cache = {}
        for key in self.normalize_obs_keys:
            self.calculate_normalization_values(self.aggregate_data(key, cache), key)
        self.save_normalization_file()

    def aggregate_data(self, norm_key, cache=None):
        """Rows [0, done_idx) of every trajectory, concatenated.  The reference visits the files in a
        fresh random order per key (data.py:188); "first row" statistics (pos - pos[0]) therefore depend
        on Python's ``random`` state there -- here too (same ``random.sample`` call)."""
        chunks = []
        for f in random.sample(self.file_list, len(self.file_list)):
            try:
                if cache is not None and f in cache:
                    d = cache[f]
                else:
                    with np.load(f) as z:
                        d = {k: z[k] for k in z.files}
                    if cache is not None:
                        cache[f] = d
                chunks.append(d[norm_key][:_trajectory_end(d['done']), :])
            except Exception as e:
                print(f"{f} could not be processed: {e}")
        return np.concatenate(chunks, axis=0)

    def _put(self, name, x):
        self.stats['mean'][name] = np.mean(x, axis=0)
        self.stats['std'][name] = np.std(x, axis=0)

    def calculate_normalization_values(self, data, norm_key):
        if norm_key == 'plug_hand_pos':
            pos = data[:, :3] if data.shape[1] == 7 else data
            self._put("plug_hand_pos", pos)
            self._put("plug_hand_pos_diff", pos - pos[0, :])
            if data.shape[1] == 7:                       # position + quaternion logged together
                quat = data[:, 3:]
                self._put("plug_hand_quat", quat)
                euler = Rotation.from_quat(quat).as_euler('xyz')
                self._put("plug_hand_euler", euler)
                self._put("plug_hand_diff_euler", euler - euler[0, :])
        elif norm_key == 'plug_hand_quat':
            self._put(norm_key, data)
            euler = Rotation.from_quat(data).as_euler('xyz')
            self._put("plug_hand_euler", euler)
            self._put("plug_hand_diff_euler", euler - euler[0, :])
            self._put("plug_hand_rot6d", self.rot_tf_from_quat.forward(data))
        elif norm_key == 'eef_pos':
            rot6d = self.rot_tf.forward(data[:, 3:].reshape(data.shape[0], 3, 3))
            self._put('eef_pos_rot6d', np.concatenate((data[:, :3], rot6d), axis=1))
            self._put(norm_key, data)
        else:
            self._put(norm_key, data)

    def save_normalization_file(self):
        with open(self.normalization_path, 'wb') as f:
            pickle.dump(self.stats, f)
        print(f'Saved new normalization file at: {self.normalization_path}')

    def run(self):
        self.ensure_directory_exists(self.normalization_path)
        self.load_or_create_normalization_file()


# ---------------------------------------------------------------------------------------------
# dataset
# ---------------------------------------------------------------------------------------------
class TactileDataset(Dataset):
    """data.py:273-440.  Item = (tactile, img, seg, lin_input, obj_pos_rpy, obs_hist, latent, action), each
    float32 with a leading ``sequence_length`` axis; absent modalities are ``zeros(1)`` like the reference.
    ``lin_input`` = [eef position + 6-D rotation (9) | socket position (3) | previous action (6)], the
    first two standardised with ``stats``; "previous action" is the sub-sequence's own actions shifted
    right by one with a zero first row (data.py:421-428)."""

    def __init__(self, traj_files, sequence_length=500, stats=None, stride=1, img_transform=None,
                 seg_transform=None, sync_transform=None, tactile_transform=None, include_img=True,
                 include_lin=True, include_tactile=True, include_seg=True, obs_keys=None):
        self.rot_tf = RotationTransformer(from_rep='matrix', to_rep='rotation_6d')
        self.all_folders = list(traj_files)
        self.sequence_length = sequence_length
        self.stride = stride
        self.stats = stats
        self.obs_keys = obs_keys
        self.include_img, self.include_seg = include_img, include_seg
        self.include_lin, self.include_tactile = include_lin, include_tactile
        self.img_transform, self.seg_transform = img_transform, seg_transform
        self.sync_transform, self.tactile_transform = sync_transform, tactile_transform
        self._traj = [self._load_trajectory(f) for f in self.all_folders]
        self.indices_per_trajectory = self._generate_indices()
        print('Total sub trajectories:', len(self.indices_per_trajectory))

    def to_torch(self, x):
        return torch.from_numpy(np.ascontiguousarray(x)).float()

    # -- one pass per file ---------------------------------------------------------------------
    def _load_trajectory(self, path):
        with np.load(path) as z:
            data = {k: z[k] for k in (self.obs_keys or z.files)}
            done = z['done']
        return {'data': data, 'end': _trajectory_end(done), 'path': path, 'norm': {}}

    def _generate_indices(self):
        out = []
        for file_idx, tr in enumerate(self._traj):
            total = tr['end']
            if total is not None and total >= self.sequence_length:
                n = (total - self.sequence_length) // self.stride + 1
                out.extend((file_idx, i * self.stride) for i in range(n))
        return out

    def __len__(self):
        return len(self.indices_per_trajectory)

    def extract_sequence(self, data, key, start_idx):
        return data[key][start_idx:start_idx + self.sequence_length]

    def _normalize_data(self, data_seq, diff, first_obs, rot6d=True):
        """data.py:354-385 on any block of rows."""
        eef_key = "eef_pos_rot6d" if rot6d else 'eef_pos'
        euler_key = "plug_hand_diff_euler" if diff else "plug_hand_euler"
        pos_key = "plug_hand_pos_diff" if diff else "plug_hand_pos"
        eef = data_seq["eef_pos"]
        if rot6d:
            eef = np.concatenate((eef[:, :3], self.rot_tf.forward(eef[:, 3:].reshape(eef.shape[0], 3, 3))), axis=1)
        socket = data_seq["socket_pos"][:, :3]
        euler = Rotation.from_quat(data_seq["plug_hand_quat"]).as_euler('xyz')
        pos = data_seq["plug_hand_pos"]
        if diff:
            euler = euler - Rotation.from_quat(first_obs["plug_hand_quat"]).as_euler('xyz')
            pos = pos - first_obs["plug_hand_pos"]
        if self.stats is not None:
            m, s = self.stats["mean"], self.stats["std"]
            eef = (eef - m[eef_key]) / s[eef_key]
            socket = (socket - m["socket_pos"][:3]) / s["socket_pos"][:3]
            euler = (euler - m[euler_key]) / s[euler_key]
            pos = (pos - m[pos_key]) / s[pos_key]
        return eef, socket, np.hstack((pos, euler))

    def _normalized(self, tr, diff):
        """whole-trajectory normalised proprioception, computed once per (file, diff)."""
        if diff not in tr['norm']:
            d = tr['data']
            first = {k: d[k][0] for k in d}
            live = {k: v[:tr['end']] for k, v in d.items()}      # rows past the episode end are zero padding
            tr['norm'][diff] = self._normalize_data(live, diff, first)
        return tr['norm'][diff]

    def _load_and_preprocess_tactile(self, tactile_folder, start_idx, diff_tac):
        frames = [np.load(os.path.join(tactile_folder, f'tactile_{i}.npz'))['tactile']
                  for i in range(start_idx, start_idx + self.sequence_length)]
        if diff_tac:
            first = np.load(os.path.join(tactile_folder, 'tactile_1.npz'))['tactile']
            frames = [(f - first) + 1e-6 for f in frames]
        x = self.to_torch(np.stack(frames))
        if self.tactile_transform is not None:
            x = self._apply_tactile_transform(x)
        return x

    def _load_and_preprocess_image(self, img_folder, seg_folder, start_idx, distinct=True, obj_id=2, socket_id=3):
        """data.py:337-352: depth frames ``img/img_{i}.npz['img']`` and segmentation ids ``seg/seg_{i}.npz['seg']``
        of the sub-sequence; only plug (id 2) and socket (id 3) pixels survive in both.  ``sync_transform(img, seg)``
        (crop / augmentation applied to both alike) runs when given."""
        idx = range(start_idx, start_idx + self.sequence_length)
        img = np.stack([np.load(os.path.join(img_folder, f'img_{i}.npz'))['img'] for i in idx])
        seg = np.stack([np.load(os.path.join(seg_folder, f'seg_{i}.npz'))['seg'] for i in idx])
        valid = ((seg == obj_id) | (seg == socket_id)).astype(np.float32)
        seg = seg * valid if distinct else valid
        img = img * valid
        img, seg = self.to_torch(img), self.to_torch(seg)
        if self.sync_transform is not None:
            img, seg = self.sync_transform(img, seg)
        return img, seg

    def _apply_tactile_transform(self, x):
        T, F, C, W, H = x.shape
        y = self.tactile_transform(x.reshape(-1, C, W, H))
        return y.reshape(T, F, C, *y.shape[2:])

    def __getitem__(self, idx, diff_tac=True, diff=False):
        file_idx, start = self.indices_per_trajectory[idx]
        tr = self._traj[file_idx]
        L = self.sequence_length
        sl = slice(start, start + L)
        if self.include_tactile:
            folder = tr['path'][:-7].replace('obs', 'tactile')       # data.py:400 (every 'obs' in the path)
            tactile = self._load_and_preprocess_tactile(folder, start, diff_tac)
        else:
            tactile = torch.zeros(1)
        if self.include_img or self.include_seg:
            img, seg = self._load_and_preprocess_image(tr['path'][:-7].replace('obs', 'img'),
                                                       tr['path'][:-7].replace('obs', 'seg'), start)
        else:
            img, seg = torch.zeros(1), torch.zeros(1)
        eef, socket, obj_pos_rpy = (a[sl] for a in self._normalized(tr, diff))
        d = tr['data']
        action = d["action"][sl]
        prev_action = np.concatenate([np.zeros((1, action.shape[-1])), action[:-1, :]], axis=0)
        lin_input = np.concatenate([eef, socket, prev_action], axis=-1)
        return (tactile, img, seg, self.to_torch(lin_input), self.to_torch(obj_pos_rpy),
                self.to_torch(d["obs_hist"][sl]), self.to_torch(d["latent"][sl]), self.to_torch(action))


class ResidentLoader:
    """Minibatch source over a ``TactileDataset`` whose items all live in device memory: built once,
    then every epoch is a ``randperm`` + row gathers on the GPU.  Iterates like the reference's
    ``DataLoader(ds, batch_size, shuffle=True)`` (runner.py:524-551): same 8-tuple per batch, last
    partial batch kept; ``len()`` = number of batches.  ``tactile_transform`` (train-time augmentation)
    is applied per batch on the device."""

    def __init__(self, dataset, batch_size, shuffle=True, device='cuda', generator=None, tactile_transform=None):
        self.batch_size, self.shuffle, self.device = int(batch_size), shuffle, device
        self.generator = generator
        self.tactile_transform = tactile_transform
        n = len(dataset)
        self.n = n
        if n == 0:
            self.fields = None
            return
        saved, dataset.tactile_transform = dataset.tactile_transform, None
        try:
            items = [dataset[i] for i in range(n)]
        finally:
            dataset.tactile_transform = saved
        self.fields = []
        for j in range(8):
            col = [it[j] for it in items]
            if all(c.numel() == 1 and c.dim() == 1 for c in col):         # absent modality: zeros(1) per item
                self.fields.append(None)
            else:
                self.fields.append(torch.stack(col).to(device))

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        if self.n == 0:
            return
        if self.shuffle:
            order = torch.randperm(self.n, generator=self.generator).to(self.device)
        else:
            order = torch.arange(self.n, device=self.device)
        for s in range(0, self.n, self.batch_size):
            idx = order[s:s + self.batch_size]
            batch = []
            for j, f in enumerate(self.fields):
                if f is None:
                    batch.append(torch.zeros(idx.numel(), 1, device=self.device))
                    continue
                x = f.index_select(0, idx)
                if j == 0 and self.tactile_transform is not None:
                    B, T, Fg, C, W, H = x.shape
                    y = self.tactile_transform(x.reshape(-1, C, W, H))
                    x = y.reshape(B, T, Fg, C, *y.shape[2:])
                batch.append(x)
            yield tuple(batch)
