"""Meters and stat aggregation with the reference's names (isaacgyminsertion/utils/misc.py:69-133)."""
import os

import numpy as np
import torch


def get_world_size():
    """misc.py:94-100 (the reference divides by LOCAL_WORLD_SIZE; single node => == WORLD_SIZE)."""
    return int(os.environ.get("LOCAL_WORLD_SIZE", "1"))


def multi_gpu_aggregate_stats(values):
    """misc.py:69-91: mean over ranks of each stat (a tensor, or a list of scalars tensors that is stacked);
    one-element results come back as python numbers.  A bare tensor in -> a bare result out."""
    import torch.distributed as dist
    world = get_world_size()
    reduce = world > 1 and dist.is_initialized()

    def one(v):
        t = torch.stack(v) if isinstance(v, list) else v
        if reduce:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t = t / world
        return t.item() if t.numel() == 1 else t

    return [one(v) for v in values] if isinstance(values, list) else one(values)


def add_to_fifo(tensor, x):
    """misc.py:102-105: push a new value at the front, drop the oldest."""
    return torch.cat((x, tensor[:, 0:-1]), dim=1)


class AverageScalarMeter:
    """Windowed mean of episode statistics with the reference's interface (misc.py:108-133):
    ``update(values)``, ``get_mean()``, ``clear()``, ``len()``.

    Semantics: a batch of k finished episodes enters with weight min(k, window); what was there keeps at
    most the remaining window.  Unlike the reference, nothing here reads the device back while a rollout is being
    collected (the trainers feed it three times per environment step): batches stay device scalars -- either a mean
    with a host-known k (``update``) or per-step (sum, count) pairs produced by the rollout kernel
    (``update_sums``) -- and the running mean is folded on the host only when ``get_mean`` / ``len`` ask for it."""

    def __init__(self, window_size):
        self.window_size = int(window_size)
        self._size = 0
        self._mean = 0.0            # folded part (python float)
        self._pending = []          # ('batch', k, device scalar mean) | ('sums', sums (S,), counts (S,))

    def update(self, values):
        k = int(values.shape[0])
        if k == 0:
            return
        self._pending.append(('batch', k, values.float().mean(dim=0).reshape(-1)[0].detach()))

    def update_sums(self, sums, counts):
        """One entry per environment step: the sum of the statistic over the episodes that ended in that step
        and how many ended (0 = nothing to add)."""
        self._pending.append(('sums', sums.detach().reshape(-1).float(), counts.detach().reshape(-1).float()))

    def _push(self, k, mean):
        w_new = min(k, self.window_size)
        w_old = min(self.window_size - w_new, self._size)
        self._mean = (self._mean * w_old + mean * w_new) / (w_old + w_new)
        self._size = w_old + w_new

    def _fold(self):
        if not self._pending:
            return
        flat = torch.cat([p[2].reshape(1) if p[0] == 'batch' else torch.cat((p[1], p[2])) for p in self._pending])
        host = flat.cpu().tolist()                                                 # one read-back
        at = 0
        for p in self._pending:
            if p[0] == 'batch':
                self._push(p[1], host[at])
                at += 1
            else:
                n = p[1].numel()
                for s, c in zip(host[at:at + n], host[at + n:at + 2 * n]):
                    if c > 0:
                        self._push(int(round(c)), s / c)
                at += 2 * n
        self._pending = []

    @property
    def current_size(self):
        self._fold()
        return self._size

    @property
    def mean(self):
        self._fold()
        return self._mean

    def get_mean(self):
        return self.mean

    def clear(self):
        self._size, self._mean, self._pending = 0, 0.0, []

    def __len__(self):
        return self.current_size
