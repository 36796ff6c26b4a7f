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
    most the remaining window.  Unlike the reference, ``update`` never reads the device back (it is called
    three times per environment step during rollouts): the batch mean stays a device scalar and the running
    mean is folded on the host only when ``get_mean`` is asked for it."""

    def __init__(self, window_size):
        self.window_size = int(window_size)
        self.current_size = 0
        self._mean = 0.0            # folded part (python float)
        self._pending = []          # [(kept_weight_of_the_past, weight_of_the_batch, device scalar)]

    def update(self, values):
        k = int(values.shape[0])
        if k == 0:
            return
        w_new = min(k, self.window_size)
        w_old = min(self.window_size - w_new, self.current_size)
        self._pending.append((w_old, w_new, values.float().mean(dim=0).reshape(-1)[0].detach()))
        self.current_size = w_old + w_new

    def _fold(self):
        if not self._pending:
            return
        batch_means = torch.stack([p[2] for p in self._pending]).cpu().tolist()    # one read-back
        for (w_old, w_new, _), m in zip(self._pending, batch_means):
            self._mean = (self._mean * w_old + m * w_new) / (w_old + w_new)
        self._pending = []

    @property
    def mean(self):
        self._fold()
        return self._mean

    def get_mean(self):
        return self.mean

    def clear(self):
        self.current_size, self._mean, self._pending = 0, 0.0, []

    def __len__(self):
        return self.current_size
