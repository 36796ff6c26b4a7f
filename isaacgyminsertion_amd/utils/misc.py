"""Meters and stat aggregation with the reference's names (isaacgyminsertion/utils/misc.py:69-133)."""
import os

import numpy as np
import torch


def get_world_size():
    """misc.py:94-100 (the reference divides by LOCAL_WORLD_SIZE; single node => == WORLD_SIZE)."""
    return int(os.environ.get("LOCAL_WORLD_SIZE", "1"))


def multi_gpu_aggregate_stats(values):
    """misc.py:69-91: all-reduce(SUM)/world of each stat tensor (lists are stacked first)."""
    import torch.distributed as dist
    single_item = not isinstance(values, list)
    if single_item:
        values = [values]
    rst = []
    for v in values:
        if isinstance(v, list):
            v = torch.stack(v)
        if get_world_size() > 1 and dist.is_initialized():
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v = v / get_world_size()
        if v.numel() == 1:
            v = v.item()
        rst.append(v)
    return rst[0] if single_item else rst


def add_to_fifo(tensor, x):
    """misc.py:102-105: push a new value at the front, drop the oldest."""
    return torch.cat((x, tensor[:, 0:-1]), dim=1)


class AverageScalarMeter(object):
    """Windowed mean of episode statistics (misc.py:108-133)."""

    def __init__(self, window_size):
        self.window_size = window_size
        self.current_size = 0
        self.mean = 0

    def update(self, values):
        size = values.size()[0]
        if size == 0:
            return
        new_mean = torch.mean(values.float(), dim=0).cpu().numpy().item()
        size = np.clip(size, 0, self.window_size)
        old_size = min(self.window_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = (self.mean * old_size + new_mean * size) / size_sum

    def clear(self):
        self.current_size = 0
        self.mean = 0

    def __len__(self):
        return self.current_size

    def get_mean(self):
        return self.mean
