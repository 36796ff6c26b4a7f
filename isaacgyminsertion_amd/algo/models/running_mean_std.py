"""RunningMeanStd with the reference's interface and state_dict (algo/models/running_mean_std.py:23-93),
computed by libigi_hip.so (torch.ops.mi355ppo.rms_update_normalize -> igi_rms_forward).

State is ONE packed fp64 device vector [mean(D), var(D), count]; the registered buffers
``running_mean`` / ``running_var`` / ``count`` (same names, dtypes and shapes as the reference, so
checkpoints interchange) are views into it.  The fused PPO kernels update the same memory.
"""
import torch
import torch.nn as nn

from ... import ops  # noqa: F401  (registers torch.ops.mi355ppo)


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        if per_channel or norm_only:
            # never used on the PPO / student path
            raise NotImplementedError("per_channel / norm_only normalisers are outside the hot path")
        if isinstance(insize, int):
            insize = (insize,)
        self.insize = tuple(insize)
        if len(self.insize) != 1:
            raise NotImplementedError("only 1-D feature normalisers are on the hot path")
        self.epsilon = epsilon
        self.axis = [0]
        d = self.insize[0]
        packed = torch.zeros(2 * d + 1, dtype=torch.float64)
        packed[d:2 * d] = 1.0   # running_var (running_mean_std.py:45)
        packed[2 * d] = 1.0     # count       (running_mean_std.py:46)
        self._ws = None
        self._set_packed(packed)

    # -- packed state <-> buffers ---------------------------------------------------------------
    def _set_packed(self, packed):
        d = self.insize[0]
        self._packed = packed
        self._buffers["running_mean"] = packed[:d]
        self._buffers["running_var"] = packed[d:2 * d]
        self._buffers["count"] = packed[2 * d]

    def bind(self, packed):
        """Adopt an externally owned packed state (the trainer's engine), keeping current values."""
        packed.copy_(self._packed.to(packed.device))
        self._set_packed(packed)

    def _apply(self, fn, recurse=True):
        self._set_packed(fn(self._packed))
        self._ws = None
        return self

    @property
    def packed(self):
        return self._packed

    # -- forward --------------------------------------------------------------------------------
    def forward(self, input, unnorm=False):
        if not input.is_cuda:
            raise RuntimeError("RunningMeanStd.forward runs on the HIP device only (no CPU fallback)")
        if self._packed.device != input.device:
            raise RuntimeError("normaliser state and input live on different devices; call .to(device)")
        d = self.insize[0]
        x = input.to(torch.float32).contiguous()
        if x.shape[-1] != d:
            raise RuntimeError(f"expected last dimension {d}, got {tuple(x.shape)}")
        return torch.ops.mi355ppo.rms_update_normalize(x, self._packed, float(self.epsilon),
                                                       bool(self.training and not unnorm), bool(unnorm))
