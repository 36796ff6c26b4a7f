"""Flat-buffer optimizer state for modules trained under torch autograd (the student): parameters
and their gradients are views into two flat fp32 vectors, so
  * the data-parallel exchange is ONE in-place all-reduce on ``flat_grad`` (the reference concatenates
    every gradient, all-reduces and copies back: ext_adapt.py:833-851), and
  * clip_grad_norm_ + Adam is one native pass (igi_clip_adam) instead of ~150 foreach launches.
"""
import torch

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)


class FlatAdam:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5, weight_decay=0.0, l2=0.0):
        """weight_decay > 0 = torch.optim.AdamW's decoupled decay (runner.py:481); l2 > 0 = torch.optim.Adam's
        coupled ``weight_decay`` (grad += l2 * param after clipping; ext_adapt.py:1139); both 0 = plain Adam.

        Every parameter handed in is stepped on every call, with a zero gradient if autograd produced none.  With both
        decays at 0 that is a no-op for such a parameter (its moments stay 0); with a decay it would shrink a parameter
        torch's optimizers skip (``grad is None``) -- so pass only parameters that take part in the loss when a decay
        is set, as ``restore_student(phase=3)`` does (the tactile branch / 'new' layers, all of which receive
        gradients; the never-used ``decoder.sa_layer.*`` template is not among them)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam runs on the HIP device only (no CPU fallback)")
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]     # 16-byte aligned slices
        n = sum(sizes)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self._views = []
        for p, sz in zip(self.params, sizes):
            v = self.flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = None
            self._views.append(self.flat_grad[off:off + p.numel()].view(p.shape))
            off += sz
        self._zero = [True] * len(self.params)     # gradient slices known to hold zeros
        self._synced = False
        self._marks = None
        self.param_groups = [{"lr": float(lr)}]
        self.betas, self.eps, self.max_norm = betas, eps, max_norm
        self.weight_decay = float(weight_decay)
        self.l2 = float(l2)
        self.t = 0
        self.stats = torch.zeros(8, dtype=torch.float32, device=dev)

    def zero_grad(self):
        """``set_to_none`` semantics: autograd then STORES each parameter's gradient (no kernel) instead of adding it
        into a zero-filled view (one ATen add per parameter tensor: 48-60 launches per student step); ``sync_grads``
        moves the gradients into the flat buffer with one multi-tensor copy."""
        for p in self.params:
            p.grad = None
        self._synced = False

    def _grad_marks(self):
        return [(None if p.grad is None else (id(p.grad), p.grad._version)) for p in self.params]

    def sync_grads(self):
        """Gather the parameters' ``.grad`` into ``flat_grad``: one ``torch._foreach_copy_`` for every parameter that
        received a gradient; slices of parameters without one (the never-used ``decoder.sa_layer.*`` template: SURVEY
        Appendix A13) hold zeros.  Idempotent while the ``.grad`` tensors are the ones already gathered: a further
        ``backward()`` without ``zero_grad()`` (gradient accumulation) changes their identity / version and is gathered
        again, so it cannot be silently ignored."""
        marks = self._grad_marks()
        if self._synced and marks == self._marks:
            return
        dst, src, clear = [], [], []
        for i, p in enumerate(self.params):
            if p.grad is not None:
                dst.append(self._views[i])
                src.append(p.grad)
                self._zero[i] = False
            elif not self._zero[i]:
                clear.append(self._views[i])
                self._zero[i] = True
        if dst:
            torch._foreach_copy_(dst, src)
        if clear:
            torch._foreach_zero_(clear)
        self._synced = True
        self._marks = marks

    def grads(self):
        """The flat gradient vector with the parameters' current ``.grad`` gathered into it -- THE buffer to all-reduce
        or edit before ``step()``.  (``p.grad`` itself stays the local, un-reduced gradient after a collective on this
        buffer; ``step()`` reads the buffer, and re-gathers only when a ``.grad`` changed since.)"""
        self.sync_grads()
        return self.flat_grad

    def step(self, grad_scale=1.0):
        """clip_grad_norm_(max_norm) + Adam (ext_adapt.py:853-855); grad_scale = 1/world after all-reduce."""
        self.sync_grads()
        self.t += 1
        torch.ops.mi355ppo.clip_adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, float(self.max_norm),
                                          float(self.param_groups[0]["lr"]), float(self.betas[0]), float(self.betas[1]),
                                          float(self.eps), self.weight_decay, self.l2, self.t, float(grad_scale),
                                          self.stats)
