"""Flat-buffer optimizer state for modules trained under torch autograd (the student): parameters
and their gradients are views into two flat fp32 vectors, so
  * the data-parallel exchange is ONE in-place all-reduce on ``flat_grad`` (the reference concatenates
    every gradient, all-reduces and copies back: ext_adapt.py:833-851), and
  * clip_grad_norm_ + Adam is one native pass (igi_clip_adam) instead of ~150 foreach launches.
"""
import torch

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)


class FlatAdam:
    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5, weight_decay=0.0, l2=0.0, late=None):
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
        # ``late``: parameters whose gradients become final LAST in a backward pass (the encoders at the bottom of the
        # graph).  They are laid out first, so that the flat gradient is [late | early]: the early range (decoder side)
        # can be handed to the gradient exchange while the encoders' backward still runs (``arm_early``).
        late_ids = {id(p) for p in late} if late is not None else set()
        self.params = [p for p in self.params if id(p) in late_ids] + [p for p in self.params if id(p) not in late_ids]
        self.n_late = sum(1 for p in self.params if id(p) in late_ids)
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
        # bumped by every step(): flat_params.py compares it between a native op's forward and its backward (the flat
        # alias those ops save has a version counter of its own, so autograd cannot see the arena change under it)
        self._epoch = [0]
        for p, sz in zip(self.params, sizes):
            v = self.flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p._igi_arena_epoch = self._epoch
            p.grad = None
            self._views.append(self.flat_grad[off:off + p.numel()].view(p.shape))
            off += sz
        self.late_floats = sum(sizes[:self.n_late])   # flat_grad[:late_floats] = late bucket, the rest = early bucket
        self._early_cb = None
        self._early_order = []
        self._early_done = False
        self._early_live = None                    # early parameters that received a gradient in the last backward
        self._early_expected = None                # ... the set the armed countdown was sized for
        self._early_flushed = None                 # ... the set that HAD a gradient when the bucket last left
        self._hooks = []
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
        self._early_done = False
        if self._early_cb is not None:      # the trigger of the next backward was learned for this set
            self._early_expected = self._early_live

    # ---- early bucket: hand the decoder-side range to the gradient exchange while the encoders' backward runs
    def arm_early(self, callback):
        """``callback(flat_grad[late_floats:])`` is called from INSIDE the next ``backward()`` as soon as every early
        parameter that takes part in the loss has its gradient and the early range of ``flat_grad`` holds them.  The
        first backward after arming runs with a post-accumulate hook on every early parameter and records the order in
        which their gradients arrive (autograd's order for a fixed graph is fixed); from then on ONE hook, on the
        parameter whose gradient arrives last, triggers the hand-over -- thirty Python hook calls per step cost more
        than the collective they hide.  Until one backward has been seen the callback runs from ``sync_grads`` instead
        (same values, no overlap); ``sync_grads`` checks after every backward that the set of early parameters with a
        gradient is the one the trigger was learned for (the never-used ``decoder.sa_layer.*`` template receives none:
        SURVEY Appendix A13) and raises otherwise.  ``None`` disarms."""
        if callback is not None and callback == self._early_cb:
            return                                   # already armed for this exchange: keep what was learned
        self._early_cb = callback
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._early_order = []
        self._early_live = None
        self._early_expected = None
        self._early_done = False
        if callback is None:
            return
        for i in range(self.n_late, len(self.params)):      # learning pass: record the arrival order
            self._hooks.append(self.params[i].register_post_accumulate_grad_hook(
                lambda _p, i=i: self._early_order.append(i)))

    def _arm_trigger(self):
        """after the learning backward: keep one hook, on the early parameter whose gradient arrived last"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._early_order:
            last = self.params[self._early_order[-1]]
            self._hooks.append(last.register_post_accumulate_grad_hook(self._on_last_early_grad))

    def _on_last_early_grad(self, _p):
        if self._early_cb is not None and self._early_expected is not None and not self._early_done:
            self._flush_early()

    def _copy_range(self, lo, hi):
        dst, src, clear = [], [], []
        for i in range(lo, hi):
            p = self.params[i]
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

    def _flush_early(self):
        # which early parameters HAVE a gradient at the moment the bucket leaves: sync_grads compares this with the set
        # that has one after backward -- a gradient that arrives later (autograd's arrival order changed although the
        # set did not: another graph, a second loss term) would otherwise be all-reduced as zeros / stale values
        self._early_flushed = [i for i in range(self.n_late, len(self.params)) if self.params[i].grad is not None]
        self._copy_range(self.n_late, len(self.params))
        self._early_done = True
        self._early_cb(self.flat_grad[self.late_floats:])

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
        if self._early_cb is not None:
            if self._synced:
                raise RuntimeError("FlatAdam: gradients changed after the early bucket went out (gradient accumulation "
                                   "is not supported with arm_early)")
            # the early range may already be with the gradient exchange (reduced in place): it must not be overwritten
            # with the local gradients again.  Not flushed from inside backward (first step, or a parameter set that
            # changed): do it now.
            live = [i for i in range(self.n_late, len(self.params)) if self.params[i].grad is not None]
            if self._early_done and (live != self._early_expected or live != self._early_flushed):
                # the trigger was learned for another set, or for another ARRIVAL ORDER of the same set (the hook fired
                # while gradients were still missing): the bucket left before every gradient was in it
                raise RuntimeError("FlatAdam: the early bucket left before every early gradient was in it (the set of "
                                   "early parameters that receive gradients, or the order in which autograd produces "
                                   "them, changed under an armed early bucket); call arm_early() again after changing "
                                   "the model or the loss")
            if not self._early_done:
                self._flush_early()
            if self._early_live is None:        # this was the learning backward
                self._arm_trigger()
            self._early_live = live
            self._copy_range(0, self.n_late)
        else:
            self._copy_range(0, len(self.params))
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
        self._epoch[0] += 1
        torch.ops.mi355ppo.clip_adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, float(self.max_norm),
                                          float(self.param_groups[0]["lr"]), float(self.betas[0]), float(self.betas[1]),
                                          float(self.eps), self.weight_decay, self.l2, self.t, float(grad_scale),
                                          self.stats)
