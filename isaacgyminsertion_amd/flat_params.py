"""The flat parameter vector a native op takes (``tactile_cnn_fwd``, ``pointnet_max_fwd``, ``token_encoder_fwd`` ...),
without a copy when the parameters already lie back to back in one storage -- which is how ``optim.FlatAdam`` lays them
out (module order, 16-byte aligned slices).  ``torch.cat`` of the parameters cost one launch per module and optimizer step
forward (CatArrayBatchedCopy: four per student step), for bytes that were already where the kernel wants them.

``flat_parameters(params)`` is differentiable: the gradient of the flat vector is handed to the parameters as views of
it (no kernels either way).  Parameters that are not adjacent (before ``FlatAdam`` adopted them, or under another
optimizer) are concatenated as before.

Ordering requirement of the zero-copy path: forward -> backward -> optimizer step.  The alias handed to the native ops
shares the arena's BYTES but not its autograd version counter, so an in-place update between a forward and its backward
would go unnoticed by autograd and the backward would use the updated weights (the ``torch.cat`` path saved a snapshot).
``_FlatParams`` therefore records the parameters' version counters and the arena's step epoch (``FlatAdam._epoch``) at
forward and raises in backward when either has moved.
"""
import torch


def _adjacent(ps):
    """total number of elements when ``ps`` are contiguous fp32 tensors lying back to back in one storage, else 0"""
    p0 = ps[0]
    base = p0.untyped_storage().data_ptr()
    off = p0.storage_offset()
    for p in ps:
        if p.dtype is not torch.float32 or not p.is_contiguous() or p.storage_offset() != off or \
                p.untyped_storage().data_ptr() != base:
            return 0
        off += p.numel()
    return off - p0.storage_offset()


class _FlatParams(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ps):
        ctx.shapes = [p.shape for p in ps]
        total = _adjacent(ps)
        ctx.guard = None
        if total:
            epoch = getattr(ps[0], "_igi_arena_epoch", None)
            ctx.guard = (ps, tuple(p._version for p in ps), epoch, epoch[0] if epoch is not None else None)
            # a fresh tensor over the same bytes (not a view of ps[0] in autograd's eyes: nothing is written through it)
            return torch.empty(0, dtype=torch.float32, device=ps[0].device).set_(
                ps[0].untyped_storage(), ps[0].storage_offset(), (total,), (1,))
        return torch.cat([p.reshape(-1) for p in ps])

    @staticmethod
    def backward(ctx, g):
        if ctx.guard is not None:
            ps, versions, epoch, at = ctx.guard
            if tuple(p._version for p in ps) != versions or (epoch is not None and epoch[0] != at):
                raise RuntimeError("flat_parameters: the parameters were modified in place (optimizer step / load_state_dict) "
                                   "between this forward and its backward; the zero-copy flat view aliases the optimizer "
                                   "arena, so backward would use the NEW weights -- run forward -> backward -> step")
        out, o = [], 0
        for s in ctx.shapes:
            n = s.numel()
            out.append(g[o:o + n].view(s))
            o += n
        return tuple(out)


def flat_parameters(params):
    ps = list(params)
    return _FlatParams.apply(*ps)
