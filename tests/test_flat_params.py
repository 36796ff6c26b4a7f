"""flat_params.flat_parameters: the flat parameter vector the native ops take is the parameters' own bytes when they lie
back to back in one storage (FlatAdam's layout), a concatenation otherwise; gradients reach every parameter either way."""
import torch

from isaacgyminsertion_amd.flat_params import _adjacent, flat_parameters


def _params_over(flat, shapes, gaps=()):
    ps, off = [], 0
    for i, s in enumerate(shapes):
        n = int(torch.Size(s).numel())
        p = torch.nn.Parameter(torch.empty(0))
        p.data = flat[off:off + n].view(s)
        ps.append(p)
        off += n + (gaps[i] if i < len(gaps) else 0)
    return ps


def test_adjacent_parameters_are_used_in_place_and_gradients_are_routed():
    flat = torch.randn(64)
    ps = _params_over(flat, [(3, 4), (8,), (2, 2, 3)])
    assert _adjacent(ps) == 32
    f = flat_parameters(ps)
    assert f.data_ptr() == ps[0].data_ptr() and f.shape == (32,) and torch.equal(f, flat[:32])
    w = torch.arange(32.0)
    (f * w).sum().backward()
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in ps]), w)
    assert [p.grad.shape for p in ps] == [p.shape for p in ps]


def test_a_gap_a_foreign_storage_or_another_dtype_falls_back_to_a_copy():
    flat = torch.randn(64)
    gap = _params_over(flat, [(3, 4), (8,)], gaps=(4,))
    assert _adjacent(gap) == 0
    other = [gap[0], torch.nn.Parameter(torch.randn(5))]
    assert _adjacent(other) == 0
    assert _adjacent([torch.nn.Parameter(torch.randn(4).double())]) == 0
    for ps in (gap, other):
        f = flat_parameters(ps)
        assert f.data_ptr() != ps[0].data_ptr()
        assert torch.equal(f, torch.cat([p.detach().reshape(-1) for p in ps]))
        for q in ps:
            q.grad = None
        f.sum().backward()
        assert all(torch.equal(q.grad, torch.ones_like(q)) for q in ps)


def test_an_in_place_update_between_forward_and_backward_is_refused():
    """the zero-copy alias has its own version counter, so autograd cannot see an optimizer step or a
    load_state_dict between a forward and its backward -- the guard (parameter versions + FlatAdam's step epoch) does."""
    import pytest
    flat = torch.randn(64)
    ps = _params_over(flat, [(3, 4), (8,)])
    f = flat_parameters(ps)
    with torch.no_grad():
        ps[1].mul_(2.0)                      # what load_state_dict / a foreign optimizer does
    with pytest.raises(RuntimeError, match="modified in place"):
        f.sum().backward()
    # the arena epoch FlatAdam bumps in step() (its native op writes the arena, not the parameters' counters)
    epoch = [0]
    for p in ps:
        p._igi_arena_epoch = epoch
        p.grad = None
    f = flat_parameters(ps)
    epoch[0] += 1
    with pytest.raises(RuntimeError, match="modified in place"):
        f.sum().backward()
    f = flat_parameters(ps)                  # forward -> backward -> step is fine
    f.sum().backward()
    epoch[0] += 1
    assert all(p.grad is not None for p in ps)
