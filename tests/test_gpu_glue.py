"""The student step's native data movement (csrc/glue.h) against ATen's: the one-launch minibatch gather, the token /
encoding concatenation with its one-launch backward, PointNet on a slice of a wider cloud tensor in place, and the flat
parameter vector as a view of FlatAdam's arena.  Copies and one add: everything bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_gather_rows_is_index_select_for_every_arena_in_one_launch():
    from isaacgyminsertion_amd import ops  # noqa: F401
    g = torch.Generator(device=DEV).manual_seed(0)
    total = 1000
    arenas = [torch.randn(total, 3, 32, 64, device=DEV, generator=g), torch.randn(total, 800, 3, device=DEV, generator=g),
              torch.randn(total, 15, device=DEV, generator=g), torch.randn(total, 6, device=DEV, generator=g),
              torch.randn(total, 1, device=DEV, generator=g), torch.randn(total, 260, device=DEV, generator=g)]
    rows = torch.randperm(total, device=DEV, generator=g)[:257]
    outs = torch.ops.mi355ppo.gather_rows(arenas, rows)
    for a, o in zip(arenas, outs):
        assert o.shape == (257,) + a.shape[1:]
        assert torch.equal(o, a.index_select(0, rows))
    # an unaligned arena (a slice starting 4 bytes into an allocation) takes the scalar path
    base = torch.randn(total * 8 + 1, device=DEV, generator=g)
    odd = base[1:].view(total, 8)
    assert torch.equal(torch.ops.mi355ppo.gather_rows([odd], rows)[0], odd.index_select(0, rows))
    # out-of-range row numbers: NaN rows (index_select asserts on the device)
    bad = torch.tensor([0, total, -1, 5], device=DEV)
    o = torch.ops.mi355ppo.gather_rows([arenas[2], arenas[5]], bad)
    for t, a in zip(o, (arenas[2], arenas[5])):
        assert torch.equal(t[0], a[0]) and torch.equal(t[3], a[5]) and torch.isnan(t[1:3]).all()
    with pytest.raises(RuntimeError):
        torch.ops.mi355ppo.gather_rows([arenas[2], torch.randn(total + 1, 4, device=DEV)], rows)
    with pytest.raises(RuntimeError):
        torch.ops.mi355ppo.gather_rows([arenas[2]], rows.to(torch.int32))


@pytest.mark.parametrize("rows,widths", [(2048, (32, 32, 32)), (37, (256, 256)), (5, (1, 7, 64, 3)), (1, (96,))])
def test_cat_cols_and_its_backward_are_torch_cat(rows, widths):
    from isaacgyminsertion_amd import ops  # noqa: F401
    g = torch.Generator(device=DEV).manual_seed(rows)
    parts = [torch.randn(rows, w, device=DEV, generator=g) for w in widths]
    add = torch.randn(sum(widths), device=DEV, generator=g)
    dy = torch.randn(rows, sum(widths), device=DEV, generator=g)
    for use_add in (False, True):
        a = [p.clone().requires_grad_() for p in parts]
        b = [p.clone().requires_grad_() for p in parts]
        y = torch.ops.mi355ppo.cat_cols(a, add if use_add else None)
        ref = torch.cat(b, dim=1)
        if use_add:
            ref = ref + add
        assert torch.equal(y, ref)
        y.backward(dy)
        ref.backward(dy)
        for p, q in zip(a, b):
            assert p.grad.is_contiguous() and torch.equal(p.grad, q.grad)
    outs = torch.ops.mi355ppo.split_cols(dy, list(widths))
    assert all(torch.equal(o, c) for o, c in zip(outs, dy.split(list(widths), dim=1)))


def test_pointnet_reads_a_slice_of_a_wider_cloud_tensor_and_of_a_wider_gradient_in_place():
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    torch.manual_seed(0)
    pn = PointNet().to(DEV)
    with torch.no_grad():
        for p in pn.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    wide = torch.randn(67, 800, 3, device=DEV)
    dwide = torch.randn(67, 512, device=DEV)
    res = []
    for in_place in (False, True):
        pn.zero_grad(set_to_none=True)
        for lo, hi, c0 in ((0, 400, 0), (400, 800, 256), (100, 137, 256)):
            x = wide[:, lo:hi] if in_place else wide[:, lo:hi].contiguous()
            dy = dwide[:, c0:c0 + 256] if in_place else dwide[:, c0:c0 + 256].contiguous()
            assert x.is_contiguous() != in_place
            y = pn(x)
            y.backward(dy)
            res.append((y.detach().clone(), [p.grad.clone() for p in pn.parameters()]))
            pn.zero_grad(set_to_none=True)
    n = len(res) // 2
    for (ya, ga), (yb, gb) in zip(res[:n], res[n:]):
        assert torch.equal(ya, yb)
        assert all(torch.equal(p, q) for p, q in zip(ga, gb))


def test_flat_parameters_of_an_adopted_module_are_the_optimizer_arena_and_train_identically():
    """PointNet + token encoder parameters under FlatAdam: the flat vector the ops get is a view of the arena (no
    concatenation), and a few optimizer steps give bit-identical parameters to the concatenating path."""
    import copy
    import torch.nn as nn
    from isaacgyminsertion_amd import flat_params
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    from isaacgyminsertion_amd.hip_token_encoder import HipTransformerEncoder
    from isaacgyminsertion_amd.optim import FlatAdam

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.pn = PointNet()
            layer = nn.TransformerEncoderLayer(d_model=32, nhead=2, dim_feedforward=128, activation="gelu", batch_first=True,
                                               norm_first=True, dropout=0.0)
            self.enc = HipTransformerEncoder(layer, num_layers=2)

        def forward(self, pts):
            f = self.pn(pts)                                    # (B, 256)
            return self.enc(f.reshape(-1, 8, 32)).sum(dim=(1, 2))

    torch.manual_seed(1)
    net_a = Net().to(DEV)
    net_b = copy.deepcopy(net_a)
    pts = torch.randn(64, 50, 3, device=DEV)
    out = []
    for net, view in ((net_a, True), (net_b, False)):
        opt = FlatAdam(net.parameters(), lr=1e-2)
        fp = net.enc.flat_parameters()
        assert (fp.data_ptr() == next(net.enc.parameters()).data_ptr()) and flat_params._adjacent(list(net.pn.parameters()))
        orig = flat_params._adjacent
        if not view:
            flat_params._adjacent = lambda ps: 0          # force the concatenation
        try:
            for _ in range(3):
                opt.zero_grad()
                (net(pts) ** 2).sum().backward()
                opt.sync_grads()
                opt.step(1.0)
        finally:
            flat_params._adjacent = orig
        out.append(opt.flat.clone())
    assert torch.isfinite(out[0]).all() and torch.equal(out[0], out[1])
