"""igi_linear_forward / igi_linear_backward (HipLinear) and igi_clip_adamw against PyTorch fp32 on the
same inputs.  Tolerances: the GEMMs are exact-fp32 fmaf chains, so forward differs from ATen only by
summation order (<= 2e-6 * sqrt(K) relative to the row scale); tanh uses the 1.5e-7-accurate fast form."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, act):
    y = torch.nn.functional.linear(x, w, b)
    return torch.tanh(y) if act == "tanh" else (torch.relu(y) if act == "relu" else y)


@pytest.mark.parametrize("rows,inf,outf,act,bias", [
    (64, 15, 64, "relu", True), (64, 64, 32, None, True), (64, 32, 6, "tanh", True),
    (1, 15, 64, "relu", True), (3, 512, 64, "relu", True), (8192, 96, 32, "relu", True),
    (8192, 32, 256, "relu", True), (2048, 256, 128, "relu", True), (1000, 33, 7, None, False),
    (16385, 64, 32, "tanh", True),
])
def test_linear_matches_torch(rows, inf, outf, act, bias):
    from isaacgyminsertion_amd.hip_linear import linear
    g = torch.Generator().manual_seed(rows + inf)
    x = torch.randn(rows, inf, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(outf, inf, generator=g) / inf ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(outf, generator=g) * 0.1).cuda().requires_grad_(True) if bias else None
    dy = torch.randn(rows, outf, generator=g).cuda()
    y = linear(x, w, b, act)
    y.backward(dy)
    got = [y.detach(), x.grad.clone(), w.grad.clone()] + ([b.grad.clone()] if bias else [])
    x.grad = w.grad = None
    if bias:
        b.grad = None
    yr = _ref(x.double(), w.double(), b.double() if bias else None, act)
    yr.backward(dy.double())
    want = [yr.detach(), x.grad, w.grad] + ([b.grad] if bias else [])
    for name, a, r in zip(("y", "dx", "dw", "db"), got, want):
        scale = r.abs().max().item() + 1e-12
        err = (a.double() - r.double()).abs().max().item()
        # fp32 accumulation over K (forward/dx) or rows (dw/db) terms vs an fp64 reference
        assert err <= 3e-6 * scale * max(1.0, (rows if name in ("dw", "db") else inf) ** 0.5) + 1e-7, (name, err, scale)


def test_linear_3d_input_and_no_input_grad():
    from isaacgyminsertion_amd.hip_linear import HipLinear
    torch.manual_seed(0)
    m = HipLinear(15, 64, act="relu").cuda()
    x = torch.randn(32, 1, 15, device="cuda")
    y = m(x)
    assert y.shape == (32, 1, 64)
    y.sum().backward()
    ref = torch.relu(torch.nn.functional.linear(x, m.weight, m.bias))
    assert torch.allclose(y, ref, atol=1e-5)
    assert m.weight.grad is not None and m.bias.grad.shape == (64,)


def test_linear_backward_is_deterministic():
    from isaacgyminsertion_amd.hip_linear import linear
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8192, 96, generator=g).cuda()
    w = torch.randn(32, 96, generator=g).cuda().requires_grad_(True)
    b = torch.zeros(32).cuda().requires_grad_(True)
    outs = []
    for _ in range(2):
        w.grad = b.grad = None
        linear(x, w, b, "relu").square().sum().backward()
        outs.append((w.grad.clone(), b.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_adamw_matches_torch():
    from isaacgyminsertion_amd.optim import FlatAdam
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(37, 5).cuda()), torch.nn.Parameter(torch.randn(11).cuda())]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FlatAdam(ps, lr=1e-2, max_norm=0.5, weight_decay=1e-2)
    ref = torch.optim.AdamW(rs, lr=1e-2, weight_decay=1e-2)
    for it in range(5):
        opt.zero_grad()
        for p, r in zip(ps, rs):
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(it * 7 + p.numel())).cuda()
            p.grad = gr.clone()          # zero_grad() sets the gradients to None; autograd (here: the test) stores them
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(rs, 0.5)
        ref.step()
        opt.step()
    for p, r in zip(ps, rs):
        assert torch.allclose(p, r, atol=2e-6, rtol=1e-6), (p - r).abs().max()


def test_adam_with_coupled_l2_matches_torch():
    """torch.optim.Adam(weight_decay=wd) adds wd * param to the (already clipped) gradient: FlatAdam(l2=wd)."""
    from isaacgyminsertion_amd.optim import FlatAdam
    torch.manual_seed(4)
    ps = [torch.nn.Parameter(torch.randn(29, 7).cuda()), torch.nn.Parameter(torch.randn(13).cuda())]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FlatAdam(ps, lr=1e-3, max_norm=0.5, l2=1e-2)
    ref = torch.optim.Adam(rs, lr=1e-3, weight_decay=1e-2)
    for it in range(5):
        opt.zero_grad()
        for p, r in zip(ps, rs):
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(it * 5 + p.numel())).cuda()
            p.grad = gr.clone()          # zero_grad() sets the gradients to None; autograd (here: the test) stores them
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(rs, 0.5)
        ref.step()
        opt.step()
    for p, r in zip(ps, rs):
        assert torch.allclose(p, r, atol=2e-6, rtol=1e-6), (p - r).abs().max()


@pytest.mark.parametrize("rows,dims,acts,need_dx,frozen", [
    (2048, [15, 64, 32], ["relu", None], False, ()),                          # lin encoder (tact.py:337-339)
    (2048, [512, 64, 32], ["relu", None], True, ()),                          # point-cloud compress (tact.py:367-369)
    (8192, [96, 32, 256, 128, 64, 32, 6], ["relu"] * 5 + ["tanh"], True, ()),  # decoder output stack + head
    (1000, [64, 256, 128, 64, 32, 6], ["relu"] * 4 + [None], True, (1,)),      # MLPDecoder + head, ragged rows, one frozen layer
    (64, [32, 6], ["tanh"], True, ()),                                         # a single layer takes the plain path
])
def test_mlp_chain_is_bitwise_the_per_layer_path(rows, dims, acts, need_dx, frozen):
    """hip_linear.mlp_chain (one autograd node, igi_mlp_backward: one grid per layer + one sum of all split partials)
    against the same HipLinear modules applied one by one under per-layer autograd (igi_linear_backward: four launches
    per layer): outputs and every gradient bit for bit -- same products, same k-order, same split factors, same order
    of additions."""
    from isaacgyminsertion_amd import _lib
    from isaacgyminsertion_amd.hip_linear import HipLinear, mlp_chain
    torch.manual_seed(rows + len(dims))
    layers = [HipLinear(i, o, act=a).cuda() for i, o, a in zip(dims[:-1], dims[1:], acts)]
    for l in frozen:
        for p in layers[l].parameters():
            p.requires_grad_(False)
    x = torch.randn(rows, dims[0], device="cuda", requires_grad=need_dx)
    dy = torch.randn(rows, dims[-1], device="cuda")

    def run(chain):
        for m in layers:
            m.zero_grad(set_to_none=True)
        x.grad = None
        if chain:
            y = mlp_chain(x, layers)
        else:
            y = x
            for m in layers:
                y = m(y)
        y.backward(dy)
        torch.cuda.synchronize()
        return [y.detach().clone(), None if x.grad is None else x.grad.clone()] + \
            [None if p.grad is None else p.grad.clone() for m in layers for p in m.parameters()]

    _lib.prof_enable(True)
    a = run(True)
    launches_chain = sum(c["launches"] for c in _lib.prof_read())
    b = run(False)
    launches_layers = sum(c["launches"] for c in _lib.prof_read())
    _lib.prof_enable(False)
    for u, v in zip(a, b):
        assert (u is None) == (v is None)
        if u is not None:
            assert torch.equal(u, v)
    assert a[0].isfinite().all() and any(g is not None and g.abs().max() > 0 for g in a[2:])
    if len(layers) > 1:
        assert launches_chain < launches_layers, (launches_chain, launches_layers)


def test_flat_adam_early_bucket_protocol():
    """optim.FlatAdam.arm_early: layout [late | early]; the first backward after arming hands the early range over from
    sync_grads (learning pass), every later one from INSIDE backward -- after the last early gradient, before the late
    parameters have theirs --; a parameter without a gradient keeps a zero slice; accumulating into an exchanged bucket
    and a changed set of live early parameters both raise instead of sending a wrong bucket."""
    from isaacgyminsertion_amd.optim import FlatAdam
    torch.manual_seed(5)
    enc = torch.nn.Linear(12, 16).cuda()          # bottom of the graph: its gradients arrive last
    dec = torch.nn.Linear(16, 4).cuda()
    unused = torch.nn.Linear(3, 3).cuda()         # never takes part in the loss (decoder.sa_layer.* in the student)
    params = list(dec.parameters()) + list(unused.parameters()) + list(enc.parameters())
    opt = FlatAdam(params, lr=1e-3, late=list(enc.parameters()))
    assert opt.n_late == 2 and opt.late_floats == 12 * 16 + 16
    assert [p.data_ptr() for p in opt.params[:2]] == [p.data_ptr() for p in enc.parameters()]
    seen = []

    def cb(bucket):
        # called with the early range; records whether the encoders already had their gradients at that moment
        assert bucket.data_ptr() == opt.flat_grad[opt.late_floats:].data_ptr() and bucket.numel() == opt.flat_grad.numel() - opt.late_floats
        seen.append((enc.weight.grad is not None, bucket.clone()))

    opt.arm_early(cb)
    x = torch.randn(32, 12, device="cuda")

    def backward():
        opt.zero_grad()
        dec(torch.tanh(enc(x))).square().sum().backward()

    backward()
    assert seen == []                              # learning pass: nothing leaves from inside backward
    g = opt.grads()
    assert len(seen) == 1 and seen[0][0] is True   # ... the bucket went out of sync_grads, after backward
    backward()
    assert len(seen) == 2 and seen[1][0] is False  # armed: handed over while the encoder's backward had not run yet
    g = opt.grads()
    assert len(seen) == 2                          # and not a second time
    for p, v in zip(opt.params, opt._views):
        if p.grad is None:
            assert not v.any()                     # the unused layer's slices hold zeros
        else:
            assert torch.equal(v, p.grad)
    assert torch.equal(seen[1][1], g[opt.late_floats:])
    assert torch.equal(seen[0][1], seen[1][1])     # same data, same graph: same bucket either way
    before = opt.flat.clone()
    opt.step()
    assert not torch.equal(before, opt.flat)
    # gradient accumulation into a bucket that already left: refused
    backward()
    opt.grads()
    dec(torch.tanh(enc(x))).square().sum().backward()
    with pytest.raises(RuntimeError, match="gradient accumulation"):
        opt.grads()
    # a changed set of live early parameters under the armed trigger: refused, re-arming learns the new set
    opt.zero_grad()
    (dec(torch.tanh(enc(x))).square().sum() + unused(torch.ones(2, 3, device="cuda")).sum()).backward()
    with pytest.raises(RuntimeError, match="changed under an armed"):
        opt.grads()
    opt.arm_early(None)
    opt.arm_early(cb)
    n = len(seen)
    opt.zero_grad()
    (dec(torch.tanh(enc(x))).square().sum() + unused(torch.ones(2, 3, device="cuda")).sum()).backward()
    opt.grads()
    assert len(seen) == n + 1 and all(torch.equal(v, p.grad) for p, v in zip(opt.params, opt._views))


def test_flat_adam_early_bucket_refuses_a_changed_arrival_order():
    """Same early parameter SET, another order in which autograd produces their gradients (another graph over the same
    modules): the one trigger hook then fires while gradients are still missing.  sync_grads must notice that the bucket
    left incomplete and raise -- not hand zero / stale slices to clip + Adam."""
    from isaacgyminsertion_amd.optim import FlatAdam
    torch.manual_seed(6)
    enc = torch.nn.Linear(8, 8).cuda()
    a, b = torch.nn.Linear(8, 4).cuda(), torch.nn.Linear(8, 4).cuda()
    opt = FlatAdam(list(a.parameters()) + list(b.parameters()) + list(enc.parameters()), lr=1e-3, late=list(enc.parameters()))
    sent = []
    opt.arm_early(lambda bucket: sent.append(bucket.clone()))
    x = torch.randn(16, 8, device="cuda")

    def graph1():            # b's node is created last: its gradients arrive first, a's last
        h = torch.tanh(enc(x))
        return a(h).square().sum() + b(h).square().sum()

    def graph2():            # the other way round
        h = torch.tanh(enc(x))
        u = b(h).square().sum()
        return a(h).square().sum() * 1.0 + u

    for _ in range(2):       # learning pass, then one armed pass: fine
        opt.zero_grad()
        graph1().backward()
        opt.grads()
    assert len(sent) == 2
    order1 = list(opt._early_order)
    opt.zero_grad()
    graph2().backward()
    if opt._early_flushed == [i for i in range(opt.n_late, len(opt.params)) if opt.params[i].grad is not None]:
        pytest.skip("autograd produced the same arrival order for both graphs on this build")
    with pytest.raises(RuntimeError, match="left before every early gradient"):
        opt.grads()
    assert order1                                   # (the learned order existed)


@pytest.mark.parametrize("rows,dims,acts,view", [
    (2048, [15, 64, 32], [2, 0], "dense"),
    (333, [96, 32, 256, 128, 64, 32, 6], [2, 2, 2, 2, 2, 1], "dense"),        # ragged last row block
    (4096, [512, 64, 32], [2, 0], "dense"),
    (640, [33, 40, 7], [1, 0], "dense"),                                        # nothing aligned: generic order, scalar loads
    (512, [64, 128, 32], [2, 2], "strided"),                                    # x = a column slice of a wider tensor (ld 80)
    (512, [60, 128, 32], [2, 2], "offset"),                                     # ... starting 4 bytes off a 16-byte boundary
    (31, [32, 32], [1], "dense"),                                               # fewer rows than one block
])
def test_mlp_fwd_is_bitwise_the_per_layer_launches(rows, dims, acts, view):
    """torch.ops.mi355ppo.mlp_fwd (igi_mlp_forward: the whole chain in one launch, csrc/mlp_fwd.h) against one
    torch.ops.mi355ppo.linear launch per layer (igi_linear_forward), every layer's output with torch.equal -- aligned
    K % 32 == 0 layers replay the LDS-DMA kernel's k order, the others the generic kernel's; ragged rows, an input that
    is a strided / misaligned view, widths that are no multiples of four."""
    g = torch.Generator().manual_seed(rows + sum(dims))
    if view == "dense":
        x = torch.randn(rows, dims[0], generator=g).cuda()
    elif view == "strided":
        x = torch.randn(rows, dims[0] + 16, generator=g).cuda()[:, 8:8 + dims[0]]
    else:
        x = torch.randn(rows, dims[0] + 5, generator=g).cuda()[:, 1:1 + dims[0]]
    ws = [(torch.randn(o, i, generator=g) / i ** 0.5).cuda() for i, o in zip(dims[:-1], dims[1:])]
    bs = [torch.randn(o, generator=g).cuda() * 0.1 for o in dims[1:]]
    ys = torch.ops.mi355ppo.mlp_fwd(x, ws, bs, acts)
    h = x
    for l, (w, b, a) in enumerate(zip(ws, bs, acts)):
        h = torch.ops.mi355ppo.linear(h, w, b, a)
        assert torch.equal(ys[l], h), (l, float((ys[l] - h).abs().max()))
    ref = x.double().cpu()
    for w, b, a in zip(ws, bs, acts):
        ref = ref @ w.double().cpu().t() + b.double().cpu()
        ref = torch.tanh(ref) if a == 1 else (torch.relu(ref) if a == 2 else ref)
    assert torch.allclose(ys[-1].double().cpu(), ref, atol=1e-5, rtol=1e-5)


def test_mlp_fwd_runs_chains_the_one_launch_kernel_does_not_take_layer_by_layer():
    """igi_mlp_forward answers IGI_E_UNSUPPORTED for a layer wider than 256 outputs, for more than 40 64-wide k-chunks in all
    and in the bf16-input mode; its contract is "run the layers one by one" and since round 6 the op does exactly that
    (the Python side used to raise, which crashed a student forward under IGI_GEMM_BF16=1 or with a wide first
    layer): same results as one ``linear`` per layer, bit for bit."""
    from isaacgyminsertion_amd import _lib
    g = torch.Generator().manual_seed(3)
    cases = [([32, 512], [0]),                     # a layer wider than 256 outputs
             ([2600, 64, 32], [2, 0])]             # 41 + 1 k-chunks: more than the kernel's step table holds
    for dims, acts in cases:
        x = torch.randn(64, dims[0], generator=g).cuda()
        ws = [(torch.randn(o, i, generator=g) / i ** 0.5).cuda() for i, o in zip(dims[:-1], dims[1:])]
        bs = [torch.randn(o, generator=g).cuda() * 0.1 for o in dims[1:]]
        ys = torch.ops.mi355ppo.mlp_fwd(x, ws, bs, acts)
        h = x
        for l, (w, b, a) in enumerate(zip(ws, bs, acts)):
            h = torch.ops.mi355ppo.linear(h, w, b, a)
            assert torch.equal(ys[l], h), (dims, l)
    # the bf16-input mode: the op must not raise (results are that mode's, compared with the per-layer launches of the same mode)
    x = torch.randn(256, 64, generator=g).cuda()
    w = (torch.randn(32, 64, generator=g) / 8).cuda()
    b = torch.zeros(32, device="cuda")
    prev = _lib.lib().igi_gemm_set_bf16_inputs(1)
    try:
        y = torch.ops.mi355ppo.mlp_fwd(x, [w], [b], [1])[0]
        ref = torch.ops.mi355ppo.linear(x, w, b, 1)
    finally:
        _lib.lib().igi_gemm_set_bf16_inputs(prev)
    assert torch.equal(y, ref)
