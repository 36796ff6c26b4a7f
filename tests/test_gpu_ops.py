"""torch.ops.mi355ppo.* on the device: torch.library.opcheck (schema vs actual mutation / aliasing, fake kernel vs real
output metadata, autograd registration, AOT dispatch) for every op, and the boundary's argument checks -- dtype,
device, contiguity and shape violations raise RuntimeError before anything reaches the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eng(N=64, T=4, E=2):
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth
    units, priv_units = [64, 48, 32], [48, 32, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, priv_units, seed=3, done_p=0.1)
    eng = TeacherEngine(N, T, E, units=units, priv_units=priv_units, perm=perm, device=DEV)
    eng.load_params(init)
    eng.set_rollout(ro)
    return eng


def _opcheck(op, args, aot=True, **kw):
    # test_aot_dispatch_dynamic traces with symbolic shapes through our ctypes-level int arguments; the static variant
    # covers functionalisation + fake + autograd under AOT.  aot=False: the three encoder forwards hand their
    # workspace (returned, saved for backward) to a backward op that uses it as scratch, i.e. a forward OUTPUT is
    # mutated in the backward graph, which AOT's partitioner rejects ("Node ... was invalid, but is output"); eager
    # autograd -- the way the trainers run them -- has no such restriction, and the backward ops pass all four tests.
    tests = ("test_schema", "test_autograd_registration", "test_faketensor") + (("test_aot_dispatch_static",) if aot else ())
    torch.library.opcheck(op, args, test_utils=tests, **kw)


def test_opcheck_teacher_ops():
    eng = _eng()
    ic, fc = eng._cfg_args()
    o = torch.ops.mi355ppo
    _opcheck(o.gae_advnorm, (eng._ro, eng.state_list(), ic, fc, True))
    _opcheck(o.ppo_minibatch_fwd_bwd, (eng._ro, eng.state_list(), ic, fc, 0, 0, -1))
    _opcheck(o.ppo_clip_adam, (eng.state_list(), ic, fc, 0, 1, 1.0))
    _opcheck(o.ppo_update, (eng._ro, eng.state_list(), ic, fc, 0))
    obs, priv = torch.randn(10, 15, device=DEV), torch.randn(10, 64, device=DEV)
    _opcheck(o.actor_critic_infer, (eng.state_list(), ic, fc, obs, priv, True, True))
    _opcheck(o.actor_critic_infer, (eng.state_list(), ic, fc, obs, priv, False, False))
    z = lambda *s: torch.zeros(*s, device=DEV)   # noqa: E731
    _opcheck(o.rollout_policy_step, (eng.state_list(), ic, fc, obs, priv, True, torch.randn(10, 6, device=DEV),
                                     torch.tensor([0.1, 2.0, 10.0], dtype=torch.float64, device=DEV),
                                     z(10, 15), z(10, 64), z(10, 6), z(10), z(10, 1), z(10, 6), z(10, 6), z(10, 6), z(10, 1)))
    _opcheck(o.rollout_policy_step, (eng.state_list(), ic, fc, obs, priv, False, torch.randn(10, 6, device=DEV), None,
                                     None, None, z(10, 6), z(10), z(10, 1), z(10, 6), z(10, 6), z(10, 6), z(10, 1)))


def test_opcheck_small_ops():
    o = torch.ops.mi355ppo
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)            # noqa: E731
    st = torch.zeros(2 * 15 + 1, dtype=torch.float64, device=DEV)
    st[15:30] = 1
    st[30] = 1
    _opcheck(o.rms_update_normalize, (r(100, 15), st, 1e-5, True, False))
    n = 1000
    _opcheck(o.clip_adam_step, (r(n), r(n), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 0.5, 3e-4, 0.9, 0.999,
                                1e-8, 0.0, 0.0, 1, 1.0, torch.zeros(8, device=DEV)))
    N, A = 32, 6
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)   # noqa: E731
    _opcheck(o.rollout_act_store, (r(N, 15), r(N, 64), r(N, A), r(N, 1), r(A), r(N, A),
                                   torch.tensor([0.1, 2.0, 10.0], dtype=torch.float64, device=DEV), 1e-5,
                                   z(N, 15), z(N, 64), z(N, A), z(N), z(N, 1), z(N, A), z(N, A), z(N, A), z(N, 1)))
    dones = (torch.rand(N, device=DEV, generator=g) < 0.3).to(torch.uint8)
    _opcheck(o.rollout_env_store, (r(N), dones, r(N, 1), dones.clone(), r(N).abs(), 0.99, True,
                                   z(N, 1), z(N, dt=torch.uint8), z(N, 1), z(N), z(N), z(4)))
    w = torch.tensor([1, 1, 0.1, 1, 1, 1.0], device=DEV)
    _opcheck(o.bc_loss_fwd_bwd, (r(50, 6), r(50, 6), w, True))
    _opcheck(o.bc_loss, (r(50, 6).requires_grad_(), r(50, 6), w))
    _opcheck(o.bc_loss_value_grad, (r(50, 6).requires_grad_(), r(50, 6), w))
    c = z(64, 32)
    _opcheck(o.gemm_f32, (True, True, 64, 32, 16, r(64, 16), 16, r(32, 16), 16, c, 32, None, None, 0, 0, False))


def test_opcheck_student_ops():
    o = torch.ops.mi355ppo
    g = torch.Generator(device=DEV).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)            # noqa: E731
    x, w, b = r(40, 15).requires_grad_(), (0.3 * r(64, 15)).requires_grad_(), r(64).requires_grad_()
    _opcheck(o.linear, (x, w, b, 2))
    _opcheck(o.linear, (x, w.detach(), None, 0))
    y = o.linear(x.detach(), w.detach(), b.detach(), 1)
    _opcheck(o.linear_bwd, (x.detach(), w.detach(), y, r(40, 64), 1, True, True, True))
    _opcheck(o.linear_bwd, (x.detach(), w.detach(), y, r(40, 64), 1, True, False, False))
    _opcheck(o.mlp_fwd, (x.detach(), [w.detach(), r(8, 64)], [b.detach(), None], [2, 0]))
    from isaacgyminsertion_amd.algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    cnn = CNNWithSpatialSoftArgmax(32).to(DEV)
    p = cnn.flat_parameters().detach().requires_grad_()
    img = torch.rand(32, 3, 32, 64, device=DEV, generator=g)
    _opcheck(o.tactile_cnn_fwd, (img, p, 32), aot=False)
    yy, ws = o.tactile_cnn_fwd(img, p.detach(), 32)
    _opcheck(o.tactile_cnn_bwd, (r(32, 32), p.detach(), ws, 32, 64))
    pn = PointNet().to(DEV)
    pp = pn.flat_parameters().detach().requires_grad_()
    pts = r(4, 37, 3)
    _opcheck(o.pointnet_max_fwd, (pts, pp))
    f, idx = o.pointnet_max_fwd(pts, pp.detach())
    _opcheck(o.pointnet_max_bwd, (pts, pp.detach(), r(4, 256), idx))
    pq = PointNet().to(DEV).flat_parameters().detach().requires_grad_()
    pts2 = r(4, 60, 3)
    _opcheck(o.pointnet_max_fwd_multi, (pts2, [pp, pq], [37, 23]))
    f2, idx2 = o.pointnet_max_fwd_multi(pts2, [pp.detach(), pq.detach()], [37, 23])
    _opcheck(o.pointnet_max_bwd_multi, (pts2, [pp.detach(), pq.detach()], [37, 23], r(4, 512), idx2))
    _opcheck(o.gather_rows, ([r(10, 3, 4), r(10, 5)], torch.tensor([3, 9, 0, 3], device=DEV)))
    _opcheck(o.cat_cols, ([r(6, 8).requires_grad_(), r(6, 3).requires_grad_()], r(11)))
    _opcheck(o.split_cols, (r(6, 11), [8, 3]))
    from isaacgyminsertion_amd.hip_token_encoder import HipTransformerEncoder
    layer = torch.nn.TransformerEncoderLayer(32, 2, 128, 0.1, activation="gelu", batch_first=True, norm_first=True)
    enc = HipTransformerEncoder(layer, 2).to(DEV)
    tp = enc.flat_parameters().detach().requires_grad_()
    tok = r(16, 3, 32).requires_grad_()
    _opcheck(o.token_encoder_fwd, (tok, tp, 2, 128, 2, 0.0, False, 0), aot=False)
    ty, tws = o.token_encoder_fwd(tok.detach(), tp.detach(), 2, 128, 2, 0.1, True, 1234)
    _opcheck(o.token_encoder_bwd, (r(16, 3, 32), tp.detach(), tws, 2, 128, 2, 0.1, True, 1234))


def test_opcheck_depth_backbone():
    from isaacgyminsertion_amd.algo.models.transformer.depth_backbone import DepthOnlyFCBackbone54x96
    o = torch.ops.mi355ppo
    m = DepthOnlyFCBackbone54x96(32).to(DEV)
    p = m.flat_parameters().detach().requires_grad_()
    x = torch.rand(32, 1, 54, 96, device=DEV)
    _opcheck(o.depth_backbone_fwd, (x, p, 32), aot=False)
    y, ws = o.depth_backbone_fwd(x, p.detach(), 32)
    _opcheck(o.depth_backbone_bwd, (x, torch.randn(32, 32, device=DEV), p.detach(), ws))


def test_argument_checks_raise_runtime_error():
    o = torch.ops.mi355ppo
    eng = _eng()
    ic, fc = eng._cfg_args()
    x, w = torch.randn(8, 15, device=DEV), torch.randn(4, 15, device=DEV)
    with pytest.raises(RuntimeError, match="dtype"):
        o.linear(x, w.double(), None, 0)
    with pytest.raises(RuntimeError, match="HIP"):
        o.linear(x.cpu(), w.cpu(), None, 0)
    with pytest.raises(RuntimeError, match="contiguous"):
        o.linear(x, torch.randn(15, 4, device=DEV).t(), None, 0)
    with pytest.raises(RuntimeError, match="shape"):
        o.linear(x, torch.randn(4, 16, device=DEV), None, 0)
    with pytest.raises(RuntimeError, match="shape"):
        o.rms_update_normalize(x, torch.zeros(7, dtype=torch.float64, device=DEV), 1e-5, True, False)
    with pytest.raises(RuntimeError, match="dtype"):
        o.rms_update_normalize(x, torch.zeros(31, device=DEV), 1e-5, True, False)
    st = eng.state_list()
    bad = list(st)
    bad[0] = st[0][:-4]                                          # params too short
    with pytest.raises(RuntimeError, match="state.params"):
        o.ppo_update(eng._ro, bad, ic, fc, 0)
    bad = list(st)
    bad[7] = st[7].to(torch.int32)                               # perm dtype
    with pytest.raises(RuntimeError, match="state.perm"):
        o.ppo_update(eng._ro, bad, ic, fc, 0)
    ro = list(eng._ro)
    ro[5] = ro[5].float()                                        # dones must be uint8
    with pytest.raises(RuntimeError, match="rollout.dones"):
        o.gae_advnorm(ro, st, ic, fc, True)
    ro = list(eng._ro)
    ro[0] = ro[0].cpu()
    with pytest.raises(RuntimeError, match="rollout.obses"):
        o.gae_advnorm(ro, st, ic, fc, True)
    with pytest.raises(RuntimeError, match="workspace"):
        bad = list(st)
        bad[15] = st[15][:1024]
        o.ppo_update(eng._ro, bad, ic, fc, 0)
    with pytest.raises(RuntimeError, match="32k"):
        o.tactile_cnn_fwd(torch.rand(5, 3, 32, 64, device=DEV), torch.zeros(10, device=DEV), 32)
    with pytest.raises(RuntimeError, match="reducer"):
        o.ppo_update_dp(eng._ro, st, ic, fc, 0, 0.5, 987654)
    with pytest.raises(RuntimeError, match="1-based"):
        n = 16
        z = torch.zeros(n, device=DEV)
        o.clip_adam_step(z, z.clone(), z.clone(), z.clone(), 0.5, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.0, 0, 1.0,
                         torch.zeros(8, device=DEV))


def test_update_dp_callback_errors_propagate():
    from isaacgyminsertion_amd import ops
    eng = _eng()
    eng.prepare()

    def boom(bucket, step):
        if step == 1 and bucket == 1:
            raise ValueError("collective failed")

    h = ops.register_reducer(boom)
    try:
        with pytest.raises(ValueError, match="collective failed"):
            torch.ops.mi355ppo.ppo_update_dp(eng._ro, eng.state_list(), *eng._cfg_args(), 0, 0.5, h)
    finally:
        ops.unregister_reducer(h)
    torch.cuda.synchronize()
