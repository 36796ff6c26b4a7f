"""CPU-side checks of the op surface (no device needed): every op SURVEY section 8(b) lists is registered in the
``mi355ppo`` namespace with a schema that declares what it mutates, the fake kernels produce the right metadata under
FakeTensorMode, and CPU tensors are refused (there is no CPU implementation to fall back to)."""
import pytest
import torch

from isaacgyminsertion_amd import ops


def test_every_op_is_registered_with_a_schema():
    for name in ops.OP_NAMES:
        packet = getattr(torch.ops.mi355ppo, name)
        schema = str(packet.default._schema)
        assert schema.startswith(f"mi355ppo::{name}("), schema
    s = str(torch.ops.mi355ppo.ppo_update.default._schema)
    assert "Tensor(a!)[] state" in s and "Tensor[] rollout" in s
    assert "Tensor(a!) state" in str(torch.ops.mi355ppo.rms_update_normalize.default._schema)
    assert "Tensor? bias" in str(torch.ops.mi355ppo.linear.default._schema)
    # the names SURVEY section 8(b) spells out
    for name in ("gae_advnorm", "rms_update_normalize", "ppo_minibatch_fwd_bwd", "clip_adam_step", "tactile_cnn_fwd",
                 "tactile_cnn_bwd", "pointnet_max_fwd", "pointnet_max_bwd", "bc_loss_fwd_bwd"):
        assert name in ops.OP_NAMES


def test_fake_kernels_give_output_metadata():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x, w, b = torch.empty(40, 15), torch.empty(64, 15), torch.empty(64)
        y = torch.ops.mi355ppo.linear(x, w, b, 2)
        assert y.shape == (40, 64) and y.dtype == torch.float32
        dx, dw, db = torch.ops.mi355ppo.linear_bwd(x, w, y, torch.empty(40, 64), 2, True, False, False)
        assert dx.shape == (40, 15) and dw.shape == (0, 15) and db.shape == (0,)
        yy, ws = torch.ops.mi355ppo.tactile_cnn_fwd(torch.empty(64, 3, 32, 64), torch.empty(100), 32)
        assert yy.shape == (64, 32) and ws.dtype == torch.uint8 and ws.numel() > 64 * 3 * 32 * 64 * 4
        f, idx = torch.ops.mi355ppo.pointnet_max_fwd(torch.empty(8, 400, 3), torch.empty(16896))
        assert f.shape == (8, 256) and idx.dtype == torch.int32
        loss, dmu = torch.ops.mi355ppo.bc_loss_fwd_bwd(torch.empty(9, 6), torch.empty(9, 6), torch.empty(6), True)
        assert loss.shape == () and dmu.shape == (9, 6)
        n = torch.ops.mi355ppo.rms_update_normalize(torch.empty(5, 15), torch.empty(31, dtype=torch.float64), 1e-5, True, False)
        assert n.shape == (5, 15)
        icfg = [15, 64, 6, 3, 48, 32, 8, 0, 3, 64, 48, 32, 0, 16, 4, 2]
        mu, v, lat = torch.ops.mi355ppo.actor_critic_infer([torch.empty(1)] * 16, icfg, [0.0] * 12, torch.empty(7, 15),
                                                           torch.empty(7, 64), True, True)
        assert mu.shape == (7, 6) and v.shape == (7, 1) and lat.shape == (7, 8)


def test_cpu_tensors_are_refused():
    with pytest.raises(RuntimeError, match="HIP"):
        torch.ops.mi355ppo.linear(torch.zeros(2, 3), torch.zeros(4, 3), None, 0)
    with pytest.raises(RuntimeError, match="HIP"):
        torch.ops.mi355ppo.rms_update_normalize(torch.zeros(2, 3), torch.zeros(7, dtype=torch.float64), 1e-5, False, False)
    with pytest.raises(RuntimeError, match="HIP"):
        torch.ops.mi355ppo.bc_loss(torch.zeros(2, 6), torch.zeros(2, 6), torch.ones(6))
    with pytest.raises(RuntimeError, match="HIP"):
        torch.ops.mi355ppo.pointnet_max_fwd(torch.zeros(2, 5, 3), torch.zeros(16896))


def test_cfg_roundtrip():
    from isaacgyminsertion_amd.teacher_native import make_cfg
    cfg, _ = make_cfg(15, 64, 6, [512, 256, 128], [256, 128, 8], 4096, 32, 8, lr=1e-3)
    ic, fc = ops.pack_cfg(cfg)
    c2 = ops._unpack_cfg(ic, fc)
    for f, _t in cfg._fields_:
        a, b = getattr(cfg, f), getattr(c2, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), f
    with pytest.raises(RuntimeError):
        ops._unpack_cfg(ic[:-1], fc)
