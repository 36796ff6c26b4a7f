"""Edge cases of the hot path against the CPU oracle: ragged sizes (nothing a multiple of a tile),
single-step horizons, every-env-done rollouts, and the stand-alone RunningMeanStd op in all its modes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,T,E,units,priv_units", [
    (37, 5, 5, [70, 36, 20], [30, 12, 8]),     # mb = 37: no dimension is a multiple of 4 / 32 / 128
    (24, 1, 2, [64, 32], [16, 8]),             # horizon 1 (GAE degenerates to one step), 2-layer nets
    (130, 3, 3, [33, 17, 9], [9, 8]),          # odd widths everywhere
])
def test_ragged_configs_match_oracle(N, T, E, units, priv_units):
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    init, ro, perm = synth.teacher_problem(N, T, units, priv_units, seed=3, done_p=0.2)
    eng = TeacherEngine(N, T, E, units=units, priv_units=priv_units, perm=perm)
    eng.load_params(init)
    orc = ot.TeacherOracle(init, perm, N, T, E, units, priv_units)
    d = orc.prepare(ro)
    eng.prepare(ro)
    torch.cuda.synchronize()
    assert torch.equal(eng.returns_raw.cpu(), orc.returns_raw)
    np.testing.assert_allclose(eng.env_major(eng.advantages).cpu().numpy(), d["advantages"].numpy(), atol=5e-5)
    st = orc.update(record_grads=1)
    eng.fwd_bwd(0, 0)
    torch.cuda.synchronize()
    ref = st["grads"][0].numpy()
    np.testing.assert_allclose(eng.packed(eng.grads).cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)
    eng.apply(0)
    slot = 1
    for e in range(E):
        for i in range(eng.n_mb):
            if e == 0 and i == 0:
                continue
            eng.fwd_bwd(i, slot)
            eng.apply(slot)
            slot += 1
    torch.cuda.synchronize()
    s = eng.stats.cpu().numpy()
    for j, nm in enumerate(["a_losses", "c_losses", "b_losses", "entropies"]):
        np.testing.assert_allclose(s[:slot, j], np.array([x.item() for x in st[nm]]), rtol=2e-4, atol=2e-6, err_msg=nm)
    np.testing.assert_allclose(eng.packed().cpu().numpy(), orc.flat_params().numpy(), atol=slot * 2.5e-4 * 0.05)


@pytest.mark.parametrize("obs_dim,priv_dim,act_dim", [(11, 20, 3), (15, 64, 8), (9, 33, 7)])
def test_other_observation_and_action_widths(obs_dim, priv_dim, act_dim):
    """Input / action widths other than the task's 15 / 64 / 6: act <= 7 runs the packed loss kernel (the scalar
    section once per four rows), act = 8 the one-row-at-a-time kernel with its separate wave sums."""
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    N, T, E, units, pu = 96, 4, 3, [48, 40, 24], [24, 16, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, pu, obs_dim=obs_dim, priv_dim=priv_dim, act_dim=act_dim,
                                           seed=5, done_p=0.1)
    eng = TeacherEngine(N, T, E, units=units, priv_units=pu, perm=perm, obs_dim=obs_dim, priv_dim=priv_dim,
                        act_dim=act_dim)
    eng.load_params(init)
    orc = ot.TeacherOracle(init, perm, N, T, E, units, pu, obs_dim=obs_dim, priv_dim=priv_dim, act_dim=act_dim)
    orc.prepare(ro)
    eng.prepare(ro)
    st = orc.update(record_grads=1)
    eng.fwd_bwd(0, 0)
    torch.cuda.synchronize()
    ref = st["grads"][0].numpy()
    np.testing.assert_allclose(eng.packed(eng.grads).cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)
    eng.apply(0)
    slot = 1
    for e in range(E):
        for i in range(eng.n_mb):
            if e == 0 and i == 0:
                continue
            eng.fwd_bwd(i, slot)
            eng.apply(slot)
            slot += 1
    torch.cuda.synchronize()
    s = eng.stats.cpu().numpy()
    for j, nm in enumerate(["a_losses", "c_losses", "b_losses", "entropies"]):
        np.testing.assert_allclose(s[:slot, j], np.array([x.item() for x in st[nm]]), rtol=2e-4, atol=2e-6, err_msg=nm)
    np.testing.assert_allclose(eng.packed().cpu().numpy(), orc.flat_params().numpy(), atol=slot * 2.5e-4 * 0.05)
    # update_mu_sigma: the scattered policy means / sigmas of the last pass
    np.testing.assert_allclose(eng.env_major(eng.mus_w).cpu().numpy(), orc.data["mus"].detach().numpy(), atol=2e-5)


def test_first_trunk_layer_with_a_full_32_wide_input():
    """obs_dim + latent == 32: the padded 32-wide xcat has no free column, so column 31 is a REAL input.  The fused
    first-layer weight gradient (gemm_dma.h kind 6) puts the ONE of the bias gradient in lane 31 and must therefore
    decline this shape (plan predicate xw < 32, re-checked at the launch: ADVICE round 5); the separate weight-gradient
    launch runs instead and dW1[:, 31] of both nets matches the oracle."""
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    N, T, E, units, pu, obs_dim = 128, 8, 2, [512, 256, 128], [256, 128, 8], 24    # mb = 512: whole 128-row tiles
    init, ro, perm = synth.teacher_problem(N, T, units, pu, obs_dim=obs_dim, seed=17, done_p=0.1)
    eng = TeacherEngine(N, T, E, units=units, priv_units=pu, perm=perm, obs_dim=obs_dim)
    eng.load_params(init)
    orc = ot.TeacherOracle(init, perm, N, T, E, units, pu, obs_dim=obs_dim)
    orc.prepare(ro)
    eng.prepare(ro)
    st = orc.update(record_grads=1, max_steps=1)
    eng.fwd_bwd(0, 0)
    torch.cuda.synchronize()
    ref = st["grads"][0].numpy()
    got = eng.packed(eng.grads).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)
    # the last input column of both nets' first trunk layer, explicitly (it came out as zeros from the fused tiles)
    names = list(init.keys())
    off = 0
    for k in names:
        n = init[k].numel()
        if k in ("actor_mlp.mlp.0.weight", "critic_mlp.mlp.0.weight"):
            g = got[off:off + n].reshape(512, 32)[:, 31]
            r = ref[off:off + n].reshape(512, 32)[:, 31]
            assert np.abs(r).max() > 0
            np.testing.assert_allclose(g, r, atol=2e-4 * np.abs(ref).max(), rtol=2e-3, err_msg=k)
        off += n


@pytest.mark.parametrize("N,T,E,act_dim,units", [
    (100, 3, 2, 3, [64, 32, 128]),      # mb = 150: two full 64-row tiles + a 22-row one; 3 actions
    (50, 5, 5, 7, [40, 96, 128]),       # mb = 50: ONE partial tile per net; 7 actions (the widest the fused kernel takes)
    (257, 2, 2, 6, [32, 128]),          # mb = 257: 4 tiles + one row; two-layer trunk
])
def test_fused_last_layer_and_loss_on_ragged_minibatches(N, T, E, act_dim, units):
    """k_trunk_loss (last trunk layer 128 wide: forward + heads + PPO loss + head backward in one launch) on minibatches
    that are not a multiple of its 64-row tiles and with action counts other than the task's six, against the oracle --
    and a check through the profiler that it is that kernel which ran."""
    from isaacgyminsertion_amd import _lib
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    pu = [24, 16, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, pu, act_dim=act_dim, seed=11, done_p=0.1)
    eng = TeacherEngine(N, T, E, units=units, priv_units=pu, perm=perm, act_dim=act_dim)
    eng.load_params(init)
    orc = ot.TeacherOracle(init, perm, N, T, E, units, pu, act_dim=act_dim)
    orc.prepare(ro)
    eng.prepare(ro)
    st = orc.update(record_grads=1)
    _lib.prof_enable(True)
    try:
        eng.fwd_bwd(0, 0)
        torch.cuda.synchronize()
        classes = {c["name"]: c["launches"] for c in _lib.prof_read()}
    finally:
        _lib.prof_enable(False)
    assert classes.get("k_trunk_loss", 0) == 1, classes
    ref = st["grads"][0].numpy()
    np.testing.assert_allclose(eng.packed(eng.grads).cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)
    eng.apply(0)
    slot = 1
    for e in range(E):
        for i in range(eng.n_mb):
            if e == 0 and i == 0:
                continue
            eng.fwd_bwd(i, slot)
            eng.apply(slot)
            slot += 1
    torch.cuda.synchronize()
    s = eng.stats.cpu().numpy()
    for j, nm in enumerate(["a_losses", "c_losses", "b_losses", "entropies"]):
        np.testing.assert_allclose(s[:slot, j], np.array([x.item() for x in st[nm]]), rtol=2e-4, atol=2e-6, err_msg=nm)
    np.testing.assert_allclose(eng.packed().cpu().numpy(), orc.flat_params().numpy(), atol=slot * 2.5e-4 * 0.05)
    np.testing.assert_allclose(eng.env_major(eng.mus_w).cpu().numpy(), orc.data["mus"].detach().numpy(), atol=2e-5)


@pytest.mark.parametrize("N,T,E,rb", [
    (1024, 8, 4, True),     # mb = 2048: row-block levels + both first-layer weight gradients from the tiles of the level above
    (96, 8, 4, False),      # mb = 192: not a multiple of 128 / below the row-block kernel's size: the tile kernels for everything
    (64, 8, 4, False),      # mb = 128: one whole 128-row tile (first trunk layer's weight gradient fused, chain of one), below the row-block kernel's 256 rows
])
def test_backward_kernel_selection_and_first_step_gradient(N, T, E, rb):
    """Which kernels run one optimizer step of the default network, through the profiler's class names, and the raw
    first-step gradient of every parameter against the oracle (2e-4 of the largest entry + 2e-3 relative) -- so that the
    persistent row-block levels (csrc/rowblock.h), the fused first-layer weight gradients (GemmArgs::lw_*,
    RbLevelArgs::lx_*) and the shapes that fall back to the tile kernels are each known to be what was tested."""
    from isaacgyminsertion_amd import _lib
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    units, pu = [512, 256, 128], [256, 128, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, pu, seed=13, done_p=0.05)
    eng = TeacherEngine(N, T, E, units=units, priv_units=pu, perm=perm)
    eng.load_params(init)
    orc = ot.TeacherOracle(init, perm, N, T, E, units, pu)
    orc.prepare(ro)
    eng.prepare(ro)
    st = orc.update(record_grads=1, max_steps=1)
    _lib.prof_enable(True)
    try:
        eng.fwd_bwd(0, 0)
        torch.cuda.synchronize()
        classes = {c["name"].split(":")[0]: c["launches"] for c in _lib.prof_read()}
    finally:
        _lib.prof_enable(False)
    mb = N * T // E
    if rb:
        assert classes.get("k_rb_level#trunk3") == 1 and classes.get("k_rb_level#env2") == 1, classes
        assert classes.get("gemm_dma_wgrad_multi_kernel#trunk2") == 1, classes
        # no other weight-gradient launch: the two first layers' products ride in the levels above
        assert not any(k.startswith("gemm_dma_wgrad_multi_kernel#env") or k.startswith("gemm_dma_wgrad_multi_kernel#trunk3")
                       for k in classes), classes
    else:
        assert not any(k.startswith("k_rb_level") for k in classes), classes
        assert classes.get("gemm_dma_wgrad_multi_kernel#trunk3") == 1, classes
        if mb % 128 == 0:      # the first trunk layer's weight gradient still comes from the dZ1 tiles: the env-side launch holds only env products
            assert classes.get("gemm_dma_wgrad_multi_kernel#env1") == 1 or classes.get("gemm_dma_wgrad_multi_kernel#env2") == 1, classes
    ref = st["grads"][0].numpy()
    np.testing.assert_allclose(eng.packed(eng.grads).cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)


def test_all_done_and_none_done_rollouts():
    """dones gate the bootstrap (experience.py:250-254): all ones -> returns = rewards + ... no carry."""
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from oracle import synth, teacher as ot
    N, T, E = 64, 6, 2
    units, pu = [32, 16], [16, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, pu, seed=9)
    for fill in (0, 1):
        ro2 = dict(ro)
        ro2["dones"] = torch.full((T, N), fill, dtype=torch.uint8)
        eng = TeacherEngine(N, T, E, units=units, priv_units=pu, perm=perm)
        eng.load_params(init)
        eng.prepare(ro2)
        torch.cuda.synchronize()
        ref = ot.gae_returns(ro2["rewards"], ro2["values"], ro2["dones"], ro2["last_values"], 0.99, 0.95)
        assert torch.equal(eng.returns_raw.cpu(), ref)
        if fill == 1:
            assert torch.equal(ref, ro2["rewards"] - ro2["values"] + ro2["values"])


@pytest.mark.parametrize("rows,D", [(1000, 15), (16384, 64), (131072, 1), (7, 3), (320000, 3), (33, 300)])
def test_rms_forward_modes(rows, D):
    from isaacgyminsertion_amd.algo.models.running_mean_std import RunningMeanStd
    from oracle.teacher import RmsState
    g = torch.Generator().manual_seed(rows + D)
    m = RunningMeanStd((D,)).cuda()
    o = RmsState(D)
    for step in range(2):
        x = torch.randn(rows, D, generator=g) * (1.0 + step) + 0.5 * step
        m.train()
        y = m(x.cuda())
        yo = o(x, train=True)
        np.testing.assert_allclose(y.cpu().numpy(), yo.numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(m.running_mean.cpu().numpy(), o.mean.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.running_var.cpu().numpy(), o.var.numpy(), rtol=1e-5)
    assert m.count.item() == o.count.item()
    m.eval()
    x = torch.randn(rows, D, generator=g) * 10
    before = m.packed.clone()
    np.testing.assert_allclose(m(x.cuda()).cpu().numpy(), o(x, train=False).numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(m(x.cuda(), True).cpu().numpy(), o.unnormalize(x).numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(before, m.packed)   # eval / unnorm never touch the state
