"""Child process of tests/test_torch_library_cpp.py: runs ONE registration of torch.ops.mi355ppo -- the C++ one
(libigi_torch_ops.so, TORCH_LIBRARY in csrc/torch_ops.cpp) or the Python one (isaacgyminsertion_amd/ops.py) -- through the
same sequence of calls, using nothing but the dispatcher ops and the ctypes view of the C ABI for sizes.

    python tests/cpp_ops_child.py schemas cpp|py            -> JSON {op: schema} (+ the CPU-tensor refusal) on stdout
    python tests/cpp_ops_child.py run cpp|py out.npz        -> tensors of the sequence (needs a GPU)
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["gae_advnorm", "ppo_minibatch_fwd_bwd", "ppo_clip_adam", "ppo_update", "actor_critic_infer", "rollout_policy_step",
         "rms_update_normalize", "clip_adam_step", "bc_loss_fwd_bwd", "tactile_cnn_fwd", "tactile_cnn_bwd",
         "spatial_softargmax_fwd", "spatial_softargmax_bwd", "pointnet_max_fwd", "pointnet_max_bwd"]


def register(which):
    if which == "cpp":
        torch.ops.load_library(os.path.join(ROOT, "isaacgyminsertion_amd", "libigi_torch_ops.so"))
        assert "isaacgyminsertion_amd.ops" not in sys.modules
    else:
        import isaacgyminsertion_amd.ops  # noqa: F401
    return torch.ops.mi355ppo


def schemas(which):
    o = register(which)
    out = {n: str(getattr(o, n).default._schema) for n in NAMES}
    try:
        o.pointnet_max_fwd(torch.zeros(2, 5, 3), torch.zeros(16896))
        out["_cpu_refused"] = False
    except RuntimeError as e:
        out["_cpu_refused"] = "HIP" in str(e)
    print(json.dumps(out))


def run(which, path):
    o = register(which)
    from isaacgyminsertion_amd import _lib          # ctypes only: struct layouts + size queries of the C ABI
    from oracle import synth
    dev = torch.device("cuda:0")
    N, T, E = 64, 8, 4
    units, priv_units = [64, 48, 32], [48, 32, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, priv_units, seed=9, done_p=0.1)
    M = _lib.IGI_MAX_LAYERS
    icfg = [15, 64, 6, 3] + priv_units + [0] * (M - 3) + [3] + units + [0] * (M - 3) + [N, T, E]
    fcfg = [0.99, 0.95, 2.5e-4, 0.9, 0.999, 1e-8, 0.2, 4.0, 0.0, 1e-4, 1.0, 1e-5]
    cfg = _lib.TeacherCfg()
    cfg.obs_dim, cfg.priv_dim, cfg.act_dim, cfg.n_priv_layers, cfg.n_layers = 15, 64, 6, 3, 3
    for i in range(3):
        cfg.priv_units[i], cfg.units[i] = priv_units[i], units[i]
    cfg.num_envs, cfg.horizon, cfg.mini_epochs = N, T, E
    L = _lib.lib()
    n = L.igi_teacher_param_offsets(C.byref(cfg), None, None, 0)
    off, sz = (C.c_int64 * n)(), (C.c_int64 * n)()
    L.igi_teacher_param_offsets(C.byref(cfg), off, sz, n)
    P = int(L.igi_teacher_param_count(C.byref(cfg)))
    f32 = dict(dtype=torch.float32, device=dev)
    params = torch.zeros(P, **f32)
    for (k, v), o_, s_ in zip(init.items(), off, sz):
        params[o_:o_ + s_] = v.reshape(-1).to(dev)

    def rms(d):
        s = torch.zeros(2 * d + 1, dtype=torch.float64, device=dev)
        s[d:2 * d] = 1.0
        s[2 * d] = 1.0
        return s

    state = [params, torch.zeros(P, **f32), torch.zeros(P, **f32), torch.zeros(P, **f32), rms(15), rms(64), rms(1),
             perm.to(dev), torch.zeros(T, N, 1, **f32), torch.zeros(T, N, **f32), torch.zeros(T, N, 1, **f32),
             torch.zeros(T, N, 1, **f32), torch.zeros(T, N, 6, **f32), torch.zeros(T, N, 6, **f32),
             torch.zeros(E * E, _lib.IGI_STATS_PER_STEP, **f32),
             torch.zeros(int(L.igi_teacher_workspace_bytes(C.byref(cfg))), dtype=torch.uint8, device=dev)]
    rollout = [ro[k].to(dev).contiguous() for k in ("obses", "priv_info", "rewards", "values", "neglogpacs", "dones",
                                                     "actions", "mus", "sigmas", "last_values")]
    out = {}
    o.gae_advnorm(rollout, state, icfg, fcfg, True)
    out["advantages"], out["returns_raw"] = state[9].clone(), state[8].clone()
    o.ppo_minibatch_fwd_bwd(rollout, state, icfg, fcfg, 0, 0, -1)
    out["grads0"] = state[1].clone()
    o.ppo_clip_adam(state, icfg, fcfg, 0, 1, 1.0)
    out["params1"] = state[0].clone()
    o.gae_advnorm(rollout, state, icfg, fcfg, True)
    o.ppo_update(rollout, state, icfg, fcfg, 1)
    out["params_after"], out["stats"] = state[0].clone(), state[14].clone()
    g = torch.Generator(device=dev).manual_seed(0)
    obs, priv = torch.randn(50, 15, device=dev, generator=g), torch.randn(50, 64, device=dev, generator=g)
    out["mu"], out["value"], out["latent"] = o.actor_critic_infer(state, icfg, fcfg, obs, priv, True, True)
    z = lambda *s: torch.zeros(*s, **f32)   # noqa: E731
    slots = [z(50, 15), z(50, 64), z(50, 6), z(50), z(50, 1), z(50, 6), z(50, 6), z(50, 6), z(50, 1)]
    o.rollout_policy_step(state, icfg, fcfg, obs, priv, True, torch.randn(50, 6, device=dev, generator=g),
                          torch.tensor([0.1, 2.0, 10.0], dtype=torch.float64, device=dev), *slots)
    for i, t in enumerate(slots):
        out[f"slot{i}"] = t
    st = rms(15)
    out["rms_y"] = o.rms_update_normalize(torch.randn(300, 15, device=dev, generator=g) * 3 + 1, st, 1e-5, True, False)
    out["rms_state"] = st
    pa, gr = torch.randn(1000, device=dev, generator=g), torch.randn(1000, device=dev, generator=g)
    m, v, stats = z(1000), z(1000), z(8)
    o.clip_adam_step(pa, gr, m, v, 0.5, 3e-4, 0.9, 0.999, 1e-8, 0.0, 0.0, 1, 1.0, stats)
    out["adam_p"], out["adam_stats"] = pa, stats
    mu_ = torch.randn(77, 6, device=dev, generator=g) * 1.5
    out["bc_loss"], out["bc_dmu"] = o.bc_loss_fwd_bwd(mu_, torch.randn(77, 6, device=dev, generator=g),
                                                      torch.tensor([1, 1, 0.1, 1, 1, 1.0], device=dev), True)
    x = torch.rand(32, 3, 32, 64, device=dev, generator=g)
    tp = torch.randn(int(L.igi_tactile_param_count(C.byref(_lib.TactileCfg(32, 32, 64, 32)))), device=dev, generator=g) * 0.05
    y, ws = o.tactile_cnn_fwd(x, tp, 32)
    out["tac_y"], out["tac_g"] = y, o.tactile_cnn_bwd(torch.randn(32, 32, device=dev, generator=g), tp, ws, 32, 64)
    xs = torch.randn(4, 16, 8, 24, device=dev, generator=g)
    so, sst = o.spatial_softargmax_fwd(xs, True)
    out["ssa"], out["ssa_dx"] = so, o.spatial_softargmax_bwd(xs, so, sst, torch.randn(4, 32, device=dev, generator=g), True)
    pc, pp = torch.randn(9, 400, 3, device=dev, generator=g) * 0.5, torch.randn(16896, device=dev, generator=g) * 0.2
    py, pidx = o.pointnet_max_fwd(pc, pp)
    out["pn_y"], out["pn_idx"] = py, pidx
    out["pn_g"] = o.pointnet_max_bwd(pc, pp, torch.randn(9, 256, device=dev, generator=g), pidx)
    torch.cuda.synchronize()
    np.savez(path, **{k: t.detach().cpu().numpy() for k, t in out.items()})


if __name__ == "__main__":
    if sys.argv[1] == "schemas":
        schemas(sys.argv[2])
    else:
        run(sys.argv[2], sys.argv[3])
