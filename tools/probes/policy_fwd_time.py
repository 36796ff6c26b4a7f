"""Device time of the rollout policy step's launches (igi_prof dispatch timestamps), 4096 / 16384 rows."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd import _lib
from isaacgyminsertion_amd.teacher_native import TeacherEngine
from isaacgyminsertion_amd.envs import synthetic_rollout as synth
out = {}
for N in (4096, 16384):
    init, ro, perm = synth.teacher_problem(64, 4, [512, 256, 128], [256, 128, 8], seed=5, device="cuda:0")
    eng = TeacherEngine(N, 8, 4, units=[512, 256, 128], priv_units=[256, 128, 8], device="cuda:0")
    eng.load_params(init)
    f = dict(dtype=torch.float32, device="cuda:0")
    obs, priv, noise = torch.randn(N, 15, **f), torch.randn(N, 64, **f), torch.randn(N, 6, **f)
    o = [torch.zeros(N, 6, **f), torch.zeros(N, **f), torch.zeros(N, 1, **f), torch.zeros(N, 6, **f), torch.zeros(N, 6, **f),
         torch.zeros(N, 6, **f), torch.zeros(N, 1, **f)]
    def step():
        torch.ops.mi355ppo.rollout_policy_step(eng.state_list(), *eng._cfg_args(), obs, priv, True, noise, None, None, None, *o)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    cl = _lib.prof_read()
    _lib.prof_enable(False)
    out[f"{N} rows"] = {c["name"]: {"launches": c["launches"], "avg_us": round(1e3 * c["total_ms"] / c["launches"], 2),
                                    "tflops": round(c["flops"] / max(c["total_ms"], 1e-9) / 1e9, 1)} for c in cl}
print(json.dumps(out, indent=1))
