"""env level duration over many separate workspace allocations (IGI_WS_TRIALS=1), for a layout experiment"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd import _lib
from isaacgyminsertion_amd.teacher_native import TeacherEngine
from isaacgyminsertion_amd.envs import synthetic_rollout as synth
dev = torch.device("cuda", 0)
UNITS, PRIV = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(4096, 32, UNITS, PRIV, seed=1234, device=dev)
eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
eng.load_params(init); eng.set_rollout(ro)
keep, out = [], []
for i in range(int(os.environ.get("N_ALLOC", "12"))):
    keep.append(eng.workspace)
    eng.workspace = torch.zeros_like(eng.workspace)
    eng.prepare(); eng.update(); torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read(); _lib.prof_enable(False)
    out.append([round(1e3 * c["total_ms"] / max(c["launches"], 1), 1) for c in cl if c["name"].startswith("k_rb_level#env2")][0])
print(os.environ.get("IGI_RB_VARIANT", "0"), out)
