"""Does the placement of anything BESIDE the workspace matter?  One engine, workspace fixed; per iteration a fresh copy of
(a) the parameter / gradient / Adam vectors, (b) the rollout arena, (c) the prepared per-update arrays -- the earlier copies
kept alive -- and the GPU time of one update's kernels."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd import _lib, ops
from isaacgyminsertion_amd.teacher_native import TeacherEngine
from isaacgyminsertion_amd.envs import synthetic_rollout as synth
dev = torch.device("cuda", 0)
UNITS, PRIV = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(4096, 32, UNITS, PRIV, seed=1234, device=dev)
eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
eng.load_params(init); eng.set_rollout(ro)
eng.tune_workspace()
keep = []

def measure():
    eng.prepare(); eng.update(); torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read(); _lib.prof_enable(False)
    return round(sum(c["total_ms"] for c in cl) / 2, 3)

out = {"base": [measure() for _ in range(3)]}
for label, names in (("small vectors", ("params", "grads", "adam_m", "adam_v", "stats")),
                     ("prepared arrays", ("returns_raw", "advantages", "values_n", "returns_n", "mus_w", "sigmas_w"))):
    res = []
    for i in range(6):
        for n in names:
            old = getattr(eng, n); keep.append(old)
            keep.append(torch.empty(4096 * (i + 1) + 512, dtype=torch.uint8, device=dev))
            setattr(eng, n, old.clone())
        res.append(measure())
    out[label] = res
res = []
for i in range(6):
    keep.append(eng._ro)
    keep.append(torch.empty((1 << 20) * (i + 1) + 4096, dtype=torch.uint8, device=dev))
    eng._ro = [t.clone() for t in eng._ro]
    res.append(measure())
out["rollout arena"] = res
print(json.dumps(out))
