"""Which small state vector's placement moves the env level?  One at a time: fresh copies (the old ones kept alive), the env
level's average and the update's kernel time after each."""
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
eng.tune_workspace()
keep = []

def measure():
    eng.prepare(); eng.update(); torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read(); _lib.prof_enable(False)
    env2 = [round(1e3 * c["total_ms"] / max(c["launches"], 1), 1) for c in cl if c["name"].startswith("k_rb_level#env2")][0]
    return env2, round(sum(c["total_ms"] for c in cl) / 2, 2)

out = {"base": [measure() for _ in range(2)]}
for n in ("params", "grads", "adam_m", "adam_v", "stats", "perm", "rms_obs", "advantages", "mus_w"):
    res = []
    for i in range(6):
        old = getattr(eng, n); keep.append(old)
        keep.append(torch.empty(4096 * (i + 1) + 512, dtype=torch.uint8, device=dev))
        setattr(eng, n, old.clone())
        res.append(measure())
    out[n] = res
print(json.dumps(out))
