"""Does the env level's duration change OVER TIME on one engine (one workspace allocation)?  Per update: its average from
the library's dispatch timestamps."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd import _lib                      # noqa: E402
from isaacgyminsertion_amd.teacher_native import TeacherEngine   # noqa: E402
from isaacgyminsertion_amd.envs import synthetic_rollout as synth   # noqa: E402

dev = torch.device("cuda", 0)
UNITS, PRIV = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(4096, 32, UNITS, PRIV, seed=1234, device=dev)
out = []
for e in range(2):
    eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
    eng.load_params(init)
    eng.set_rollout(ro)
    series = []
    for u in range(int(os.environ.get("N_UPDATES", "40"))):
        _lib.prof_enable(True)
        eng.prepare(); eng.update()
        torch.cuda.synchronize()
        cl = _lib.prof_read()
        _lib.prof_enable(False)
        series.append([round(1e3 * c["total_ms"] / max(c["launches"], 1), 1) for c in cl if c["name"].startswith("k_rb_level#env2")][0])
    out.append({"engine": e, "env2_us_per_update": series})
print(json.dumps(out))
