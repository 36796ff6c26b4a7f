"""Is the env level's two-mode duration (33 vs 41 us) a property of the PROCESS or of the ALLOCATION?  Several engines in one
process, each with its own workspace (the earlier ones kept alive, an odd-sized spacer between them), the env level's average
from the library's dispatch timestamps for each.  Run the script a few times."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd import _lib                      # noqa: E402
from isaacgyminsertion_amd.teacher_native import TeacherEngine   # noqa: E402
from isaacgyminsertion_amd.envs import synthetic_rollout as synth   # noqa: E402  (product-side arena generator)

dev = torch.device("cuda", 0)
UNITS, PRIV = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(4096, 32, UNITS, PRIV, seed=1234, device=dev)
keep, out = [], []
for i in range(int(os.environ.get("N_ENGINES", "5"))):
    eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
    eng.load_params(init)
    eng.set_rollout(ro)
    eng.tune_workspace()
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(3):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read()
    _lib.prof_enable(False)
    lv = {c["name"].split(":")[0]: round(1e3 * c["total_ms"] / max(c["launches"], 1), 2) for c in cl if c["name"].startswith("k_rb_level")}
    ws = [t for t in vars(eng).values() if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() * t.element_size() > (64 << 20)]
    out.append({"engine": i, "levels_us": lv, "trial_ms": eng.workspace_trial_ms, "workspace": hex(eng.workspace.data_ptr())})
    keep.append(eng)
    keep.append(torch.empty((3 << 20) + 4096 * (i + 1), dtype=torch.uint8, device=dev))   # spacer: the next workspace lands elsewhere
print(json.dumps(out))
