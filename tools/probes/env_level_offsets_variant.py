import json, os, sys
import torch
sys.path.insert(0, os.getcwd())
from isaacgyminsertion_amd import _lib
from isaacgyminsertion_amd.teacher_native import TeacherEngine
from isaacgyminsertion_amd.envs import synthetic_rollout as synth
dev = torch.device("cuda", 0)
UNITS, PRIV = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(4096, 32, UNITS, PRIV, seed=1234, device=dev)
eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
eng.load_params(init); eng.set_rollout(ro)
wbytes = eng.workspace.numel()
giant = torch.zeros(6 << 30, dtype=torch.uint8, device=dev)
MB = 1 << 20
out = []
for off in [0, 32 * MB, 16 * MB, 5000 * MB, 1536 * MB, 64 * MB]:
    eng.workspace = giant[off:off + wbytes]
    eng.prepare(); eng.update(); torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read(); _lib.prof_enable(False)
    d = {c["name"].split(":")[0].split("#")[-1]: round(1e3 * c["total_ms"] / max(c["launches"], 1), 1) for c in cl if c["name"].startswith("k_rb_level")}
    out.append((off // MB, d))
print(os.environ.get("IGI_RB_VARIANT", "0"), out)
