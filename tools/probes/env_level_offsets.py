"""The env level's duration as a function of WHERE inside one big allocation the workspace sits: the same engine, its
workspace a window of a 6 GB buffer at different byte offsets.  Looks for the address bits the effect depends on."""
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
eng = TeacherEngine(4096, 32, 8, units=UNITS, priv_units=PRIV, perm=perm, device=dev)
eng.load_params(init)
eng.set_rollout(ro)
wbytes = eng.workspace.numel()
giant = torch.zeros(6 << 30, dtype=torch.uint8, device=dev)
KB, MB = 1 << 10, 1 << 20
offs = [0, 4 * KB, 64 * KB, 256 * KB, 1 * MB, 2 * MB, 4 * MB, 6 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB, 256 * MB, 512 * MB,
        1024 * MB, 1536 * MB, 2048 * MB, 3072 * MB, 4096 * MB, 5000 * MB, 0, 2 * MB, 1024 * MB]
offs += [int(x) * MB for x in os.environ.get("EXTRA_MB", "").split(",") if x]
out = []
for off in offs:
    eng.workspace = giant[off:off + wbytes]
    eng.prepare(); eng.update()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(2):
        eng.prepare(); eng.update()
    torch.cuda.synchronize()
    cl = _lib.prof_read()
    _lib.prof_enable(False)
    env2 = [round(1e3 * c["total_ms"] / max(c["launches"], 1), 1) for c in cl if c["name"].startswith("k_rb_level#env2")][0]
    tot = round(sum(c["total_ms"] for c in cl) / 2, 3)
    out.append((off // KB, env2, tot))
print(json.dumps({"base": hex(giant.data_ptr()), "offset_KB__env2_us__update_ms": out}))
