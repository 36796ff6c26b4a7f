"""What the data-parallel schedule costs on ONE GPU before any inter-GPU traffic (BASELINE configs[1] size): the
single-GPU update against (a) the two-phase schedule with a no-op reducer through the Python callback, (b) the native
RCCL path on a one-rank communicator, overlapped (two grouped bucket all-reduces on the communication stream, event
fences) and (c) serial (one all-reduce of the flat gradient on the compute stream).  Median of 7 updates each, JSON."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from isaacgyminsertion_amd.envs import synthetic_rollout as synth  # noqa: E402
from isaacgyminsertion_amd.teacher_native import TeacherEngine  # noqa: E402
from isaacgyminsertion_amd.utils.dist import NativeComm  # noqa: E402

N, T, E = 4096, 32, 8
units, priv = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(N, T, units, priv)
eng = TeacherEngine(N, T, E, units=units, priv_units=priv, perm=perm, device="cuda:0")
eng.load_params(init)
eng.set_rollout(ro)
comm = NativeComm(rank=0, world=1)


class W:
    def wait(self):
        pass


out = {}
for name, fn in (("update (single GPU)", lambda: eng.update()),
                 ("two-phase schedule, no-op reducer via Python callback", lambda: eng.update_dp(None, 1, all_reduce_async=lambda t: W())),
                 ("native RCCL, 1-rank communicator, overlapped buckets", lambda: eng.update_dp_native(comm, overlap=True)),
                 ("native RCCL, 1-rank communicator, serial", lambda: eng.update_dp_native(comm, overlap=False)),
                 ("update (single GPU) again", lambda: eng.update())):
    eng.prepare()
    fn()
    torch.cuda.synchronize()
    ts, host = [], []
    for _ in range(7):
        eng.prepare()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        host.append(t1 - t0)
    out[name] = {"ms_per_update": round(1e3 * sorted(ts)[3], 3), "host_enqueue_ms": round(1e3 * sorted(host)[3], 3)}

# ---- the student's exchange (configs[3] single-rank share): single-GPU update against the native exchange on the same
# one-rank communicator, overlapped (decoder-side bucket from inside backward on the communication stream) and serial
import os  # noqa: E402
sys.path.insert(0, "tests")
import test_gpu_student_scale as T  # noqa: E402

del eng
torch.cuda.empty_cache()
for name, mode in (("student update (single GPU)", "single"), ("student, native RCCL 1-rank, overlapped buckets", "overlap"),
                   ("student, native RCCL 1-rank, serial", "serial"), ("student update (single GPU) again", "single")):
    torch.manual_seed(7)
    agent = T._student_agent(4, 512)
    if mode != "single":
        agent.multi_gpu, agent.rank_size, agent._comm = True, 1, comm
    os.environ["IGI_DP_OVERLAP"] = "1" if mode == "overlap" else "0"
    agent.update()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        agent.update()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[name] = {"ms_per_update": round(1e3 * sorted(ts)[2], 2), "ms_per_optimizer_step": round(1e3 * sorted(ts)[2] / 64, 3)}
    del agent
    torch.cuda.empty_cache()
os.environ.pop("IGI_DP_OVERLAP", None)
txt = json.dumps(out, indent=1)
if len(sys.argv) > 1:            # RCCL prints its version banner on stdout at communicator creation: keep the file pure JSON
    with open(sys.argv[1], "w") as f:
        f.write(txt + "\n")
print(txt)
