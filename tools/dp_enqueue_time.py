#!/usr/bin/env python
"""Host enqueue time of the data-parallel teacher update vs the device time it has to stay ahead of (one GPU).

The whole update is ONE native call (igi_teacher_update_dp); between the stages of every optimizer step the library
calls back so that the caller can issue its two gradient all-reduces asynchronously.  Measured here with a
stand-in reducer that does what ``dist.all_reduce(..., async_op=True)`` + ``work.wait()`` do on the host side of a
single-GPU box (an in-place op on a side stream, an event, a stream wait), and -- under ``IGI_DIST_BACKEND=gloo`` with
two processes -- with the real collective calls:

    python tools/dp_enqueue_time.py            # prints one JSON line
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from isaacgyminsertion_amd.envs import synthetic_rollout as synth  # noqa: E402
from isaacgyminsertion_amd.teacher_native import TeacherEngine  # noqa: E402


def main():
    N, T, E = 4096, 32, 8
    units, priv = [512, 256, 128], [256, 128, 8]
    init, ro, perm = synth.teacher_problem(N, T, units, priv)
    eng = TeacherEngine(N, T, E, units=units, priv_units=priv, perm=perm, device="cuda:0")
    eng.load_params(init)
    eng.set_rollout(ro)
    side = torch.cuda.Stream()

    class Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def reduce_async(t):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            t.mul_(1.0)
            ev = torch.cuda.Event()
            ev.record(side)
        return Work(ev)

    res = {}
    for name, fn in (("single_gpu_update", lambda: eng.update()),
                     ("dp_update_one_native_call", lambda: eng.update_dp(None, 1, all_reduce_async=reduce_async))):
        eng.prepare()
        fn()
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(5):
            eng.prepare()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            enq.append(t1 - t0)
            tot.append(t2 - t0)
        enq.sort(); tot.sort()
        res[name] = {"host_enqueue_ms_per_update": round(1e3 * enq[2], 2), "device_ms_per_update": round(1e3 * tot[2], 2),
                     "host_enqueue_us_per_optimizer_step": round(1e6 * enq[2] / 64, 1),
                     "device_us_per_optimizer_step": round(1e6 * tot[2] / 64, 1)}
    res["note"] = ("median of 5; 64 optimizer steps per update; the host returns from the call long before the device "
                   "finishes when enqueue < device time, i.e. the launch queue never runs dry")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
