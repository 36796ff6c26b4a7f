"""Which ATen kernels still run inside one optimizer step of the student, and which framework op launches each
(torch.profiler with stacks).  python tools/probes/student_aten_ops.py [config=4] [envs=512]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_student_scale as T
from torch.profiler import profile, ProfilerActivity
config, envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 512
agent = T._student_agent(config, envs)
for i in range(3):
    agent.update_step(i); agent.optim.step(1.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    agent.update_step(3); agent.optim.step(1.0)
    torch.cuda.synchronize()
ev = prof.events()
# kernels launched by each CPU op (leaf ops that own device time)
rows = {}
for e in ev:
    if e.device_type.name == "CPU" and e.self_device_time_total > 0:
        key = e.name
        shapes = str(e.input_shapes)[:80]
        stack = [s for s in (e.stack or []) if "isaacgyminsertion_amd" in s or "tact.py" in s or "ext_adapt" in s or "experience" in s]
        k = (key, shapes, stack[0][-70:] if stack else "")
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1; r[1] += e.self_device_time_total
for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t:8.1f} us  x{n:3d}  {k[0][:38]:38s} {k[1]:60s} {k[2]}")
