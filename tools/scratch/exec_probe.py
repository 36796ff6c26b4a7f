import subprocess, sys, torch
torch.zeros(1, device="cuda").sum().item()
print("gpu initialised", flush=True)
r = subprocess.run([sys.executable, "-c", "print('child ran')"], capture_output=True, text=True)
print("rc", r.returncode, "out", r.stdout.strip(), "err", r.stderr.strip()[-300:])
import multiprocessing as mp
def f(q): q.put("spawned child ran")
ctx = mp.get_context("spawn"); q = ctx.Queue(); p = ctx.Process(target=f, args=(q,)); p.start()
try: print(q.get(timeout=60))
except Exception as e: print("spawn failed", repr(e))
p.join(10); print("exit", p.exitcode)
