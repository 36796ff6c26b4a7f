import sys, time, torch
sys.path.insert(0, ".")
from isaacgyminsertion_amd.envs import synthetic_rollout as synth
from isaacgyminsertion_amd.teacher_native import TeacherEngine
N, T, E = 4096, 32, 8
units, priv = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(N, T, units, priv)
eng = TeacherEngine(N, T, E, units=units, priv_units=priv, perm=perm, device="cuda:0")
eng.load_params(init); eng.set_rollout(ro)
class W:
    def wait(self): pass
for name, fn in (("update", lambda: eng.update()), ("update_dp noop async", lambda: eng.update_dp(None, 1, all_reduce_async=lambda t: W())),
                 ("update_dp serial noop", lambda: eng.update_dp(lambda t: None, 1))):
    eng.prepare(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.prepare(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, "%.2f ms" % (1e3 * sorted(ts)[2]))
