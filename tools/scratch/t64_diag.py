import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import synth, teacher as ot
from tests.test_gpu_teacher import _engine
N,T,E=2048,64,8
units, priv_units = [512, 256, 128], [256, 128, 8]
init, ro, perm = synth.teacher_problem(N, T, units, priv_units, seed=4321)
meta = dict(num_envs=N, horizon=T, mini_epochs=E, units=units, priv_units=priv_units)
eng=_engine(meta, init, perm)
orc = ot.TeacherOracle(init, perm, N, T, E, units, priv_units)
orc.prepare(ro); eng.prepare(ro)
for step in range(4):
    # per-sample boundary proximity in the oracle BEFORE its step
    d=orc.data; i=step%orc.n_mb
    idx=orc.perm[i*orc.mb:(i+1)*orc.mb]
    with torch.no_grad():
        ro_o=orc.rms_obs.clone(); rp=orc.rms_priv.clone()
        obs=ro_o(d["obses"][idx],True); priv=rp(d["priv_info"][idx],True)
        nlp, values, ent, mu, sigma = ot.forward_train(orc.p, obs, priv, d["actions"][idx], 3, 3)
        ratio=torch.exp(d["neglogpacs"][idx]-nlp)
        dv=(values-d["values"][idx]).abs().squeeze(1)
        print(f"step {step}: ratio range [{ratio.min():.4f},{ratio.max():.4f}] near-clip(1e-4): {(((ratio-1.2).abs()<1e-4)|((ratio-0.8).abs()<1e-4)).sum().item()} outside: {((ratio>1.2)|(ratio<0.8)).sum().item()}; |dV| max {dv.max():.4f} near 0.2 (1e-4): {((dv-0.2).abs()<1e-4).sum().item()} outside {(dv>0.2).sum().item()}")
    st = orc.update(record_grads=1, max_steps=1, start_step=step)
    eng.fwd_bwd(step % eng.n_mb, step); torch.cuda.synchronize()
    views=eng.param_views(eng.grads)
    off=0
    g_ref=st["grads"][0]
    for k,v in views.items():
        n=v.numel(); r=g_ref[off:off+n].view(v.shape); off+=n
        e=(v.cpu()-r).abs().max().item()
        if e>1e-5*g_ref.abs().max(): print(f"   {k:28s} maxerr {e:.3e} max|ref| {r.abs().max():.3e}")
    eng.apply(step)
    eng.load_params({kk: v.detach() for kk, v in orc.p.items()})
    mv, vv = eng.param_views(eng.adam_m), eng.param_views(eng.adam_v)
    for kk, (m_, v_) in orc.adam_state().items():
        mv[kk].copy_(m_); vv[kk].copy_(v_)
