"""Checkpoint-interop golden vectors from the REFERENCE implementation (build container only).

The reference's own ``PPO.save`` (frozen_ppo.py:448-463) and ``ExtrinsicAdapt.save`` (ext_adapt.py:1150-1170)
write ``last.pth`` / ``last_stud.pth``; the files are read back with ``torch.load`` and their TENSORS are stored
in ``checkpoint.npz`` (data, not the pickle), key order and dtypes included.  A second, freshly constructed
reference agent then restores those files through its own ``restore_test`` (frozen_ppo.py:477-484,
ext_adapt.py:1087-1099) and evaluates recorded frames: the deterministic teacher action / latent
(``act_inference`` on eval-mode normalised inputs, as deploy_s1 does) and the student's action
(``process_obs`` + ``Runner.predict``, as deploy_s2 does).  The GPU test rebuilds the two ``.pth`` files from
the fixture, loads them through this repo's restore paths and must reproduce those outputs.

    python tests/golden/make_golden_checkpoint.py  ->  tests/golden/checkpoint.npz
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from algo.ppo.frozen_ppo import PPO  # noqa: E402  (reference)
import make_golden_student as mgs  # noqa: E402
import make_golden_rollout as mgr  # noqa: E402
from algo.ext_adapt.ext_adapt import ExtrinsicAdapt  # noqa: E402  (reference)

UNITS, PRIV_UNITS = (64, 48, 32), (48, 32, 8)


def dump_ckpt(out, tag, path):
    ck = torch.load(path)
    out[f"{tag}/top_keys"] = np.array(list(ck.keys()))
    for top, sd in ck.items():
        out[f"{tag}/keys/{top}"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            out[f"{tag}/t/{top}/{k}"] = v.numpy().copy()
            out[f"{tag}/dtype/{top}/{k}"] = np.array(str(v.dtype))


def make_student(cfg, env):
    orig_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as d:
            agent = ExtrinsicAdapt(env, d, cfg)
    finally:
        torch.nn.Module.to = orig_to
    agent.student.device = "cpu"
    agent.student.eval_process_tactile = lambda t: t
    return agent


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    g = torch.Generator().manual_seed(77)
    N = 4
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "stage1_nn"))
        os.makedirs(os.path.join(d, "stage2_nn"))
        # ---- teacher: written by the reference's PPO.save -------------------------------------------------
        cfg = rh.teacher_config(N, 4, 2, units=UNITS, priv_units=PRIV_UNITS)
        torch.manual_seed(5)
        a = PPO(None, d, cfg)
        with torch.no_grad():
            a.model.sigma.copy_(0.2 * torch.randn(6, generator=g))
            a.model.mu.weight.mul_(30.0)
        for m in (a.running_mean_std, a.priv_mean_std, a.value_mean_std):
            mgr.set_rms(m, g)
        s1 = os.path.join(d, "stage1_nn", "last")
        a.save(s1)
        dump_ckpt(out, "s1", s1 + ".pth")
        # a fresh reference agent restores it and acts deterministically on recorded frames
        torch.manual_seed(6)
        b = PPO(None, d, cfg)
        b.restore_test(s1 + ".pth")
        b.set_eval()
        obs, priv = 2.0 * torch.randn(5, 15, generator=g), 2.0 * torch.randn(5, 64, generator=g)
        with torch.no_grad():
            mu, latent = b.model.act_inference({"obs": b.running_mean_std(obs), "priv_info": b.priv_mean_std(priv)})
            val = b.model_act({"obs": obs, "priv_info": priv})["values"]     # de-normalised (frozen_ppo.py:365)
        out["s1/frames/obs"], out["s1/frames/priv_info"] = obs.numpy(), priv.numpy()
        out["s1/expect/mu"], out["s1/expect/latent"] = mu.numpy(), latent.numpy()
        out["s1/expect/value_denorm"] = val.numpy()
        out["s1/units"], out["s1/priv_units"] = np.array(UNITS), np.array(PRIV_UNITS)

        # ---- student: written by the reference's ExtrinsicAdapt.save --------------------------------------
        scfg = mgs.student_config(N, 4, 2, tactile=True, pcl=True)
        scfg.train.network.mlp.units = list(UNITS)
        scfg.train.network.priv_mlp.units = list(PRIV_UNITS)
        env = mgs.FakeEnv(N, True, True)
        torch.manual_seed(7)
        sa = make_student(scfg, env)
        with torch.no_grad():
            for m in sa.student.model.modules():
                if isinstance(m, torch.nn.Linear):
                    torch.nn.init.xavier_uniform_(m.weight, generator=g)
                    m.bias.uniform_(-0.1, 0.1, generator=g)
        sa.agent.load_state_dict(a.model.state_dict())
        for m, src in ((sa.running_mean_std, a.running_mean_std), (sa.priv_mean_std, a.priv_mean_std)):
            m.load_state_dict(src.state_dict())
        mgr.set_rms(sa.stud_obs_mean_std, g)
        mgr.set_rms(sa.pcl_mean_std, g)
        s2 = os.path.join(d, "stage2_nn", "last")
        sa.save(s2)
        dump_ckpt(out, "s2t", s2 + ".pth")               # the teacher file ExtrinsicAdapt.save writes (no value_mean_std)
        dump_ckpt(out, "s2", s2 + "_stud.pth")
        # fresh reference agent: restore_test(stage1_nn/last.pth) pulls stage2_nn/last_stud.pth (ext_adapt.py:1087-1099)
        torch.manual_seed(8)
        sb = make_student(scfg, env)
        sb.restore_test(s1 + ".pth")
        frames = {"student_obs": 2.0 * torch.randn(5, 15, generator=g),
                  "tactile": torch.rand(5, 1, 3, 2048, generator=g),
                  "pcl": (0.05 * torch.randn(5, 1, 800, 3, generator=g) + torch.tensor([0.5, 0.0, 0.1])).reshape(5, 1, 2400)}
        # the deployment player holds these two in eval mode (deploy_s2.py:157-159, 196-199); the trainer's
        # set_student_eval leaves them in train mode
        sb.stud_obs_mean_std.eval()
        sb.pcl_mean_std.eval()
        with torch.no_grad():
            sd = sb.process_obs(frames)
            act, _ = sb.student.predict(sd, requires_grad=False)
        for k, v in frames.items():
            out[f"s2/frames/{k}"] = v.numpy()
        out["s2/expect/student_obs_n"] = sd["student_obs"].numpy()
        out["s2/expect/pcl_n"] = sd["pcl"].numpy()
        out["s2/expect/action"] = act.numpy()
        # eval mode: restoring and evaluating must not move the normalisers
        assert torch.equal(sb.stud_obs_mean_std.count, sa.stud_obs_mean_std.count)
    path = os.path.join(HERE, "checkpoint.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; s1 keys {list(out['s1/top_keys'])}, "
          f"s2 keys {list(out['s2/top_keys'])}; |mu|max {np.abs(out['s1/expect/mu']).max():.3f} "
          f"|action|max {np.abs(out['s2/expect/action']).max():.3f}")
