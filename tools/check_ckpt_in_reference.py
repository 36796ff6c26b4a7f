"""BUILD CONTAINER ONLY (needs /root/reference): loads checkpoint files written by THIS package on the GPU box
(tests/test_gpu_checkpoint.py leaves them under its --basetemp) into the REFERENCE's own classes --
PPO.restore_test (frozen_ppo.py:477-484) and ExtrinsicAdapt.restore_test (ext_adapt.py:1087-1099, strict teacher
load, student + normalisers) -- evaluates the fixture's frames and compares with what the reference computed from
its own files.  Writes profiles/r02_ckpt_interop.json.

    python tools/check_ckpt_in_reference.py gpurun_out/ckpt/tmp
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness as rh  # noqa: E402

rh.install()
from algo.ppo.frozen_ppo import PPO  # noqa: E402  (reference)
import make_golden_student as mgs  # noqa: E402
import make_golden_checkpoint as mgc  # noqa: E402

# the files hold cuda tensors (as the reference's own files do when it trains on a GPU) and the reference calls
# torch.load(fn) without map_location: this container has no GPU, so map to cpu here
_load = torch.load
torch.load = lambda f, *a, **k: _load(f, *a, **{**k, "map_location": "cpu"})

G = np.load(os.path.join(ROOT, "tests", "golden", "checkpoint.npz"))
base = sys.argv[1]
t_dir = os.path.join(base, "test_teacher_checkpoint_writte0")
s_dir = os.path.join(base, "test_student_checkpoint_writte0")
res = {}
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "stage1_nn")); os.makedirs(os.path.join(d, "stage2_nn"))
    shutil.copy(os.path.join(t_dir, "ours.pth"), os.path.join(d, "stage1_nn", "last.pth"))
    shutil.copy(os.path.join(s_dir, "ours_stud.pth"), os.path.join(d, "stage2_nn", "last_stud.pth"))
    cfg = rh.teacher_config(4, 4, 2, units=mgc.UNITS, priv_units=mgc.PRIV_UNITS)
    a = PPO(None, d, cfg)
    a.restore_test(os.path.join(d, "stage1_nn", "last.pth"))       # strict load_state_dict inside
    a.set_eval()
    obs, priv = torch.from_numpy(G["s1/frames/obs"]), torch.from_numpy(G["s1/frames/priv_info"])
    with torch.no_grad():
        mu, lat = a.model.act_inference({"obs": a.running_mean_std(obs), "priv_info": a.priv_mean_std(priv)})
    res["teacher_file_written_by_repo_loaded_by_reference_PPO.restore_test"] = True
    res["teacher_mu_max_abs_diff_vs_reference_own_file"] = float(np.abs(mu.numpy() - G["s1/expect/mu"]).max())
    res["teacher_latent_max_abs_diff"] = float(np.abs(lat.numpy() - G["s1/expect/latent"]).max())
    scfg = mgs.student_config(4, 4, 2, tactile=True, pcl=True)
    scfg.train.network.mlp.units = list(mgc.UNITS)
    scfg.train.network.priv_mlp.units = list(mgc.PRIV_UNITS)
    sb = mgc.make_student(scfg, mgs.FakeEnv(4, True, True))
    shutil.copy(os.path.join(s_dir, "ours.pth"), os.path.join(d, "stage1_nn", "last.pth"))   # ExtrinsicAdapt.save's teacher file
    sb.restore_test(os.path.join(d, "stage1_nn", "last.pth"))
    sb.stud_obs_mean_std.eval(); sb.pcl_mean_std.eval()
    frames = {k: torch.from_numpy(G[f"s2/frames/{k}"]) for k in ("student_obs", "tactile", "pcl")}
    with torch.no_grad():
        act, _ = sb.student.predict(sb.process_obs(frames), requires_grad=False)
    missing, unexpected = sb.student.model.load_state_dict(
        torch.load(os.path.join(d, "stage2_nn", "last_stud.pth"))["student"], strict=True), None
    res["student_files_written_by_repo_loaded_by_reference_ExtrinsicAdapt.restore_test"] = True
    res["student_strict_load_ok"] = True
    res["student_action_max_abs_diff_vs_reference_own_file"] = float(np.abs(act.numpy() - G["s2/expect/action"]).max())
res["files"] = "written on the MI355X box by tests/test_gpu_checkpoint.py (PPO.save / ExtrinsicAdapt.save of this package)"
out = os.path.join(ROOT, "profiles", "r02_ckpt_interop.json")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
