"""Checkpoint interchange with the reference.

``tests/golden/checkpoint.npz`` holds the tensors of ``stage1_nn/last.pth`` / ``stage2_nn/last{,_stud}.pth`` as the
REFERENCE's own PPO.save / ExtrinsicAdapt.save wrote them (frozen_ppo.py:448-463, ext_adapt.py:1150-1170), with key
order and dtypes, plus what a freshly restored reference agent computes on recorded frames
(tests/golden/make_golden_checkpoint.py).  Here the ``.pth`` files are rebuilt from the fixture, loaded through every
restore path of this package (PPO.restore_test / restore_train, ExtrinsicAdapt.restore_test, both HardwarePlayers)
and evaluated on the same frames: actions 2e-5 abs (fp32 network on [-1, 1] outputs).  The other direction --
files written by THIS package carry exactly the reference's structure (top-level keys, state_dict key order, dtypes,
shapes), which is all ``load_state_dict(strict=True)`` on the reference side looks at -- is checked here too; the
literal load into the reference's classes is recorded in profiles/r02_ckpt_interop.json (tools/check_ckpt_in_reference.py,
build container only)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint.npz"))
DEV = "cuda:0"


def _write_pth(tag, path):
    ck = {}
    for top in [str(x) for x in G[f"{tag}/top_keys"]]:
        sd = {}
        for k in [str(x) for x in G[f"{tag}/keys/{top}"]]:
            t = torch.from_numpy(G[f"{tag}/t/{top}/{k}"].copy())
            assert str(t.dtype) == str(G[f"{tag}/dtype/{top}/{k}"])
            sd[k] = t
        ck[top] = sd
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(ck, path)
    return ck


def _cfg(**kw):
    from isaacgyminsertion_amd.utils.config import default_config
    cfg = default_config(rl_device=DEV, num_points=8, **kw)
    cfg.train.network.mlp.units = [int(x) for x in G["s1/units"]]
    cfg.train.network.priv_mlp.units = [int(x) for x in G["s1/priv_units"]]
    return cfg


def _assert_same_structure(written, tag):
    """our torch.save'd dict vs the reference-written file: keys in order, dtypes, shapes."""
    assert list(written.keys()) == [str(x) for x in G[f"{tag}/top_keys"]]
    for top, sd in written.items():
        assert list(sd.keys()) == [str(x) for x in G[f"{tag}/keys/{top}"]], top
        for k, v in sd.items():
            assert str(v.dtype) == str(G[f"{tag}/dtype/{top}/{k}"]), (top, k)
            assert tuple(v.shape) == G[f"{tag}/t/{top}/{k}"].shape, (top, k)


def test_teacher_checkpoint_written_by_reference(tmp_path):
    from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO
    from isaacgyminsertion_amd.algo.deploy.deploy_s1 import HardwarePlayer
    fn = str(tmp_path / "stage1_nn" / "last.pth")
    _write_pth("s1", fn)
    obs = torch.from_numpy(G["s1/frames/obs"]).to(DEV)
    priv = torch.from_numpy(G["s1/frames/priv_info"]).to(DEV)
    cfg = _cfg(num_envs=4, horizon_length=4, mini_epochs=2)
    for how in ("restore_test", "restore_train"):
        agent = PPO(None, None, cfg)
        vms_before = agent.value_mean_std.state_dict()
        getattr(agent, how)(fn)
        agent.set_eval()
        mu, latent = agent.model.act_inference({"obs": agent.running_mean_std(obs), "priv_info": agent.priv_mean_std(priv)})
        np.testing.assert_allclose(mu.cpu().numpy(), G["s1/expect/mu"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(latent.cpu().numpy(), G["s1/expect/latent"], atol=2e-5, rtol=0)
        # neither reference restore path touches value_mean_std (frozen_ppo.py:465-484)
        for k, v in agent.value_mean_std.state_dict().items():
            assert torch.equal(v, vms_before[k]), k
    # model_act's de-normalised value (frozen_ppo.py:365) with the value statistics a restored agent has: the fresh
    # ones, on both sides (the golden agent was restored by the reference's restore_test)
    np.testing.assert_allclose(agent.model_act({"obs": obs, "priv_info": priv})["values"].cpu().numpy(),
                               G["s1/expect/value_denorm"], atol=5e-5, rtol=1e-5)
    # deployment player (deploy_s1.py:114-131): batch-1 ticks
    player = HardwarePlayer(cfg)
    player.restore(fn)
    for i in range(obs.shape[0]):
        a, lat = player.policy_step(obs[i:i + 1], priv[i:i + 1])
        np.testing.assert_allclose(a.cpu().numpy()[0], np.clip(G["s1/expect/mu"][i], -1, 1), atol=2e-5, rtol=0)
        np.testing.assert_allclose(lat.cpu().numpy()[0], G["s1/expect/latent"][i], atol=2e-5, rtol=0)
    # and back: what this package writes has the reference file's structure, and round-trips its content bit for bit
    agent.save(str(tmp_path / "ours"))
    ours = torch.load(str(tmp_path / "ours.pth"), map_location="cpu")
    _assert_same_structure(ours, "s1")
    for top, sd in ours.items():
        if top == "value_mean_std":          # never restored (see above): structure only
            continue
        for k, v in sd.items():
            assert np.array_equal(v.numpy(), G[f"s1/t/{top}/{k}"]), (top, k)


def test_student_checkpoint_written_by_reference(tmp_path):
    from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    from isaacgyminsertion_amd.algo.deploy.deploy_s2 import HardwarePlayer
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    s1 = str(tmp_path / "stage1_nn" / "last.pth")
    _write_pth("s1", s1)
    _write_pth("s2", str(tmp_path / "stage2_nn" / "last_stud.pth"))
    frames = {k: torch.from_numpy(G[f"s2/frames/{k}"]).to(DEV) for k in ("student_obs", "tactile", "pcl")}
    cfg = _cfg(num_envs=4, horizon_length=4, mini_epochs=2, obs_info=True, tactile_info=True, pcl_info=True)
    env = SyntheticInsertionEnv(4, device=DEV, tactile_hw=(32, 64), pcl_points=800)
    agent = ExtrinsicAdapt(env, None, cfg)
    agent.restore_test(s1)                       # pulls stage2_nn/last_stud.pth (ext_adapt.py:1087-1099)
    agent.stud_obs_mean_std.eval()               # as the deployment player holds them (golden was made that way)
    agent.pcl_mean_std.eval()
    with torch.no_grad():
        sd = agent.process_obs(frames)
        act, _ = agent.student.predict(sd, requires_grad=False)
    np.testing.assert_allclose(sd["student_obs"].cpu().numpy(), G["s2/expect/student_obs_n"], atol=2e-6, rtol=1e-6)
    np.testing.assert_allclose(sd["pcl"].cpu().numpy(), G["s2/expect/pcl_n"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(act.cpu().numpy(), G["s2/expect/action"], atol=2e-5, rtol=0)
    # the teacher half of restore_test
    mu, _ = agent.agent.act_inference({"obs": agent.running_mean_std(torch.from_numpy(G["s1/frames/obs"]).to(DEV)),
                                       "priv_info": agent.priv_mean_std(torch.from_numpy(G["s1/frames/priv_info"]).to(DEV))})
    np.testing.assert_allclose(mu.cpu().numpy(), G["s1/expect/mu"], atol=2e-5, rtol=0)
    # deployment player (deploy_s2.py:167-217): batch-1 ticks
    cfg.deploy.ppo.obs_info, cfg.deploy.ppo.tactile_info, cfg.deploy.ppo.pcl_info = True, True, True
    player = HardwarePlayer(cfg)
    player.restore(s1)
    for i in range(5):
        a, raw = player.policy_step({k: v[i:i + 1] for k, v in frames.items()})
        np.testing.assert_allclose(raw.cpu().numpy()[0], G["s2/expect/action"][i], atol=2e-5, rtol=0)
        np.testing.assert_allclose(a.cpu().numpy()[0], np.clip(G["s2/expect/action"][i], -1, 1), atol=2e-5, rtol=0)
    # and back: files written by this package have the reference files' structure and content
    agent.save(str(tmp_path / "ours"))
    _assert_same_structure(torch.load(str(tmp_path / "ours.pth"), map_location="cpu"), "s2t")
    ours = torch.load(str(tmp_path / "ours_stud.pth"), map_location="cpu")
    _assert_same_structure(ours, "s2")
    for top, sd_ in ours.items():
        for k, v in sd_.items():
            assert np.array_equal(v.numpy(), G[f"s2/t/{top}/{k}"]), (top, k)
