"""Deployment players and evaluation loops: checkpoints written by the trainers are
consumed by HardwarePlayer (deploy_s1 / deploy_s2) and the closed-loop policy step at batch 1 -- the robot's
batch -- matches the CPU oracle on the same weights (fp32 tolerance 2e-5 abs on actions in [-1, 1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from isaacgyminsertion_amd.utils.config import default_config
    return default_config(rl_device="cuda:0", num_points=8, **kw)


def test_stage1_player_restores_ppo_checkpoint_and_matches_oracle(tmp_path):
    from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO
    from isaacgyminsertion_amd.algo.deploy.deploy_s1 import HardwarePlayer
    from isaacgyminsertion_amd.algo.deploy import ReplayRobot
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    from oracle import teacher as O
    cfg = _cfg(num_envs=128, horizon_length=8, mini_epochs=2)
    env = SyntheticInsertionEnv(num_envs=128, device="cuda:0")
    agent = PPO(env, str(tmp_path), cfg)
    agent.obs = env.reset()
    agent.train_epoch()                       # moves the weights and the normaliser statistics off their init
    agent.save(str(tmp_path / "stage1_nn" / "last"))
    g = torch.Generator().manual_seed(3)
    frames = [{"obs": torch.randn(1, 15, generator=g), "priv_info": torch.randn(1, 64, generator=g)}
              for _ in range(6)]
    robot = ReplayRobot(frames, episode_length=3)
    player = HardwarePlayer(cfg, robot=robot)
    player.restore(str(tmp_path / "stage1_nn" / "last.pth"))
    assert torch.equal(player.model.flat_params, agent.model.flat_params)
    assert player.deploy(num_episodes=2) == 6
    got = robot.stacked_actions().cpu()[:, 0]
    # oracle: eval-mode normalisers + tanh MLPs on the checkpoint's tensors
    ck = torch.load(str(tmp_path / "stage1_nn" / "last.pth"), map_location="cpu")
    p = {k: v.float() for k, v in ck["model"].items()}

    def norm(sd, x):
        return torch.clamp((x - sd["running_mean"].float()) / torch.sqrt(sd["running_var"].float() + 1e-5), -5, 5)
    for t, f in enumerate(frames):
        mu, _, _, _ = O.actor_critic(p, norm(ck["running_mean_std"], f["obs"]),
                                     norm(ck["priv_mean_std"], f["priv_info"]), 3, 3)
        np.testing.assert_allclose(got[t].numpy(), torch.clamp(mu, -1, 1)[0].numpy(), atol=2e-5, rtol=0)
    # the same rows through the trainer's own batched inference
    obs = torch.cat([f["obs"] for f in frames]).cuda()
    priv = torch.cat([f["priv_info"] for f in frames]).cuda()
    a, latent = player.policy_step(obs, priv)
    np.testing.assert_allclose(a.cpu().numpy(), got.numpy(), atol=2e-6, rtol=0)
    assert latent.shape == (6, 8)
    with pytest.raises(RuntimeError):
        HardwarePlayer(cfg).deploy()


def test_stage2_player_restores_student_checkpoint_and_matches_oracle(tmp_path):
    """lin-only student (the oracle restates that network): ExtrinsicAdapt.save -> HardwarePlayer.restore ->
    batch-1 ticks == numpy forward on the checkpoint; eval-mode normaliser, clamp."""
    from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    from isaacgyminsertion_amd.algo.deploy.deploy_s2 import HardwarePlayer
    from isaacgyminsertion_amd.algo.deploy import ReplayRobot
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    from oracle import offline as OO
    cfg = _cfg(num_envs=64, horizon_length=4, mini_epochs=2, obs_info=True)
    env = SyntheticInsertionEnv(64, device="cuda:0")
    agent = ExtrinsicAdapt(env, str(tmp_path), cfg)
    with torch.no_grad():
        for m in agent.student.model.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
    agent.obs = env.reset()
    agent.train_epoch()
    agent.save(str(tmp_path / "stage2_nn" / "last"))
    cfg.deploy.ppo.tactile_info = False
    cfg.deploy.ppo.pcl_info = False
    g = torch.Generator().manual_seed(5)
    frames = [{"student_obs": 2.0 * torch.randn(1, 15, generator=g)} for _ in range(5)]
    robot = ReplayRobot(frames)
    player = HardwarePlayer(cfg, robot=robot)
    player.restore(str(tmp_path / "stage1_nn" / "last.pth"))        # resolves to stage2_nn/last_stud.pth
    assert player.deploy(num_episodes=1) == 5
    got = robot.stacked_actions().cpu()[:, 0].numpy()
    ck = torch.load(str(tmp_path / "stage2_nn" / "last_stud.pth"), map_location="cpu")
    params = {k: v.numpy() for k, v in ck["student"].items()}
    sd = ck["stud_obs_mean_std"]
    for t, f in enumerate(frames):
        x = torch.clamp((f["student_obs"] - sd["running_mean"].float()) / torch.sqrt(sd["running_var"].float() + 1e-5),
                        -5, 5).numpy()
        out, _ = OO.forward(params, x)
        np.testing.assert_allclose(got[t], np.clip(out[0], -1, 1), atol=2e-5, rtol=0)
    # the normalisers did not move during deployment (eval mode)
    assert player.stud_obs_mean_std.count.item() == sd["count"].item()


def test_stage2_player_visuotactile_tick_equals_trainer_step(tmp_path):
    """tactile + point cloud + proprioception at batch 1: the player's tick equals the trainer's student_act on
    the same observation (bitwise: same kernels, same weights), and the point-cloud normaliser stays frozen."""
    from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    from isaacgyminsertion_amd.algo.deploy.deploy_s2 import HardwarePlayer
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    cfg = _cfg(num_envs=32, horizon_length=4, mini_epochs=2, obs_info=True, tactile_info=True, pcl_info=True)
    env = SyntheticInsertionEnv(32, device="cuda:0", tactile_hw=(32, 64), pcl_points=800)
    agent = ExtrinsicAdapt(env, str(tmp_path), cfg)
    agent.obs = env.reset()
    agent.train_epoch()
    agent.save(str(tmp_path / "stage2_nn" / "last"))
    player = HardwarePlayer(cfg)
    player.restore_student(str(tmp_path / "stage2_nn" / "last_stud.pth"))
    player.set_student_eval()
    obs = env.reset()
    one = {k: v[:1] for k, v in obs.items()}
    a1, raw1 = player.policy_step(one)
    assert a1.shape == (1, 6) and torch.isfinite(a1).all() and a1.abs().max() <= 1
    agent.set_student_eval()
    agent.stud_obs_mean_std.eval()
    agent.pcl_mean_std.eval()
    a_ref, _ = agent.student_act(obs)
    a_all, _ = player.policy_step(obs)
    np.testing.assert_allclose(a_all.cpu().numpy(), a_ref.cpu().numpy(), atol=0, rtol=0)
    np.testing.assert_allclose(a1.cpu().numpy(), a_all[:1].cpu().numpy(), atol=2e-6, rtol=0)
    assert player.pcl_mean_std.count.item() == agent.pcl_mean_std.count.item()


def test_trainer_test_loops_report_success_counts(tmp_path):
    """PPO.test / ExtrinsicAdapt.test / test_log: deterministic policy roll-outs over one episode clock."""
    from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO
    from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    cfg = _cfg(num_envs=64, horizon_length=4, mini_epochs=2, obs_info=True, pcl_info=True)
    env = SyntheticInsertionEnv(64, device="cuda:0", max_episode_length=12, pcl_points=800, done_p=0.05)
    ppo = PPO(env, None, cfg)
    ns, nd = ppo.test()
    assert nd == 64 and 0 <= ns <= nd                     # the clock ran out for every env
    stud = ExtrinsicAdapt(env, str(tmp_path), cfg)
    c0 = stud.stud_obs_mean_std.count.item()
    ns, nd = stud.test()
    assert nd == 64 and 0 <= ns <= nd and abs(stud.test_success - ns / nd) < 1e-9
    assert stud.stud_obs_mean_std.count.item() == c0 + 11 * 64     # train-mode normaliser at test (Appendix A17)
    ns2, nd2 = stud.test(total_steps=3)
    assert nd2 <= 64
    res = stud.test_log(noise_levels=[0.0, 0.01], trials_per_noise=2)
    assert set(res) == {0.0, 0.01} and all(0 <= v["mean"] <= 1 for v in res.values())
    import json, os
    assert "results" in json.load(open(os.path.join(str(tmp_path), "stage2_nn", "pcl_noise_success.json")))
