"""Rollout-bookkeeping golden vectors from the REFERENCE implementation (build container only).

Runs the reference's own, unmodified ``PPO.play_steps`` (frozen_ppo.py:648-725) and
``ExtrinsicAdapt.play_steps`` (ext_adapt.py:658-767) on CPU against a *scripted* environment: the
observations, rewards, dones, time-outs and successes of every step are pre-drawn tensors, so the only
thing under test is what the trainers do with them (policy sampling, neglogp, value de-normalisation,
arena writes, shaped reward ``0.01 r + gamma V timeout``, episode accumulators, windowed meters, ingest-time
normaliser updates, DAgger action choice, the post-rollout tail).  The Gaussian noise of
``torch.distributions.Normal.sample`` is replaced by a pre-drawn tensor per step so that the HIP path can
replay the same draws.

    python tests/golden/make_golden_rollout.py  ->  tests/golden/rollout.npz
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
import make_golden_student as mgs  # noqa: E402  (harness pieces: config, transforms patch; imports ExtrinsicAdapt)
from algo.ext_adapt.ext_adapt import ExtrinsicAdapt  # noqa: E402  (reference)


class ScriptedEnv:
    """Replays pre-drawn step results; records the actions it is given."""

    def __init__(self, script, extra_obs=(), queues=None):
        self.s = script
        self.t = 0
        self.extra = tuple(extra_obs)
        self.actions = []
        for k, v in (queues or {}).items():
            setattr(self, k, v)
        self.cfg_task = rh.to_attr({"env": {"record_video_every": 10 ** 9, "record_ft_every": 10 ** 9},
                                    "data_logger": {"collect_data": False}, "external_cam": {"display": False}})

    def obs(self):
        d = {"obs": self.s["obs"][self.t], "priv_info": self.s["priv_info"][self.t]}
        for k in self.extra:
            d[k] = self.s[k][self.t]
        return d

    def reset(self, **k):
        self.t = 0
        return self.obs()

    def step(self, actions):
        self.actions.append(actions.clone())
        t = self.t
        self.t += 1
        infos = {"time_outs": self.s["time_outs"][t], "successes": self.s["successes"][t],
                 "scalar_metric": 0.5 + t}          # a scalar entry: lands in extra_info
        return self.obs(), self.s["rewards"][t], self.s["dones"][t], infos

    # recording hooks used by log_video (frozen_ppo.py:814-851): nothing recorded
    def start_recording(self): pass
    def start_recording_ft(self): pass
    def pause_recording(self): pass
    def pause_recording_ft(self): pass
    def get_complete_frames(self): return []
    def get_ft_frames(self): return []


def draw_script(g, N, T, done_p=0.3):
    s = {"obs": torch.randn(T + 1, N, 15, generator=g), "priv_info": torch.randn(T + 1, N, 64, generator=g),
         "rewards": torch.randn(T, N, generator=g),
         "dones": (torch.rand(T, N, generator=g) < done_p).long(),          # vec_task reset_buf is int64
         "successes": (torch.rand(T, N, generator=g) < 0.5).float(),
         "noise": torch.randn(T + 1, N, 6, generator=g)}
    s["time_outs"] = (s["dones"] > 0) & (torch.rand(T, N, generator=g) < 0.5)   # bool, only where done
    return s


class FixedNoise:
    """Normal.sample() -> loc + scale * noise[call index] for the duration of the context."""

    def __init__(self, noise):
        self.noise, self.i = noise, 0

    def __enter__(self):
        self.orig = torch.distributions.Normal.sample
        me = self

        def sample(dist, sample_shape=torch.Size()):
            e = me.noise[me.i]
            me.i += 1
            return dist.loc + dist.scale * e

        torch.distributions.Normal.sample = sample
        return self

    def __exit__(self, *a):
        torch.distributions.Normal.sample = self.orig


def set_rms(m, g):
    """non-trivial normaliser state (as after some training)"""
    with torch.no_grad():
        m.running_mean.copy_(0.3 * torch.randn(m.running_mean.shape, generator=g, dtype=torch.float64))
        m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g, dtype=torch.float64))
        m.count.fill_(1000.0)


def rms_state(m):
    return np.concatenate([m.running_mean.numpy().reshape(-1), m.running_var.numpy().reshape(-1),
                           np.array([m.count.item()])]).astype(np.float64)


def meter_state(m):
    return np.array([float(m.mean), float(m.current_size)], dtype=np.float64)


def teacher_case(out, tag, N, T, seed, rollouts=2):
    units, priv_units = (64, 48, 32), (48, 32, 8)
    cfg = rh.teacher_config(N, T, 2, units=units, priv_units=priv_units)
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    scripts = [draw_script(g, N, T) for _ in range(rollouts)]
    env = ScriptedEnv(scripts[0])
    with tempfile.TemporaryDirectory() as d:
        agent = PPO(env, d, cfg)
        with torch.no_grad():
            agent.model.sigma.copy_(0.2 * torch.randn(6, generator=g))
            agent.model.mu.weight.mul_(30.0)        # std-0.01 init -> actions that reach the +-1 clamp
        for m in (agent.running_mean_std, agent.priv_mean_std, agent.value_mean_std):
            set_rms(m, g)
        out[f"{tag}/meta"] = np.array([N, T, rollouts], dtype=np.int64)
        out[f"{tag}/units"] = np.array(units, dtype=np.int64)
        out[f"{tag}/priv_units"] = np.array(priv_units, dtype=np.int64)
        for k, v in agent.model.state_dict().items():
            out[f"{tag}/init/{k}"] = v.numpy().copy()
        for nm in ("running_mean_std", "priv_mean_std", "value_mean_std"):
            out[f"{tag}/rms_in/{nm}"] = rms_state(getattr(agent, nm))
        agent.set_eval()
        agent.agent_steps = agent.batch_size
        for r, sc in enumerate(scripts):
            env.s, env.t, env.actions = sc, 0, []
            agent.obs = env.reset()
            for k, v in sc.items():
                out[f"{tag}/r{r}/script/{k}"] = v.numpy().copy()
            with FixedNoise(sc["noise"]):
                agent.play_steps()
            for k in ["obses", "priv_info", "actions", "neglogpacs", "values", "mus", "sigmas", "dones", "rewards",
                      "returns"]:
                out[f"{tag}/r{r}/storage/{k}"] = agent.storage.storage_dict[k].numpy().copy()
            dd = agent.storage.data_dict
            out[f"{tag}/r{r}/advantages"] = dd["advantages"].numpy().copy()
            out[f"{tag}/r{r}/values_norm"] = dd["values"].numpy().copy()
            out[f"{tag}/r{r}/returns_norm"] = dd["returns"].numpy().copy()
            out[f"{tag}/r{r}/env_actions"] = torch.stack(env.actions).numpy().copy()
            out[f"{tag}/r{r}/current_rewards"] = agent.current_rewards.numpy().copy()
            out[f"{tag}/r{r}/current_lengths"] = agent.current_lengths.numpy().copy()
            out[f"{tag}/r{r}/current_success"] = agent.current_success.numpy().copy()
            for nm in ("episode_rewards", "episode_lengths", "episode_success"):
                out[f"{tag}/r{r}/meter/{nm}"] = meter_state(getattr(agent, nm))
            out[f"{tag}/r{r}/value_mean_std"] = rms_state(agent.value_mean_std)
            out[f"{tag}/r{r}/agent_steps"] = np.array(agent.agent_steps, dtype=np.int64)
            out[f"{tag}/r{r}/extra_info"] = np.array(agent.extra_info["scalar_metric"], dtype=np.float64)
        # eval-mode obs / priv normalisers must be untouched by rollouts
        for nm in ("running_mean_std", "priv_mean_std"):
            assert np.array_equal(rms_state(getattr(agent, nm)), out[f"{tag}/rms_in/{nm}"])
    print(tag, "meters", out[f"{tag}/r{rollouts - 1}/meter/episode_rewards"], "clamped frac",
          float((np.abs(out[f"{tag}/r0/storage/actions"]) > 1).mean()))


def student_case(out, tag, N, T, seed, tactile, pcl, agent_steps0):
    cfg = mgs.student_config(N, T, 2, tactile, pcl)
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 100)
    sc = draw_script(g, N, T)
    sc["student_obs"] = torch.randn(T + 1, N, 15, generator=g)
    extra = ["student_obs"]
    queues = {"tactile_queue": None, "pcl_queue": None, "img_queue": None, "seg_queue": None}
    if tactile:
        sc["tactile"] = torch.rand(T + 1, N, 1, 3, 2048, generator=g)
        queues["tactile_queue"] = torch.zeros(N, 1, 3, 2048)
        extra.append("tactile")
    if pcl:
        sc["pcl"] = (0.05 * torch.randn(T + 1, N, 1, 800, 3, generator=g) + torch.tensor([0.5, 0.0, 0.1])).reshape(T + 1, N, 1, 2400)
        queues["pcl_queue"] = torch.zeros(N, 1, 2400)
        extra.append("pcl")
    env = ScriptedEnv(sc, extra, queues)
    orig_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as d:
            agent = ExtrinsicAdapt(env, d, cfg)
    finally:
        torch.nn.Module.to = orig_to
    agent.student.device = "cpu"
    agent.student.eval_process_tactile = lambda t: t
    model = agent.student.model
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight, generator=g)
                m.bias.uniform_(-0.1, 0.1, generator=g)
        agent.agent.sigma.copy_(0.2 * torch.randn(6, generator=g))
        agent.agent.mu.weight.mul_(30.0)
    for m in (agent.running_mean_std, agent.priv_mean_std, agent.stud_obs_mean_std):
        set_rms(m, g)
    if pcl:
        set_rms(agent.pcl_mean_std, g)
    out[f"{tag}/meta"] = np.array([N, T, int(tactile), int(pcl), int(agent_steps0)], dtype=np.int64)
    for k, v in model.state_dict().items():
        out[f"{tag}/student/{k}"] = v.numpy().copy()
    for k, v in agent.agent.state_dict().items():
        out[f"{tag}/teacher/{k}"] = v.numpy().copy()
    for nm in ("running_mean_std", "priv_mean_std", "stud_obs_mean_std") + (("pcl_mean_std",) if pcl else ()):
        out[f"{tag}/rms_in/{nm}"] = rms_state(getattr(agent, nm))
    for k, v in sc.items():
        out[f"{tag}/script/{k}"] = v.numpy().copy()
    agent.set_student_eval()
    agent.agent_steps = agent_steps0
    agent.obs = env.reset()
    with FixedNoise(sc["noise"]):
        agent.play_steps()
    for k, v in agent.storage.storage_dict.items():
        out[f"{tag}/storage/{k}"] = v.numpy().copy()
    out[f"{tag}/env_actions"] = torch.stack(env.actions).numpy().copy()
    out[f"{tag}/step_reward"] = agent.step_reward.numpy().copy()
    out[f"{tag}/step_length"] = agent.step_length.numpy().copy()
    out[f"{tag}/step_success"] = agent.step_success.numpy().copy()
    for nm in ("mean_eps_reward", "mean_eps_length", "mean_eps_success"):
        out[f"{tag}/meter/{nm}"] = meter_state(getattr(agent, nm))
    for nm in ("running_mean_std", "priv_mean_std", "stud_obs_mean_std") + (("pcl_mean_std",) if pcl else ()):
        out[f"{tag}/rms_out/{nm}"] = rms_state(getattr(agent, nm))
    out[f"{tag}/agent_steps"] = np.array(agent.agent_steps, dtype=np.int64)
    same = np.allclose(out[f"{tag}/env_actions"], np.clip(out[f"{tag}/storage/teacher_actions"], -1, 1))
    print(tag, "env driven by", "teacher" if same else "student", "meters", out[f"{tag}/meter/mean_eps_reward"])


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    teacher_case(out, "ppo", N=16, T=6, seed=0)                      # two consecutive rollouts (meter / accumulator carry-over)
    student_case(out, "s2_lin", N=16, T=6, seed=1, tactile=False, pcl=False, agent_steps0=16 * 6)
    # tactile student past the DAgger horizon: beta = 0, the STUDENT's clamped action drives the env (:718-728)
    student_case(out, "s2_tac_pcl", N=4, T=3, seed=2, tactile=True, pcl=True, agent_steps0=int(4e6))
    path = os.path.join(HERE, "rollout.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")
