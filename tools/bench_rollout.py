#!/usr/bin/env python
"""Rollout-side throughput: PPO.play_steps with the synthetic environment (policy inference +
sampling + buffer writes + GAE / prepare at the end) and the bare inference call, 4096 envs x 32 steps.
    python tools/bench_rollout.py [--envs 4096] [--horizon 32]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO
from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
from isaacgyminsertion_amd.utils.config import default_config

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--horizon", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
cfg = default_config(num_envs=a.envs, horizon_length=a.horizon, rl_device="cuda:0")
env = SyntheticInsertionEnv(a.envs, device="cuda:0")
agent = PPO(env, None, cfg)
agent.obs = env.reset()
agent.set_eval()
for _ in range(2):
    agent.play_steps()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    agent.play_steps()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
obs = agent.obs
for _ in range(5):
    agent.model_act(obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    agent.model_act(obs)
torch.cuda.synchronize()
di = (time.perf_counter() - t0) / 200
print(json.dumps({"workload": f"PPO.play_steps, {a.envs} envs x {a.horizon} steps, synthetic environment",
                  "ms_per_rollout": round(dt * 1e3, 3), "env_steps_per_s": round(a.envs * a.horizon / dt),
                  "model_act_us": round(di * 1e6, 1), "policy_inferences_per_s": round(a.envs / di)}))
