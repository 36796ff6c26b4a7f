"""Where the host time of one environment step of PPO.play_steps goes (4096 envs): wall time of each call with the GPU
queue drained before and after the whole loop only -- enqueue cost, not device time."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO
from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
from isaacgyminsertion_amd.utils.config import default_config
cfg = default_config(num_envs=4096, horizon_length=32, rl_device="cuda:0")
env = SyntheticInsertionEnv(4096, device="cuda:0")
agent = PPO(env, None, cfg)
agent.obs = env.reset()
agent.set_eval()
for _ in range(3):
    agent.play_steps()
torch.cuda.synchronize()
sd = agent.storage.storage_dict
f32 = dict(dtype=torch.float32, device="cuda:0")
clamped = torch.empty((4096, 6), **f32); values = torch.empty((4096, 1), **f32); meter = torch.zeros((32, 4), **f32)
policy_step, env_store = torch.ops.mi355ppo.rollout_policy_step, torch.ops.mi355ppo.rollout_env_store
state, (icfg, fcfg) = agent.engine.state_list(), agent.engine._cfg_args()
rms_v = agent.value_mean_std._packed
acc = {k: 0.0 for k in ("randn", "policy_step", "env.step", "env_store", "other")}
R = 20
t_all = time.perf_counter()
for _ in range(R):
    for n in range(32):
        t0 = time.perf_counter()
        obs = agent.obs['obs'].to(**f32).contiguous(); priv = agent.obs['priv_info'].to(**f32).contiguous()
        t1 = time.perf_counter()
        noise = torch.randn_like(clamped)
        t2 = time.perf_counter()
        policy_step(state, icfg, fcfg, obs, priv, True, noise, rms_v, sd['obses'][n], sd['priv_info'][n], sd['actions'][n],
                    sd['neglogpacs'][n], sd['values'][n], sd['mus'][n], sd['sigmas'][n], clamped, values)
        t3 = time.perf_counter()
        agent.obs, rewards, dones, infos = env.step(clamped)
        t4 = time.perf_counter()
        touts = infos['time_outs'].view(torch.uint8).contiguous(); succ = infos['successes'].to(**f32).contiguous()
        env_store(rewards, dones.contiguous(), values, touts, succ, 0.99, True, sd['rewards'][n], sd['dones'][n],
                  agent.current_rewards, agent.current_lengths, agent.current_success, meter[n])
        t5 = time.perf_counter()
        acc["other"] += t1 - t0; acc["randn"] += t2 - t1; acc["policy_step"] += t3 - t2; acc["env.step"] += t4 - t3; acc["env_store"] += t5 - t4
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / R
print(json.dumps({"ms_per_rollout_loop": round(wall * 1e3, 3), "host_us_per_step": {k: round(v / R / 32 * 1e6, 1) for k, v in acc.items()}}))
