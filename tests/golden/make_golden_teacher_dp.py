"""Two-rank data-parallel golden vectors from the REFERENCE implementation (build container only).

Each of two processes imports the reference's ``PPO`` on CPU, is initialised with the same seed
(identical parameters, the state after frozen_ppo.py:376-381's broadcast), fills its own rollout
(data seed + rank, train.py:58-64) and runs the reference's unmodified ``train_epoch`` with
``multi_gpu`` switched on after construction, so the reference's own gradient-averaging code
(frozen_ppo.py:586-603, 624-627) executes -- over a gloo process group instead of the hard-coded
"nccl"/cuda device (frozen_ppo.py:121-122 cannot run without a GPU).

    python tests/golden/make_golden_teacher_dp.py   ->  tests/golden/teacher_dp2.npz
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

WORLD = 2
CASE = dict(num_envs=32, horizon=8, mini_epochs=4, units=(64, 48, 32), priv_units=(48, 32, 8))


def worker(rank, port, out_dir):
    import ref_harness as rh
    rh.install()
    from algo.ppo.frozen_ppo import PPO
    from make_golden_teacher import synth_fill, ref_tail, flat_params
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    cfg = rh.teacher_config(CASE["num_envs"], CASE["horizon"], CASE["mini_epochs"], units=CASE["units"],
                            priv_units=CASE["priv_units"])
    torch.manual_seed(42)
    agent = PPO(None, None, cfg)
    agent.multi_gpu, agent.rank, agent.rank_size = True, rank, WORLD   # switch the DP branches on
    # each rank draws its own permutation in the reference (per-rank seed); make it explicit
    agent.storage.indices = torch.randperm(agent.batch_size, generator=torch.Generator().manual_seed(100 + rank))
    gen = torch.Generator().manual_seed(1234 + rank)
    out = {"perm": agent.storage.indices.numpy().copy()}
    if rank == 0:
        out["meta"] = np.array([CASE["num_envs"], CASE["horizon"], CASE["mini_epochs"], 1], dtype=np.int64)
        out["units"] = np.array(CASE["units"], dtype=np.int64)
        out["priv_units"] = np.array(CASE["priv_units"], dtype=np.int64)
        for k, v in agent.model.state_dict().items():
            out[f"init/{k}"] = v.numpy().copy()

    def fake_play_steps():
        last_values = synth_fill(agent, gen, 0.05)
        for k in ["obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus", "sigmas"]:
            out[f"in/{k}"] = agent.storage.storage_dict[k].numpy().copy()
        out["in/last_values"] = last_values.numpy().copy()
        ref_tail(agent, last_values)

    agent.play_steps = fake_play_steps
    a_losses, c_losses, b_losses, entropies, kls, grad_norms, _ = agent.train_epoch()
    out["a_losses"] = np.array([x.item() for x in a_losses], dtype=np.float32)
    out["c_losses"] = np.array([x.item() for x in c_losses], dtype=np.float32)
    out["kls"] = np.array([x.item() for x in kls], dtype=np.float32)
    out["params_after"] = flat_params(agent.model)
    out["priv_var"] = agent.priv_mean_std.running_var.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f"_dp_rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(29533, HERE), nprocs=WORLD, join=True)
    merged = {}
    for r in range(WORLD):
        p = os.path.join(HERE, f"_dp_rank{r}.npz")
        z = np.load(p)
        for k in z.files:
            if k in ("meta", "units", "priv_units") or k.startswith("init/"):
                merged[k] = z[k]
            else:
                merged[f"r{r}/{k}"] = z[k]
        os.remove(p)
    path = os.path.join(HERE, "teacher_dp2.npz")
    np.savez_compressed(path, **merged)
    same = np.array_equal(merged["r0/params_after"], merged["r1/params_after"])
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); ranks end with identical parameters: {same}")
