"""Two-rank data-parallel STUDENT golden vectors from the REFERENCE implementation (build container only).

Each of two processes builds the reference's ``ExtrinsicAdapt`` on CPU with the same seed (identical student), fills
its own ``StudentBuffer`` (data seed + rank) and runs the reference's unmodified ``train_epoch`` with ``multi_gpu``
switched on after construction, so the reference's own student gradient exchange (ext_adapt.py:833-851: flatten the
gradients of the parameters that have one, all-reduce SUM, copy back / rank_size) executes -- over a gloo process
group instead of the hard-coded "nccl" (ext_adapt.py:176).

    python tests/golden/make_golden_student_dp.py   ->  tests/golden/student_dp2.npz
"""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

WORLD = 2
N, T, E = 8, 4, 2


def worker(rank, port, out_dir):
    import ref_harness as rh
    rh.install()
    import make_golden_student as mgs
    from algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    cfg = mgs.student_config(N, T, E, tactile=False, pcl=True)
    env = mgs.FakeEnv(N, False, True)
    torch.manual_seed(9)
    orig_to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as d:
            agent = ExtrinsicAdapt(env, d, cfg)
    finally:
        torch.nn.Module.to = orig_to
    agent.student.device = "cpu"
    agent.multi_gpu, agent.rank, agent.rank_size = True, rank, WORLD     # switch the DP branch on
    model = agent.student.model
    g0 = torch.Generator().manual_seed(10)                               # same on both ranks
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight, generator=g0)
                m.bias.uniform_(-0.1, 0.1, generator=g0)
    out = {}
    if rank == 0:
        out["meta"] = np.array([N, T, E], dtype=np.int64)
        for k, v in model.state_dict().items():
            out[f"init/{k}"] = v.numpy().copy()
    g = torch.Generator().manual_seed(500 + rank)                        # per-rank data
    st = agent.storage
    st.indices = torch.randperm(N * T, generator=torch.Generator().manual_seed(100 + rank))
    for t in range(T):
        st.update_data('n_obs', t, torch.randn(N, 15, generator=g))
        st.update_data('n_priv_info', t, torch.randn(N, 64, generator=g))
        st.update_data('latent_gt', t, torch.randn(N, 8, generator=g))
        st.update_data('teacher_actions', t, torch.rand(N, 6, generator=g) * 2.4 - 1.2)
        st.update_data('student_actions', t, torch.rand(N, 6, generator=g) * 2.4 - 1.2)
        st.update_data('n_student_obs', t, torch.randn(N, 15, generator=g))
        st.update_data('n_pcl', t, (torch.randn(N, 1, 800, 3, generator=g) * 0.5).reshape(N, 1, 2400))
    st.prepare_training()
    for k, v in st.storage_dict.items():
        out[f"in/{k}"] = v.numpy().copy()
    out["perm"] = st.indices.numpy().copy()
    agent.play_steps = lambda: None
    losses, _ = agent.train_epoch()
    out["action_losses"] = np.array([x.item() for x in losses], dtype=np.float32)
    for k, v in model.state_dict().items():
        out[f"final/{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(out_dir, f"_sdp_rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(29541, HERE), nprocs=WORLD, join=True)
    merged = {}
    for r in range(WORLD):
        p = os.path.join(HERE, f"_sdp_rank{r}.npz")
        z = np.load(p)
        for k in z.files:
            if k == "meta" or k.startswith("init/"):
                merged[k] = z[k]
            elif k.startswith("final/"):
                if r == 0:
                    merged[k] = z[k]
                else:
                    assert np.array_equal(merged[k], z[k]), k      # ranks end identical
            else:
                merged[f"r{r}/{k}"] = z[k]
        os.remove(p)
    path = os.path.join(HERE, "student_dp2.npz")
    np.savez_compressed(path, **merged)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); losses r0 {merged['r0/action_losses']} r1 {merged['r1/action_losses']}")
