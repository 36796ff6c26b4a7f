"""World-size-2 data-parallel semantics on CPU (gloo): gradient SUM then /world, per-rank local
advantage / normaliser statistics and permutations, KL averaged over ranks (frozen_ppo.py:586-603,
624-627).  The oracle run under a real 2-process gloo group must reproduce the two-rank goldens captured
from the reference (tests/golden/make_golden_teacher_dp.py); the stat-aggregation helper of the host API
is exercised under the same group."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_io import load_teacher_dp, rollout_dp


def _worker(rank, port, q):
    try:
        torch.set_num_threads(1)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE="2")
        dist.init_process_group("gloo", rank=rank, world_size=2)
        from oracle import teacher as ot
        from isaacgyminsertion_amd.utils.misc import multi_gpu_aggregate_stats
        g, meta, init = load_teacher_dp()
        orc = ot.TeacherOracle(init, torch.from_numpy(g[f"r{rank}/perm"]), meta["num_envs"], meta["horizon"],
                               meta["mini_epochs"], meta["units"], meta["priv_units"], world_size=2,
                               all_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        orc.prepare(rollout_dp(g, rank))
        st = orc.update()
        got_a = np.array([x.item() for x in st["a_losses"]], dtype=np.float32)
        got_k = np.array([x.item() for x in st["kls"]], dtype=np.float32)
        np.testing.assert_allclose(got_a, g[f"r{rank}/a_losses"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(got_k, g[f"r{rank}/kls"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(orc.flat_params().numpy(), g[f"r{rank}/params_after"], atol=2e-6)
        np.testing.assert_allclose(orc.rms_priv.var.numpy(), g[f"r{rank}/priv_var"], rtol=1e-10)
        # host helper: mean over ranks (utils/misc.py:69-91)
        agg = multi_gpu_aggregate_stats([torch.tensor([float(rank)]), [torch.tensor(1.0 + rank), torch.tensor(3.0)]])
        assert agg[0] == pytest.approx(0.5) and torch.allclose(agg[1], torch.tensor([1.5, 3.0]))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


def test_two_rank_oracle_matches_reference_dp_golden():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _vote_worker(rank, port, q):
    """utils.dist.native_comm_or_none's protocol with a stand-in communicator class over gloo: whatever goes wrong on ONE
    rank (no id, constructor failure, wrong probe sum), BOTH ranks come back with None -- nobody waits in a collective the
    other skipped -- and with nothing wrong both get a communicator."""
    try:
        torch.set_num_threads(1)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE="2")
        os.environ.pop("IGI_DP_NATIVE", None)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        from isaacgyminsertion_amd.utils import dist as D
        D.dist.get_backend = lambda *a, **k: "nccl"          # the helper only goes native over RCCL: pretend, on CPU
        mode = {"m": "ok"}
        closed = []

        class FakeComm:
            def __init__(self, rank=None, world=None, ident=None):
                assert ident == b"x" * 128
                if mode["m"] == "ctor_fails_on_1" and rank == 1:
                    raise RuntimeError("igi_comm_create: simulated")
                self.rank, self.world = rank, world

            @staticmethod
            def draw_id():
                return None if mode["m"] == "no_id" else b"x" * 128

            def all_reduce_(self, t):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                if mode["m"] == "bad_sum_on_1" and self.rank == 1:
                    t.add_(1.0)
                return t

            def close(self):
                closed.append(mode["m"])

        D.NativeComm = FakeComm
        out = {}
        for m in ("ok", "no_id", "ctor_fails_on_1", "bad_sum_on_1", "ok"):
            mode["m"] = m
            c = D.native_comm_or_none("cpu", 2)
            out[m] = c is not None
            if m == "ctor_fails_on_1" and rank == 0:
                assert closed and closed[-1] == m            # the rank that did get a communicator gave it back
            dist.barrier()
        assert out == {"ok": True, "no_id": False, "ctor_fails_on_1": False, "bad_sum_on_1": False}, out
        # IGI_DP_NATIVE=0 and a non-RCCL group: None without touching a collective
        os.environ["IGI_DP_NATIVE"] = "0"
        assert D.native_comm_or_none("cpu", 2) is None
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))


def test_native_comm_rendezvous_never_strands_a_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_vote_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
