#!/usr/bin/env python
"""Student-distillation update throughput on one GPU (BASELINE configs[2] / configs[3]),
the companion of bench.py's headline teacher metric (bench.py runs ``student_bench`` for its ``student`` section).
One "update" = mini_epochs x n_minibatch optimizer steps of ExtrinsicAdapt.update() on a synthetic StudentBuffer
resident in HBM.

    python tools/bench_student.py --config 3 [--hw 32 64] [--envs 2048] [--updates 2]
    python tools/bench_student.py --config 4 --envs 512        # tactile + PointNet(plug+socket) + lin
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_F32_MFMA_TFLOPS = 157.3

# forward MACs per sample (SURVEY section 8d "Algorithmic FLOPs"; train = 6 x forward MACs)
MAC_TACTILE = {(32, 64): 17_917_952, (64, 64): 48_556_032}
MAC_POINTNET_OBJ, MAC_PCL_COMPRESS, MAC_LIN = 6_630_400, 34_816, 3_008
MAC_DEC_TRANSFORMER_S3, MAC_DEC_OUT_S3, MAC_HEAD = 73_728, 54_464, 192


def algorithmic_macs(config, hw):
    """forward MACs per sample of the student of ``config`` (3: tactile + lin; 4: tactile + pcl + lin)."""
    if config == 5:
        return None                                   # depth/segmentation student: not a BASELINE config, no figure
    m = MAC_TACTILE.get(tuple(hw))
    if m is None:
        return None
    m += MAC_LIN + MAC_HEAD
    if config == 4:
        m += 2 * MAC_POINTNET_OBJ + MAC_PCL_COMPRESS + MAC_DEC_TRANSFORMER_S3 + MAC_DEC_OUT_S3
    else:                                             # two tokens: 2/3 of the S = 3 token-transformer work, S*32 -> 32 first layer
        m += MAC_DEC_TRANSFORMER_S3 * 2 // 3 + (MAC_DEC_OUT_S3 - 32 * 32)
    return m


def student_bench(config=3, envs=2048, horizon=32, hw=(32, 64), updates=2, device="cuda:0", multi_gpu=False,
                  profile=True):
    """multi_gpu: one call per rank under torchrun (the process group is up): per-rank synthetic buffers (seed + rank),
    rank 0's initial student on every rank, ExtrinsicAdapt.update() with its gradient exchange; the time is the maximum
    over the ranks between barriers and the record says whether the parameter vectors ended bit-identical."""
    import torch
    import torch.distributed as dist
    from isaacgyminsertion_amd import _lib
    from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt
    from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv
    from isaacgyminsertion_amd.utils.config import default_config
    H, W = hw
    pcl = config == 4
    img = config == 5
    rank = dist.get_rank() if multi_gpu else 0
    world = dist.get_world_size() if multi_gpu else 1
    if multi_gpu:                                     # this rank's GPU (the synthetic environment is built before the agent)
        from isaacgyminsertion_amd.utils.dist import init_rank_device
        device = init_rank_device()[2]
    cfg = default_config(num_envs=envs, horizon_length=horizon, rl_device=device, multi_gpu=multi_gpu, obs_info=True,
                         tactile_info=not img, pcl_info=pcl, img_info=img, seg_info=img, num_points=8)
    cfg.offline_train.tactile_width, cfg.offline_train.tactile_height = H, W
    env = SyntheticInsertionEnv(envs, device=device, tactile_hw=None if img else (H, W),
                                pcl_points=800 if pcl else 0, img_hw=(54, 96) if img else None)
    agent = ExtrinsicAdapt(env, None, cfg)
    device = agent.device
    g = torch.Generator(device=device).manual_seed(rank)
    st = agent.storage.storage_dict
    if img:
        st["n_img"].uniform_(0, 1, generator=g)
        st["n_seg"].uniform_(0, 3, generator=g).round_()
    else:
        st["n_tactile"].uniform_(0, 1, generator=g)
    st["n_student_obs"].normal_(generator=g)
    st["teacher_actions"].uniform_(-1.2, 1.2, generator=g)
    if pcl:
        st["n_pcl"].normal_(0, 0.5, generator=g)
    with torch.no_grad():   # O(1)-scale student
        for m in agent.student.model.modules():
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.kaiming_uniform_(m.weight, a=5 ** 0.5)
    agent.storage.prepare_training()
    agent.set_student_train()
    if multi_gpu:
        dist.broadcast(agent.optim.flat, 0)          # ext_adapt.py:861-870: every rank starts from rank 0's student

    def fence():
        if multi_gpu:
            dist.barrier()
        torch.cuda.synchronize()

    agent.update()   # warm-up
    fence()
    t0 = time.perf_counter()
    for _ in range(updates):
        losses, _ = agent.update()
    fence()
    dt = (time.perf_counter() - t0) / updates
    identical = None
    if multi_gpu:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        bits = agent.optim.flat.view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=device) % 8191 + 1)).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        identical = bool((lo == hi).all().item())
    # per-kernel figures from a separate instrumented update (the dispatch timestamps serialise nothing, but keep the
    # timed region clean)
    kern = []
    if profile:
        _lib.prof_enable(True)
        agent.update()
        torch.cuda.synchronize()
        kern = _lib.prof_read()
        _lib.prof_enable(False)
    steps = agent.mini_epochs_num * len(agent.storage)
    macs = algorithmic_macs(config, hw)
    out = {"workload": f"student distillation: "
                       + ("depth 54x96 + segmentation 54x96" if img else f"tactile {H}x{W}") + (" + pcl 2x400" if pcl else "")
                       + f" + lin, {envs} envs x {horizon}, minibatch {agent.minibatch_size}, {steps} optimizer steps per update",
           "updates_per_s": round(1.0 / dt, 4), "ms_per_update": round(1e3 * dt, 1),
           "ms_per_optimizer_step": round(1e3 * dt / steps, 2),
           "samples_per_s": round(world * envs * horizon * agent.mini_epochs_num / dt),
           "finite": bool(torch.isfinite(torch.stack(losses)).all())}
    if multi_gpu:
        comm = getattr(agent, "_comm", None)
        out["n_gpus"] = world
        out["envs_per_gpu"] = envs
        out["params_identical_across_ranks"] = identical
        if comm is not None:
            try:
                out["rccl_ranks"], out["rccl_version"] = comm.rccl_ranks(), comm.rccl_version()
            except Exception as e:   # noqa: BLE001
                out["rccl_ranks"], out["rccl_error"] = None, f"{type(e).__name__}: {e}"
        out["grad_allreduce"] = ("rccl issued by libigi_hip.so" if comm is not None else "torch.distributed") + \
            (", decoder-side bucket overlapped with the encoders' backward"
             if comm is not None and getattr(agent.optim, "_early_cb", None) is not None else ", serial")
    if macs:
        fl = 6.0 * macs * world * envs * horizon * agent.mini_epochs_num
        out["algorithmic_tflop_per_update"] = round(fl / 1e12, 2)
        out["tflops"] = round(fl / dt / 1e12, 2)                 # whole job, all `world` GPUs
        # the fraction is PER GPU: the job's flops over the job's `world` peaks (round 5 divided the aggregate by ONE GPU's
        # peak, which reads ~`world` on a multi-GPU record)
        out["tflops_per_gpu"] = round(fl / world / dt / 1e12, 2)
        out["frac_of_f32_mfma_peak"] = round(fl / world / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    if kern:
        # "operand_gbs": the launch site's algorithmic operand bytes (im2col operands counted once per tap) over the
        # launch time -- NOT HBM traffic (it exceeds the 8 TB/s peak for the convolutions); the HBM-side bytes of these
        # kernels are the PMC figures under profiles/*_student_c*_hbm_traffic.json
        out["native_kernels"] = [{"name": k["name"], "launches_per_update": k["launches"],
                                  "ms_per_update": round(k["total_ms"], 2),
                                  "avg_us": round(1e3 * k["total_ms"] / max(k["launches"], 1), 1),
                                  "tflops": round(k["flops"] / max(k["total_ms"], 1e-9) / 1e9, 1),
                                  "operand_gbs": round(k["bytes"] / max(k["total_ms"], 1e-9) / 1e6, 1)}
                                 for k in sorted(kern, key=lambda k: -k["total_ms"])[:12]]
    del agent, env
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="3 tactile+lin, 4 tactile+pcl+lin, 5 depth+segmentation+lin (README.md:153-155)")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--hw", type=int, nargs=2, default=[32, 64])
    ap.add_argument("--updates", type=int, default=2)
    args = ap.parse_args()
    print(json.dumps(student_bench(args.config, args.envs, args.horizon, tuple(args.hw), args.updates)))


if __name__ == "__main__":
    main()
