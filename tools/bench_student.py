#!/usr/bin/env python
"""Student-distillation update throughput on one GPU, as a companion to
bench.py (which measures the headline teacher metric).  One "update" = mini_epochs x n_minibatch
optimizer steps of ExtrinsicAdapt.update() on a synthetic StudentBuffer resident in HBM.

    python tools/bench_student.py --config 3 [--hw 32 64] [--envs 2048] [--updates 2]
    python tools/bench_student.py --config 4 --envs 512        # tactile + PointNet(plug+socket) + lin
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from isaacgyminsertion_amd import _lib  # noqa: E402
from isaacgyminsertion_amd.algo.ext_adapt.ext_adapt import ExtrinsicAdapt  # noqa: E402
from isaacgyminsertion_amd.envs.synthetic import SyntheticInsertionEnv  # noqa: E402
from isaacgyminsertion_amd.utils.config import default_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="3 tactile+lin, 4 tactile+pcl+lin, 5 depth+segmentation+lin (README.md:153-155)")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--hw", type=int, nargs=2, default=[32, 64])
    ap.add_argument("--updates", type=int, default=2)
    args = ap.parse_args()
    H, W = args.hw
    pcl = args.config == 4
    img = args.config == 5
    cfg = default_config(num_envs=args.envs, horizon_length=args.horizon, rl_device="cuda:0", obs_info=True,
                         tactile_info=not img, pcl_info=pcl, img_info=img, seg_info=img, num_points=8)
    cfg.offline_train.tactile_width, cfg.offline_train.tactile_height = H, W
    env = SyntheticInsertionEnv(args.envs, device="cuda:0", tactile_hw=None if img else (H, W),
                                pcl_points=800 if pcl else 0, img_hw=(54, 96) if img else None)
    agent = ExtrinsicAdapt(env, None, cfg)
    g = torch.Generator(device="cuda").manual_seed(0)
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
    agent.update()   # warm-up
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.updates):
        losses, _ = agent.update()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.updates
    kern = _lib.prof_read()
    _lib.prof_enable(False)
    steps = agent.mini_epochs_num * len(agent.storage)
    out = {"workload": f"student distillation config {args.config}: "
                       + ("depth 54x96 + segmentation 54x96" if img else f"tactile {H}x{W}") + (" + pcl 2x400" if pcl else "")
                       + f" + lin, {args.envs} envs x {args.horizon}, minibatch {agent.minibatch_size}",
           "updates_per_s": round(1.0 / dt, 4), "ms_per_update": round(1e3 * dt, 1),
           "ms_per_optimizer_step": round(1e3 * dt / steps, 2),
           "samples_per_s": round(args.envs * args.horizon * agent.mini_epochs_num / dt),
           "final_loss": float(torch.stack(losses).mean()),
           "native_kernels": [{"name": k["name"], "launches_per_update": k["launches"] // args.updates,
                               "ms_per_update": round(k["total_ms"] / args.updates, 2),
                               "tflops": round(k["flops"] / max(k["total_ms"], 1e-9) / 1e9, 1)}
                              for k in sorted(kern, key=lambda k: -k["total_ms"])]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
