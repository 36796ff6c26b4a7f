#!/usr/bin/env python
"""Per-launch-shape roofline table of the teacher update from a rocprofv3 per-dispatch kernel trace.

``rocprofv3 --kernel-trace --stats`` averages every launch of a symbol; the teacher step launches the same symbol with
several grids (the four backward levels share ``gemm_dma_wgrad_multi_kernel``, the three trunk layers share
``gemm_dma_kernel<128,true,true,...>``).  This tool groups the dispatches of one ``python3 bench.py`` run by
(kernel symbol, grid size), attaches the ALGORITHMIC flops of the launch that has that grid at the bench's configuration
(BASELINE configs[1]: minibatch 16384, obs 15, priv 64, latent 8, trunk 512-256-128, env_mlp 256-128-8; zero-padded
columns are NOT credited: K = 23 of 32 for the first trunk layer) and prints one CSV row per shape:

    python tools/levels_from_trace.py <..._kernel_trace.csv> > profiles/r04_bench_levels.csv

columns: kernel, grid (workgroups), calls, avg_us, min_us, max_us, what, algorithmic GFLOP per launch, TFLOP/s,
fraction of the 157.3 TFLOP/s fp32-MFMA peak.  Rows without an entry in the table below carry no flops (HBM / issue
bound kernels): their time is what matters.
"""
import collections
import csv
import re
import sys

PEAK = 157.3
MB = 16384


def gf(m, n, k, batch=1):
    return 2.0 * m * n * k * batch / 1e9


# (symbol prefix, workgroups) -> (what, algorithmic GFLOP).  Grids: 128-row tiles, see DESIGN.md section 4.
SHAPES = {
    ("k_env_fwd", 256): ("env_mlp forward 64->256->128->8, one launch", gf(MB, 256, 64) + gf(MB, 128, 256) + gf(MB, 8, 128)),
    # round 6: env_mlp + the first trunk layer of both nets as one persistent launch (csrc/fwd12.h)
    ("k_fwd12", 256): ("env_mlp forward 64->256->128->8 + trunk layer 1 forward 23(32)->512 x2, one persistent launch",
                       gf(MB, 256, 64) + gf(MB, 128, 256) + gf(MB, 8, 128) + gf(MB, 512, 23, 2)),
    ("gemm_dma_kernel<128,true,true", 1024): ("trunk layer 1 forward, 23(32)->512 x2", gf(MB, 512, 23, 2)),
    ("gemm_dma_kernel<128,true,true", 512): ("trunk layer 2 forward, 512->256 x2", gf(MB, 256, 512, 2)),
    ("gemm_dma_kernel<128,true,true", 256): ("trunk layer 3 forward, 256->128 x2 (IGI_LOSS_FUSED=0)", gf(MB, 128, 256, 2)),
    # round 4: the last trunk layer's forward with the heads, the PPO loss and the head backward in its 64-row tiles
    ("k_trunk_loss", 512): ("trunk layer 3 forward 256->128 x2 + heads + PPO loss + head backward",
                            gf(MB, 128, 256, 2) + 3 * gf(MB, 7, 128)),
    ("gemm_dma_wgrad_multi_kernel", 768): ("trunk-3 level: dW 256->128 x2 + dgrad 128->256 x2",
                                           gf(128, 256, MB, 2) + gf(MB, 256, 128, 2)),
    # 640 workgroups since the env layer's weight gradient that shares this launch runs half the split (teacher.h)
    ("gemm_dma_wgrad_multi_kernel", 640): ("env level: dW 23->512 x2 + dW env 256->128 + env dgrad 128->256",
                                           gf(512, 23, MB, 2) + gf(128, 256, MB) + gf(MB, 256, 128)),
    ("gemm_dma_wgrad_multi_kernel", 1280): ("trunk-2 level: dW 512->256 x2 + dgrad 256->512 x2 + latent row dots",
                                            gf(256, 512, MB, 2) + gf(MB, 512, 256, 2) + gf(MB, 8, 512, 2)),
    ("gemm_dma_wgrad_multi_kernel", 256): ("dW env 64->256", gf(256, 64, MB)),
    # round 5: the two 128-wide levels as the persistent row-block kernel (csrc/rowblock.h), one workgroup per CU; the two
    # FIRST layers' weight gradients come from the tiles that produce their dZ (trunk: four chained 128-row tiles per
    # workgroup in the trunk-2 level, 256 + 256 workgroups; env: the second product of k_rb_level<2,1>)
    ("gemm_dma_wgrad_multi_kernel", 512): ("trunk-2 level: dW 512->256 x2 + dgrad 256->512 x2 + latent row dots + dW 23->512 x2 "
                                           "from the dZ1 tiles", gf(256, 512, MB, 2) + gf(MB, 512, 256, 2) + gf(MB, 8, 512, 2) + gf(512, 23, MB, 2)),
    ("k_rb_level<0,2>", 256): ("trunk-3 level (row-block kernel): dW 256->128 x2 + dgrad 128->256 x2",
                               gf(128, 256, MB, 2) + gf(MB, 256, 128, 2)),
    ("k_rb_level<0,1>", 256): ("env level (row-block kernel): dW env 256->128 + env dgrad 128->256",
                               gf(128, 256, MB) + gf(MB, 256, 128)),
    ("k_rb_level<2,1>", 256): ("env level (row-block kernel): dW env 256->128 + env dgrad 128->256 + dW env 64->256 from its tiles",
                               gf(128, 256, MB) + gf(MB, 256, 128) + gf(256, 64, MB)),
    # round 6 (LATZ): + the last env layer's backward at the head of every block (dl . W3 in each of the four column slices,
    # dW3 = dl^T e2 once)
    ("k_rb_level<3,1>", 256): ("env level (row-block kernel): dW env 256->128 + env dgrad 128->256 + dW env 64->256 from its tiles "
                               "+ the 8-wide latent layer's backward at the head of every block",
                               gf(128, 256, MB) + gf(MB, 256, 128) + gf(256, 64, MB) + gf(MB, 128, 8) + gf(8, 128, MB)),
}


def short(name):
    name = re.sub(r"^void ", "", name.split("(")[0]).replace("igi::", "")
    return name.replace(" ", "")


def main():
    rows = collections.OrderedDict()
    seen = collections.Counter()
    recs = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
    for r in recs:
        k = short(r["Kernel_Name"])
        if not (k.startswith("gemm") or k.startswith("k_")):
            continue
        wgs = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * \
              (int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)) * \
              (int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        sub = ""
        rows.setdefault((k, wgs, sub), []).append(us)
    # --skip-first-updates N: leave the first N updates of the process out of every row (bench.py's workspace choice runs one
    # warm-up update + one update per CANDIDATE allocation before anything that counts: the slow candidates' launches do not
    # belong in the averages of the allocation the run then keeps).  Updates are counted by k_gae (one launch per update);
    # a kernel with c launches in u updates loses its first N * round(c / u) launches, kernels launched less than once per
    # update keep everything.
    skip = 0
    if "--skip-first-updates" in sys.argv:
        skip = int(sys.argv[sys.argv.index("--skip-first-updates") + 1])
    n_updates = sum(len(t) for (k, _, _), t in rows.items() if k == "k_gae")
    if skip > 0 and n_updates > skip:
        for key, t in rows.items():
            per = int(round(len(t) / n_updates))
            if per >= 1 and len(t) > skip * per:
                del t[:skip * per]
        print(f"# the first {skip} of {n_updates} updates (workspace choice) left out", file=sys.stdout)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "workgroups", "calls", "avg_us", "min_us", "max_us", "what", "algorithmic_gflop_per_launch",
                "tflops", "frac_of_157.3"])
    for (k, wgs, sub), t in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        what, g = "", None
        for key, (desc, gflop) in SHAPES.items():
            if k.startswith(key[0]) and key[1] == wgs and (len(key) == 2 or key[2] == sub):
                what, g = desc, gflop
        avg = sum(t) / len(t)
        tf = g / avg * 1e3 if g else None          # GFLOP / us = 1000 TFLOP/s
        w.writerow([k, wgs, len(t), round(avg, 2), round(min(t), 2), round(max(t), 2), what,
                    round(g, 3) if g else "", round(tf, 1) if g else "", round(tf / PEAK, 3) if g else ""])


if __name__ == "__main__":
    main()
