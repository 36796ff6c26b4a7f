#!/usr/bin/env python
"""Headline benchmark: PPO update steps/sec, teacher PPO, 4096 envs x 32 horizon per GPU
(BASELINE.json configs[1]; BASELINE.md section 3 synthetic arena).

One "step" = one PPO update = GAE + advantage/value normalisation + mini_epochs x n_minibatch
(8 x 8 = 64) optimizer steps (gather, running-stat update, forward, clipped-surrogate loss, backward,
[gradient all-reduce], global-norm clip, Adam, mu/sigma write-back); environment stepping excluded
(frozen_ppo.py:495-646, experience.py:242-263).  The rollout arena is resident in HBM before the timed
region.  N > 1: one process per GPU (torchrun), every rank owns its own 4096-env arena (weak scaling,
as the reference gives every rank its own numEnvs) and the flat gradient is all-reduced with RCCL on
every optimizer step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline] [--no-roofline] [--no-student]

The JSON line also carries ``roofline`` (dominant kernel: algorithmic flops / the kernel's own dispatch-timestamp
duration, the same figure recomputed from the tracked rocprofv3 summary under profiles/, and ``levels``: the backward
levels that share that symbol, each with its own GFLOP / us / fraction), ``cpu_baseline`` (the oracle on this host's
cores: 1 warm-up update + the median of 5 full updates, CPU model printed) and ``student`` (BASELINE configs[2] and the
single-rank share of configs[3], a few updates each, outside ``value``).
"""
import argparse
import csv
import json
import os
import sys
import time

# dmabuf IPC is the only mode the host driver supports; the runtime reads this when HSA initialises (first GPU call)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM_ENVS, HORIZON, MINI_EPOCHS = 4096, 32, 8
UNITS, PRIV_UNITS = [512, 256, 128], [256, 128, 8]
OBS, PRIV, ACT = 15, 64, 6
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix), spec
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (same guide); AMD's headline figure includes 2:1 sparsity
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak BW, spec
N_PARAMS = 404501              # ActorCriticSplit with the default widths


def fwd_macs():
    m, d = 0, PRIV
    for u in PRIV_UNITS:
        m += d * u
        d = u
    for _ in range(2):
        d = OBS + PRIV_UNITS[-1]
        for u in UNITS:
            m += d * u
            d = u
    return m + UNITS[-1] * (ACT + 1)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(init, ro, perm, budget_s=75.0):
    """Oracle (PyTorch-CPU restatement of the reference loop, pinned to the reference's goldens) timed on this host,
    BASELINE.md section 3 protocol: CPU model and core count printed, the intra-op thread count calibrated first (one
    optimizer step per candidate: eager ATen on 16384-row minibatches gets SLOWER with hundreds of threads, the
    fastest setting is the fair baseline), then 1 warm-up update and the MEDIAN of 5 full updates (prepare + all 64
    optimizer steps each, from the same initial state).  Only if that would not fit the time budget is the number of
    timed updates reduced (>= 1; the line says how many)."""
    import statistics
    import torch
    from oracle import teacher as ot
    cores = os.cpu_count() or 1
    cands = sorted({min(cores, c) for c in (8, 16, 32, 64)})

    def fresh():
        return ot.TeacherOracle(init, perm, NUM_ENVS, HORIZON, MINI_EPOCHS, UNITS, PRIV_UNITS)

    best_t, best_n = None, cands[0]
    for n in cands:
        torch.set_num_threads(n)
        orc = fresh()
        orc.prepare(ro)
        orc.update(max_steps=1)  # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        orc.update(max_steps=1, start_step=1)
        t = time.perf_counter() - t0
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        if t > 8.0:  # this and larger settings are hopeless; stay inside the time budget
            break
    torch.set_num_threads(best_n)
    steps = MINI_EPOCHS * MINI_EPOCHS

    def one_update():
        orc = fresh()
        t0 = time.perf_counter()
        orc.prepare(ro)
        orc.update(max_steps=steps)
        return time.perf_counter() - t0

    warm = one_update()                                        # 1 warm-up update
    n_timed = max(1, min(5, int((budget_s - warm) / max(warm, 1e-3))))
    times = [one_update() for _ in range(n_timed)]
    med = statistics.median(times)
    return {"value": round(1.0 / med, 5), "unit": "updates/s", "cores": best_n, "kind": "port",
            "host_cores": cores, "cpu_model": cpu_model(),
            "sample": f"oracle/teacher.py (PyTorch-CPU restatement of frozen_ppo.py:495-646) at minibatch "
                      f"{NUM_ENVS * HORIZON // MINI_EPOCHS} with {best_n} threads (fastest of {cands}): 1 warm-up update "
                      f"({warm:.2f}s), then the median of {n_timed} full updates (prepare + all {steps} optimizer steps)",
            "s_per_update": round(med, 2), "s_per_update_all": [round(t, 2) for t in times],
            "updates_timed": n_timed, "full_update_timed": True}


PROFILE_TAG = "r06"


def profile_build(path):
    """the build hash tools/stamp_profiles.py wrote into a profile summary (JSON key "build" / CSV first line), or None"""
    from tools.stamp_profiles import read_build
    return read_build(path)


def profile_is_current(path):
    """True when the summary was collected with the library this process has loaded (igi_build_info() == its stamp)"""
    from isaacgyminsertion_amd import _lib
    want = _lib.lib().igi_build_info().decode()
    return profile_build(path) == want


def rocprof_row(kernel):
    """(calls, average ns) of ``kernel`` in the tracked ``rocprofv3 --kernel-trace --stats`` summary of this same
    command (profiles/<tag>_bench_kernel_stats.csv); our class names are prefixes of the demangled symbols."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_kernel_stats.csv")
    want = kernel.replace(" ", "").rstrip(">").rstrip("*").rstrip("<")
    calls, total = 0, 0.0
    try:
        with open(path) as f:
            for r in csv.DictReader(line for line in f if not line.startswith("#")):
                name = r["Name"].replace(" ", "")
                name = name[name.find("igi::") + 5:] if "igi::" in name else name
                if name.startswith(want):
                    calls += int(r["Calls"])
                    total += float(r["TotalDurationNs"])
    except (OSError, KeyError, ValueError):
        return None
    return (calls, total / calls, os.path.relpath(path, ROOT)) if calls else None


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, tools/hbm_traffic.py): counters cannot be read from inside this process, so the
    figure comes from profiles/ (same command, same build) and says so; null when the file is absent."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.json")
    try:
        rows = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return {"traffic": None}
    want = kernel.replace(" ", "").rstrip(">")
    for r in rows:
        if r["kernel"].replace(" ", "").startswith(want):
            return {"traffic": round((r["fetch_MB_per_launch_x2"] + r["write_MB_per_launch"]) * 1e6),
                    "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE)",
                    "traffic_source": os.path.relpath(path, ROOT)}
    return {"traffic": None}


def pmc_bytes_per_step():
    """HBM-side bytes of ONE optimizer step summed over every kernel of the committed PMC passes (same command, same
    build): sum of launches x (FETCH_SIZE x2 + WRITE_SIZE) / optimizer steps of that run (= launches of k_slab_reduce,
    one per step); the few per-update kernels (GAE, normaliser scan) are in it and are noise at this scale."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.json")
    try:
        rows = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    steps = [r["launches"] for r in rows if r["kernel"].startswith("k_slab_reduce")]
    if not steps or steps[0] < 1:
        return None
    total = sum(r["launches"] * (r["fetch_MB_per_launch_x2"] + r["write_MB_per_launch"]) for r in rows)
    return {"value": round(total / steps[0] * 1e6), "source": os.path.relpath(path, ROOT), "optimizer_steps": steps[0]}


def measured_peaks(dev, seconds=1.0):
    """The box's own figures beside the 157.3 TFLOP/s spec peak, measured before the timed region:
    (i) >= `seconds` of a register-only v_mfma_f32_32x32x2_f32 loop on every SIMD (igi_mfma_peak_probe: no LDS, no
    memory, non-trivial operands) -> the matrix rate and the in-kernel clock the chip holds under pure fp32 MFMA load;
    (ii) the library's own LDS-DMA GEMM k-loop on a 262144 x 256 x 4096 product (550 GFLOP per launch, the epilogue and
    the fill are < 1 % of it) -> what a k-loop sustains with operands streaming from HBM."""
    import torch
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream(dev)
    blocks, iters = 256 * 8, 2048                       # 8 waves per SIMD everywhere; ~8 ms per launch
    clocks = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)

    def probe(shape=0):
        _lib.check(L.igi_mfma_peak_probe(shape, blocks, iters, _lib.ptr(clocks), _lib.ptr(sink), st.cuda_stream), "igi_mfma_peak_probe")

    probe(); torch.cuda.synchronize(dev)
    t_end = time.perf_counter() + 0.5 * seconds          # reach the loaded power state first
    while time.perf_counter() < t_end:
        for _ in range(8):
            probe()
        torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    e0.record(st)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            probe()
        n += 8
        torch.cuda.synchronize(dev)
    e1.record(st); torch.cuda.synchronize(dev)
    dt = e0.elapsed_time(e1) * 1e-3
    flops = float(blocks) * 4 * iters * 16 * 4096 * n
    c = clocks.view(-1, 2).double()
    ghz = float((c[:, 0] / c[:, 1].clamp(min=1)).median()) * 0.1
    out = {"peak_measured_mfma": round(flops / dt / 1e12, 2), "peak_measured_unit": "TFLOP/s",
           "peak_measured_clock_ghz": round(ghz, 3),
           "peak_measured_how": f"register-only v_mfma_f32_32x32x2_f32 loop, {blocks} x 256 threads (8 waves per SIMD), "
                                f"{n} launches over {dt:.2f} s after {0.5 * seconds:.1f} s of the same load; clock = median over "
                                "blocks of s_memtime / s_memrealtime",
           "peak_at_measured_clock": round(256 * 4 * 64 * ghz / 1e3, 2)}
    # the same loop on the other fp32 MFMA shape and on the bf16 pipe (0.25 s each): what the chip holds there
    for shape, key, fl_mfma in ((1, "peak_measured_mfma_16x16x4_f32", 2048.0), (2, "peak_measured_mfma_32x32x16_bf16", 32768.0)):
        for _ in range(8):
            probe(shape)
        torch.cuda.synchronize(dev)
        n2 = 0
        e0.record(st)
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            for _ in range(8):
                probe(shape)
            n2 += 8
            torch.cuda.synchronize(dev)
        e1.record(st); torch.cuda.synchronize(dev)
        dt2 = e0.elapsed_time(e1) * 1e-3
        c2 = clocks.view(-1, 2).double()
        out[key] = {"tflops": round(float(blocks) * 4 * iters * 16 * fl_mfma * n2 / dt2 / 1e12, 2),
                    "clock_ghz": round(float((c2[:, 0] / c2[:, 1].clamp(min=1)).median()) * 0.1, 3)}
    del clocks
    # (ii) the GEMM k-loop
    M, N, K = 262144, 256, 4096
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    cc = torch.empty(M, N, device=dev); b = torch.zeros(N, device=dev)

    def gemm():
        _lib.check(L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(cc), N, _lib.ptr(b), None, 0, 0, 0,
                                  st.cuda_stream), "igi_gemm_f32")

    for _ in range(20):
        gemm()
    torch.cuda.synchronize(dev)
    e0.record(st)
    reps = 60
    for _ in range(reps):
        gemm()
    e1.record(st); torch.cuda.synchronize(dev)
    dtg = e0.elapsed_time(e1) * 1e-3 / reps
    out["gemm_kloop_tflops"] = round(2.0 * M * N * K / dtg / 1e12, 2)
    out["gemm_kloop_how"] = f"igi_gemm_f32 {M} x {N} x {K} (LDS-DMA kernel), {reps} launches of {dtg * 1e3:.2f} ms after 20 warm-up launches"
    out["gemm_kloop_over_measured_peak"] = round(out["gemm_kloop_tflops"] / max(out["peak_measured_mfma"], 1e-9), 4)
    del a, w, cc, b
    torch.cuda.empty_cache()
    return out


def bf16x3_experiment(dev):
    """EXPERIMENT, outside `value`: fp32 products on the bf16 matrix pipe through the exact
    three-plane split (x = hi + mid + lo in bf16, every plane product exact in fp32, fp32 accumulation:
    igi_gemm_set_bf16x3) against the native fp32 MFMA kernel, on the trunk-2 forward shape of the teacher step (2 nets x
    16384 rows, 512 -> 256, bias + tanh; the split is done in the loop by every consuming wave).  Time per launch and the
    error of the pre-activation against fp64, per mode; the headline runs with it off."""
    import torch
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    M, N, K = 32768, 256, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.tanh(torch.randn(M, K, generator=g)).to(dev)
    w = (torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    c = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream(dev)
    ref = a[:2048].double() @ w.double().t() + b.double()

    def run(epi):
        _lib.check(L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(c), N, _lib.ptr(b), None, 0, epi, 0,
                                  st.cuda_stream), "igi_gemm_f32")

    out = {"shape": f"{M} x {N} x {K}, bias + tanh (the trunk-2 forward of both nets)", "modes": {}}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        for name, prod in (("f32_mfma (headline arithmetic)", 0), ("bf16x3, all 9 plane products (exact products)", 9),
                           ("bf16x3, 6 plane products (without mid*lo, lo*mid, lo*lo)", 6)):
            L.igi_gemm_set_bf16x3(prod)
            run(3); torch.cuda.synchronize(dev)
            err = (c[:2048].double() - ref).abs()
            t_end = time.perf_counter() + 0.6        # the loaded power state first (a cold start flatters whichever mode runs later)
            while time.perf_counter() < t_end:
                for _ in range(50):
                    run(1)
                torch.cuda.synchronize(dev)
            e0.record(st)
            for _ in range(200):
                run(1)
            e1.record(st); torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / 200
            out["modes"][name] = {"us_per_launch": round(us, 2), "mean_abs_err_vs_fp64": float(err.mean()), "max_abs_err_vs_fp64": float(err.max())}
    finally:
        L.igi_gemm_set_bf16x3(0)
    base = out["modes"]["f32_mfma (headline arithmetic)"]["us_per_launch"]
    for v in out["modes"].values():
        v["time_vs_f32_mfma"] = round(v["us_per_launch"] / base, 3)
    out["note"] = ("off by default (igi_gemm_set_bf16x3 / IGI_GEMM_X3); the split costs 44 vector instructions per 8-k fragment per "
                   "consuming wave in this form; DESIGN.md section 4, round 6")
    return out


def params_identical(t, dev):
    """every rank holds bit-identical values of ``t`` (min == max over ranks of two order-independent integer checksums)"""
    import torch
    import torch.distributed as dist
    bits = t.view(torch.int32).to(torch.int64)
    chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool((lo == hi).all().item())


def teacher_dp_leg(envs, horizon, comm, world, rank, dev, steps=5, warmup=2):
    """One data-parallel teacher configuration on every rank (own arena per rank, weak scaling), timed under each gradient
    exchange schedule this process group offers: the library's RCCL communicator overlapped / serial, and the
    torch.distributed callback path.  -> {schedule: {updates_per_s, ms_per_update, params_identical_across_ranks}}"""
    import torch
    import torch.distributed as dist
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from isaacgyminsertion_amd.envs import synthetic_rollout as synth
    init, ro, perm = synth.teacher_problem(envs, horizon, UNITS, PRIV_UNITS, seed=4321 + rank, device=dev)
    out = {}
    schedules = ([("rccl_native_overlapped", True, True), ("rccl_native_serial", True, False)] if comm is not None else []) \
        + [("torch_distributed_overlapped", False, True), ("torch_distributed_serial", False, False)]
    for name, native, overlap in schedules:
        eng = TeacherEngine(envs, horizon, MINI_EPOCHS, units=UNITS, priv_units=PRIV_UNITS, perm=perm, device=dev)
        eng.load_params(init)
        dist.broadcast(eng.params, 0)
        eng.set_rollout(ro)

        def one():
            eng.prepare()
            if native:
                eng.update_dp_native(comm, overlap=overlap)
            else:
                eng.update_dp(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), world,
                              all_reduce_async=(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
                              if overlap else None)

        for _ in range(warmup):
            one()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        out[name] = {"updates_per_s": round(world * steps / dt, 3), "ms_per_update": round(1e3 * dt / steps, 3),
                     "params_identical_across_ranks": params_identical(eng.params, dev),
                     "finite": bool(torch.isfinite(eng.params).all())}
        del eng
    return out


def multi_gpu_configs(comm, world, rank, dev):
    """The configurations BASELINE.json defines on 8 GPUs, at whatever N this job runs on (outside `value`):
    configs[4] teacher 16384 envs x 64 = 2048 x 64 per rank at N = 8 (weak: 2048 envs per rank at any N), and configs[3]
    student tactile + PointNet, strong (4096 envs over the ranks) and weak (4096 envs per rank, as the reference gives
    every rank its own numEnvs), each under the overlapped and the serial gradient exchange."""
    import torch.distributed as dist
    from tools.bench_student import student_bench
    rec = {"note": "outside `value`; every record: time = max over ranks between barriers, parameters compared bit for bit "
                   "across ranks afterwards; schedules: the library's RCCL communicator (overlapped with backward | serial) "
                   "and the torch.distributed path"}
    import torch
    # wall-clock budget of the whole section (rank 0's clock decides for everyone): a leg that would start after it is
    # recorded as skipped -- the headline line is already out by then (main() prints it before this section starts)
    budget_s = float(os.environ.get("IGI_MULTI_BUDGET_S", "420"))
    t_section = time.perf_counter()

    def leg(name, fn):
        """a failing leg is recorded and ends the section on every rank (MIN vote) instead of taking the headline line
        down with it; a rank that fails INSIDE a collective cannot be helped (the others wait for the backend's time-out)"""
        go = torch.tensor([1 if time.perf_counter() - t_section < budget_s else 0], dtype=torch.int32, device=dev)
        dist.broadcast(go, 0)
        if not bool(go.item()):
            rec[name] = {"skipped": f"wall-clock budget of the multi-GPU sub-records spent ({budget_s:.0f} s, IGI_MULTI_BUDGET_S)"}
            return True
        ok, res = 1, None
        t_leg = time.perf_counter()
        try:
            res = fn()
        except Exception as e:   # noqa: BLE001
            ok, res = 0, {"error": f"{type(e).__name__}: {e}"}
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not bool(flag.item()):
            rec[name] = res if not ok else {"error": "failed on another rank"}
            return False
        if isinstance(res, dict):
            res["leg_wall_s"] = round(time.perf_counter() - t_leg, 1)
        rec[name] = res
        return True

    if not leg("configs[1] teacher 4096 x 32 per rank, by schedule",
               lambda: teacher_dp_leg(NUM_ENVS, HORIZON, comm, world, rank, dev)):
        return rec
    if not leg("configs[4] teacher 2048 x 64 per rank (16384 x 64 over 8 ranks), by schedule",
               lambda: teacher_dp_leg(2048, 64, comm, world, rank, dev)):
        return rec
    prev = os.environ.get("IGI_DP_OVERLAP")
    try:
        for label, envs, updates in (("strong: 4096 envs over the ranks", max(32, 4096 // world // 32 * 32), 2),
                                     ("weak: 4096 envs per rank", 4096, 1)):
            for sched in ("1", "0"):
                os.environ["IGI_DP_OVERLAP"] = sched
                name = f"configs[3] student tactile + PointNet, {label}, {'overlapped' if sched == '1' else 'serial'}"
                if not leg(name, lambda: student_bench(4, envs, 32, (32, 64), updates=updates, multi_gpu=True,
                                                       profile=False)):
                    return rec
    finally:
        if prev is None:
            os.environ.pop("IGI_DP_OVERLAP", None)
        else:
            os.environ["IGI_DP_OVERLAP"] = prev
    return rec


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` started as ONE process (the way the driver starts every bench): this parent -- which
    has imported nothing that touches the GPU and makes no GPU call -- starts the N ranks the way the reference's
    scripts/train_s1.sh:16 does (torchrun, one process per GPU, rendezvous on 127.0.0.1) as a CHILD, lets rank 0's JSON
    line through on the shared stdout and exits with the children's code.  Nothing is re-executed in place."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 4096 = BASELINE configs[1]; "
                    "2048 with --horizon 64 is one rank of configs[4])")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--bf16-inputs", action="store_true",
                    help="opt-in: bf16-rounded operands on the bf16 MFMA pipe with fp32 accumulation for the large "
                         "products (NOT the reference arithmetic; the line says so in dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-peak-probe", action="store_true", help="skip the ~3 s measured-peak probes (MFMA register loop, GEMM k-loop)")
    ap.add_argument("--no-student", action="store_true", help="skip the student section (configs[2] / [3] legs)")
    ap.add_argument("--no-experiments", action="store_true", help="skip the `experiments` section (the bf16 three-plane split probe)")
    ap.add_argument("--no-multi-configs", action="store_true",
                    help="N > 1: skip the sub-records of the multi-GPU configurations (configs[4] teacher, configs[3] student)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    global NUM_ENVS, HORIZON
    if args.envs:
        NUM_ENVS = args.envs
    if args.horizon:
        HORIZON = args.horizon
    import torch
    import torch.distributed as dist
    from isaacgyminsertion_amd import _lib
    from isaacgyminsertion_amd.teacher_native import TeacherEngine
    from isaacgyminsertion_amd.envs import synthetic_rollout as synth   # product-side arena generator (no oracle/)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} processes (WORLD_SIZE={world})")
    # IGI_DIST_BACKEND=gloo lets the multi-rank path be exercised on a single-GPU box (all ranks on one
    # device, gradients carried by gloo); the real run is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("IGI_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        # a finite time-out: a rank lost inside a collective ends the job (the process-group watchdog aborts the others,
        # torchrun then stops every rank and the parent exits non-zero) instead of hanging the node
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("IGI_PG_TIMEOUT_S", "600")))
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)

    _lib.lib()  # fail loudly if the HIP library is missing
    if args.bf16_inputs:
        _lib.lib().igi_gemm_set_bf16_inputs(1)
    # per-rank arena (seed + rank, train.py:58-64), identical initial parameters on every rank
    init, ro, perm = synth.teacher_problem(NUM_ENVS, HORIZON, UNITS, PRIV_UNITS, seed=1234 + rank, device=dev)
    eng = TeacherEngine(NUM_ENVS, HORIZON, MINI_EPOCHS, units=UNITS, priv_units=PRIV_UNITS, perm=perm, device=dev)
    eng.load_params(init)
    # the gradient exchange is issued by the library over its own RCCL communicator (csrc/comm.h); IGI_DP_NATIVE=0 or
    # a non-RCCL backend selects the torch.distributed callback path instead
    comm, native_note = None, None
    if world > 1:
        # every rank takes the same path: utils.dist.native_comm_or_none distributes the id, creates and probes the
        # communicator and votes after every stage (no rank is left in a collective another one skipped)
        from isaacgyminsertion_amd.utils.dist import native_comm_or_none
        comm = native_comm_or_none(dev, world)
        if comm is None and backend == "nccl" and os.environ.get("IGI_DP_NATIVE", "1") != "0":
            native_note = "probe failed on at least one rank"
    native = comm is not None
    if native:
        comm.broadcast_(eng.params, 0)           # frozen_ppo.py:376-381 as one flat vector
    elif world > 1:
        dist.broadcast(eng.params, 0)            # frozen_ppo.py:376-381
    eng.set_rollout(ro)                          # arena resident in HBM from here on
    # which workspace ALLOCATION the update runs fastest on (up to 2 %: the env level's duration depends on where its
    # buffers landed in HBM); what the trainer does once before its first update (experience.py), outside any timed region
    ws_trial_ms = eng.tune_workspace()

    def all_reduce(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def all_reduce_async(t):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    overlap = os.environ.get("IGI_DP_OVERLAP", "1") != "0"
    schedule_pick = None

    def one_update():
        eng.prepare()
        if native:
            eng.update_dp_native(comm, overlap=overlap)
        elif world > 1:
            eng.update_dp(all_reduce, world, all_reduce_async=all_reduce_async if overlap else None)
        else:
            eng.update()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and "IGI_DP_OVERLAP" not in os.environ:
        # Which exchange schedule -- the early bucket overlapped with the rest of backward, or one exchange behind it --
        # is faster depends on the node (on ONE rank the overlap costs +0.8 ms per update and hides nothing,
        # profiles/r04_dp_phase_cost.json; across 8 GPUs it hides a 1.3 MB all-reduce per optimizer step).  The job
        # times two updates of each, once, before the warm-up, every rank takes the max over ranks and so the same choice.
        # IGI_DP_OVERLAP=0 / 1 pins it.
        trial = {}
        for cand in (True, False):
            overlap = cand
            one_update()
            fence()
            t0 = time.perf_counter()
            one_update(); one_update()
            fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[cand] = float(t.item()) / 2
        overlap = trial[True] <= trial[False]
        schedule_pick = {"overlapped_ms_per_update": round(1e3 * trial[True], 3),
                         "serial_ms_per_update": round(1e3 * trial[False], 3),
                         "picked": "overlapped" if overlap else "serial"}

    # the box's own matrix rate, beside the spec peak the fractions are quoted against (every rank runs it: each has a GPU)
    peaks = None
    if not args.no_roofline and not args.no_peak_probe:
        peaks = measured_peaks(dev)
        fence()

    for _ in range(args.warmup):
        one_update()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_update()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = eng.stats.cpu()
    finite = bool(torch.isfinite(stats).all() and torch.isfinite(eng.params).all())
    # data-parallel sanity, outside the timed region: every rank applied the same summed gradient to the same
    # parameters, so the parameter vectors must be IDENTICAL bit for bit (min == max over ranks of an order-independent
    # integer checksum of the bits)
    ranks_identical = params_identical(eng.params, dev) if world > 1 else None

    # ---- per-kernel-class roofline: a second, identical, event-instrumented region
    roof, classes = None, None
    if not args.no_roofline:
        # every rank runs the SAME step as the timed region (at N > 1 the data-parallel schedule with its exchange and
        # its second slab-sum launch); rank 0 carries the per-launch timestamps
        k = min(args.steps, 5)
        if rank == 0:
            _lib.prof_enable(True)
        for _ in range(k):
            one_update()
        fence()
        if rank == 0:
            classes = _lib.prof_read()
            _lib.prof_enable(False)
    if classes is not None:
        # one rocprofv3 symbol may carry several of our classes ("symbol#level": the four backward levels share
        # gemm_dma_wgrad_multi_kernel): the dominant KERNEL is chosen by symbol, the levels are listed beside it
        levels = [c for c in classes if "#" in c["name"]]
        by_sym = {}
        for c in classes:
            sym = c["name"].split("#")[0]
            a = by_sym.setdefault(sym, {"name": sym, "launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            for f in ("launches", "total_ms", "flops", "bytes"):
                a[f] += c[f]
        classes = list(by_sym.values())
        for c in classes + levels:
            c["avg_us"] = 1e3 * c["total_ms"] / max(c["launches"], 1)
            c["tflops"] = c["flops"] / (c["total_ms"] * 1e-3) / 1e12 if c["total_ms"] > 0 else 0.0
            c["gbs"] = c["bytes"] / (c["total_ms"] * 1e-3) / 1e9 if c["total_ms"] > 0 else 0.0
            c["ms_per_update"] = c["total_ms"] / k
        dom = max(classes, key=lambda c: c["total_ms"])
        gemms = [c for c in classes if c["name"].startswith("gemm_") or c["name"] in ("k_env_fwd", "k_trunk_loss", "k_rb_level")]
        g_ms = sum(c["total_ms"] for c in gemms)
        g_fl = sum(c["flops"] for c in gemms)
        gemm_all = {"ms_per_update": round(g_ms / k, 3), "tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 2),
                    "frac_of_f32_mfma_peak": round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        if dom["flops"] > 0:
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(dom["tflops"], 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(dom["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "avg_launch_us": round(dom["avg_us"], 2), "launches": dom["launches"],
                    "all_gemm_kernels": gemm_all,
                    "levels": [{"kernel": c["name"].split("#", 1)[0], "level": c["name"].split("#", 1)[1], "launches": c["launches"],
                                "gflop_per_launch": round(c["flops"] / max(c["launches"], 1) / 1e9, 3),
                                "avg_us": round(c["avg_us"], 2), "tflops": round(c["tflops"], 2),
                                "frac": round(c["tflops"] / PEAK_F32_MFMA_TFLOPS, 4)}
                               for c in levels]}
        else:
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(dom["gbs"], 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(dom["gbs"] / PEAK_HBM_GBS, 4),
                    "traffic": None, "avg_launch_us": round(dom["avg_us"], 2), "launches": dom["launches"]}

    if roof is not None:
        roof["timing"] = "dispatch start/stop timestamps of each launch (hipExtLaunchKernelGGL events), live in this run"
        roof["algorithmic_gflop_per_launch"] = round(dom["flops"] / max(dom["launches"], 1) / 1e9, 4)
        # the KERNEL's own operand + output + partial-slab footprint per launch (what its tiling has to move) ...
        roof["operand_bytes_per_launch"] = round(dom["bytes"] / max(dom["launches"], 1))
        # ... against what the ALGORITHM has to move per optimizer step: 452 B gathered / written
        # back per sample-pass + 28 B per parameter of Adam; everything above that is the layer-by-layer design's
        # activations, split-K partials and re-reads -- the PMC figure beside it is measured over ALL kernels of a step
        mb = NUM_ENVS * HORIZON // MINI_EPOCHS
        roof["algorithmic_bytes_per_step"] = mb * 452 + 28 * N_PARAMS
        # profile-sourced fields (counters and the rocprofv3 summary cannot be read from inside this process) are attached
        # only when the committed summaries carry THIS library's build hash (tools/stamp_profiles.py); otherwise the line
        # says so instead of letting a stale profile ride in a fresh record
        traffic_json = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.json")
        stats_csv = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_bench_kernel_stats.csv")
        roof["build"] = _lib.lib().igi_build_info().decode()
        if profile_is_current(traffic_json):
            pmc = pmc_bytes_per_step()
            if pmc is not None:
                roof["pmc_bytes_per_step"] = pmc
                roof["pmc_over_algorithmic"] = round(pmc["value"] / roof["algorithmic_bytes_per_step"], 1)
            roof.update(pmc_traffic(roof["kernel"]))
        else:
            roof["traffic"] = None
            roof["traffic_stale"] = True
            roof["traffic_stale_why"] = (f"profiles/{PROFILE_TAG}_hbm_traffic.json is stamped with build "
                                         f"{profile_build(traffic_json)}, the loaded library is {roof['build']}")
        row = rocprof_row(roof["kernel"]) if profile_is_current(stats_csv) else None
        if row is not None and roof.get("bound") == "mfma" and not args.bf16_inputs:
            calls, avg_ns, src = row
            ach = dom["flops"] / max(dom["launches"], 1) / (avg_ns * 1e-9) / 1e12
            roof["rocprof"] = {"source": src, "calls": calls, "avg_launch_us": round(avg_ns / 1e3, 2),
                               "achieved": round(ach, 2), "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4)}
            roof["frac_rocprof"] = roof["rocprof"]["frac"]
        elif not profile_is_current(stats_csv):
            roof["frac_rocprof"] = None
            roof["frac_rocprof_stale"] = True
        if peaks is not None:
            roof.update(peaks)
            roof["frac_of_measured_peak"] = round(roof["achieved"] / max(peaks["peak_measured_mfma"], 1e-9), 4) \
                if roof.get("bound") == "mfma" else None
        if args.bf16_inputs and roof.get("bound") == "mfma":   # the opt-in mode runs the large products on the bf16 pipe
            roof["peak"] = PEAK_BF16_MFMA_TFLOPS
            roof["frac"] = round(roof["achieved"] / PEAK_BF16_MFMA_TFLOPS, 4)
            roof["note"] = "bf16-input MFMA (dense bf16 peak); fp32 loaders/LDS traffic unchanged"
            roof["traffic"] = None

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and args.gpus == 1:
        cpu = cpu_baseline(init, {k: v.cpu() for k, v in ro.items()}, perm)

    student = None
    if not args.no_student and rank == 0 and args.gpus == 1 and not args.bf16_inputs:
        from tools.bench_student import student_bench
        eng = None                                   # free the teacher's arenas before the student legs
        torch.cuda.empty_cache()
        student = {
            "note": "outside `value`; one GPU; synthetic StudentBuffer resident in HBM; fp32",
            "configs[2] tactile CNN (3 x 32x64 gray, the reference default) + lin, 2048 envs x 32":
                student_bench(3, 2048, 32, (32, 64), updates=2),
            "configs[2] with 64x64 images (BASELINE wording) + lin, 2048 envs x 32":
                student_bench(3, 2048, 32, (64, 64), updates=1),
            "configs[3] single-rank share: tactile + PointNet(2 x 400) + lin, 512 envs x 32 (4096 envs over 8 GPUs)":
                student_bench(4, 512, 32, (32, 64), updates=3),
        }

    # what RCCL itself says about the communicator the gradients went through (WORLD_SIZE is only what the launcher said)
    rccl = None
    if native:
        try:
            rccl = {"rccl_ranks": comm.rccl_ranks(), "rccl_version": comm.rccl_version()}
        except Exception as e:   # noqa: BLE001
            rccl = {"rccl_ranks": None, "rccl_error": f"{type(e).__name__}: {e}"}
    elif world > 1:
        rccl = {"rccl_ranks": None, "rccl_note": "gradients went through torch.distributed (" + backend + "), not the library's communicator"}

    # the engine's workspace placement trial (TeacherEngine.tune_workspace, before anything here was timed: one update per
    # candidate allocation, GPU time of its kernels in ms; the first entry is the allocation the engine was built with)
    rccl = dict(rccl or {}, workspace_trial_ms=ws_trial_ms)

    def record(multi):
        return build_record(args, world, backend, native, native_note, overlap, schedule_pick, dt, finite, ranks_identical,
                            roof, cpu, student, multi, classes, rccl, experiments)

    experiments = None
    if rank == 0 and args.gpus == 1 and not args.bf16_inputs and not args.no_experiments:
        try:
            experiments = {"bf16x3": bf16x3_experiment(dev)}
        except Exception as e:   # noqa: BLE001  (an experiment must not cost the line)
            experiments = {"bf16x3": {"error": f"{type(e).__name__}: {e}"}}

    multi = None
    if world > 1 and not args.no_multi_configs and not args.bf16_inputs:
        if rank == 0:
            # the COMPLETE headline record is on stdout before the (longer, newer) multi-GPU sub-records start: a hang or a
            # lost rank in a sub-record cannot cost the headline.  A second, extended line (the same record +
            # multi_gpu_configs) follows when the section returns.
            print(json.dumps(record(None)), flush=True)
        eng = None
        torch.cuda.empty_cache()
        multi = multi_gpu_configs(comm, world, rank, dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    print(json.dumps(record(multi)), flush=True)


def build_record(args, world, backend, native, native_note, overlap, schedule_pick, dt, finite, ranks_identical, roof, cpu,
                 student, multi, classes, rccl, experiments=None):
    ms = 1e3 * dt / args.steps
    upd_per_s = world * args.steps / dt
    flops_update = 6.0 * fwd_macs() * NUM_ENVS * HORIZON * MINI_EPOCHS  # train = 6 x fwd MACs
    out = {
        "metric": f"PPO update steps/sec ({NUM_ENVS} envs x {HORIZON} horizon)", "value": round(upd_per_s, 3),
        "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16-in/f32-acc (opt-in, not the reference arithmetic)" if args.bf16_inputs else "f32",
        "data": "synthetic",
        "config": {"workload": f"teacher PPO update, MLP actor-critic 404,501 params, {NUM_ENVS} envs x {HORIZON} horizon "
                               f"per GPU, 8 mini-epochs x 8 minibatches of {NUM_ENVS * HORIZON // MINI_EPOCHS}"
                               + (" (BASELINE configs[1])" if (NUM_ENVS, HORIZON) == (4096, 32) else ""),
                   "envs_per_gpu": NUM_ENVS, "horizon": HORIZON, "optimizer_steps_per_update": MINI_EPOCHS ** 2,
                   "parallelism": f"dp{world}",
                   "grad_allreduce": ((backend if backend != "nccl" else "rccl")
                                      + (" issued by libigi_hip.so (own communicator + comm stream)" if native
                                         else " through torch.distributed")
                                      + (", 2 buckets overlapped with backward" if overlap else ", serial")
                                      + (f" [native communicator unavailable: {native_note}]" if native_note else ""))
                   if world > 1 else "none",
                   **(rccl or {}),
                   **({"grad_allreduce_schedule_trial": schedule_pick} if schedule_pick else {})},
        "optimizer_steps_per_s": round(upd_per_s * MINI_EPOCHS ** 2, 1),
        "sample_passes_per_s": round(upd_per_s * NUM_ENVS * HORIZON * MINI_EPOCHS, 0),
        "whole_update_tflops": round(flops_update / (dt / args.steps) / 1e12, 2),
        "whole_update_mfma_frac": round(flops_update / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        "finite": finite,
        **({"params_identical_across_ranks": ranks_identical} if world > 1 else {}),
        "roofline": roof, "cpu_baseline": cpu,
    }
    if student is not None:
        out["student"] = student
    if experiments is not None:
        out["experiments"] = experiments
    if multi is not None:
        out["multi_gpu_configs"] = multi
    if classes:
        out["kernels"] = [{"name": c["name"], "launches_per_update": c["launches"] // min(args.steps, 5),
                           "avg_us": round(c["avg_us"], 2), "ms_per_update": round(c["ms_per_update"], 3),
                           "tflops": round(c["tflops"], 2), "operand_gbs": round(c["gbs"], 1)}
                          for c in sorted(classes, key=lambda c: -c["total_ms"])]
    if multi is None and world > 1 and not args.no_multi_configs and not args.bf16_inputs:
        out["multi_gpu_configs"] = "follow on a second, extended line of this record"
    return out


if __name__ == "__main__":
    main()
