// Optional per-kernel-class timing with HIP events on the launch stream (used by bench.py for the
// roofline figures: algorithmic FLOPs / bytes of every launch are known at the launch site).
// Disabled by default: zero events are recorded unless igi_prof_enable(1) was called.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

namespace igi {

enum ProfClass {
  PC_GEMM_FWD = 0, PC_GEMM_DGRAD, PC_GEMM_WGRAD, PC_GATHER_STATS, PC_RMS_FINAL, PC_NORMALIZE,
  PC_LOSS, PC_SLAB_REDUCE, PC_SUMSQ, PC_ADAM, PC_PREPARE, PC_OTHER, PC_COUNT
};

static const char* const kProfNames[PC_COUNT] = {
    "gemm_f32_fwd", "gemm_f32_dgrad", "gemm_f32_wgrad", "gather_stats", "rms_final", "normalize",
    "heads_loss", "slab_reduce", "gradnorm_stats", "clip_adam", "prepare(gae+norm)", "other"};

struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes; };

struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  std::mutex mu;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
};

static Profiler& profiler() { static Profiler p; return p; }

struct ProfScope {
  bool active = false;
  hipStream_t s;
  ProfRec r;
  ProfScope(int cls, hipStream_t stream, double flops, double bytes) : s(stream) {
    Profiler& p = profiler();
    if (!p.on) return;
    std::lock_guard<std::mutex> g(p.mu);
    active = true;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    r.a = p.get(); r.b = p.get();
    (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(r.b, s);
    Profiler& p = profiler();
    std::lock_guard<std::mutex> g(p.mu);
    p.recs.push_back(r);
  }
};

}  // namespace igi
