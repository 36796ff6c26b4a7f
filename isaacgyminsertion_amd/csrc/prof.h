// Optional per-kernel-class timing with HIP events on the launch stream (used by bench.py for the
// roofline figures: algorithmic FLOPs / bytes of every launch are known at the launch site).
// Disabled by default: zero events are recorded unless igi_prof_enable(1) was called.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <mutex>
#include <vector>

namespace igi {

// Classes are keyed by kernel symbol (the same names rocprofv3 --kernel-trace reports), so the
// live figures and the committed rocprof summary can be compared line by line.
enum ProfClass {
  PC_DMA_256_TT = 0, PC_DMA_256_TF, PC_DMA_256_FF, PC_DMA_256_FT,
  PC_DMA_128_TT, PC_DMA_128_TF, PC_DMA_128_FF, PC_DMA_128_FT,
  PC_DMA_64_TT, PC_DMA_64_TF, PC_DMA_64_FF, PC_DMA_64_FT,
  PC_GROUP_128_FF, PC_GROUP_64_FF,
  // gemm_dma_wgrad_multi_kernel by backward level (one symbol in rocprofv3; "#level" is ours): the launch sites tag it
  PC_WGRAD_MULTI, PC_WGRAD_MULTI_L1, PC_WGRAD_MULTI_L2, PC_WGRAD_MULTI_L3, PC_WGRAD_MULTI_L4, PC_WGRAD_MULTI_L5,
  // the tall (256-row) im2col instantiations of the student's convolutions
  PC_CONV_TALL64_TT, PC_CONV_TALL32_TT, PC_CONV_TALL64_TF, PC_CONV_TALL32_TF, PC_CONV_TALL64_SSA, PC_CONV_WG_TALL32, PC_CONV_WG_TALL64,
  PC_CONV_PM64, PC_CONV_PM32,   // position-major data-gradient tiles (GATHER == 4)
  PC_CONV_PW32, PC_CONV_PW64, PC_CONV_PW64_192,   // weight gradients with a position-major reduction (GATHER == 5), 256-tap tiles
  PC_POINTNET_FWD, PC_POINTNET_BWD, PC_SOFTARGMAX_FWD, PC_SOFTARGMAX_BWD,
  PC_DMA_HEAD, PC_TRUNK_LOSS, PC_RB_TRUNK, PC_RB_ENV, PC_MLP_FWD, PC_POLICY_FWD, PC_FWD12, PC_ENV_FWD, PC_GEMM_GENERIC, PC_GATHER_NORMALIZE, PC_RMS_FINAL, PC_NORMALIZE,
  PC_LOSS, PC_LATENT_BWD, PC_SLAB_REDUCE, PC_SUMSQ, PC_ADAM, PC_ADAM_GATHER, PC_PREPARE, PC_OTHER, PC_COUNT
};

static const char* const kProfNames[PC_COUNT] = {
    "gemm_dma_kernel<256,true,true>", "gemm_dma_kernel<256,true,false>", "gemm_dma_kernel<256,false,false>",
    "gemm_dma_kernel<256,false,true>", "gemm_dma_kernel<128,true,true>", "gemm_dma_kernel<128,true,false>",
    "gemm_dma_kernel<128,false,false>", "gemm_dma_kernel<128,false,true>", "gemm_dma_kernel<64,true,true>",
    "gemm_dma_kernel<64,true,false>", "gemm_dma_kernel<64,false,false>", "gemm_dma_kernel<64,false,true>",
    "gemm_dma_group_kernel<128,false,false>", "gemm_dma_group_kernel<64,false,false>",
    "gemm_dma_wgrad_multi_kernel#trunk3: dW 256->128 x2 + dgrad 128->256 x2",
    "gemm_dma_wgrad_multi_kernel#trunk2: dW 512->256 x2 + dgrad 256->512 x2 + latent row dots (+ dW 23->512 x2 from the dZ1 tiles)",
    "gemm_dma_wgrad_multi_kernel#env2: dW 23->512 x2 + dW env 256->128 + env dgrad 128->256",
    "gemm_dma_wgrad_multi_kernel#env1: dW env 64->256",
    "gemm_dma_wgrad_multi_kernel#other",
    "gemm_dma_wgrad_multi_kernel#env1+: dW 23->512 x2 + dW env 64->256",
    "gemm_dma_kernel<64,true,true,1,2,256>", "gemm_dma_kernel<32,true,true,1,2,256>",
    "gemm_dma_kernel<64,true,false,1,2,256>", "gemm_dma_kernel<32,true,false,1,2,256>",
    "gemm_dma_kernel<64,true,true,6,2,256>",
    "gemm_dma_kernel<32,false,false,3,2,256>", "gemm_dma_kernel<64,false,false,3,2,256>",
    "gemm_dma_kernel<64,true,true,4,2,256>", "gemm_dma_kernel<32,true,true,4,2,256>",
    "gemm_dma_kernel<32,false,false,5,2,256>", "gemm_dma_kernel<64,false,false,5,2,256>", "gemm_dma_kernel<64,false,false,5,2,192>",
    "k_pointnet_fwd", "k_pointnet_bwd", "k_softargmax_fwd", "k_softargmax_bwd",
    "gemm_dma_head_kernel<true>", "k_trunk_loss",
    "k_rb_level#trunk3: dW 256->128 x2 + dgrad 128->256 x2", "k_rb_level#env2: dW env 256->128 + env dgrad 128->256 (+ dW env 64->256 from its tiles)",
    "k_mlp_fwd", "k_policy_fwd", "k_fwd12",
    "k_env_fwd", "gemm_f32_kernel<*>", "k_gather_normalize", "k_rms_final", "k_normalize",
    "k_loss", "k_latent_bwd", "k_slab_reduce", "k_sumsq_stats", "k_clip_adam", "k_adam_gather", "k_gae+k_prep_final+k_prep_norm", "other"};

// which backward level the next gemm_dma_wgrad_multi_kernel launch belongs to (set by the teacher's orchestration)
static thread_local int g_multi_level = 4;

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

// A scope brackets ONE kernel launch (ext = true, the default): the launch itself (IGI_LAUNCH below) carries the two
// events as the dispatch's own start / stop timestamps (hipExtLaunchKernelGGL), i.e. the kernel's execution time as
// rocprofv3 --kernel-trace reports it -- events recorded around the launch would add the ~3.5 us dispatch gap to
// every figure.  ext = false: events recorded around whatever the scope encloses (several launches).
struct ProfScope;
static thread_local ProfScope* g_prof_cur = nullptr;

struct ProfScope {
  bool active = false, ext = true, used = false;
  hipStream_t s;
  ProfRec r;
  ProfScope* prev = nullptr;
  ProfScope(int cls, hipStream_t stream, double flops, double bytes, bool ext_ = true) : ext(ext_), s(stream) {
    Profiler& p = profiler();
    if (!p.on) return;
    std::lock_guard<std::mutex> g(p.mu);
    active = true;
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    r.a = p.get(); r.b = p.get();
    if (ext) { prev = g_prof_cur; g_prof_cur = this; }
    else (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!active) return;
    if (ext) {
      g_prof_cur = prev;
      if (!used) { (void)hipEventRecord(r.a, s); (void)hipEventRecord(r.b, s); }  // nothing was launched inside
    } else {
      (void)hipEventRecord(r.b, s);
    }
    Profiler& p = profiler();
    std::lock_guard<std::mutex> g(p.mu);
    p.recs.push_back(r);
  }
};

// the innermost open single-launch scope hands its events to the first launch made inside it
static inline bool prof_take(hipEvent_t* a, hipEvent_t* b) {
  ProfScope* c = g_prof_cur;
  if (!c || !c->active || !c->ext || c->used) return false;
  c->used = true;
  *a = c->r.a; *b = c->r.b;
  return true;
}

#define IGI_LAUNCH(kernel, grid, block, shm, stream, ...)                                              \
  do {                                                                                                 \
    hipEvent_t _pa, _pb;                                                                               \
    if (igi::prof_take(&_pa, &_pb))                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, shm, stream, _pa, _pb, 0, __VA_ARGS__);               \
    else                                                                                               \
      hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);                               \
  } while (0)

}  // namespace igi
