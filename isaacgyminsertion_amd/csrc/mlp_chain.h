// Backward of a chain of nn.Linear (+ fused activation) layers as ONE native call -- the student's small MLPs (lin
// encoder 15-64-32, point-cloud compress 512-64-32, decoder output stack 96-32-256-128-64-32 + action head 32-6:
// tact.py:137-158, 196-212, 337-339, 367-369, 407-410) under torch autograd ran, per layer, k_act_grad + data gradient +
// weight gradient + split sum = four launches of 4 - 8 us for a few MFLOP each (40 launches per optimizer step).  Here
//   * dz of the top layer: k_act_grad once;
//   * per layer ONE grid (gemm_dma_mlp_level_kernel): the weight gradient dW_l = dz_l^T x_l (split over the rows, slab
//     partials) next to the data gradient into the layer below, whose epilogue applies that layer's act' -- it writes
//     dz_{l-1} directly (no k_act_grad, no d(output) round trip);
//   * ONE fixed-order sum of every layer's weight / bias partials at the end (k_split_sum_multi) into a single flat
//     gradient buffer [dW_0 | db_0 | dW_1 | ...] (offsets multiples of four floats).
// Same products, same k-order, same split factors and the same order of additions as linear_backward (linear.h): the
// results are bit-identical to the per-layer path (tests/test_gpu_linear.py).  Layers the LDS-DMA kernel cannot take
// (row pitch not a multiple of 16 bytes: the 15-wide student observation, the 6-wide head) run the generic kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_dma.h"
#include "linear.h"

namespace igi {

constexpr int MLP_MAX_LAYERS = 8;

struct MlpPlan {
  int n;
  int in[MLP_MAX_LAYERS], out[MLP_MAX_LAYERS], sk[MLP_MAX_LAYERS], kchunk[MLP_MAX_LAYERS];
  long long o_w[MLP_MAX_LAYERS], o_b[MLP_MAX_LAYERS], grad_floats;      // flat gradient layout
  size_t w_dz[2], w_slabw[MLP_MAX_LAYERS], w_slabb[MLP_MAX_LAYERS], w_total;   // workspace (floats)
};

static int mlp_plan(long long rows, int n, const int* dims, MlpPlan* p) {
  if (n < 1 || n > MLP_MAX_LAYERS || rows < 1 || rows > (1LL << 30) || !dims) return IGI_E_BADARG;
  p->n = n;
  long long g = 0;
  size_t w = 0;
  int maxout = 0;
  for (int l = 0; l < n; ++l) {
    p->in[l] = dims[l]; p->out[l] = dims[l + 1];
    if (p->in[l] < 1 || p->out[l] < 1) return IGI_E_BADARG;
    if (p->out[l] > maxout) maxout = p->out[l];
    p->o_w[l] = g; g += ((long long)p->out[l] * p->in[l] + 3) & ~3LL;
    p->o_b[l] = g; g += (p->out[l] + 3) & ~3LL;
    int sk = linear_splitk(rows, p->in[l], p->out[l]), kc = 0;
    if (sk > 1) {   // whole 32-row k-tiles per split, no empty split (as linear_backward)
      kc = (int)(((rows + sk - 1) / sk + DMA_BK - 1) / DMA_BK * DMA_BK);
      sk = (int)((rows + kc - 1) / kc);
    }
    p->sk[l] = sk; p->kchunk[l] = kc;
  }
  p->grad_floats = g;
  for (int q = 0; q < 2; ++q) { p->w_dz[q] = w; w += ((size_t)rows * maxout + 3) / 4 * 4; }
  for (int l = 0; l < n; ++l) {
    p->w_slabw[l] = w; w += ((size_t)p->sk[l] * p->out[l] * p->in[l] + 3) / 4 * 4;
    p->w_slabb[l] = w; w += ((size_t)p->sk[l] * p->out[l] + 3) / 4 * 4;
  }
  p->w_total = w;
  return 0;
}

// x [rows][ldx]; W[l] (out_l, in_l) row-major; y[l] [rows][out_l] = the saved layer outputs (after the activation);
// dy [rows][out_{n-1}] = gradient w.r.t. the chain's output; dx [rows][in_0] or NULL; grads: the flat gradient buffer
// (mlp_plan: o_w / o_b); need_w[l] == 0: layer l is frozen (no weight / bias gradient, its range of grads is not
// written).
static int mlp_backward(const float* x, int ldx, long long rows, int n, const int* dims, const int* acts,
                        const float* const* W, const float* const* y, const float* dy, float* dx, float* grads,
                        const int* need_w, void* workspace, size_t workspace_bytes, hipStream_t s) {
  MlpPlan p;
  int rc = mlp_plan(rows, n, dims, &p);
  if (rc) return rc;
  if (!x || !W || !y || !dy || !acts || ldx < p.in[0] || (!grads && need_w)) return IGI_E_BADARG;
  if (!workspace || workspace_bytes < p.w_total * sizeof(float) + 16) return IGI_E_WORKSPACE;
  float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15);
  for (int l = 0; l < n; ++l)
    if (acts[l] < 0 || acts[l] > 2 || !W[l] || !y[l]) return IGI_E_BADARG;
  // dz of the top layer
  const float* dz = dy;
  int cur = 0;
  if (acts[n - 1] != LIN_NONE) {
    long long nb = (rows * p.out[n - 1] + 255) / 256;
    if (nb > 2048) nb = 2048;
    float* d = ws + p.w_dz[0];
    hipLaunchKernelGGL(k_act_grad, dim3((unsigned)nb), dim3(256), 0, s, dy, p.out[n - 1], y[n - 1], p.out[n - 1], d, rows,
                       p.out[n - 1], acts[n - 1]);
    dz = d;
    cur = 1;
  }
  SplitSumTable st;
  static_assert(2 * MLP_MAX_LAYERS <= MLP_SUM_SEGS, "one weight and one bias segment per layer must fit the sum table");
  auto add_seg = [&](float* dst, const float* src, long long cnt, long long stride, int parts) {
    if (!split_sum_defer(st, dst, src, cnt, stride, parts)) {   // table full (cannot happen: see the static_assert):
      split_sum_flush(st, s);                                    // what is queued goes out, then this segment
      split_sum_defer(st, dst, src, cnt, stride, parts);
    }
  };
  for (int l = n - 1; l >= 0; --l) {
    const int in = p.in[l], out = p.out[l];
    const float* xin = l == 0 ? x : y[l - 1];
    const int ldin = l == 0 ? ldx : p.in[l];
    const bool wneed = !need_w || need_w[l];
    GemmArgs wg, dg;
    if (wneed) {
      const int sk = p.sk[l];
      wg.A = dz; wg.lda = out;
      wg.B = xin; wg.ldb = ldin;
      wg.M = out; wg.N = in; wg.K = (int)rows;
      if (sk > 1) {
        wg.C = ws + p.w_slabw[l]; wg.ldc = in;
        wg.Cbias = ws + p.w_slabb[l];
        wg.splitk = sk; wg.kchunk = p.kchunk[l];
        wg.sCsplit = (long long)out * in; wg.sCbiasSplit = out;
        add_seg(grads + p.o_w[l], ws + p.w_slabw[l], (long long)out * in, (long long)out * in, sk);
        add_seg(grads + p.o_b[l], ws + p.w_slabb[l], out, out, sk);
      } else {
        wg.C = grads + p.o_w[l]; wg.ldc = in;
        wg.Cbias = grads + p.o_b[l];
      }
    }
    float* dnext = nullptr;
    const bool dneed = l > 0 || dx;
    if (dneed) {
      dg.A = dz; dg.lda = out;
      dg.B = W[l]; dg.ldb = in;
      dg.M = (int)rows; dg.N = in; dg.K = out;
      if (l > 0) {
        dnext = ws + p.w_dz[cur];
        dg.C = dnext; dg.ldc = in;
        const int a = acts[l - 1];
        if (a != LIN_NONE) {
          dg.aux = y[l - 1]; dg.ldaux = in;
          dg.epilogue = a == LIN_TANH ? EPI_TANHGRAD : EPI_RELUGRAD;
        }
      } else {
        dg.C = dx; dg.ldc = in;
      }
    }
    if ((rc = mlp_level(wneed ? &wg : nullptr, dneed ? &dg : nullptr, s))) return rc;
    if (l > 0) { dz = dnext; cur ^= 1; }
  }
  split_sum_flush(st, s);
  return (int)hipGetLastError();
}

}  // namespace igi
