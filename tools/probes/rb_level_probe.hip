// Standalone check + timing of k_rb_level (csrc/rowblock.h) against a double-precision host evaluation of the level:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/probes/rb_level_probe.hip -o /tmp/rb_probe
//   /tmp/rb_probe [rows=16384] [IN=256] [nets=2] [iters=200] [check=1] [lowx=0]   (lowx = 1, nets = 1: the layer below's weight
//   gradient from the data-gradient tiles, dX not stored)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../isaacgyminsertion_amd/csrc/rowblock.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 16384, IN = argc > 2 ? atoi(argv[2]) : 256, nets = argc > 3 ? atoi(argv[3]) : 2;
  const int iters = argc > 4 ? atoi(argv[4]) : 200, check = argc > 5 ? atoi(argv[5]) : 1, lowx = argc > 6 ? atoi(argv[6]) : 0;
  const int KO = igi::RB_KO;
  const int ranges = igi::rb_level_ranges(rows, IN, nets);
  printf("rows %d IN %d nets %d ranges %d grid %d LDS %zu B\n", rows, IN, nets, ranges, nets * ranges * (IN / 64),
         sizeof(float) * igi::RB_LDS_FLOATS);
  const size_t nZ = (size_t)nets * rows * KO, nX = (size_t)nets * rows * IN, nW = (size_t)nets * KO * IN;
  std::vector<float> hZ(nZ), hX(nX), hW(nW);
  unsigned seed = 12345;
  for (auto& v : hZ) v = 0.01f * frand(seed);
  for (auto& v : hX) v = tanhf(1.5f * frand(seed));
  for (auto& v : hW) v = 0.1f * frand(seed);
  float *dZ, *dXin, *dW, *dOut, *dWp, *dBp;
  CK(hipMalloc(&dZ, nZ * 4)); CK(hipMalloc(&dXin, nX * 4)); CK(hipMalloc(&dW, nW * 4)); CK(hipMalloc(&dOut, nX * 4));
  CK(hipMalloc(&dWp, (size_t)ranges * nW * 4)); CK(hipMalloc(&dBp, (size_t)ranges * nets * KO * 4));
  CK(hipMemcpy(dZ, hZ.data(), nZ * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dXin, hX.data(), nX * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), nW * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dOut, 0xff, nX * 4)); CK(hipMemset(dWp, 0xff, (size_t)ranges * nW * 4));
  CK(hipMemset(dBp, 0xff, (size_t)ranges * nets * KO * 4));
  igi::RbLevelArgs a;
  a.dZ = dZ; a.ldz = KO; a.sZ = (long long)rows * KO;
  a.W = dW; a.ldw = IN; a.sW = (long long)KO * IN;
  a.X = dXin; a.ldx = IN; a.sX = (long long)rows * IN;
  a.dX = dOut; a.lddx = IN; a.sdX = (long long)rows * IN;
  a.dWp = dWp; a.ldwp = IN; a.sWpart = (long long)nets * KO * IN; a.sWnet = (long long)KO * IN;
  a.dBp = dBp; a.sBpart = (long long)nets * KO; a.sBnet = KO;
  a.rows = rows; a.IN = IN; a.nets = nets; a.ranges = ranges;
  std::vector<float> hXb;
  float *dXb = nullptr, *dLW = nullptr, *dLB = nullptr;
  if (lowx) {
    if (nets != 1) { printf("lowx needs nets = 1\n"); return 1; }
    hXb.resize((size_t)rows * 64);
    for (auto& v : hXb) v = frand(seed);
    CK(hipMalloc(&dXb, hXb.size() * 4)); CK(hipMalloc(&dLW, (size_t)ranges * IN * 64 * 4)); CK(hipMalloc(&dLB, (size_t)ranges * IN * 4));
    CK(hipMemcpy(dXb, hXb.data(), hXb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dLW, 0xff, (size_t)ranges * IN * 64 * 4)); CK(hipMemset(dLB, 0xff, (size_t)ranges * IN * 4));
    a.lx_X = dXb; a.lx_ld = 64; a.lx_W = dLW; a.lx_ldw = 64; a.lx_sPart = (long long)IN * 64; a.lx_B = dLB; a.lx_bsPart = IN;
  }
  hipStream_t s; CK(hipStreamCreate(&s));
  hipError_t e = igi::rb_level_backward(a, s, igi::PC_OTHER);
  if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; }
  CK(hipStreamSynchronize(s));
  int bad = 0;
  if (check) {
    std::vector<float> oX(nX), oW((size_t)ranges * nW), oB((size_t)ranges * nets * KO);
    CK(hipMemcpy(oX.data(), dOut, nX * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(oW.data(), dWp, oW.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(oB.data(), dBp, oB.size() * 4, hipMemcpyDeviceToHost));
    if (lowx) {
      // the second product against fp64: dWb[c][j] = sum_r dX[r][c] Xb[r][j], dBb[c] = sum_r dX[r][c], dX evaluated in double
      std::vector<float> oLW((size_t)ranges * IN * 64), oLB((size_t)ranges * IN);
      CK(hipMemcpy(oLW.data(), dLW, oLW.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(oLB.data(), dLB, oLB.size() * 4, hipMemcpyDeviceToHost));
      std::vector<double> rW((size_t)IN * 64, 0.0), rB(IN, 0.0), mW((size_t)IN * 64, 0.0);
      std::vector<double> dx(IN);
      for (int r = 0; r < rows; ++r) {
        for (int n = 0; n < IN; ++n) {
          double acc = 0;
          for (int k = 0; k < KO; ++k) acc += (double)hZ[(size_t)r * KO + k] * hW[(size_t)k * IN + n];
          const double x = hX[(size_t)r * IN + n];
          dx[n] = acc * (1.0 - x * x);
        }
        for (int n = 0; n < IN; ++n) {
          rB[n] += dx[n];
          for (int j = 0; j < 64; ++j) { rW[(size_t)n * 64 + j] += dx[n] * hXb[(size_t)r * 64 + j]; mW[(size_t)n * 64 + j] += fabs(dx[n] * hXb[(size_t)r * 64 + j]); }
        }
      }
      double e1 = 0, m1 = 0;
      for (size_t i = 0; i < rW.size(); ++i) {
        double got = 0;
        for (int p = 0; p < ranges; ++p) got += oLW[(size_t)p * IN * 64 + i];
        if (!(fabs(got - rW[i]) <= 2e-6 * mW[i] + 1e-7)) { if (bad < 5) printf("dWbelow[%zu] %g vs %g\n", i, got, rW[i]); ++bad; }
        e1 = fmax(e1, fabs(got - rW[i])); m1 = fmax(m1, fabs(rW[i]));
      }
      printf("dW below: max err %.3g (max |ref| %.3g)\n", e1, m1);
      e1 = m1 = 0;
      for (int n = 0; n < IN; ++n) {
        double got = 0;
        for (int p = 0; p < ranges; ++p) got += oLB[(size_t)p * IN + n];
        if (!(fabs(got - rB[n]) <= 2e-5 + 1e-4 * fabs(rB[n]))) { if (bad < 10) printf("dBbelow[%d] %g vs %g\n", n, got, rB[n]); ++bad; }
        e1 = fmax(e1, fabs(got - rB[n])); m1 = fmax(m1, fabs(rB[n]));
      }
      printf("dB below: max err %.3g (max |ref| %.3g)\n", e1, m1);
    }
    // data gradient on a sample of rows, weight gradient in full
    double emax = 0, rmax = 0;
    for (int net = 0; net < nets; ++net)
      for (int r = 0; r < rows && !lowx; r += 37) {
        for (int n = 0; n < IN; ++n) {
          double acc = 0;
          for (int k = 0; k < KO; ++k) acc += (double)hZ[((size_t)net * rows + r) * KO + k] * hW[((size_t)net * KO + k) * IN + n];
          const double x = hX[((size_t)net * rows + r) * IN + n];
          const double ref = acc * (1.0 - x * x);
          const double got = oX[((size_t)net * rows + r) * IN + n];
          if (!(fabs(got - ref) <= 1e-6 + 1e-4 * fabs(ref))) { if (bad < 5) printf("dX[%d][%d][%d] %g vs %g\n", net, r, n, got, ref); ++bad; }
          emax = fmax(emax, fabs(got - ref)); rmax = fmax(rmax, fabs(ref));
        }
      }
    printf("dX: max err %.3g (max |ref| %.3g)\n", emax, rmax);
    emax = rmax = 0;
    std::vector<double> refW(nW, 0.0), refB((size_t)nets * KO, 0.0);
    for (int net = 0; net < nets; ++net)
      for (int r = 0; r < rows; ++r) {
        const float* z = &hZ[((size_t)net * rows + r) * KO];
        const float* x = &hX[((size_t)net * rows + r) * IN];
        for (int o = 0; o < KO; ++o) {
          const double zz = z[o];
          refB[(size_t)net * KO + o] += zz;
          double* w = &refW[((size_t)net * KO + o) * IN];
          for (int n = 0; n < IN; ++n) w[n] += zz * x[n];
        }
      }
    for (size_t i = 0; i < nW; ++i) {
      double got = 0;
      for (int p = 0; p < ranges; ++p) got += oW[(size_t)p * nW + i];
      const double ref = refW[i];
      if (!(fabs(got - ref) <= 2e-5 + 1e-4 * fabs(ref))) { if (bad < 10) printf("dW[%zu] %g vs %g\n", i, got, ref); ++bad; }
      emax = fmax(emax, fabs(got - ref)); rmax = fmax(rmax, fabs(ref));
    }
    printf("dW: max err %.3g (max |ref| %.3g)\n", emax, rmax);
    emax = rmax = 0;
    for (size_t i = 0; i < (size_t)nets * KO; ++i) {
      double got = 0;
      for (int p = 0; p < ranges; ++p) got += oB[(size_t)p * nets * KO + i];
      if (!(fabs(got - refB[i]) <= 2e-5 + 1e-4 * fabs(refB[i]))) { if (bad < 15) printf("dB[%zu] %g vs %g\n", i, got, refB[i]); ++bad; }
      emax = fmax(emax, fabs(got - refB[i])); rmax = fmax(rmax, fabs(refB[i]));
    }
    printf("dB: max err %.3g (max |ref| %.3g)\n", emax, rmax);
    printf(bad ? "MISMATCH: %d\n" : "parity ok\n", bad);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) igi::rb_level_backward(a, s, igi::PC_OTHER);
  CK(hipStreamSynchronize(s));
  igi::profiler().on = true;   // every launch then carries its own start / stop timestamps (the kernel's duration, as rocprofv3 reports it)
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) igi::rb_level_backward(a, s, igi::PC_OTHER);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / iters, fl = 4.0 * nets * (double)rows * KO * IN + (lowx ? 2.0 * rows * (double)IN * 64 : 0.0);
  {
    double sum = 0, mn = 1e9;
    for (auto& r : igi::profiler().recs) { float t = 0; CK(hipEventElapsedTime(&t, r.a, r.b)); sum += t; mn = fmin(mn, t); }
    printf("k_rb_level: kernel duration avg %.2f us, min %.2f us over %zu launches\n", 1e3 * sum / igi::profiler().recs.size(), 1e3 * mn,
           igi::profiler().recs.size());
  }
  printf("k_rb_level: %.2f us per launch (back to back), %.1f TFLOP/s = %.3f of 157.3\n", us, fl / us * 1e-6, fl / us * 1e-6 / 157.3);
  return bad ? 2 : 0;
}
