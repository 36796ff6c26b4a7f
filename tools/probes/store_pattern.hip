#include <hip/hip_runtime.h>
#include <stdio.h>
// rows x 512 floats (2 KB rows); each 512-thread block owns a 128-row x 128-col tile like the GEMM epilogue
// SEG = contiguous bytes written per row by one wave-instruction: 128 (8 rows x 8 float4), 256, 512 (2 rows... )
template <int SEGF4>  // float4 per row segment per instruction: 8 -> 128 B, 16 -> 256 B, 32 -> 512 B
__global__ __launch_bounds__(512) void k(float* out, int ld, int mtiles, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = blockIdx.x % ntiles, mt = blockIdx.x / ntiles;
  constexpr int RPI = 64 / SEGF4;            // rows per instruction
  constexpr int WCOLS = SEGF4 * 4;           // columns a wave covers
  constexpr int WPR = 128 / WCOLS;           // waves per tile row-group
  constexpr int WROWS = 128 / (8 / WPR);     // rows a wave covers
  const int wn = wave % WPR, wm = wave / WPR;
  const int c4 = lane % SEGF4, rl = lane / SEGF4;
  float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
  for (int it = 0; it < WROWS / RPI; ++it) {
    const int row = mt * 128 + wm * WROWS + it * RPI + rl;
    const int col = nt * 128 + wn * WCOLS + 4 * c4;
    *reinterpret_cast<float4*>(out + (long long)row * ld + col) = v;
  }
}
__global__ void fill(float4* o, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) o[i] = make_float4(1, 2, 3, 4);
}
int main() {
  const int M = 32768, N = 512;  // 67 MB
  float* d; hipMalloc(&d, (size_t)M * N * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto fn) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(a); fn(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    printf("%-28s %.2f us  %.2f TB/s\n", name, best * 1e3, (double)M * N * 4 / (best * 1e-3) / 1e12);
  };
  const int mt = M / 128, nt = N / 128;
  run("fill contiguous", [&] { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (float4*)d, (long long)M * N / 4); });
  run("tile 128B segments (now)", [&] { hipLaunchKernelGGL(k<8>, dim3(mt * nt), dim3(512), 0, 0, d, N, mt, nt); });
  run("tile 256B segments", [&] { hipLaunchKernelGGL(k<16>, dim3(mt * nt), dim3(512), 0, 0, d, N, mt, nt); });
  run("tile 512B segments", [&] { hipLaunchKernelGGL(k<32>, dim3(mt * nt), dim3(512), 0, 0, d, N, mt, nt); });
  return 0;
}
