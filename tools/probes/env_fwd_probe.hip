// phase stamps of one workgroup of k_env_fwd + launch timing:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o /tmp/envp tools/probes/env_fwd_probe.hip
#include <hip/hip_runtime.h>
__device__ long long g_ts[8][128];
#ifdef NO_TS
#define ENV_TS(i)
#else
#define ENV_TS(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 100) g_ts[threadIdx.x >> 6][(i)] = clock64(); } while (0)
#endif
#include "../../isaacgyminsertion_amd/csrc/env_mlp.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace igi;
int main(int argc, char** argv) {
  const int ldw2 = argc > 1 ? atoi(argv[1]) : 256;
  const int M = 16384, K1 = 64, N1 = 256, N2 = 128, N3 = 8;
  float *priv, *W1, *b1, *W2, *b2, *W3, *b3, *e1, *e2, *out;
  auto al = [](float** p, size_t n) { hipMalloc(p, n * 4); std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 5000.f - 0.1f; hipMemcpy(*p, h.data(), n * 4, hipMemcpyHostToDevice); };
  al(&priv, (size_t)M * K1); al(&W1, N1 * K1); al(&b1, N1); al(&W2, N2 * ldw2); al(&b2, N2); al(&W3, N3 * N2); al(&b3, N3);
  al(&e1, (size_t)(M + 64) * N1); al(&e2, (size_t)M * N2); al(&out, (size_t)M * 32);
  EnvFwdArgs a{priv, K1, W1, b1, W2, b2, W3, b3, e1, N1, e2, N2, out, 32, M, K1, N1, N2, N3, ldw2};
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 5; ++i) if (env_mlp_forward(a, s) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  hipEventRecord(ea, s);
  for (int i = 0; i < 50; ++i) env_mlp_forward(a, s);
  hipEventRecord(eb, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, ea, eb);
  printf("ldw2 %d: k_env_fwd %.2f us per launch (back to back)\n", ldw2, ms * 1000 / 50);
  long long ts[8][128]; hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof(ts));
  for (int w = 0; w < 4; ++w) {
    printf("wave %d (cycles from kernel entry):", w);
    for (int c = 0; c < 12; ++c) printf("  c%d mm %lld epi %lld |", c, ts[w][2 * c + 1] - ts[w][2 * c], (c < 11 ? ts[w][2 * c + 2] : ts[w][41]) - ts[w][2 * c + 1]);
    printf("\n  tail: z-epilogue %lld barrier %lld e2 stores %lld head %lld out %lld", ts[w][24] - ts[w][23], ts[w][25] - ts[w][24], ts[w][26] - ts[w][25], ts[w][27] - ts[w][26], ts[w][41] - ts[w][27]);
    printf("\n  total %lld, prologue %lld, first barrier at %lld\n", ts[w][41] - ts[w][40], ts[w][0] - ts[w][40], ts[w][1] - ts[w][40]);
  }
  return 0;
}
