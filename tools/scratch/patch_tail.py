p = 'isaacgyminsertion_amd/csrc/gemm_f32.h'
s = open(p).read()
old = "  int head_ld = 0, head_n = 0;\n"
new = '''  int head_ld = 0, head_n = 0;
  // "tail" layer fused behind the head (gemm_dma.h, HEAD kernel): the workgroup that just produced head rows m0..m0+63
  // also computes tail_C[z][m][n] = tanh(sum_k x[m][k] tail_W[z][n][k] + tail_b[z][n]) for z < tail_nb, n < tail_n, with
  // x[m] = [tail_x[m][0 .. tail_xcols) | head output (head_n values) | zeros] (32 wide = one k-tile): the first
  // actor / critic trunk layer on [obs | latent] (models_split.py:199-221).  tail_W is [tail_nb][tail_n][32], zero padded.
  const float* tail_x = nullptr;
  const float* tail_W = nullptr;
  const float* tail_b = nullptr;
  float* tail_C = nullptr;
  int tail_ldx = 0, tail_xcols = 0, tail_n = 0, tail_nb = 0, tail_ldc = 0;
  long long tail_sW = 0, tail_sB = 0, tail_sC = 0;
'''
assert old in s
s = s.replace(old, new, 1)
open(p, 'w').write(s)

p = 'isaacgyminsertion_amd/csrc/gemm_dma.h'
s = open(p).read()
old = '''        g.head_out[(long long)(m0 + r) * g.head_ld + q] = fast_tanh(hacc + g.head_b[q]);
      }
      return;
    }'''
new = '''        const float hv = fast_tanh(hacc + g.head_b[q]);
        g.head_out[(long long)(m0 + r) * g.head_ld + q] = hv;
        if (g.tail_C) {  // k-contiguous swizzled A image of the tail layer's input row (see below)
          const int k = g.tail_xcols + q;
          xs_tail[(r * 8 + ((k >> 2) ^ ((r >> 1) & 7))) * 4 + (k & 3)] = hv;
        }
      }
      if (!g.tail_C) return;
      // ---- tail layer: out[64][tail_nb * tail_n] = tanh(x[64][32] . W^T + b), one 32-k tile, N in 128-column slices.
      // x lives in LDS as the k-contiguous XOR-swizzled image the main loop reads ([row][8 units of 16 B], unit p of
      // row m holds k-group p ^ ((m >> 1) & 7)); the weight slices arrive by LDS-DMA into a two-deep ring; the MFMA
      // order (pairs k, k+4 inside every group of eight) is the main loop's, so the result is bit-identical to the
      // layer run as its own launch.
      {
        float* bst = xs_tail + 64 * DMA_BK;                       // two stages of 128 x 32 floats
        const int ntl = (g.tail_n + 127) / 128, nsl = ntl * g.tail_nb;
        auto issue_w = [&](int j) {
          const int z = j / ntl, c0 = (j - z * ntl) * 128;
          dma_tile<128, true>(g.tail_W + z * g.tail_sW, DMA_BK, c0, g.tail_n, 0, bst + (j & 1) * (128 * DMA_BK), wave, lane);
        };
        issue_w(0);
        // the other columns of x: tail_xcols leading values from tail_x, zeros behind the head outputs
        for (int e = tid; e < 64 * DMA_BK; e += DMA_THREADS) {
          const int r = e >> 5, k = e & 31;
          if (k >= g.tail_xcols && k < g.tail_xcols + g.head_n) continue;   // written by the head lanes above
          const int row = min(m0 + r, g.M - 1);
          const float v = (k < g.tail_xcols) ? g.tail_x[(long long)row * g.tail_ldx + k] : 0.f;
          xs_tail[(r * 8 + ((k >> 2) ^ ((r >> 1) & 7))) * 4 + (k & 3)] = v;
        }
        for (int j = 0; j < nsl; ++j) {
          if (j + 1 < nsl) { issue_w(j + 1); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();                                      // slice j landed for every wave; x is complete
          const float* bs2 = bst + (j & 1) * (128 * DMA_BK);
          f32x16 acc2;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
          const int ma = wm * WTM + l31, mb_ = wn * WTN + l31;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(xs_tail + (ma * 8 + ((2 * c + h) ^ ((ma >> 1) & 7))) * 4);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bs2 + (mb_ * 8 + ((2 * c + h) ^ ((mb_ >> 1) & 7))) * 4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bv[jj], acc2, 0, 0, 0);
          }
          // stage through this wave's own LDS slice (the head is done with it) -> bias + tanh -> 16-byte stores
#pragma unroll
          for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EPLD + l31] = acc2[r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int z = j / ntl, c0 = (j - z * ntl) * 128;
          epilogue_rows<EPI_BIAS_TANH, WTM, WTN>(ep, g.tail_C + z * g.tail_sC, g.tail_ldc, g.tail_b + z * g.tail_sB, nullptr,
                                                 0, m0 + wm * WTM, c0 + wn * WTN, g.M, g.tail_n, lane);
          __syncthreads();                                      // every wave is past its reads of stage j & 1
        }
      }
      return;
    }'''
assert old in s
s = s.replace(old, new, 1)
# xs_tail declaration: right after wsh is defined in the HEAD block
old = '''      float* wsh = smem + DMA_WAVES * (WTM * EPLD);
      for (int e = tid; e < g.head_n * g.N; e += DMA_THREADS) wsh[e] = g.head_W[e];'''
new = '''      float* wsh = smem + DMA_WAVES * (WTM * EPLD);
      float* xs_tail = wsh + 8 * 128;   // [64][32] input image of the fused tail layer, then its weight ring
      for (int e = tid; e < g.head_n * g.N; e += DMA_THREADS) wsh[e] = g.head_W[e];'''
assert old in s
s = s.replace(old, new, 1)
# launcher: LDS size + validation + flops
old = '''  constexpr size_t epi = sizeof(float) * (DMA_WAVES * (BM / 2) * (32 + 4) + 8 * 128);  // staging + head weights
  constexpr size_t shm = ring > epi ? ring : epi;
  const double fl = 2.0 * g.M * (double)g.N * (g.K + g.head_n);
  const double by = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);'''
new = '''  // staging + head weights + (tail layer) its 64 x 32 input image and two 128 x 32 weight stages
  constexpr size_t epi = sizeof(float) * (DMA_WAVES * (BM / 2) * (32 + 4) + 8 * 128 + 64 * DMA_BK + 2 * 128 * DMA_BK);
  constexpr size_t shm = ring > epi ? ring : epi;
  if (g.tail_C) {
    if (!g.tail_x || !g.tail_W || !g.tail_b || g.tail_nb < 1 || g.tail_n < 4 || (g.tail_n & 3) || (g.tail_ldc & 3) ||
        (g.tail_sC & 3) || (g.tail_sB & 3) || (g.tail_sW & 3) || !aligned16(g.tail_C) || !aligned16(g.tail_W) ||
        !aligned16(g.tail_b) || g.tail_xcols < 0 || g.tail_xcols + g.head_n > DMA_BK || g.tail_ldx < g.tail_xcols)
      return hipErrorInvalidValue;
  }
  const double fl = 2.0 * g.M * (double)g.N * (g.K + g.head_n) +
                    (g.tail_C ? 2.0 * g.M * (double)g.tail_n * g.tail_nb * DMA_BK * g.flop_credit : 0.0);
  const double by = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N) +
                    (g.tail_C ? 4.0 * (double)g.M * g.tail_n * g.tail_nb : 0.0);'''
assert old in s
s = s.replace(old, new, 1)
open(p, 'w').write(s)

p = 'isaacgyminsertion_amd/csrc/teacher.h'
s = open(p).read()
old = '''      gh.head_out = xcat + p.obs; gh.head_ld = p.xld; gh.head_n = p.pu[l + 1];
      if (gemm_with_head(gh, s) == hipSuccess) break;'''
new = '''      gh.head_out = xcat + p.obs; gh.head_ld = p.xld; gh.head_n = p.pu[l + 1];
      static int fuse_tail = -1;
      if (fuse_tail < 0) { const char* e = getenv("IGI_FUSE_TRUNK1"); fuse_tail = e ? atoi(e) : 1; }
      if (fuse_tail && p.xld == DMA_BK && p.xw == p.obs + gh.head_n && (p.u0p & 3) == 0 && !bf16_mode()) {
        // ... and so does the first actor / critic trunk layer on [obs | latent] (one k-tile): the workgroup that
        // has just produced 64 latent rows computes their 2 x 512 first-layer activations too (a launch less)
        GemmArgs gt = gh;
        gt.tail_x = xcat; gt.tail_ldx = p.xld; gt.tail_xcols = p.obs;
        gt.tail_W = w1p; gt.tail_sW = (long long)p.u0p * p.xld;
        gt.tail_b = P + p.o_acB[0]; gt.tail_sB = p.ac_block;
        gt.tail_C = wsp<float>(st, p.w_h[0]); gt.tail_ldc = ru4(p.u[0]); gt.tail_sC = mbs * ru4(p.u[0]);
        gt.tail_n = p.u[0]; gt.tail_nb = 2;
        gt.flop_credit = (double)p.xw / p.xld;
        if (gemm_with_head(gt, s) == hipSuccess) { trunk1_done = true; break; }
      }
      if (gemm_with_head(gh, s) == hipSuccess) break;'''
assert old in s
s = s.replace(old, new, 1)
old = '''  static int fuse_head = -1;
  if (fuse_head < 0) { const char* e = getenv("IGI_FUSE_HEAD"); fuse_head = e ? atoi(e) : 1; }'''
new = '''  static int fuse_head = -1;
  if (fuse_head < 0) { const char* e = getenv("IGI_FUSE_HEAD"); fuse_head = e ? atoi(e) : 1; }
  bool trunk1_done = false;'''
assert old in s
s = s.replace(old, new, 1)
old = '''  for (int l = 0; l < p.nl; ++l) {
    GemmArgs g;
    g.A = in; g.lda = ldin; g.sA = sIn;
    if (l == 0) { g.B = w1p;'''
new = '''  for (int l = 0; l < p.nl; ++l) {
    GemmArgs g;
    g.A = in; g.lda = ldin; g.sA = sIn;
    if (l == 0 && trunk1_done) {   // computed behind the latent head
      in = wsp<float>(st, p.w_h[0]); ldin = ru4(p.u[0]); sIn = mbs * ru4(p.u[0]);
      continue;
    }
    if (l == 0) { g.B = w1p;'''
assert old in s
s = s.replace(old, new, 1)
open(p, 'w').write(s)
print("ok")
