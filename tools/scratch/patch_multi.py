p = 'isaacgyminsertion_amd/csrc/gemm_dma.h'
s = open(p).read()
s = s.replace('''template <int BN, bool A_KC, bool B_KC, int GATHER, int NS, int BM = DMA_BM, bool STORE_ONLY = false, bool HEAD = false,
          bool BF16IN = false>
__device__ __forceinline__ void gemm_dma_body(''', '''template <int BN, bool A_KC, bool B_KC, int GATHER, int NS, int BM = DMA_BM, bool STORE_ONLY = false, bool HEAD = false,
          bool BF16IN = false, bool TANHGRAD_ONLY = false>
__device__ __forceinline__ void gemm_dma_body(''')
s = s.replace("    if (STORE_ONLY) IGI_EPI_ROWS(EPI_STORE);\n",
              "    if (STORE_ONLY) IGI_EPI_ROWS(EPI_STORE);\n    else if (TANHGRAD_ONLY) IGI_EPI_ROWS(EPI_TANHGRAD);\n")
s = s.replace("  if (STORE_ONLY) { IGI_EPI_CALL(EPI_STORE, false); }\n",
              "  if (STORE_ONLY) { IGI_EPI_CALL(EPI_STORE, false); }\n  else if (TANHGRAD_ONLY) { IGI_EPI_CALL(EPI_TANHGRAD, false); }\n")
s = s.replace('''  int kind[DMA_MULTI_MAX];   // 0: 128 x 128 tiles, 1: 128 x 64 tiles''',
              '''  int kind[DMA_MULTI_MAX];   // weight gradients (reduction-major operands, plain store): 0 = 128 x 128 tiles, 1 = 128 x 64;
                             // 2 = data gradient dZ.W times tanh' (A k-contiguous, B reduction-major), 128 x 128 tiles''')
old = '''  if (gr->kind[p] == 0) gemm_dma_body<128, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else gemm_dma_body<64, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
}'''
new = '''  const int kind = gr->kind[p];
  if (kind == 0) gemm_dma_body<128, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else if (kind == 1) gemm_dma_body<64, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else gemm_dma_body<128, true, false, 0, 2, DMA_BM, false, false, false, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
}'''
assert old in s
s = s.replace(old, new)
old = '''static hipError_t gemm_wgrad_multi(GemmArgs* list, int count, hipStream_t s) {
  GemmMulti mt_;
  double fl = 0, by = 0;
'''
new = '''// dgrad (optional): a data-gradient product (A k-contiguous, B reduction-major, tanh' epilogue) whose tiles lead the
// grid -- it is on the critical path of the backward chain, the weight-gradient workgroups fill its fill / drain
// bubbles and the tail.  Returns hipErrorNotSupported when dgrad cannot ride (the caller launches it on its own).
static hipError_t gemm_wgrad_multi(GemmArgs* list, int count, hipStream_t s, const GemmArgs* dgrad = nullptr) {
  GemmMulti mt_;
  double fl = 0, by = 0;
  if (dgrad) {
    GemmArgs g = *dgrad;
    const long long mtl = (g.M + DMA_BM - 1) / DMA_BM, ntl = (g.N + 127) / 128;
    if (g.epilogue != EPI_TANHGRAD || g.splitk != 1 || g.gather || !dma_eligible(g, true, false) ||
        mtl * ntl * g.nbatch > (1 << 20))
      return hipErrorNotSupported;
    g.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.N & 3) == 0 &&
                 (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0 && (g.sAux & 3) == 0));
    mt_.g[0] = g; mt_.n_tiles[0] = (int)ntl; mt_.m_tiles[0] = (int)mtl; mt_.kind[0] = 2;
    mt_.tile_end[0] = (int)(mtl * ntl * g.nbatch);
    mt_.n = 1;
    fl += 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
    by += 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + 2.0 * (double)g.M * g.N);
  }
'''
assert old in s
s = s.replace(old, new)
old = '''  ProfScope ps(PC_WGRAD_MULTI, s, fl, by);
  IGI_LAUNCH(gemm_dma_wgrad_multi_kernel,'''
new = '''  ProfScope ps(dgrad ? PC_DGRAD_WGRAD_MULTI : PC_WGRAD_MULTI, s, fl, by);
  IGI_LAUNCH(gemm_dma_wgrad_multi_kernel,'''
assert old in s
s = s.replace(old, new)
open(p, 'w').write(s)
p = 'isaacgyminsertion_amd/csrc/prof.h'
s = open(p).read()
s = s.replace("PC_GROUP_128_FF, PC_GROUP_64_FF, PC_WGRAD_MULTI, PC_DMA_HEAD,",
              "PC_GROUP_128_FF, PC_GROUP_64_FF, PC_WGRAD_MULTI, PC_DGRAD_WGRAD_MULTI, PC_DMA_HEAD,")
s = s.replace('"gemm_dma_wgrad_multi_kernel", "gemm_dma_head_kernel<true>",',
              '"gemm_dma_wgrad_multi_kernel", "gemm_dma_wgrad_multi_kernel (dgrad + wgrad)", "gemm_dma_head_kernel<true>",')
open(p, 'w').write(s)
print("patched")
