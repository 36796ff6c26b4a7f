p = 'isaacgyminsertion_amd/csrc/teacher.h'
s = open(p).read()
# trunk dgrad: ride with the pending weight gradients
old = '''      g.aux = wsp<float>(st, p.w_h[l - 1]); g.ldaux = ru4(p.u[l - 1]); g.sAux = mbs * ru4(p.u[l - 1]);
      g.nbatch = 2;
      g.epilogue = EPI_TANHGRAD;
      IGI_HIP_TRY(gemm(g, true, false, s));
    } else if (do1) {'''
new = '''      g.aux = wsp<float>(st, p.w_h[l - 1]); g.ldaux = ru4(p.u[l - 1]); g.sAux = mbs * ru4(p.u[l - 1]);
      g.nbatch = 2;
      g.epilogue = EPI_TANHGRAD;
      // this layer's weight gradient needs the same dZ: it shares the data gradient's launch (gemm_level)
      IGI_HIP_TRY(gemm_level(g, wgrads, n_wgrads, s));
      n_wgrads = 0;
    } else if (do1) {'''
assert old in s
s = s.replace(old, new)
old = '''      g.aux = wsp<float>(st, p.w_e[l - 1]); g.ldaux = ru4(in);
      g.epilogue = EPI_TANHGRAD;
      IGI_HIP_TRY(gemm(g, true, false, s));
    }
  }

  IGI_HIP_TRY(gemm_wgrad_group(wgrads, n_wgrads, s));
'''
new = '''      g.aux = wsp<float>(st, p.w_e[l - 1]); g.ldaux = ru4(in);
      g.epilogue = EPI_TANHGRAD;
      IGI_HIP_TRY(gemm_level(g, wgrads, n_wgrads, s));   // + this layer's (and the first trunk layer's) weight gradient
      n_wgrads = 0;
    }
  }

  IGI_HIP_TRY(gemm_wgrad_group(wgrads, n_wgrads, s));
'''
assert old in s
s = s.replace(old, new)
s = s.replace('''  // ---- backward through the actor / critic trunk.  The weight-gradient products are only
  //      collected here; they run as grouped launches once every dZ exists (gemm_wgrad_group)''',
'''  // ---- backward through the actor / critic trunk.  Level fusion: the weight gradient of layer l and the data
  //      gradient INTO layer l-1 both consume dZ_l and are independent of each other, so they share one grid
  //      (gemm_level -> gemm_dma_wgrad_multi_kernel: the data-gradient tiles lead, the weight-gradient workgroups fill
  //      their fill / drain bubbles).  The first trunk layer's weight gradient rides with the env_mlp data gradient.''')
open(p, 'w').write(s)
p = 'isaacgyminsertion_amd/csrc/gemm_dma.h'
s = open(p).read()
old = "}  // namespace igi"
i = s.rindex(old)
add = '''// One level of a backward chain: the data gradient `dgrad` (k-contiguous dZ times reduction-major W, tanh' epilogue)
// together with the weight-gradient products that are ready at this point, in one grid when the shapes allow it.
static hipError_t gemm_level(const GemmArgs& dgrad, GemmArgs* wgrads, int count, hipStream_t s) {
  static int fuse = -1;
  if (fuse < 0) { const char* e = getenv("IGI_LEVEL_FUSE"); fuse = e ? atoi(e) : 1; }
  if (fuse && count > 0 && !bf16_mode()) {
    hipError_t e = gemm_wgrad_multi(wgrads, count, s, &dgrad);
    if (e != hipErrorNotSupported) return e;
  }
  hipError_t e = gemm(dgrad, true, false, s);
  if (e != hipSuccess) return e;
  return count > 0 ? gemm_wgrad_group(wgrads, count, s) : hipSuccess;
}

'''
s = s[:i] + add + s[i:]
open(p, 'w').write(s)
print("ok")
