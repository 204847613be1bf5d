// Instantiations of the fused kernels for ONE padded mode count (AOG_INST_APAD), all table counts and sin/cos flavours.
#include "aogym_internal.h"
#include "aogym_kernels.h"

#ifndef AOG_INST_APAD
#error "compile with -DAOG_INST_APAD=16|32|64|128"
#endif

namespace {
using aog_host::round_up;
// ---- fused kernel dispatch -------------------------------------------------------------------------
template <int A_PAD, int MRW, int SC>
void launch_valu(aog_env* e, hipStream_t s) {
  const int n_groups = e->Bp / 64;
  dim3 grid(e->valu_chunks, (n_groups + 3) / 4);
  const float ratio = (float)(e->cfg.wavelength_wfs / e->cfg.wavelength_sci);
  hipLaunchKernelGGL((aog::k_fused_valu<A_PAD, MRW, 1, SC>), grid, dim3(256), 0, s, e->modes_f32, e->tabs_f32,
                     reinterpret_cast<const float4*>(e->psi_rev), e->act_rev, e->partials, e->n_quads, e->Bp, n_groups,
                     e->valu_qpc, ratio);
}

template <int A_PAD, int MRW, int SC>
void launch_mfma(aog_env* e, hipStream_t s) {
  aog::MfmaGeom g;
  g.n_ptiles = e->n_ptiles;
  g.n_etiles = e->n_etiles;
  g.Bp = e->Bp;
  g.P = e->mfma_chunks_x;
  g.we = e->mfma_we;
  g.wg_y = (e->n_etiles + e->mfma_we - 1) / e->mfma_we;
  g.max_tiles = e->mfma_tpc;
  static const int skew = getenv("AOG_SKEW_NOPS") ? atoi(getenv("AOG_SKEW_NOPS")) : aog::kSkewNops;
  g.skew = skew;
  dim3 grid(round_up(g.P, 8) * g.wg_y);
  const float ratio = (float)(e->cfg.wavelength_wfs / e->cfg.wavelength_sci);
  if constexpr (MRW <= 8) {
    if (e->tab_mfma && e->fused_t16 && e->mfma_we == 4) {   // (we == 4: one slab per pixel chunk, as this kernel writes them)
      aog::MfmaGeom g16 = g;
      g16.wg_y = (e->n_etiles + 1) / 2;
      const size_t lds16 = (size_t)e->mfma_tpc * 8 * 16 + (size_t)2 * (A_PAD / 16) * 2 * 64 * 16;   // science rows + two mode-tile buffers
#define AOG_T16_LAUNCH(ABL)                                                                                                                      \
  hipLaunchKernelGGL((aog::k_fused_t16<A_PAD, MRW, ABL>), dim3(round_up(g16.P, 8) * g16.wg_y), dim3(256), lds16, s,                                 \
                     reinterpret_cast<const _Float16*>(e->modes16), reinterpret_cast<const _Float16*>(e->tab16),                                   \
                     reinterpret_cast<const aog::f32x4*>(e->sci_tile), reinterpret_cast<const aog::f32x4*>(e->psi_tile),                           \
                     reinterpret_cast<const _Float16*>(e->act16), e->partials, g16, ratio)
      if constexpr (A_PAD == 64 && MRW == 7) {   // timing-only ablations of the developer variant (AOG_ABLATE=1..4)
        if (e->ablate == 1) { AOG_T16_LAUNCH(1); return; }
        if (e->ablate == 2) { AOG_T16_LAUNCH(2); return; }
        if (e->ablate == 3) { AOG_T16_LAUNCH(3); return; }
        if (e->ablate == 4) { AOG_T16_LAUNCH(4); return; }
      }
      AOG_T16_LAUNCH(0);
#undef AOG_T16_LAUNCH
      return;
    }
  }
  if (e->tab_mfma) {
    const size_t lds_t = (size_t)e->mfma_tpc * 8 * 16 + ((MRW <= 8 || A_PAD > 64) ? (size_t)4 * (A_PAD / 16) * 2 * 64 * 16 : 0);   // science rows (+ actuator operands)
    if (lds_t > 64 * 1024 && !e->tab_attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(aog::k_fused_tab<A_PAD, MRW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
      e->tab_attr_set = true;
    }
    hipLaunchKernelGGL((aog::k_fused_tab<A_PAD, MRW>), grid, dim3(256), lds_t, s, reinterpret_cast<const aog::f16x8*>(e->modes16),
                       reinterpret_cast<const aog::f16x8*>(e->tab16), reinterpret_cast<const aog::f32x4*>(e->sci_tile),
                       reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio);
    return;
  }
  size_t lds = (size_t)e->mfma_tpc * 8 * (MRW + 1) * 16;
  if (const char* pad = getenv("AOG_LDS_PAD_KB")) lds = std::max(lds, (size_t)atoi(pad) * 1024);   // developer aid: force one workgroup per CU
  g.max_tiles = getenv("AOG_NO_HOIST") ? -e->mfma_tpc : e->mfma_tpc;
  if constexpr (A_PAD == 64 && MRW == 7 && SC == 2) {
    if (e->ablate == 1) {
      hipLaunchKernelGGL((aog::k_fused_mfma<A_PAD, MRW, 1, SC, 1>), grid, dim3(256), lds, s,
                         reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->tabs_tile),
                         reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio);
      return;
    }
#define AOG_ABL_CASE(N)                                                                                                   \
    if (e->ablate == N) {                                                                                                 \
      hipLaunchKernelGGL((aog::k_fused_mfma<A_PAD, MRW, 1, SC, N>), grid, dim3(256), lds, s,                              \
                         reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->tabs_tile), \
                         reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio);           \
      return;                                                                                                             \
    }
    AOG_ABL_CASE(2)
    AOG_ABL_CASE(3)
    AOG_ABL_CASE(4)
    AOG_ABL_CASE(5)
    AOG_ABL_CASE(6)
#undef AOG_ABL_CASE
  }
  if constexpr (SC == 2 && (MRW == 28 || MRW == 7)) {
    if (e->ablate == 7) {   // self-checking build (prints on a mismatch)
      hipLaunchKernelGGL((aog::k_fused_mfma<A_PAD, MRW, 1, SC, 7>), grid, dim3(256), lds, s,
                         reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->tabs_tile),
                         reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio);
      return;
    }
  }
  hipLaunchKernelGGL((aog::k_fused_mfma<A_PAD, MRW, 1, SC>), grid, dim3(256), lds, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->tabs_tile),
                     reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio);
}

template <int A_PAD, int MRW>
void launch_fast2(aog_env* e, hipStream_t s) {
  if (e->kernel == AOG_KERNEL_MFMA) {
    if (e->sincos_hw == 2) launch_mfma<A_PAD, MRW, 2>(e, s);
    else if (e->sincos_hw == 1) launch_mfma<A_PAD, MRW, 1>(e, s);
    else launch_mfma<A_PAD, MRW, 0>(e, s);
  } else {
    if (e->sincos_hw) launch_valu<A_PAD, MRW, 1>(e, s); else launch_valu<A_PAD, MRW, 0>(e, s);
  }
}


template <int A_PAD>
void launch_fast1(aog_env* e, hipStream_t s) {
#ifdef AOG_FAST_BUILD  // developer builds: only the 8-table kernels
  launch_fast2<A_PAD, 7>(e, s);
#else
  switch (e->MRW) {
    case 7: launch_fast2<A_PAD, 7>(e, s); break;
    case 12: launch_fast2<A_PAD, 12>(e, s); break;
    case 20: launch_fast2<A_PAD, 20>(e, s); break;
    default: launch_fast2<A_PAD, 28>(e, s); break;
  }
#endif
}
}  // namespace

namespace aog_host {
#define AOG_CAT2(a, b) a##b
#define AOG_CAT(a, b) AOG_CAT2(a, b)
void AOG_CAT(launch_fused_apad, AOG_INST_APAD)(aog_env* e, hipStream_t s) { launch_fast1<AOG_INST_APAD>(e, s); }
}  // namespace aog_host
