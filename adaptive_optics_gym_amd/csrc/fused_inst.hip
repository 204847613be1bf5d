// Instantiations of the fused kernels for ONE padded mode count (AOG_INST_APAD), all table counts.
#include "host_common.h"
#include "k_fused.h"

#ifndef AOG_INST_APAD
#error "compile with -DAOG_INST_APAD=16|32|64|128"
#endif

namespace {
using aog_host::round_up;
// ---- fused kernel dispatch -------------------------------------------------------------------------
template <int A_PAD, int MRW, int SC>
int launch_valu(aog_env* e, hipStream_t s) {
  const int n_groups = e->Bp / 64;
  dim3 grid(e->valu_chunks, (n_groups + 3) / 4);
  const float ratio = (float)(e->cfg.wavelength_wfs / e->cfg.wavelength_sci);
  hipLaunchKernelGGL((aog::k_fused_valu<A_PAD, MRW, 1, SC>), grid, dim3(256), 0, s, e->modes_f32, e->tabs_f32,
                     reinterpret_cast<const float4*>(e->psi_rev), e->act_rev, e->partials, e->n_quads, e->Bp, n_groups,
                     e->valu_qpc, ratio);
  return 0;
}

template <int A_PAD, int MRW>
int launch_tab(aog_env* e, hipStream_t s) {
  aog::MfmaGeom g;
  g.n_ptiles = e->n_ptiles;
  g.n_etiles = e->n_etiles;
  g.Bp = e->Bp;
  g.P = e->mfma_chunks_x;
  g.we = e->mfma_we;
  g.wg_y = (e->n_etiles + e->mfma_we - 1) / e->mfma_we;
  g.max_tiles = e->mfma_tpc;
  g.skew = aog::kSkewNops;
  g.heavy = e->mfma_heavy;
  const int threads = 64 * e->mfma_waves;
  g.pair = (e->mfma_waves == 4 && g.wg_y % 2 == 0 && 64 % g.wg_y == 0 && g.wg_y >= 2) ? 1 : 0;
  g.dev = 0;
  g.timeline = nullptr;
#ifdef AOG_DEV
  if (getenv("AOG_DEV_TIMELINE")) {
    static long long* buf = nullptr;
    if (!buf) (void)hipMalloc(&buf, sizeof(long long) * 8 * 4 * 8192);
    g.timeline = buf;
    aog_host::dev_timeline = buf;
  }
  if (const char* v = getenv("AOG_DEV_FLAGS")) g.dev = atoi(v);
  if (const char* v = getenv("AOG_DEV_PAIR")) g.pair = g.pair && atoi(v);
  if (const char* v = getenv("AOG_DEV_SKEW")) g.skew = atoi(v);
  if (const char* v = getenv("AOG_DEV_HEAVY")) g.heavy = g.heavy ? atoi(v) : 0;
#endif
  const int chunks_per_xcd = round_up(g.P, 8) / 8;
  const int wgs_per_xcd = g.pair ? round_up(chunks_per_xcd, 64 / g.wg_y) * g.wg_y : chunks_per_xcd * g.wg_y;
  dim3 grid(8 * wgs_per_xcd);
  const float ratio = (float)(e->cfg.wavelength_wfs / e->cfg.wavelength_sci);
  const size_t lds_0 = (size_t)e->mfma_tpc * 8 * 16 + ((A_PAD > 64 || e->ring_direct) ? (size_t)e->mfma_waves * (A_PAD / 16) * 2 * 64 * 16 : 0) +   // science rows (+ actuator operands)
                       (e->ring_direct ? (size_t)e->mfma_waves * 32 * 36 * 4 : 0);                                             // (+ ring-direct transpose tiles)
  g.acc_off = (int)((lds_0 + 15) / 16 * 16);
  // many-table variants: float64 table sums per wave, [2 LIVE][64 lanes]
  const size_t lds_t = g.acc_off + (aog::TabGeom<MRW>::kF64 ? 0 : (size_t)e->mfma_waves * 2 * aog::TabGeom<MRW>::kLiveRegs * 64 * sizeof(double));
  aog::DynPsi dyn{};
  if (e->ring_direct) {   // dynamic atmosphere: the screens come straight from the fp32 ring copy of the master screens
    dyn.ring = e->psi_ring;
    dyn.origin = e->origin;
    dyn.desc = reinterpret_cast<const uint4*>(e->quad_desc);
    dyn.cont = reinterpret_cast<const uint4*>(e->quad_cont);
    dyn.N = e->cfg.n_pupil;
    dyn.RS = e->cfg.n_pupil + 4;
    dyn.B = e->B;
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_fused_tab<A_PAD, MRW, true>), lds_t, e->device)) return rc;
    hipLaunchKernelGGL((aog::k_fused_tab<A_PAD, MRW, true>), grid, dim3(threads), lds_t, s, reinterpret_cast<const aog::f16x8*>(e->modes16),
                       reinterpret_cast<const aog::f16x8*>(e->tab16), reinterpret_cast<const aog::f32x4*>(e->sci_tile),
                       reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio, dyn);
    return 0;
  }
  if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_fused_tab<A_PAD, MRW, false>), lds_t, e->device)) return rc;
  hipLaunchKernelGGL((aog::k_fused_tab<A_PAD, MRW, false>), grid, dim3(threads), lds_t, s, reinterpret_cast<const aog::f16x8*>(e->modes16),
                     reinterpret_cast<const aog::f16x8*>(e->tab16), reinterpret_cast<const aog::f32x4*>(e->sci_tile),
                     reinterpret_cast<const aog::f32x4*>(e->psi_tile), reinterpret_cast<const aog::f16x8*>(e->act16), e->partials, g, ratio, dyn);
  return 0;
}

template <int A_PAD, int MRW>
int launch_fast2(aog_env* e, hipStream_t s) {
  if (e->kernel == AOG_KERNEL_MFMA) return launch_tab<A_PAD, MRW>(e, s);
  return e->sincos_hw ? launch_valu<A_PAD, MRW, 1>(e, s) : launch_valu<A_PAD, MRW, 0>(e, s);
}

template <int A_PAD>
int launch_fast1(aog_env* e, hipStream_t s) {
#ifdef AOG_FAST_BUILD  // developer builds: only the 8-table kernels
  return launch_fast2<A_PAD, 7>(e, s);
#else
  switch (e->MRW) {
    case 7: return launch_fast2<A_PAD, 7>(e, s);
    case 12: return launch_fast2<A_PAD, 12>(e, s);
    case 20: return launch_fast2<A_PAD, 20>(e, s);
    default: return launch_fast2<A_PAD, 28>(e, s);
  }
#endif
}
}  // namespace

namespace aog_host {
#define AOG_CAT2(a, b) a##b
#define AOG_CAT(a, b) AOG_CAT2(a, b)
int AOG_CAT(launch_fused_apad, AOG_INST_APAD)(aog_env* e, hipStream_t s) { return launch_fast1<AOG_INST_APAD>(e, s); }

// ---- phase-only contraction u = psi + Mt a (k_phase_mfma) for this padded mode count ----------------------------------------------
int AOG_CAT(launch_phase_apad, AOG_INST_APAD)(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile) {
  constexpr int A_PAD = AOG_INST_APAD;
  hipLaunchKernelGGL((aog::k_phase_mfma<A_PAD>), dim3((e->n_ptiles + 3) / 4, e->n_etiles), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile),
                     reinterpret_cast<const aog::f16x8*>(act16), reinterpret_cast<aog::f32x4*>(out_tile), e->n_ptiles, e->n_etiles);
  return 0;
}
// field (or, grid = true, one float of reduced phase per pixel) of every env on its own pupil grid: see k_phase_mfma<.., FIELD, GRID>
int AOG_CAT(launch_phase_field_apad, AOG_INST_APAD)(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, bool grid) {
  constexpr int A_PAD = AOG_INST_APAD;
  auto kern = grid ? aog::k_phase_mfma<A_PAD, true, true> : aog::k_phase_mfma<A_PAD, true, false>;
  const int epw = e->n_etiles >= 8 ? 2 : 1;   // env tiles per wave (k_phase_mfma; measured at 64 env tiles, N = 512: 975 / 875 / 889 / 947 us for 1 / 2 / 4 / 8)
  hipLaunchKernelGGL(kern, dim3((e->n_ptiles + 3) / 4, (e->n_etiles + epw - 1) / epw), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile),
                     reinterpret_cast<const aog::f16x8*>(act16), static_cast<aog::f32x4*>(nullptr), e->n_ptiles, e->n_etiles, fa, epw);
  return 0;
}
// K4: reduced phases of env tiles [etile0, etile0 + n_et) as one float per pixel on a dense [env][rows][row_stride] grid (no micro-lens term)
int AOG_CAT(launch_phase_grid_apad, AOG_INST_APAD)(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, int etile0, int n_et) {
  constexpr int A_PAD = AOG_INST_APAD;
  const int epw = n_et >= 8 ? 2 : 1;   // (as launch_phase_field_apad)
  hipLaunchKernelGGL((aog::k_phase_mfma<A_PAD, true, true>), dim3((e->n_ptiles + 3) / 4, (n_et + epw - 1) / epw), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile) + (size_t)etile0 * e->n_ptiles * 4 * 64,
                     reinterpret_cast<const aog::f16x8*>(act16) + (size_t)etile0 * (A_PAD / 16) * 2 * 64, static_cast<aog::f32x4*>(nullptr), e->n_ptiles,
                     n_et, fa, epw);
  return 0;
}
}  // namespace aog_host
