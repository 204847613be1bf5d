// The fused pupil pass (VALU form, f16 matrix-core form) and the phase-only contraction.
#pragma once
#include "k_common.h"

namespace aog {

// ------------------------------------------------------------------------------------------------
// K3a  fused pupil pass, VALU form.  lane = env (64 envs per wave), every per-pixel operand (mode row,
// table row) is wave-uniform and comes through the scalar cache (s_load), the screen is one float4 per lane
// per pixel quad.  Per (pixel, env): A_PAD fma (surface), 2 sincos, 2*(MRW+MRS) fma.
//   grid = (pixel chunks, ceil(env groups / 4)), block = 4 waves = 4 env groups sharing the pixel range.
//   partials[chunk][s][env] float64, s < NS.
// ------------------------------------------------------------------------------------------------
template <int A_PAD, int MRW, int MRS, int SINCOS>
__global__ __launch_bounds__(256) void k_fused_valu(const float* __restrict__ modes, const float* __restrict__ tabs,
                                                    const float4* __restrict__ psi4, const float* __restrict__ act_rev,
                                                    double* __restrict__ partials, int n_quads, int Bp, int n_groups,
                                                    int quads_per_chunk, float ratio) {
  constexpr int NS = 2 * (MRW + MRS);
  constexpr int TROW = (MRW + MRS + 3) & ~3;
  constexpr int TQ = 8;  // quads per fp32 tile-sum before the float64 flush
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (group >= n_groups) return;
  const int env = group * 64 + lane;

  float a[A_PAD];
#pragma unroll
  for (int k = 0; k < A_PAD; ++k) a[k] = act_rev[(size_t)k * Bp + env];

  double acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = 0.0;

  const int q0 = blockIdx.x * quads_per_chunk;
  const int q1 = min(n_quads, q0 + quads_per_chunk);
  for (int qb = q0; qb < q1; qb += TQ) {
    float t[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) t[i] = 0.f;
    const int qe = min(q1, qb + TQ);
    for (int q = qb; q < qe; ++q) {
      const float4 u4 = psi4[(size_t)q * Bp + env];
      float u[4] = {u4.x, u4.y, u4.z, u4.w};
      const float* __restrict__ mrow = modes + (size_t)q * 4 * A_PAD;
#pragma unroll
      for (int k = 0; k < A_PAD; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = fmaf(mrow[j * A_PAD + k], a[k], u[j]);
      }
      const float* __restrict__ trow = tabs + (size_t)q * 4 * TROW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s, c;
        sincos_rev<(SINCOS == 2 ? 1 : SINCOS)>(u[j], s, c);
#pragma unroll
        for (int m = 0; m < MRW; ++m) {
          const float g = trow[j * TROW + m];
          t[2 * m] = fmaf(c, g, t[2 * m]);
          t[2 * m + 1] = fmaf(s, g, t[2 * m + 1]);
        }
        sincos_rev<(SINCOS == 2 ? 1 : SINCOS)>(u[j] * ratio, s, c);
#pragma unroll
        for (int m = 0; m < MRS; ++m) {
          const float g = trow[j * TROW + MRW + m];
          t[2 * (MRW + m)] = fmaf(c, g, t[2 * (MRW + m)]);
          t[2 * (MRW + m) + 1] = fmaf(s, g, t[2 * (MRW + m) + 1]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] += (double)t[i];
  }
  double* out = partials + (size_t)blockIdx.x * NS * Bp + env;
#pragma unroll
  for (int i = 0; i < NS; ++i) out[(size_t)i * Bp] = acc[i];
}

// ------------------------------------------------------------------------------------------------
// K3b  fused pupil pass on the f16 matrix cores.  One wave owns a 32-env tile and walks 32-pixel tiles:
//     u[pixel i][env j] = psi[i][j] + sum_k Mt[i][k] * a[k][j]                         (K = A_PAD modes)
// The contraction runs on the f16 matrix cores with BOTH operands split in two halves that together carry
// fp32 precision:   x = x_hi + x_lo,  x_hi = half(x),  x_lo = half(x - x_hi)   (the matrix pipe keeps f16 subnormals, measured:
// tools/microbench/mfma_f16_denorm.hip — so the low halves need no scale of their own and ALL products share one accumulator)
//     D = Mh.ah + Mh.al + Ml.ah           u = psi + 2^-22 D                      (the 2^-22-relative Ml.al term is dropped)
// = 3 x v_mfma_f32_32x32x16_f16 per 16 modes, products exact in fp32, fp32 accumulation.  Unlike v_mfma_f32_32x32x2_f32
// (which was measured NOT to overlap with vector instructions: fused = vector-only + matrix-only time, profiles/r01), the
// f16 matrix pipe co-executes with the VALU.
// C/D register map: lane l holds env j = l&31 and pixels i = (r&3) + 8*(r>>2) + 4*(l>>5), r < 16.
//   psi_tile   [env tile][pixel tile][g=r>>2][lane][r&3]        one float4 per lane per g, 1 KiB per instruction
//   modes16    [pixel tile][s][hi|lo][lane][8 halfs]: lane (pixel i = l&31, h = l>>5), element e <-> mode 16 s + 8 h + e
//   act16      [env tile][s][hi|lo][lane][8 halfs]:   lane (env j = l&31, h),          element e <-> mode 16 s + 8 h + e
// ------------------------------------------------------------------------------------------------
// Launch geometry of k_fused_tab (host side fills it; see aog_create):
//   1-D grid of 8 * ceil(P/8) * wg_y workgroups.  Workgroup L runs on XCD L % 8 (round-robin dispatch, speed only):
//   xcd = L & 7, j = L >> 3, env group = j % wg_y, pixel chunk c = (j / wg_y) * 8 + xcd, so the wg_y workgroups that
//   share a pixel chunk (= the same mode-matrix and table tiles) sit on ONE XCD back to back and each XCD's L2 only ever
//   sees 1/8 of the mode matrix.  Chunk c owns pixel tiles [c*n_ptiles/P, (c+1)*n_ptiles/P).
struct MfmaGeom {
  int n_ptiles, n_etiles, Bp, P, wg_y, we, max_tiles;
  int skew;   // start-up skew of every second workgroup, x 16 cycles
  int pair;   // 1: the two workgroups that share a CU walk the SAME pixel chunk (different env groups), see fused_wg_map
  int heavy;  // > 0: asymmetric wave pairs, sub-chunk 0 takes heavy / 1024 of a chunk's tiles (see k_fused_tab); 0: interleaved
  int acc_off;   // many-table variants: byte offset of the float64 table-sum accumulators in the workgroup's dynamic LDS (behind everything else)
  int dev;    // developer experiments (AOG_DEV builds only; 0 in the product)
  long long* timeline;   // AOG_DEV builds: per-wave time stamps (wall_clock64, 10 ns ticks) [wave][8], or null
};

// workgroup L -> (pixel chunk c, env group eg).  Workgroup L runs on XCD L % 8; inside an XCD workgroups j = L >> 3 fill the 32 CUs
// round-robin, two per CU (j and j + 32 share a CU).  pair = 0: eg = j % wg_y, c = (j / wg_y) * 8 + xcd (the wg_y workgroups of a
// chunk sit on wg_y different CUs).  pair = 1 (wg_y even and a divisor of 64): the two workgroups of a CU take the same chunk, so the
// eight waves of a CU pull one copy of the chunk's mode / table operands through the CU's L1 instead of two.  Placement is a speed
// matter only: every (chunk, env group) pair is covered exactly once either way.
__device__ __forceinline__ void fused_wg_map(const MfmaGeom& geo, int L, int& c, int& eg) {
  const int j = L >> 3, xcd = L & 7;
  if (geo.pair) {
    const int cpr = 64 / geo.wg_y;            // chunks per round of 64 workgroups (32 CUs x 2)
    const int r = j >> 6, k = j & 31, half = (j >> 5) & 1;
    c = (r * cpr + k % cpr) * 8 + xcd;
    eg = k / cpr + (geo.wg_y >> 1) * half;
  } else {
    c = (j / geo.wg_y) * 8 + xcd;
    eg = j % geo.wg_y;
  }
}

// ---- the fused kernel: BOTH contractions on the f16 matrix cores --------------------------------------------------------------
// Phase stage as above.  The table reduction  Z_m(env) = sum_p G_m(p) (cos, sin)(u_p,env)  is a second MFMA:
//   A = table rows (m < 32) x 16 pixels, f16 hi + lo (unscaled), pre-arranged on the host in the pixel order in which the phase
//       accumulator hands its 16 values per lane to the B operand (tab16);
//   B = cos / sin of this lane's 8 pixels of the step, f16 hi + lo;  Gh Eh + Gh El + Gl Eh accumulate into ONE fp32 accumulator.
// Vector work per (pixel, env): u = fma(D, 2^-22, psi), u_sci = u * ratio, 4 hardware sin/cos (they take revolutions), the two
// science-table FMAs, and the hi/lo split of cos and sin as  hi = x & 0xffffe000 (an fp32 with 11 significant bits: exact in f16),
// lo = x - hi, two values packed per v_cvt_pkrtz_f16_f32 — 3 ops per component instead of the 5 of convert / convert back /
// subtract / convert / pack.
// Sums: tables m < MRW in the 32x32 accumulators (lane (env, h) holds rows (a & 3) + 8 (a >> 2) + 4 h); the 8-table variant
// folds its (few) live rows into float64 every kFlushTiles tiles, the others run fp32 over a chunk of bounded length.
// Dynamic atmosphere: the fused kernel reads the screens STRAIGHT from the fp32 ring-buffer copy of the float64 master screens instead
// of a per-step repack into psi_tile (which re-read 537 MB and re-wrote 211 MB per step to move ~29 KB of new samples per env):
//   ring   [B][N][RS] fp32, RS = N + 4: revolutions at lambda_wfs minus the env's reference piston, stored at the master's physical
//          (toroidal) position; columns 0..3 are duplicated at N..N+3 so that 4 consecutive x never wrap
//   origin [B][2] (ox, oy): logical (iy, ix) lives at physical ((iy + oy) mod N, (ix + ox) mod N)
//   desc   [n_ptiles * 2][4]: for pixel tile t, half-wave h, register group g: the packed 4-pixel group starts at logical
//          (iy, ix) = (d >> 18, (d >> 4) & 0x3fff); d & 7 = k = how many of its pixels lie in that row (4 = all); bit 3 = some group
//          of this (t, g) (either half-wave) continues in another row (wave-uniform)
//   cont   [n_ptiles * 2][4]: for a group with k < 4, where its pixel k sits MINUS k columns, (iy2 << 18 | ((ix2 - k) mod N) << 4):
//          a second 16-byte load from there has the right values in elements k..3
// Each lane (env, h) makes its own 16-byte load per group (4-byte aligned); the 32 envs of a tile hit 32 different lines, each of
// which holds this tile's 32 pixels of that env, so no byte is fetched twice.
struct DynPsi {
  const float* ring;
  const int32_t* origin;
  const uint4* desc;
  const uint4* cont;
  int N, RS, B;
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MRW>
struct TabGeom {
  static constexpr int kLiveRegs = MRW <= 8 ? 4 : (MRW <= 16 ? 8 : (MRW <= 24 ? 12 : 16));   // accumulator registers a < kLiveRegs hold real tables
  static constexpr bool kF64 = MRW <= 8;
};
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {   // (half(a), half(b)) in one register; callers pass values exact in f16
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
template <int A_PAD, int MRW, bool DYN>
__global__ __launch_bounds__(512, 2) void k_fused_tab(const f16x8* __restrict__ modes16, const f16x8* __restrict__ tab16,
                                                      const f32x4* __restrict__ sci_tile, const f32x4* __restrict__ psi_tile,
                                                      const f16x8* __restrict__ act16, double* __restrict__ partials, MfmaGeom geo, float ratio,
                                                      DynPsi dyn) {
  constexpr int NSTEP = A_PAD / 16, NM = 3 * NSTEP, NS = 2 * (MRW + 1);
  constexpr int LIVE = TabGeom<MRW>::kLiveRegs;
  constexpr bool F64 = TabGeom<MRW>::kF64;
  extern __shared__ f32x4 lds_sci[];   // [tile in chunk][h][4] float4 = the science table in accumulator order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = blockIdx.x, j = L >> 3;
#ifdef AOG_DEV
  long long tl[5] = {geo.timeline ? (long long)wall_clock64() : 0, 0, 0, 0, 0};
#endif
  int c, eg;
  fused_wg_map(geo, L, c, eg);
  if (c >= geo.P) return;
  const int we = geo.we, wp = (int)(blockDim.x >> 6) / we;   // env tiles x pixel sub-chunks = the waves of the workgroup
  const int w_e = wave % we, w_p = wave / we;
  const int etile = eg * we + w_e;
  const int t0 = (int)(((long long)c * geo.n_ptiles) / geo.P);
  const int t1 = (int)(((long long)(c + 1) * geo.n_ptiles) / geo.P);
  const int h = lane >> 5;
  const int etile_c = min(etile, geo.n_etiles - 1);
  // This wave's tiles: first, first + stride, ... (n of them).  Interleaved sub-chunks by default.  Asymmetric pairs (geo.heavy > 0,
  // 8 waves, wp = 2): the workgroup's waves sit two per SIMD, and of two co-resident waves the one with priority runs at ~1.3x the
  // rate of the other (vector issue is arbitrated by priority, then age: MI355X_MICROARCH.md, "Two waves per SIMD") — so sub-chunk 0
  // takes geo.heavy / 1024 of the chunk's tiles AND the priority, sub-chunk 1 the rest, and the two finish together instead of
  // leaving every SIMD to a single wave (which fills ~40 % of its issue slots) for the last third of the launch.
  const int nt_c = t1 - t0;
  const int n_heavy = geo.heavy > 0 ? min(nt_c, (nt_c * geo.heavy + 512) >> 10) : 0;
  const int stride = geo.heavy > 0 ? 1 : wp;
  const int first = geo.heavy > 0 ? (w_p == 0 ? t0 : t0 + n_heavy) : t0 + w_p;
  const int n = geo.heavy > 0 ? (w_p == 0 ? n_heavy : nt_c - n_heavy) : (first < t1 ? (t1 - first + wp - 1) / wp : 0);
  const int last = n > 0 ? first + (n - 1) * stride : min(t0, geo.n_ptiles - 1);
  f16x8 bh[NSTEP], bl[NSTEP];
  {
    const f16x8* asrc = act16 + ((size_t)etile_c * NSTEP * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bh[s] = asrc[(2 * s) * 64];
      bl[s] = asrc[(2 * s + 1) * 64];
    }
  }
  // The float64-flush variant (few tables) and the 128-mode variants are short of registers: their actuator operands live in LDS (this
  // wave's own 2 NSTEP KB, behind the science rows) and are read back right before each phase MFMA.
  constexpr bool BLDS = A_PAD > 64 || DYN;   // (the ring-direct variant needs its registers for addresses)
  f16x8* lds_b = reinterpret_cast<f16x8*>(lds_sci + (size_t)geo.max_tiles * 8) + (size_t)wave * NSTEP * 2 * 64 + lane;
  if constexpr (BLDS) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      lds_b[(2 * s) * 64] = bh[s];
      lds_b[(2 * s + 1) * 64] = bl[s];
    }
  }
  const size_t psi_base = (size_t)etile_c * geo.n_ptiles;
  // Ring-direct loads (DYN only).  A tile is 32 envs x 32 pixels = one 128-byte line per env; load instruction i (of four) covers
  // envs 8 i .. 8 i + 7 with EIGHT LANES PER LINE: lane l fetches the 16-byte piece l & 7 (register group g = piece >> 1 of half-wave
  // piece & 1) of env 8 i + (l >> 3), so an instruction touches 8 lines (per-lane loads in the accumulator layout touched 32 and ran
  // the launch at 100 us against 51).  The pieces reach the accumulator layout (lane = env, 16 pixels) through this wave's private
  // [32][36] float tile in LDS right before the tile is reduced.
  const float* ring_env[4] = {nullptr, nullptr, nullptr, nullptr};
  int dyn_ox[4] = {0, 0, 0, 0}, dyn_oy[4] = {0, 0, 0, 0};
  float* dyn_x = nullptr;
  const int dyn_piece = lane & 7;
  if constexpr (DYN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int env = min(etile_c * 32 + 8 * i + (lane >> 3), dyn.B - 1);
      ring_env[i] = dyn.ring + (size_t)env * dyn.N * dyn.RS;
      dyn_ox[i] = dyn.origin[2 * env];
      dyn_oy[i] = dyn.origin[2 * env + 1];
    }
    dyn_x = reinterpret_cast<float*>(lds_b - lane + (size_t)((int)(blockDim.x >> 6) - wave) * NSTEP * 2 * 64) + (size_t)wave * 32 * 36;
  }
  auto load_modes = [&](f16x8 (&mh)[NSTEP], f16x8 (&ml)[NSTEP], int t) {
#ifdef AOG_DEV
    if (geo.dev & 16) t = first;   // timing experiment: every tile's operands come from the same (cache-hot) addresses
#endif
    const f16x8* ms = modes16 + ((size_t)min(t, last) * NSTEP * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      mh[s] = ms[(2 * s) * 64];
      ml[s] = ms[(2 * s + 1) * 64];
    }
  };
  // screen values of tile t, register groups [G0, G0 + NG) (4 registers = one 16-byte load each)
  auto load_psi = [&](auto g0c, auto ngc, f32x16& d, int t) {
    constexpr int G0 = decltype(g0c)::v, NG = decltype(ngc)::v;
    if constexpr (DYN) {
      static_assert(G0 == 0 && NG == 4, "ring-direct tiles are requested whole");
      const int tt = min(t, last);
      // this lane's piece of the tile: register group g = piece >> 1 of half-wave piece & 1
      const uint32_t* dsc = reinterpret_cast<const uint32_t*>(dyn.desc) + ((size_t)tt * 2 + (dyn_piece & 1)) * 4 + (dyn_piece >> 1);
      const uint32_t code = *dsc;
      const bool straddle = __any((code & 7u) < 4u ? 1 : 0);   // some piece of this tile continues in another aperture row
      uint32_t ccode = 0;
      if (straddle) ccode = *(reinterpret_cast<const uint32_t*>(dyn.cont) + (dsc - reinterpret_cast<const uint32_t*>(dyn.desc)));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        auto fetch = [&](uint32_t cd) {   // 4 consecutive x from logical (cd >> 18, (cd >> 4) & 0x3fff) of env 8 i + (lane >> 3)
          uint32_t py = (cd >> 18) + (uint32_t)dyn_oy[i], px = ((cd >> 4) & 0x3fffu) + (uint32_t)dyn_ox[i];
          py = min(py, py - (uint32_t)dyn.N);   // (unsigned: the wrapped candidate is huge unless py >= N)
          px = min(px, px - (uint32_t)dyn.N);
          return *reinterpret_cast<const f32x4u*>(ring_env[i] + (size_t)py * dyn.RS + px);
        };
        f32x4 v = fetch(code);
        if (straddle) {
          const int k = (int)(code & 7u);
          const f32x4 w = fetch(k < 4 ? ccode : code);
          v[1] = k <= 1 ? w[1] : v[1];
          v[2] = k <= 2 ? w[2] : v[2];
          v[3] = k <= 3 ? w[3] : v[3];
        }
        d[4 * i + 0] = v[0]; d[4 * i + 1] = v[1]; d[4 * i + 2] = v[2]; d[4 * i + 3] = v[3];
      }
    } else {
      const f32x4* ps = psi_tile + ((psi_base + min(t, last)) * 4) * 64 + lane;
#pragma unroll
      for (int g = G0; g < G0 + NG; ++g) {
        const f32x4 v = ps[g * 64];
        d[4 * g + 0] = v[0]; d[4 * g + 1] = v[1]; d[4 * g + 2] = v[2]; d[4 * g + 3] = v[3];
      }
    }
  };
  // ring-direct: pieces (lane = env-of-eight x piece, register group = load instruction) -> accumulator layout (lane = env, 16 pixels)
  auto dyn_transpose = [&](f32x16& d) {
    if constexpr (DYN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = {d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]};
        *reinterpret_cast<f32x4*>(dyn_x + (size_t)(8 * i + (lane >> 3)) * 36 + 4 * dyn_piece) = v;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tile is private to this wave
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dyn_x + (size_t)(lane & 31) * 36 + 8 * g + 4 * h);
        d[4 * g] = v[0]; d[4 * g + 1] = v[1]; d[4 * g + 2] = v[2]; d[4 * g + 3] = v[3];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  };
  // Table rows m >= MRW are zero: the lanes that would fetch them all read ONE zero entry (row 31 of the first half) instead, so a
  // variant with few tables pulls 2-3 cache lines per operand through L1 instead of 8.
  const int tlane = (lane & 31) <= MRW ? lane : 31;
  auto load_tab = [&](f16x8 (&ta)[4], int t) {   // [step][hi|lo]
    const f16x8* ts = tab16 + ((size_t)min(t, last) * 4) * 64 + tlane;
#pragma unroll
    for (int q = 0; q < 4; ++q) ta[q] = ts[q * 64];
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f16x8 mh[NSTEP], ml[NSTEP], ta[4];
  f32x16 Pa = zero16, Pb = zero16;   // screens: Pa = tile being reduced, Pb = the tile after it (two tiles of the HBM stream in flight)
  load_modes(mh, ml, first);
  load_psi(IC<0>{}, IC<4>{}, Pa, first);
  load_tab(ta, first);
  __builtin_amdgcn_sched_barrier(0);
  {
    const int n4 = (t1 - t0) * 8;
    const f32x4* src = sci_tile + (size_t)t0 * 8;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) lds_sci[i] = src[i];
  }
  __syncthreads();
  if (etile >= geo.n_etiles) return;
  if (geo.heavy > 0) {
    if (w_p == 0) __builtin_amdgcn_s_setprio(1);
  } else if ((j & 32) != 0) {
    // 4-wave workgroups, two per CU: the second one dispatched to a CU (j and j + 32 share it) is the younger and would lose every
    // arbitration, starting its loop ~8 us late; with the priority it starts on time and the older one fills the gaps (measured at
    // B = 4096, o = 5: 266 against 287 us per launch)
    __builtin_amdgcn_s_setprio(1);
  }
#ifdef AOG_DEV
  if (geo.timeline) tl[1] = wall_clock64();
#endif
  f32x16 Dc = zero16, Ds = zero16;          // table sums (cos, sin), rows by register
  float sc_c = 0.f, sc_s = 0.f;             // science-table sums of this lane's pixels
  // Many-table variants (o >= 3): 2 x LIVE live accumulator registers per lane are too many to mirror in float64 registers, so the float64
  // sums live in this wave's own plane of LDS, [2 LIVE][64 lanes], and the fp32 registers are folded into it every kTabF32Tiles tiles (the fp32
  // run length the tolerances were set for).  Round 3 instead cut the pixel range into chunks of <= 13 tiles per wave and wrote a float slab
  // per chunk (124+ slabs, a fold kernel, and every wave's 5 us of set-up amortised over 13 tiles: 267 us per 4096 envs at o = 5 against 4 x 52).
  double* lds_acc = reinterpret_cast<double*>(reinterpret_cast<char*>(lds_sci) + geo.acc_off) + (size_t)wave * 2 * LIVE * 64 + lane;
  if constexpr (!F64) {
#pragma unroll
    for (int a = 0; a < 2 * LIVE; ++a) lds_acc[a * 64] = 0.0;
  }
  double acc_t[F64 ? 2 * LIVE : 1];
  double acc_sc = 0.0, acc_ss = 0.0;
#pragma unroll
  for (int i = 0; i < (F64 ? 2 * LIVE : 1); ++i) acc_t[i] = 0.0;
  if (n > 0) {
    // phase MFMA q of a tile into `acc`: s = q / 3; Mh.ah, Mh.al, Ml.ah.  The first one of a tile starts the sum (C = 0).
    auto mfma_q = [&](auto qc, f32x16& acc) {
      constexpr int q = decltype(qc)::v, s = q / 3, w = q % 3;
      f16x8 xh, xl;
      if constexpr (BLDS) {
        if constexpr (w != 1) xh = lds_b[(2 * s) * 64];
        if constexpr (w == 1) xl = lds_b[(2 * s + 1) * 64];
      } else {
        xh = bh[s];
        xl = bl[s];
      }
      if constexpr (q == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], xh, zero16, 0, 0, 0);
      else if constexpr (w == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], xh, acc, 0, 0, 0);
      else if constexpr (w == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], xl, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[s], xh, acc, 0, 0, 0);
    };
    f32x16 X, Y = zero16;   // phase accumulators (scaled by 2^22): X = tile being reduced, Y = the next tile's contraction in flight
    static_for<NM>([&](auto qc) { mfma_q(qc, X); });
    // Registers are refilled just in time: the mode halves and the step-0 table operands of the NEXT stage are requested right
    // after the matrix ops that read the current ones have been issued (middle of the stage), the step-1 table operands at the end;
    // a screen register group is re-requested (for the tile after next) as soon as its four pixels have been reduced.
    load_psi(IC<0>{}, IC<4>{}, Pb, first + stride);
    __builtin_amdgcn_sched_barrier(0);
    load_modes(mh, ml, first + stride);
    // Matrix instructions are dealt BETWEEN the pixels of a vector step (one wave issues in order: a block of 18 MFMAs would keep
    // it from issuing vector work for ~600 cycles, and both waves of a SIMD tend to be in the same phase).  The queue of a stage:
    //   during step 0 of tile t : the 12 phase MFMAs of the next tile  and  the 6 step-1 table MFMAs of the previous tile (operands kept)
    //   during step 1 of tile t : the 6 step-0 table MFMAs of tile t
    u32x4 c0, l0, s0, m0;   // step-0 B operands of the current tile (cos hi, cos lo, sin hi, sin lo), two f16 per register
    u32x4 c1, l1, s1, m1;   // step-1 B operands, consumed during the NEXT stage's step 0
    auto tab_one = [&](auto kc, const f16x8& tah, const f16x8& tal, const u32x4& ch, const u32x4& cl, const u32x4& sh, const u32x4& sl) {
      constexpr int k = decltype(kc)::v;   // 0..5
      if constexpr (k == 0) Dc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(f16x8, ch), Dc, 0, 0, 0);
      else if constexpr (k == 1) Ds = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(f16x8, sh), Ds, 0, 0, 0);
      else if constexpr (k == 2) Dc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(f16x8, cl), Dc, 0, 0, 0);
      else if constexpr (k == 3) Ds = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(f16x8, sl), Ds, 0, 0, 0);
      else if constexpr (k == 4) Dc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tal, __builtin_bit_cast(f16x8, ch), Dc, 0, 0, 0);
      else Ds = __builtin_amdgcn_mfma_f32_32x32x16_f16(tal, __builtin_bit_cast(f16x8, sh), Ds, 0, 0, 0);
    };
    // one pixel e of step s: phase, sin/cos, science sums, hi/lo split; the halves of an even pixel wait in `st` for their odd
    // neighbour and the pair goes into element pair e >> 1 of the four B operands
    float st[4];
    f32x2 uw2 = {0.f, 0.f}, us2 = {0.f, 0.f};
    auto vec_pixel = [&](auto sc, auto ec, const f32x16& D, const f32x16& P, const f32x4& g0, const f32x4& g1, u32x4& ch, u32x4& cl, u32x4& sh,
                         u32x4& sl) {
      constexpr int s = decltype(sc)::v, e = decltype(ec)::v;
      // phases of a pixel pair with one packed FMA and one packed multiply (their inputs are matrix-pipe results and loaded screen
      // values, never fresh transcendental results: the packed-read hazard of DESIGN.md section 5 does not apply)
      if constexpr ((e & 1) == 0) {
        const f32x2 d2 = {D[8 * s + e], D[8 * s + e + 1]}, p2 = {P[8 * s + e], P[8 * s + e + 1]};
        const f32x2 k2 = {kPhaseUnscale, kPhaseUnscale}, r2 = {ratio, ratio};
        uw2 = __builtin_elementwise_fma(d2, k2, p2);
        us2 = uw2 * r2;
      }
      const float u = uw2[e & 1];
      const float cw = __builtin_amdgcn_cosf(u), sw = __builtin_amdgcn_sinf(u);
      const float us = us2[e & 1];
      const float cs = __builtin_amdgcn_cosf(us), ss = __builtin_amdgcn_sinf(us);
      const float g = e < 4 ? g0[e & 3] : g1[e & 3];
      sc_c = fmaf(cs, g, sc_c);
      sc_s = fmaf(ss, g, sc_s);
      // hi/lo split by mask: hi = x & 0xffffe000 (an fp32 with 11 significant bits: exact in f16), lo = x - hi, two values per
      // v_cvt_pkrtz.  (Tried: hi of a pair in ONE v_cvt_pkrtz and lo = x - hi as v_fma_mix_f32 reading the f16 half in place — 2
      // instructions per value instead of 3, 12.7 M instead of 13.5 M vector instructions per launch, and 4 us SLOWER (54.7 vs 50.5):
      // convert -> mixed FMA -> convert is a dependent chain per pair, the mask form's and / subtract pairs are independent.)
      const float chf = __uint_as_float(__float_as_uint(cw) & 0xffffe000u), shf = __uint_as_float(__float_as_uint(sw) & 0xffffe000u);
      const float clf = cw - chf, slf = sw - shf;
      if constexpr ((e & 1) == 0) {
        st[0] = chf; st[1] = clf; st[2] = shf; st[3] = slf;
      } else {
        ch[e >> 1] = pk_f16(st[0], chf);
        cl[e >> 1] = pk_f16(st[1], clf);
        sh[e >> 1] = pk_f16(st[2], shf);
        sl[e >> 1] = pk_f16(st[3], slf);
      }
    };
    auto flush = [&] {
      acc_sc += (double)sc_c; acc_ss += (double)sc_s;
      sc_c = 0.f; sc_s = 0.f;
      if constexpr (F64) {
        static_for<LIVE>([&](auto ac) {
          constexpr int a = decltype(ac)::v;
          acc_t[2 * a] += (double)Dc[a];
          acc_t[2 * a + 1] += (double)Ds[a];
          Dc[a] = 0.f;
          Ds[a] = 0.f;
        });
      }
    };
    auto fold_lds = [&] {   // (!F64) fp32 table sums of the last run of tiles -> this wave's float64 plane in LDS
      static_for<LIVE>([&](auto ac) {
        constexpr int a = decltype(ac)::v;
        lds_acc[(2 * a) * 64] += (double)Dc[a];
        lds_acc[(2 * a + 1) * 64] += (double)Ds[a];
        Dc[a] = 0.f;
        Ds[a] = 0.f;
      });
    };
    // stage: vector work of tile t (accumulator D, screen P); PREV: the previous tile still owes its step-1 table MFMAs; NEXT: the next tile
    // gets its phase contraction into Dn (its screen is already in the other screen set)
    auto stage = [&](auto prevc, auto nextc, int i, int t, f32x16& D, f32x16& Dn, f32x16& P) {
      constexpr bool PREV = decltype(prevc)::v != 0, NEXT = decltype(nextc)::v != 0;
      constexpr int NQ0 = (NEXT ? NM : 0) + (PREV ? 6 : 0);   // matrix ops dealt over the 8 pixels of step 0
      dyn_transpose(P);
      const f32x4* gs = lds_sci + (size_t)(t - t0) * 8 + h * 4;
      {
        const f32x4 g0 = gs[0], g1 = gs[1];
        static_for<8>([&](auto ec) {
          constexpr int e = decltype(ec)::v;
          vec_pixel(IC<0>{}, ec, D, P, g0, g1, c0, l0, s0, m0);
          constexpr int qa = NQ0 * e / 8, qb = NQ0 * (e + 1) / 8;
          static_for<qb - qa>([&](auto kc) {
            constexpr int q = qa + decltype(kc)::v;
            if constexpr (PREV && q < 6) tab_one(IC<q>{}, ta[2], ta[3], c1, l1, s1, m1);
            else mfma_q(IC<q - (PREV ? 6 : 0)>{}, Dn);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if constexpr (NEXT) {   // the operands those matrix ops read are free again: request the next stage's
        load_modes(mh, ml, t + 2 * stride);
        if constexpr (!DYN) load_psi(IC<0>{}, IC<2>{}, P, t + 2 * stride);   // (pixels of register groups 0 and 1 are done)
      }
      {
        const f16x8* ts = tab16 + ((size_t)t * 4) * 64 + tlane;   // this tile's step-1 table operands (consumed next stage)
        ta[2] = ts[128];
        ta[3] = ts[192];
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const f32x4 g0 = gs[2], g1 = gs[3];
        static_for<8>([&](auto ec) {
          constexpr int e = decltype(ec)::v;
          vec_pixel(IC<1>{}, ec, D, P, g0, g1, c1, l1, s1, m1);
          constexpr int qa = 6 * e / 8, qb = 6 * (e + 1) / 8;
          static_for<qb - qa>([&](auto kc) { tab_one(IC<qa + decltype(kc)::v>{}, ta[0], ta[1], c0, l0, s0, m0); });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if constexpr (NEXT) {
        const f16x8* ts = tab16 + ((size_t)min(t + stride, last) * 4) * 64 + tlane;   // next tile's step-0 table operands
        ta[0] = ts[0];
        ta[1] = ts[64];
        // ring-direct: the four 16-byte pieces a lane takes from its env's 128-byte line go out together, while the line is in the L1
        // (ring-direct: the lines come from HBM — the 0.8 GB of master screens and ring copy the extrusion rewrites every step do not
        // stay in the Infinity Cache — and the waves wait on memory for half their cycles; touching the lines of tile t + 4 with a
        // throw-away dword load made it worse, 175 against 96 us: the touches retire in order in front of the real loads)
        if constexpr (DYN) load_psi(IC<0>{}, IC<4>{}, P, t + 2 * stride);
        else load_psi(IC<2>{}, IC<2>{}, P, t + 2 * stride);
      }
      if ((i % kFlushTiles) == kFlushTiles - 1) flush();
      if constexpr (!F64) {
        if ((i % kTabF32Tiles) == kTabF32Tiles - 1) fold_lds();
      }
    };
    if (n == 1) {
      stage(IC<0>{}, IC<0>{}, 0, first, X, Y, Pa);
    } else {
      stage(IC<0>{}, IC<1>{}, 0, first, X, Y, Pa);
#ifdef AOG_DEV
      if (geo.timeline) { asm volatile("" ::"v"(X[0]), "v"(Y[0])); tl[2] = wall_clock64(); }
#endif
      int i = 1, t = first + stride;
      for (; i + 2 < n; i += 2, t += 2 * stride) {
        stage(IC<1>{}, IC<1>{}, i, t, Y, X, Pb);
        stage(IC<1>{}, IC<1>{}, i + 1, t + stride, X, Y, Pa);
      }
      if (i + 1 < n) {
        stage(IC<1>{}, IC<1>{}, i, t, Y, X, Pb);
        stage(IC<1>{}, IC<0>{}, i + 1, t + stride, X, Y, Pa);
      } else {
        stage(IC<1>{}, IC<0>{}, i, t, Y, X, Pb);
      }
    }
    // the last tile's step-1 table MFMAs
    static_for<6>([&](auto kc) { tab_one(kc, ta[2], ta[3], c1, l1, s1, m1); });
    flush();
    if constexpr (!F64) fold_lds();
  }
#ifdef AOG_DEV
  if (geo.timeline) { asm volatile("" ::"v"(acc_sc)); tl[3] = wall_clock64(); }
#endif
  const int chunk = c * wp + w_p;
  if constexpr (F64) {
    double* out = partials + (size_t)chunk * NS * geo.Bp + (size_t)etile * 32 + (lane & 31);
    static_for<LIVE>([&](auto ac) {
      constexpr int a = decltype(ac)::v;
      const int m = (a & 3) + 8 * (a >> 2) + 4 * h;
      if (m < MRW) {
        out[(size_t)(2 * m) * geo.Bp] = acc_t[2 * a];
        out[(size_t)(2 * m + 1) * geo.Bp] = acc_t[2 * a + 1];
      }
    });
    const double vc = acc_sc + __shfl_down(acc_sc, 32, 64), vs = acc_ss + __shfl_down(acc_ss, 32, 64);
    if (h == 0) {
      out[(size_t)(2 * MRW) * geo.Bp] = vc;
      out[(size_t)(2 * MRW + 1) * geo.Bp] = vs;
    }
  } else {
    double* out = partials + (size_t)chunk * NS * geo.Bp + (size_t)etile * 32 + (lane & 31);
    static_for<LIVE>([&](auto ac) {
      constexpr int a = decltype(ac)::v;
      const int m = (a & 3) + 8 * (a >> 2) + 4 * h;
      if (m < MRW) {
        out[(size_t)(2 * m) * geo.Bp] = lds_acc[(2 * a) * 64];
        out[(size_t)(2 * m + 1) * geo.Bp] = lds_acc[(2 * a + 1) * 64];
      }
    });
    const double vc = acc_sc + __shfl_down(acc_sc, 32, 64), vs = acc_ss + __shfl_down(acc_ss, 32, 64);
    if (h == 0) {
      out[(size_t)(2 * MRW) * geo.Bp] = vc;
      out[(size_t)(2 * MRW + 1) * geo.Bp] = vs;
    }
  }
#ifdef AOG_DEV
  if (geo.timeline && (threadIdx.x & 63) == 0) {
    long long* rec = geo.timeline + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
    rec[0] = tl[0]; rec[1] = tl[1]; rec[2] = tl[2]; rec[3] = tl[3]; rec[4] = wall_clock64(); rec[5] = n;
  }
#endif
}

// Phase-only form of the contraction: u = psi + Mt a for every (pixel, env), written back in the psi_tile layout.  Used by the
// Shack-Hartmann chain, whose mirror (deformable_mirror_shack) carries its own actuators.  One wave per (env tile, pixel tile).
// FIELD: instead of the phases, the Shack-Hartmann chain's input field E = amplitude e^{2 pi i u} x micro-lens phase goes out, complex64 at
// (iy, ix) of the env's image (compact N x N for the pruned passes), through a [32 envs][32 pixels] tile in LDS so that a store instruction
// writes 256 contiguous bytes per env (k_sh_field re-read the phases through the tile layout and wrote 8 bytes per thread: 0.23 ms per
// 1024 envs at N = 256 on top of this kernel's 0.11).
// GRID (with FIELD): only a phase leaves this kernel — w = u + (micro-lens phase of the pixel), reduced to [-1/2, 1/2] revolutions, as ONE
// float at (iy, ix) — and the first propagation pass forms E = amplitude e^{2 pi i w} itself while it loads: 4 bytes written and read per
// pixel instead of 8, one load per pixel as before.  Pixels outside the aperture hold kShOutside (written once at upload): field 0.
// (A first form kept the micro-lens factor as a complex table multiplied in by the pass: its second load per pixel cost the pass 1.4 ms.)
template <int A_PAD, bool FIELD = false, bool GRID = false>
__global__ __launch_bounds__(256) void k_phase_mfma(const f16x8* __restrict__ modes16, const f32x4* __restrict__ psi_tile,
                                                    const f16x8* __restrict__ act16, f32x4* __restrict__ out_tile, int n_ptiles,
                                                    int n_etiles, PhaseFieldArgs fa = PhaseFieldArgs{}, int etiles_per_wave = 1) {
  constexpr int NSTEP = A_PAD / 16;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  __shared__ float2 field_lds[(FIELD && !GRID) ? 4 * 32 * 33 : 1];
  __shared__ float grid_lds[GRID ? 4 * 32 * 33 : 1];   // (GRID: one float per pixel — half the LDS, twice the workgroups per CU)
  [[maybe_unused]] float2* field_tile = field_lds + ((FIELD && !GRID) ? (threadIdx.x >> 6) * 32 * 33 : 0);
  [[maybe_unused]] float* grid_tile = grid_lds + (GRID ? (threadIdx.x >> 6) * 32 * 33 : 0);
  const int et0 = blockIdx.y * etiles_per_wave, et1 = min(et0 + etiles_per_wave, n_etiles);
  if (t >= n_ptiles || et0 >= n_etiles) return;
  // A wave takes ONE pixel tile through `etiles_per_wave` env tiles: the mode operands are loaded once, and the next env tile's actuator
  // operands and screen values are requested while this one is reduced and stored (one env tile per wave put both round trips and the wave's
  // start-up in front of ~1 k cycles of work: 0.95 ms per 2048 envs at N = 512 for 3.4 GB)
  const f16x8* ms = modes16 + ((size_t)t * NSTEP * 2) * 64 + lane;
  f16x8 mh[NSTEP], ml[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) { mh[s] = ms[(2 * s) * 64]; ml[s] = ms[(2 * s + 1) * 64]; }
  f16x8 bh[NSTEP], bl[NSTEP], bhn[NSTEP], bln[NSTEP];
  f32x4 pc[4], pn[4];
  auto request = [&](int etile, f16x8 (&ah)[NSTEP], f16x8 (&al)[NSTEP], f32x4 (&pp)[4]) {
    const f16x8* asrc = act16 + ((size_t)etile * NSTEP * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) { ah[s] = asrc[(2 * s) * 64]; al[s] = asrc[(2 * s + 1) * 64]; }
    const size_t base = (((size_t)etile * n_ptiles + t) * 4) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) pp[g] = psi_tile[base + g * 64];
  };
  request(et0, bh, bl, pc);
  [[maybe_unused]] const int h = lane >> 5;
  // (per-pixel constants of this tile, the same for every env tile)
  [[maybe_unused]] const int qs = lane & 31, pix_s = t * 32 + qs;
  [[maybe_unused]] const int yx_s = (FIELD && pix_s < fa.n_ap) ? fa.ap_yx[pix_s] : 0;
  [[maybe_unused]] const size_t at_s = (size_t)(yx_s >> 16) * fa.row_stride + (yx_s & 0xffff);
  for (int etile = et0; etile < et1; ++etile) {
    if (etile + 1 < et1) request(etile + 1, bhn, bln, pn);   // (wave-uniform)
    f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], bh[s], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], bl[s], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml[s], bh[s], d, 0, 0, 0);
      if constexpr (GRID) {
        // K4: the actuators to 33 bits.  A rounding error of an ACTUATOR is a smooth phase error over the whole pupil — it does not average
        // down over the pixels like the per-pixel rounding of a mode value does — and at 2^-23 of an actuator of half a revolution it was
        // most of the error of the focal fields (7e-8 of the peak amplitude, the whole tolerance of a pixel 30 dB down)
        if (fa.act_ll) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh[s], fa.act_ll[((size_t)etile * NSTEP + s) * 64 + lane], d, 0, 0, 0);
      }
    }
    const size_t base = (((size_t)etile * n_ptiles + t) * 4) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 p = pc[g];
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(d[4 * g + r], kPhaseUnscale, p[r]);
      if constexpr (!FIELD) {
        out_tile[base + g * 64] = o;
      } else {
        // this lane's four field values of register group g -> the wave's [32 envs][32 pixels] tile in LDS; written out below with the
        // lanes along the PIXELS of an env (256 contiguous bytes per env and instruction: 32-byte pieces straight from the accumulator
        // layout ran the kernel at 1.2 TB/s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 8 * g + 4 * h + r;
          const int pix = min(t * 32 + q, fa.n_ap - 1);
          [[maybe_unused]] const int yx = GRID ? 0 : fa.ap_yx[pix], iy = yx >> 16, ix = yx & 0xffff;
          if constexpr (GRID) {
            // screen and mirror phase are reduced to a revolution EACH before they are added: their sum then rounds at 2^-25 .. 2^-24 of a
            // revolution instead of at the ulp of a phase of several revolutions (which was most of the error of the K4 focal fields: 0.93
            // -> 0.5 of the test tolerance at N = 64, where the image is a speckle field and every pixel's phase error counts)
            const float dm = d[4 * g + r] * kPhaseUnscale, ps = p[r];
            const float w = ((ps - rintf(ps)) + (dm - rintf(dm))) + (fa.mla_rev ? fa.mla_rev[pix] : 0.f);   // + the micro-lens phase of this pixel (K4: none)
            grid_tile[(lane & 31) * 33 + q] = w - rintf(w);
          } else {
            float sn, cs;
            sincospif(2.0f * (o[r] - rintf(o[r])), &sn, &cs);
            const float2 m = fa.mla32[iy * fa.N + ix];
            field_tile[(lane & 31) * 33 + q] = make_float2(fa.amplitude * (cs * m.x - sn * m.y), fa.amplitude * (cs * m.y + sn * m.x));
          }
        }
      }
    }
    if constexpr (FIELD) {
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tile is private to the wave
      __builtin_amdgcn_wave_barrier();
      if (pix_s < fa.n_ap) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int el = 2 * j + (lane >> 5), env_j = etile * 32 + el;
          if (env_j < fa.B) {
            if constexpr (GRID) reinterpret_cast<float*>(fa.field)[(size_t)env_j * fa.env_stride + at_s] = grid_tile[el * 33 + qs];
            else fa.field[(size_t)env_j * fa.env_stride + at_s] = field_tile[el * 33 + qs];
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // (the tile is rewritten by the next env tile)
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) { bh[s] = bhn[s]; bl[s] = bln[s]; }
#pragma unroll
    for (int g = 0; g < 4; ++g) pc[g] = pn[g];
  }
}


}  // namespace aog
