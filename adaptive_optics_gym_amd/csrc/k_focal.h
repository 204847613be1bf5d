// K4: focal-plane field export (single env float64 form and the batched split-f16 matrix-core form).
#pragma once
#include "k_common.h"

namespace aog {

// ------------------------------------------------------------------------------------------------
// K4  focal-plane field of one env (propagator_fiber, AO_env.py:138), off the step() path.
//   E[y][x]  = exp(2 pi i (psi + M a))  on the aperture (amplitude folded into focal_m1), 0 outside
//   T[v][x]  = sum_y m1[v][y] E[y][x];     F[v][u] = sum_x T[v][x] m2[x][u]        (float64 accumulation)
// ------------------------------------------------------------------------------------------------
__global__ void k_focal_field(const float* __restrict__ psi_tile, const double* __restrict__ psi64, const float* __restrict__ modes_f32,
                              const double* __restrict__ modes64, const float* __restrict__ act_rev, const double* __restrict__ act_dm,
                              const int32_t* __restrict__ ap_index, double2* __restrict__ E, int env, int n_ap, int n_ptiles, int A,
                              int A_pad, int Bp, double lambda_wfs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_ap) return;
  double rev;
  if (psi64) {  // float64 validation handle
    double surf = 0;
    for (int k = 0; k < A; ++k) surf = fma(modes64[(size_t)p * A + k], act_dm[(size_t)env * A + k], surf);
    rev = (psi64[(size_t)env * n_ap + p] + 4.0 * M_PI * surf) / (2.0 * M_PI * lambda_wfs);
  } else {
    double acc = (double)psi_tile[psi_tile_index(env, p, n_ptiles)];
    const double two_over_lambda = 2.0 / lambda_wfs;   // actuators (metres) -> revolutions per unit mode, as the prologue does
    for (int k = 0; k < A; ++k) acc = fma((double)modes_f32[(size_t)p * A_pad + k], act_dm[(size_t)env * A + k] * two_over_lambda, acc);
    rev = acc;
  }
  double sn, cs;
  sincospi(2.0 * (rev - rint(rev)), &sn, &cs);
  E[ap_index[p]] = make_double2(cs, sn);
}

// K4, batched (aog_focal_images): both products on the f16 matrix cores with every operand split hi + lo (the step kernel's
// contraction: 3 x v_mfma_f32_32x32x16_f16 per real product, 22 significant bits per factor, exact products, fp32 sums).  The fp32
// matrix instruction this path used in round 2 runs at 1/16 of the f16 rate and does not co-execute with vector work.
//   pass 1  (k_focal_pass1):  T'^T[x][v] = sum_y E[y][x] m1'[v][y],  E = e^{2 pi i w} formed from the dense phase grid k_phase_mfma<GRID>
//           writes (one float per pixel, kShOutside outside the aperture -> E = 0) while it is loaded: E never exists in memory
//   pass 2  (k_focal_pass2):  F[v][u] = sum_x T'[v][x] m2'[x][u] / scale
// m1' = m1 2^e1, m2' = m2 2^e2 (largest component in [1/2, 1): the f16 halves stay normal), scale = 2^(e1 + e2).
// Operand tiles are stored MFMA-ready: one tile = [part: re hi, re lo, im hi, im lo][lane 64][8 f16] = 4 KiB; lane l carries row / column
// l & 31 and the 8 k-slots of k-group l >> 5.  Pass 1 leaves T' already split, in tiles [x tile of 32][v block][s][part][lane]: the 16
// accumulator registers of a lane (column v = l & 31, rows x = (r & 3) + 8 (r >> 2) + 4 (l >> 5)) are two k-groups of 8 (s = r >> 3) for
// pass 2, whose m2' table is laid out in the same order of x — the matrix instruction sums over k whatever order the slots are in, so no
// transposition happens anywhere.  Workgroup = 4 waves = 4 x 32 columns (v blocks / u blocks) of ONE 128-row span; per k-step the four
// waves produce the span's four A tiles into LDS (pass 1: one x tile each — 8 loads, 16 transcendentals, mask split; pass 2: one copied
// T' tile each), every wave then runs 4 tiles x 12 matrix instructions against its own B tile from the L2-resident table.
constexpr int kFocalTile = 4 * 64;   // f16x8 per operand tile
// hi = x rounded to nearest f16, lo = x - hi rounded to nearest: an unbiased 22-bit operand.  (The step kernel's cheaper split by mask
// truncates both halves; here the truncation error — a fixed non-linear function of cos / sin of the phase — showed up as ghost terms of
// 1e-7 of the peak amplitude, the whole error budget of a pixel 30 dB down; this path has the vector slots to round properly.)
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)x[j];
    hi[j] = h;
    lo[j] = (_Float16)(x[j] - (float)h);
  }
}
__device__ __forceinline__ f16x8 neg8(f16x8 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, v) ^ 0x80008000u);
}
// one A tile (LDS, [part][lane]) against a wave's B tile (registers; nbh, nbl = -Bi): Cr += Ar Br - Ai Bi, Ci += Ar Bi + Ai Br
__device__ __forceinline__ void focal_mma_tile(const f16x8* __restrict__ a_tile, int lane, const f16x8 (&b)[4], f16x8 nbh, f16x8 nbl, f32x16& cr, f32x16& ci) {
  const f16x8 arh = a_tile[0 * 64 + lane], arl = a_tile[1 * 64 + lane], aih = a_tile[2 * 64 + lane], ail = a_tile[3 * 64 + lane];
  // (the two accumulation chains alternate; small terms first)
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, b[0], cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, b[2], ci, 0, 0, 0);
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, b[1], cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, b[3], ci, 0, 0, 0);
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, nbh, cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, b[0], ci, 0, 0, 0);
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, nbl, cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, b[1], ci, 0, 0, 0);
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, b[0], cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, b[2], ci, 0, 0, 0);
  cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, nbh, cr, 0, 0, 0);
  ci = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, b[0], ci, 0, 0, 0);
}
// one k-step of a wave: its four A tiles (LDS, [tile][part][lane]) against its B tile
__device__ __forceinline__ void focal_mma(const f16x8* __restrict__ a_lds, int lane, const f16x8 (&b)[4], f32x16 (&cr)[4], f32x16 (&ci)[4]) {
  const f16x8 nbh = neg8(b[2]), nbl = neg8(b[3]);
#pragma unroll
  for (int t = 0; t < 4; ++t) focal_mma_tile(a_lds + t * kFocalTile, lane, b, nbh, nbl, cr[t], ci[t]);
}
// pass 1.  grid (Nxp / 128, nfp / 128, envs); phase [env][Nyp][Nxp]; m1s [nfp / 32][Nyp / 16] tiles; T16 [env][Nxp / 32][nfp / 32][2] tiles
__global__ __launch_bounds__(256, 2) void k_focal_pass1(const float* __restrict__ phase, const f16x8* __restrict__ m1s, f16x8* __restrict__ T16, int Nxp,
                                                        int Nyp, int nfp) {
  __shared__ f16x8 a_lds[2][4 * kFocalTile];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int env = blockIdx.z, x0 = blockIdx.x * 128, vb = blockIdx.y * 4 + wave;
  const int nk = Nyp / 16;
  const float* __restrict__ src = phase + ((size_t)env * Nyp + 8 * (lane >> 5)) * Nxp + x0 + 32 * wave + (lane & 31);
  const f16x8* __restrict__ bsrc = m1s + (size_t)vb * nk * kFocalTile + lane;
  f32x16 cr[4], ci[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { cr[t][r] = 0.f; ci[t][r] = 0.f; }
  float w[8];   // phases of the k-step being produced next (requested one k-step ahead, before the matrix instructions; two ahead: no gain)
  f16x8 b[4], bn[4];
  auto load_w = [&](int ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = src[(size_t)(ks * 16 + j) * Nxp];
  };
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = bsrc[(size_t)ks * kFocalTile + q * 64];
  };
  auto produce = [&](int buf) {   // this wave's x tile of the k-step whose phases are in w -> LDS
    float c[8], s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool in = w[j] < 1.5f;
      c[j] = in ? __builtin_amdgcn_cosf(w[j]) : 0.f;   // (the instructions take revolutions; sincospif changed nothing measurable)
      s[j] = in ? __builtin_amdgcn_sinf(w[j]) : 0.f;
    }
    f16x8 ch, cl, sh, sl;
    split8(c, ch, cl);
    split8(s, sh, sl);
    f16x8* dst = a_lds[buf] + wave * kFocalTile + lane;
    dst[0] = ch; dst[64] = cl; dst[128] = sh; dst[192] = sl;
  };
  load_w(0);
  load_b(0, b);
  produce(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int nxt = min(ks + 1, nk - 1);
    // the loads of the coming k-steps go out BEFORE this k-step's matrix instructions (left alone the compiler sinks them to their first
    // use, after the matrix instructions, and every k-step then waits a full memory round trip between two bursts of matrix work)
    load_w(nxt);
    load_b(nxt, bn);
    __builtin_amdgcn_sched_barrier(0);
    focal_mma(a_lds[ks & 1], lane, b, cr, ci);
    __builtin_amdgcn_sched_barrier(0);
    produce((ks + 1) & 1);   // (unconditional: under `if (ks + 1 < nk)` the loads above are sunk into the branch, behind the matrix instructions; the last
                             // k-step re-produces its own tile into the buffer nobody reads any more)
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = bn[q];
    __syncthreads();
  }
  // T' leaves split and in pass 2's operand order: registers 8 s .. 8 s + 7 of a lane = the 8 k-slots of k-step s of this x tile
  const int nvb = nfp / 32;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int xt = (x0 >> 5) + t;
    f16x8* dst = T16 + ((((size_t)env * (Nxp / 32) + xt) * nvb + vb) * 2) * kFocalTile + lane;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float vr[8], vi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { vr[j] = cr[t][8 * s2 + j]; vi[j] = ci[t][8 * s2 + j]; }
      f16x8 rh, rl, ih, il;
      split8(vr, rh, rl);
      split8(vi, ih, il);
      f16x8* d = dst + (size_t)s2 * kFocalTile;
      d[0] = rh; d[64] = rl; d[128] = ih; d[192] = il;
    }
  }
}
// pass 2.  grid (nfp / 128 [u], nfp / 128 [v], envs); m2s [nfp / 32][Nxp / 32][2] tiles; F [env][nf][nf] complex64
__global__ __launch_bounds__(256, 2) void k_focal_pass2(const f16x8* __restrict__ T16, const f16x8* __restrict__ m2s, float2* __restrict__ F, int Nxp, int nfp,
                                                        int nf, float unscale) {
  __shared__ f16x8 a_lds[2][4 * kFocalTile];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int env = blockIdx.z, ub = blockIdx.x * 4 + wave, vb0 = blockIdx.y * 4;
  const int nvb = nfp / 32, nk = (Nxp / 32) * 2;
  // k-step ks = (x tile ks >> 1, s = ks & 1); this wave copies the tile of v block vb0 + wave
  const f16x8* __restrict__ asrc = T16 + ((size_t)env * (Nxp / 32) * nvb + vb0 + wave) * 2 * kFocalTile + lane;
  const f16x8* __restrict__ bsrc = m2s + (size_t)ub * nk * kFocalTile + lane;
  f32x16 cr[4], ci[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { cr[t][r] = 0.f; ci[t][r] = 0.f; }
  // (fp32 sums over the whole of x.  Folding them into float64 every one or two k-steps was built and measured: worst error 0.84 -> 0.45 of
  // the test tolerance at N = 256, but 384 accumulator registers mean one wave per SIMD and the kernel went from 100 to 250 us.)
  f16x8 a[4], b[4], bn[4];
  auto load_a = [&](int ks) {
    const f16x8* p = asrc + ((size_t)(ks >> 1) * nvb * 2 + (ks & 1)) * kFocalTile;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = p[q * 64];
  };
  auto load_b = [&](int ks, f16x8 (&dst)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = bsrc[(size_t)ks * kFocalTile + q * 64];
  };
  auto produce = [&](int buf) {
    f16x8* dst = a_lds[buf] + wave * kFocalTile + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * 64] = a[q];
  };
  load_a(0);
  load_b(0, b);
  produce(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int nxt = min(ks + 1, nk - 1);
    load_a(nxt);   // (ahead of the matrix instructions: see pass 1)
    load_b(nxt, bn);
    __builtin_amdgcn_sched_barrier(0);
    focal_mma(a_lds[ks & 1], lane, b, cr, ci);
    __builtin_amdgcn_sched_barrier(0);
    produce((ks + 1) & 1);   // (unconditional: under `if (ks + 1 < nk)` the loads above are sunk into the branch, behind the matrix instructions; the last
                             // k-step re-produces its own tile into the buffer nobody reads any more)
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = bn[q];
    __syncthreads();
  }
  const int u = ub * 32 + (lane & 31);
  if (u < nf) {
    float2* Fe = F + (size_t)env * nf * nf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int v = (vb0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (v < nf) Fe[(size_t)v * nf + u] = make_float2(cr[t][r] * unscale, ci[t][r] * unscale);
      }
  }
}

// out[r][c] = sum_k a[r][k] * b[k][c]  (complex, row-major), one thread per output
__global__ void k_cgemm_small(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ out, float2* __restrict__ out32,
                              int R, int K, int Cn) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * Cn) return;
  const int r = idx / Cn, c = idx - r * Cn;
  double re = 0, im = 0;
  for (int k = 0; k < K; ++k) {
    const double2 x = a[(size_t)r * K + k], y = b[(size_t)k * Cn + c];
    re = fma(x.x, y.x, fma(-x.y, y.y, re));
    im = fma(x.x, y.y, fma(x.y, y.x, im));
  }
  if (out) out[idx] = make_double2(re, im);
  if (out32) out32[idx] = make_float2((float)re, (float)im);
}

}  // namespace aog
