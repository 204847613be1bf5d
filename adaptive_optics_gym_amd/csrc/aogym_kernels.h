// Device kernels of libaogym.so (gfx950 only).  See DESIGN.md for the data layout and the roofline of each.
//
// Notation: B envs (padded to Bp, a multiple of 64), n_ap aperture pixels packed row-major (padded to a
// multiple of 32 = one MFMA pixel tile), A modes (padded to A_PAD), MRW / MRS real pupil-plane tables at the
// wavefront-sensing / science wavelength, NS = 2*(MRW+MRS) real sums per env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace aog {

constexpr float kLog2Dummy = 0.f;

// ------------------------------------------------------------------------------------------------
// sin/cos of 2*pi*u for u in revolutions.  Range reduction is exact in fp32 (u - rint(u), then the
// octant split), so accuracy does not degrade with |u|; max abs error 7.4e-8, zero mean bias (checked
// on the host against float64 and on the device by tests/test_gpu_kernels.py).
//   SINCOS = 0: polynomial (degree 7 / 8 in the reduced argument)
//   SINCOS = 1: hardware v_sin_f32 / v_cos_f32, which take revolutions directly
// ------------------------------------------------------------------------------------------------
template <int SINCOS>
__device__ __forceinline__ void sincos_rev(float u, float& s, float& c) {
  const float r = u - rintf(u);  // [-0.5, 0.5], exact
  if constexpr (SINCOS == 1) {
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
  } else {
    const float q = rintf(4.0f * r);       // -2..2
    const float t = fmaf(q, -0.25f, r);    // [-1/8, 1/8], exact
    const float z = t * t;
    const float ps = fmaf(z, fmaf(z, -75.43880659180556f, 81.5934996521887f), -41.34166926730038f);
    // sin(2 pi t) = t*(2pi_hi) + t*(2pi_lo + z*ps)
    const float sp = fmaf(t, 6.2831854820251465f, t * fmaf(z, ps, -1.8420333e-07f));
    const float pc = fmaf(z, fmaf(z, fmaf(z, 59.43078516585609f, -85.44897459881716f), 64.93936989759587f),
                          -19.739208790231338f);
    const float cp = fmaf(z, pc, 1.0f);
    const int qi = (int)q;
    const bool swap = (qi & 1) != 0;
    const float ss = swap ? cp : sp;
    const float cc = swap ? sp : cp;
    s = (qi & 2) ? -ss : ss;
    c = ((qi + 1) & 2) ? -cc : cc;
  }
}

// ------------------------------------------------------------------------------------------------
// K0  pack_screens: achromatic screens [count][N][N] (T = double|float) -> internal layouts.
//   psi_rev   fp32, revolutions at lambda_wfs, aperture mean removed, layout [quad q][env][4]
//             (lane = env reads one float4 = 4 consecutive packed pixels; 1 KiB per wave instruction)
//   psi_tile  fp32, same values in MFMA accumulator order (see k_fused_tab)
//   psi64     (validation mode) float64 [env][n_ap], aperture mean removed, hcipy units
// One workgroup per env.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_sum(double v, double* sm) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}

// index of (env, packed pixel p) in the MFMA-tiled screen layout:
//   [env_tile = env/32][pixel tile = p/32][g = (p%32)/8][lane = 32*h + env%32][r = p%4],  h = ((p%32)/4)&1
__device__ __host__ __forceinline__ size_t psi_tile_index(int env, int p, int n_ptiles) {
  const int et = env >> 5, e = env & 31, pt = p >> 5, i = p & 31;
  const int g = i >> 3, h = (i >> 2) & 1, r = i & 3;
  return ((((size_t)et * n_ptiles + pt) * 4 + g) * 64 + (h * 32 + e)) * 4 + r;
}

// `origin` (nullable, [env][2] = (ox, oy)): the source screens are toroidal ring buffers (dynamic atmosphere) whose
// logical pixel (iy, ix) lives at physical ((iy + oy) mod N, (ix + ox) mod N).
template <typename T>
__global__ __launch_bounds__(256) void k_pack_screens(const T* __restrict__ psi, const int32_t* __restrict__ ap_index,
                                                      float* __restrict__ psi_rev, float* __restrict__ psi_tile,
                                                      double* __restrict__ psi64, int first, int n_pix2, int n_ap,
                                                      int n_ap_pad, int Bp, double inv_two_pi_lambda,
                                                      const int32_t* __restrict__ origin, int N,
                                                      double* __restrict__ offset_out = nullptr,
                                                      double* __restrict__ sum_out = nullptr) {
  __shared__ double sm[8];
  const int e = blockIdx.x;
  const int env = first + e;
  const T* src = psi + (size_t)e * n_pix2;
  int ox = 0, oy = 0;
  if (origin) {
    ox = origin[2 * env];
    oy = origin[2 * env + 1];
  }
  auto phys = [&](int flat) {
    if (!origin) return flat;
    const int iy = flat / N, ix = flat - iy * N;
    int py = iy + oy, px = ix + ox;
    if (py >= N) py -= N;
    if (px >= N) px -= N;
    return py * N + px;
  };
  double acc = 0;
  for (int p = threadIdx.x; p < n_ap; p += blockDim.x) acc += (double)src[phys(ap_index[p])];
  const double total = block_reduce_sum(acc, sm);
  const double mean = total / (double)n_ap;
  if (threadIdx.x == 0) {
    if (offset_out) offset_out[env] = mean;
    if (sum_out) sum_out[env] = total;  // as if a repack had just measured this screen
  }
  const int n_ptiles = n_ap_pad >> 5;
  for (int p = threadIdx.x; p < n_ap_pad; p += blockDim.x) {
    const double v = (p < n_ap) ? ((double)src[phys(ap_index[p])] - mean) : 0.0;
    const float vr = (float)(v * inv_two_pi_lambda);
    if (psi_rev) psi_rev[((size_t)(p >> 2) * Bp + env) * 4 + (p & 3)] = vr;
    if (psi_tile) psi_tile[psi_tile_index(env, p, n_ptiles)] = vr;
    if (psi64 && p < n_ap) psi64[(size_t)env * n_ap + p] = v;
  }
}

// The same conversion for whole batches (semi_dynamic resets install thousands of screens at once): k_pack_screens writes 4 bytes per
// thread into layouts whose contiguous runs are 16 bytes per env, so its stores are what it waits for.  Here a workgroup takes one env
// tile (32 envs) x kPackTiles pixel tiles, gathers the aperture pixels env by env (coalesced along the packed index), transposes through
// LDS and stores whole 1-KiB MFMA register groups (psi_tile) / 512-byte quad rows (psi_rev).  Aperture means come from k_screen_means.
template <typename T>
__global__ __launch_bounds__(256) void k_screen_means(const T* __restrict__ psi, const int32_t* __restrict__ ap_index, double* __restrict__ mean,
                                                      int n_pix2, int n_ap) {
  __shared__ double sm[8];
  const T* src = psi + (size_t)blockIdx.x * n_pix2;
  double acc = 0;
  // (same order of additions as k_pack_screens' loop — the two conversion paths must agree bit for bit — but eight gathers in flight)
  for (int p0 = threadIdx.x; p0 < n_ap; p0 += 8 * blockDim.x) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = p0 + u * (int)blockDim.x;
      v[u] = p < n_ap ? src[ap_index[p]] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u * (int)blockDim.x < n_ap) acc += (double)v[u];
  }
  const double total = block_reduce_sum(acc, sm);
  if (threadIdx.x == 0) mean[blockIdx.x] = total / (double)n_ap;
}

constexpr int kPackTiles = 8;   // pixel tiles (of 32 packed pixels) per workgroup
template <typename T>
__global__ __launch_bounds__(256) void k_pack_tiles(const T* __restrict__ psi, const int32_t* __restrict__ ap_index, const double* __restrict__ mean,
                                                    float* __restrict__ psi_rev, float* __restrict__ psi_tile, int first, int count, int n_pix2,
                                                    int n_ap, int n_ptiles, int Bp, double inv_two_pi_lambda) {
  __shared__ float tile[kPackTiles * 32][33];   // [packed pixel of the block][env of the tile]
  const int et = (first >> 5) + blockIdx.y;      // env tile
  const int pt0 = blockIdx.x * kPackTiles;
  const int npix = min(kPackTiles, n_ptiles - pt0) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // gather: one wave = one env at a time, lanes along the packed pixel index; a wave's 8 envs x 4 pixels per lane are requested together
  constexpr int PL = kPackTiles * 32 / 64;   // pixels per lane
  int flat[PL];
#pragma unroll
  for (int u = 0; u < PL; ++u) {
    const int p = pt0 * 32 + lane + 64 * u;
    flat[u] = (lane + 64 * u < npix && p < n_ap) ? ap_index[p] : -1;
  }
  T raw[8][PL];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int env = et * 32 + wave + 4 * j;
    const bool live = env >= first && env < first + count;   // (wave-uniform)
    const T* src = psi + (size_t)(live ? env - first : 0) * n_pix2;
#pragma unroll
    for (int u = 0; u < PL; ++u) raw[j][u] = (live && flat[u] >= 0) ? src[flat[u]] : (T)0;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int el = wave + 4 * j, env = et * 32 + el;
    const bool live = env >= first && env < first + count;
    const double mu = live ? mean[env - first] : 0.0;
#pragma unroll
    for (int u = 0; u < PL; ++u) {
      const int pl = lane + 64 * u;
      if (pl < npix) tile[pl][el] = (live && flat[u] >= 0) ? (float)(((double)raw[j][u] - mu) * inv_two_pi_lambda) : 0.f;
    }
  }
  __syncthreads();
  // psi_tile: [env tile][pixel tile][g 4][lane = 32 h + e][r 4], pixel of the tile = 8 g + 4 h + r
  const int h = lane >> 5, e = lane & 31;
  const int env = et * 32 + e;
  const bool mine = env >= first && env < first + count;
  if (psi_tile && mine) {
    for (int c = wave; c < (npix >> 5) * 4; c += 4) {
      const int ptl = c >> 2, g = c & 3;
      const int pl = ptl * 32 + 8 * g + 4 * h;
      const float4 v = make_float4(tile[pl][e], tile[pl + 1][e], tile[pl + 2][e], tile[pl + 3][e]);
      *reinterpret_cast<float4*>(psi_tile + ((((size_t)et * n_ptiles + pt0 + ptl) * 4 + g) * 64 + lane) * 4) = v;
    }
  }
  // psi_rev: [quad][env][4]: two quads per wave instruction (lanes 0-31 / 32-63)
  if (psi_rev && mine) {
    for (int c = wave; c < (npix >> 3); c += 4) {
      const int ql = 2 * c + h;
      const float4 v = make_float4(tile[4 * ql][e], tile[4 * ql + 1][e], tile[4 * ql + 2][e], tile[4 * ql + 3][e]);
      *reinterpret_cast<float4*>(psi_rev + ((size_t)(pt0 * 8 + ql) * Bp + env) * 4) = v;
    }
  }
}

// one actuator value (revolutions) -> the hi/lo f16 B-operand layout of k_fused_tab:
//   act16[env tile][s = i/16][hi|lo][lane = 32*((i/8)&1) + env%32][i%8]   (A_pad is a multiple of 16 for this layout)
__device__ __forceinline__ void store_act16(_Float16* __restrict__ act16, int env, int i, int A_pad, float ar) {
  if (act16 == nullptr || (A_pad & 15)) return;
  const float sc = ar * 256.0f;  // kActScale
  const _Float16 hi = (_Float16)sc;
  const _Float16 lo = (_Float16)(sc - (float)hi);   // unscaled low half (see split_f16)
  const int s = i >> 4, h = (i >> 3) & 1, e = i & 7, nstep = A_pad >> 4;
  const size_t base = (((size_t)(env >> 5) * nstep + s) * 2) * 64 + (h * 32 + (env & 31));
  act16[base * 8 + e] = hi;
  act16[(base + 64) * 8 + e] = lo;
}

// ------------------------------------------------------------------------------------------------
// K1  prologue: action -> actuators (AO_env.py:115-120).  One wave per env, four per workgroup.
//   a'_i = action_i / (i + 10);  var = a'^T G a'  (G = centred Gram, float64);  a'' = a' * target / sqrt(var)
//   act_dm  [B][A] float64 (metres)         — deformable_mirror.actuators
//   act_rev [A_PAD][Bp] float32             — 2 a''/lambda_wfs (revolutions of wfs phase per unit mode)
//   act16   (MFMA B-operand order, hi/lo f16 halves) [env/32][A_PAD/16][hi|lo][64 lanes][8]
// A zero action gives 0/0 = NaN exactly like numpy (documented in DESIGN.md).
// ------------------------------------------------------------------------------------------------
#ifdef AOG_MAIN_TU
constexpr int kProEnvs = 4;   // envs (= waves) per workgroup: they share one copy of the Gram matrix in LDS
// SHARED_GRAM = false: the Gram matrix is read through the caches instead of a 32 KB LDS copy — same arithmetic in the same order (bit-
// identical), 1 us slower, but the workgroup then fits beside a resident extrusion workgroup (146 KB of a CU's 160 KB LDS): the form
// aog_step uses while the next step's extrusion runs on the library's stream (aog_set_lookahead)
template <bool SHARED_GRAM, int ENVS>
__device__ __forceinline__ void prologue_body(const float* __restrict__ action, const double* __restrict__ gram,
                                              double* __restrict__ act_dm, float* __restrict__ act_rev,
                                              _Float16* __restrict__ act16, int B, int A, int A_pad, int Bp,
                                              int sh_operation, double target, double two_over_lambda, int block) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = block * ENVS + wave;
  const bool live = env < B;
  __shared__ double Gs[SHARED_GRAM ? 64 * 64 : 1];   // G[j][i] at j * 64 + i (A <= 64); lane i then reads a conflict-free row per j
  __shared__ double aps[ENVS][256];
  double* ap = aps[wave];
  // A <= 64 (every fast-path config of the reference): the Gram matrix crosses L2 -> LDS ONCE per workgroup, every load of it in
  // flight together with the action loads: one memory round trip in front of the arithmetic (4 K multiply-adds per env).  Round 1
  // had every env pull its own 32 KB copy through L2 (33 MB per step at B = 1024).
  const bool pre = SHARED_GRAM && !sh_operation && A <= 64;
  if (pre) {
    constexpr int PER = 64 * 64 / (64 * ENVS);
    double g[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = threadIdx.x + 64 * ENVS * u;
      const int j = idx >> 6, i = idx & 63;
      g[u] = gram[(size_t)min(j, A - 1) * A + min(i, A - 1)];
    }
    for (int i = lane; i < A; i += 64) {
      const double a = live ? (double)action[(size_t)env * A + i] : 1.0;
      ap[i] = a / (double)(i + 10);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) Gs[threadIdx.x + 64 * ENVS * u] = g[u];
  } else {
    for (int i = lane; i < A; i += 64) {
      const double a = live ? (double)action[(size_t)env * A + i] : 1.0;
      ap[i] = sh_operation ? a : a / (double)(i + 10);
    }
  }
  __syncthreads();
  if (!live) return;
  double scale = 1.0;
  if (!sh_operation) {
    double part = 0;
    if (pre) {
      if (lane < A) {
        double r = 0;
        for (int j = 0; j < A; ++j) r = fma(Gs[j * 64 + lane], ap[j], r);
        part = ap[lane] * r;
      }
    } else {
      for (int i = lane; i < A; i += 64) {
        const double* gcol = gram + i;   // G is symmetric: column i read with the lanes along a row (coalesced)
        double r = 0;
        for (int j = 0; j < A; ++j) r = fma(gcol[(size_t)j * A], ap[j], r);
        part = fma(ap[i], r, part);
      }
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    part = __shfl(part, 0, 64);
    scale = target / sqrt(part);
  }
  for (int i = lane; i < A_pad; i += 64) {
    const double a = (i < A) ? ap[i] * scale : 0.0;
    if (i < A) act_dm[(size_t)env * A + i] = a;
    const float ar = (float)(a * two_over_lambda);
    if (act_rev) act_rev[(size_t)i * Bp + env] = ar;
    store_act16(act16, env, i, A_pad, ar);
  }
}
template <bool SHARED_GRAM>
__global__ __launch_bounds__(64 * kProEnvs) void k_prologue(const float* __restrict__ action, const double* __restrict__ gram,
                                                            double* __restrict__ act_dm, float* __restrict__ act_rev,
                                                            _Float16* __restrict__ act16, int B, int A, int A_pad, int Bp,
                                                            int sh_operation, double target, double two_over_lambda) {
  prologue_body<SHARED_GRAM, kProEnvs>(action, gram, act_dm, act_rev, act16, B, A, A_pad, Bp, sh_operation, target, two_over_lambda, (int)blockIdx.x);
}
#endif  // AOG_MAIN_TU

// actuators (metres, float64) -> the two fp32 operand layouts (used by reset / set_actuators)
#ifdef AOG_MAIN_TU
// act16_ll (nullable, K4): what the two f16 halves of act16 leave of the float64 actuator, as a third f16 term in the same operand order
// without the hi | lo dimension: [env tile][A_pad / 16][lane][8]
__global__ void k_load_actuators(const double* __restrict__ act_dm, float* __restrict__ act_rev,
                                 _Float16* __restrict__ act16, int B, int A, int A_pad, int Bp, double two_over_lambda,
                                 _Float16* __restrict__ act16_ll = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A_pad) return;
  const int env = idx / A_pad, i = idx % A_pad;
  const double a64 = (i < A) ? act_dm[(size_t)env * A + i] * two_over_lambda : 0.0;
  const float ar = (float)a64;
  act_rev[(size_t)i * Bp + env] = ar;
  store_act16(act16, env, i, A_pad, ar);
  if (act16_ll && !(A_pad & 15)) {
    const float sc = ar * 256.0f;   // (as store_act16)
    const _Float16 hi = (_Float16)sc, lo = (_Float16)(sc - (float)hi);
    const int s = i >> 4, h = (i >> 3) & 1, el = i & 7, nstep = A_pad >> 4;
    act16_ll[((((size_t)(env >> 5) * nstep + s)) * 64 + (h * 32 + (env & 31))) * 8 + el] = (_Float16)(float)(a64 * 256.0 - (double)(float)hi - (double)(float)lo);
  }
}
#endif  // AOG_MAIN_TU

// AOEnv.reset bookkeeping (AO_env.py:79-83)
#ifdef AOG_MAIN_TU
__global__ void k_reset_state(const uint8_t* __restrict__ mask, double* __restrict__ act_dm, int32_t* __restrict__ t_render,
                              int B, int A, int flatten) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int env = idx / A, i = idx % A;
  if (mask && !mask[env]) return;
  if (flatten) act_dm[idx] = 0.0;
  if (i == 0) t_render[env] = 0;
}
#endif  // AOG_MAIN_TU

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
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));
// Operands are pre-scaled into the middle of the f16 range: modes (|M| <= 1) by 2^14, actuators in revolutions (|a| < 255) by 2^8;
// a low half is half(x - hi) at the same scale (<= 2^-11 |hi|, subnormal below 2^-14: absolute error <= 2^-25 there, i.e. <= 2^-39 of
// a unit mode value and <= 2^-33 revolutions of an actuator).  u = psi + 2^-22 D.
constexpr float kModeScale = 16384.0f;               // 2^14
constexpr float kActScale = 256.0f;                  // 2^8
constexpr float kPhaseUnscale = 1.0f / (16384.0f * 256.0f);        // 2^-22

// x (already multiplied by its operand scale) -> hi + lo
__device__ __host__ inline void split_f16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

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

// compile-time loops (indices usable as template arguments)
template <int K>
struct IC { static constexpr int v = K; };
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// The 8-table variant of k_fused_tab folds its fp32 table sums into float64 every kFlushTiles tiles (32 terms per tile and accumulator
// element); the many-table variants run fp32 over a chunk of at most kTabF32Tiles tiles.
constexpr int kFlushTiles = 4;
constexpr int kTabF32Tiles = 13;
constexpr int kSkewNops = 150;   // x 16 cycles: start-up skew of the second workgroup of a CU (about half a stage)

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
    // the sums of these variants are fp32 anyway: the slabs are written (and read by the epilogue) as float, half the traffic
    float* out = reinterpret_cast<float*>(partials) + (size_t)chunk * NS * geo.Bp + (size_t)etile * 32 + (lane & 31);
    static_for<LIVE>([&](auto ac) {
      constexpr int a = decltype(ac)::v;
      const int m = (a & 3) + 8 * (a >> 2) + 4 * h;
      if (m < MRW) {
        out[(size_t)(2 * m) * geo.Bp] = Dc[a];
        out[(size_t)(2 * m + 1) * geo.Bp] = Ds[a];
      }
    });
    const double vc = acc_sc + __shfl_down(acc_sc, 32, 64), vs = acc_ss + __shfl_down(acc_ss, 32, 64);
    if (h == 0) {
      out[(size_t)(2 * MRW) * geo.Bp] = (float)vc;
      out[(size_t)(2 * MRW + 1) * geo.Bp] = (float)vs;
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
constexpr float kShOutside = 2.0f;   // phase-grid value of a pixel outside the aperture (reduced phases lie in [-1/2, 1/2])
struct PhaseFieldArgs {
  const int32_t* ap_yx;      // [n_ap] iy << 16 | ix
  const float2* mla32;       // [N*N] micro-lens phase factor, complex64
  const float* mla_rev;      // [n_ap] its argument in revolutions per packed aperture pixel (GRID form)
  const f16x8* act_ll;       // nullable (K4): third f16 term of the actuators, [env tile][A_pad / 16][64][8] (see k_load_actuators)
  float2* field;
  size_t env_stride;
  int row_stride, n_ap, B, N;
  float amplitude;
};
// GRID (with FIELD): only a phase leaves this kernel — w = u + (micro-lens phase of the pixel), reduced to [-1/2, 1/2] revolutions, as ONE
// float at (iy, ix) — and the first propagation pass forms E = amplitude e^{2 pi i w} itself while it loads: 4 bytes written and read per
// pixel instead of 8, one load per pixel as before.  Pixels outside the aperture hold kShOutside (written once at upload): field 0.
// (A first form kept the micro-lens factor as a complex table multiplied in by the pass: its second load per pixel cost the pass 1.4 ms.)
template <int A_PAD, bool FIELD = false, bool GRID = false>
__global__ __launch_bounds__(256) void k_phase_mfma(const f16x8* __restrict__ modes16, const f32x4* __restrict__ psi_tile,
                                                    const f16x8* __restrict__ act16, f32x4* __restrict__ out_tile, int n_ptiles,
                                                    int n_etiles, PhaseFieldArgs fa = PhaseFieldArgs{}) {
  constexpr int NSTEP = A_PAD / 16;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int etile = blockIdx.y;
  __shared__ float2 field_lds[(FIELD && !GRID) ? 4 * 32 * 33 : 1];
  __shared__ float grid_lds[GRID ? 4 * 32 * 33 : 1];   // (GRID: one float per pixel — half the LDS, twice the workgroups per CU)
  [[maybe_unused]] float2* field_tile = field_lds + ((FIELD && !GRID) ? (threadIdx.x >> 6) * 32 * 33 : 0);
  [[maybe_unused]] float* grid_tile = grid_lds + (GRID ? (threadIdx.x >> 6) * 32 * 33 : 0);
  if (t >= n_ptiles || etile >= n_etiles) return;
  const f16x8* asrc = act16 + ((size_t)etile * NSTEP * 2) * 64 + lane;
  const f16x8* ms = modes16 + ((size_t)t * NSTEP * 2) * 64 + lane;
  f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const f16x8 mh = ms[(2 * s) * 64], ml = ms[(2 * s + 1) * 64], bh = asrc[(2 * s) * 64], bl = asrc[(2 * s + 1) * 64];
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, bh, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, bl, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml, bh, d, 0, 0, 0);
    if constexpr (GRID) {
      // K4: the actuators to 33 bits.  A rounding error of an ACTUATOR is a smooth phase error over the whole pupil — it does not average
      // down over the pixels like the per-pixel rounding of a mode value does — and at 2^-23 of an actuator of half a revolution it was
      // most of the error of the focal fields (7e-8 of the peak amplitude, the whole tolerance of a pixel 30 dB down)
      if (fa.act_ll) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, fa.act_ll[((size_t)etile * NSTEP + s) * 64 + lane], d, 0, 0, 0);
    }
  }
  const size_t base = (((size_t)etile * n_ptiles + t) * 4) * 64 + lane;
  [[maybe_unused]] const int h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 p = psi_tile[base + g * 64];
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
    const int q = lane & 31, pix = t * 32 + q;
    if (pix < fa.n_ap) {
      const int yx = fa.ap_yx[pix];
      const size_t at = (size_t)(yx >> 16) * fa.row_stride + (yx & 0xffff);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int el = 2 * j + (lane >> 5), env_j = etile * 32 + el;
        if (env_j < fa.B) {
          if constexpr (GRID) reinterpret_cast<float*>(fa.field)[(size_t)env_j * fa.env_stride + at] = grid_tile[el * 33 + q];
          else fa.field[(size_t)env_j * fa.env_stride + at] = field_tile[el * 33 + q];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K3c  float64 validation form (AOG_PRECISION_FP64): one workgroup per env, everything in float64 from
// float64 tables; also the general path for shapes the fast kernels are not instantiated for.
// ------------------------------------------------------------------------------------------------
constexpr int kRefMaxSums = 2 * 80;

#ifdef AOG_MAIN_TU
__global__ __launch_bounds__(256) void k_fused_ref(const double* __restrict__ modes64, const double* __restrict__ tabs64,
                                                   const double* __restrict__ psi64, const double* __restrict__ act_dm,
                                                   double* __restrict__ partials, int n_ap, int A, int MRW, int MRS,
                                                   int Bp, double lambda_wfs, double lambda_sci) {
  __shared__ double sm[8];
  __shared__ double sa[256];
  const int env = blockIdx.x;
  for (int i = threadIdx.x; i < A; i += blockDim.x) sa[i] = act_dm[(size_t)env * A + i];
  __syncthreads();
  const int MR = MRW + MRS;
  const int NS = 2 * MR;
  double acc[kRefMaxSums];
  for (int i = 0; i < NS; ++i) acc[i] = 0;
  for (int p = threadIdx.x; p < n_ap; p += blockDim.x) {
    const double* mrow = modes64 + (size_t)p * A;
    double surf = 0;
    for (int k = 0; k < A; ++k) surf = fma(mrow[k], sa[k], surf);
    const double theta = psi64[(size_t)env * n_ap + p] + 4.0 * M_PI * surf;  // achromatic phase (rad * m)
    double sw, cw, ss, cs;
    sincos(theta / lambda_wfs, &sw, &cw);
    sincos(theta / lambda_sci, &ss, &cs);
    const double* trow = tabs64 + (size_t)p * MR;
    for (int m = 0; m < MRW; ++m) {
      acc[2 * m] += cw * trow[m];
      acc[2 * m + 1] += sw * trow[m];
    }
    for (int m = MRW; m < MR; ++m) {
      acc[2 * m] += cs * trow[m];
      acc[2 * m + 1] += ss * trow[m];
    }
  }
  for (int i = 0; i < NS; ++i) {
    const double v = block_reduce_sum(acc[i], sm);
    if (threadIdx.x == 0) partials[(size_t)i * Bp + env] = v;
  }
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K9  epilogue: chunk partials -> complex amplitudes -> observation, fiber power, Strehl, reward, done.
// (AO_env.py:142-153, 468-503.)  One thread per env.
// ------------------------------------------------------------------------------------------------
struct EpilogueArgs {
  const double* partials;
  const double* wfs_coef;  // [n_out][MRW][2]
  const double* sci_coef;  // [MRS][2]
  float* obs_raw;
  uint16_t* obs;
  float* reward;
  uint8_t* done;
  float* power;
  float* strehl;
  int32_t* t_render;
  float* ret_acc;   // nullable: episode-return accumulator [B] (aog_set_return_accumulator)
  int B, Bp, n_chunks, MRW, MRS, MRW_used, MRS_used, n_obs, n_fiber, reward_type, has_thr, max_steps, is_step;
  int partials_f32;   // slabs hold float (table-MFMA variants with fp32-only sums) instead of double
  double thr, ssim_peak, ssim_alpha;
};

__device__ inline double ssim_1d_delta_ref(const double* x, int stride, int n, double peak, int peak_idx) {
  // skimage.metrics.structural_similarity, 1-D, win 7, uniform filter, sample covariance; the reference image
  // is peak at peak_idx and 0 elsewhere (AO_env.py:491-495).  Mean over the interior windows.
  const double C1 = (0.01 * peak) * (0.01 * peak), C2 = (0.03 * peak) * (0.03 * peak);
  const double cov_norm = 7.0 / 6.0;
  double sum = 0;
  int cnt = 0;
  for (int i = 3; i < n - 3; ++i) {
    double ux = 0, uxx = 0, uy = 0, uyy = 0, uxy = 0;
    for (int k = -3; k <= 3; ++k) {
      const double a = x[(size_t)(i + k) * stride];
      const double b = (i + k == peak_idx) ? peak : 0.0;
      ux += a; uxx += a * a; uy += b; uyy += b * b; uxy += a * b;
    }
    ux /= 7; uxx /= 7; uy /= 7; uyy /= 7; uxy /= 7;
    const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    sum += ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    ++cnt;
  }
  return sum / cnt;
}

// Pre-reduction of float slabs (table-MFMA variants write many short chunks): out[s][env] = sum_c part[c][s][env], one thread per
// (s, env), envs along the lanes (256-B rows), eight independent loads in flight.  The epilogue then sees a single float64 slab.
#ifdef AOG_MAIN_TU
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ part, double* __restrict__ out, int n_chunks, int NS, int Bp) {
  const int env = blockIdx.x * 64 + (threadIdx.x & 63);
  const int s = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (env >= Bp || s >= NS) return;
  const size_t cstride = (size_t)NS * Bp;
  const float* src = part + (size_t)s * Bp + env;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int c = 0;
  for (; c + 7 < n_chunks; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += (double)src[(size_t)(c + u) * cstride];
  }
  for (; c < n_chunks; ++c) a[0] += (double)src[(size_t)c * cstride];
  out[(size_t)s * Bp + env] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
#endif  // AOG_MAIN_TU

// block = 16 envs x 16 sum slots x 4 chunk groups (1024 threads, grid = Bp / 16: 64 workgroups at B = 1024):
// thread (e, q, cq) adds sums s = q, q + 16, ... over chunks cq, cq + 16, ... (independent loads in flight); the chunk groups meet
// in LDS; then one thread per (env, output) forms |coef . sums|^2, and one thread per env finishes reward / done / power.
// dynamic LDS: see epilogue_lds_bytes().
constexpr int kEpiEnvs = 16;     // 16 envs x 8 B = one 128-byte line of a slab row per (chunk, sum): round 1's 4 envs fetched 32-byte pieces
constexpr int kEpiGroups = 4;
constexpr int kEpiOutSlots = 16;   // threads per env in the output phase
__host__ __device__ inline size_t epilogue_lds_bytes(int NS, int n_obs, int n_fiber, int MRW_used, int MRS_used) {
  return ((size_t)kEpiGroups * NS * kEpiEnvs + (size_t)NS * kEpiEnvs + (size_t)(n_obs + n_fiber + 1) * kEpiEnvs +
          (size_t)(n_obs + n_fiber) * MRW_used * 2 + (size_t)MRS_used * 2) * sizeof(double);
}
#ifdef AOG_MAIN_TU
__device__ __forceinline__ void epilogue_body(const EpilogueArgs& p, int block, double* __restrict__ sm) {
  const int e = threadIdx.x & (kEpiEnvs - 1);
  const int q = (threadIdx.x / kEpiEnvs) & 15;            // sum slot
  const int cq = threadIdx.x / (kEpiEnvs * 16);           // chunk group (= wave index)
  const int env = block * kEpiEnvs + e;              // < Bp: padded envs read defined (ignored) slabs
  const int MR = p.MRW + p.MRS;
  const int NS = 2 * MR;
  const int n_out = p.n_obs + p.n_fiber;
  const size_t cstride = (size_t)NS * p.Bp;
  double* part = sm;                                              // [group][NS][4]
  double* U = part + (size_t)kEpiGroups * NS * kEpiEnvs;          // [NS][4]: U_m = U[(2m) * 4 + e], V_m = U[(2m + 1) * 4 + e]
  double* pw = U + (size_t)NS * kEpiEnvs;                         // [n_out + 1][4]: powers of the outputs, then Strehl
  double* cfs = pw + (size_t)(n_out + 1) * kEpiEnvs;              // [n_out][MRW_used][2] then [MRS_used][2]
  double* cfsci = cfs + (size_t)n_out * p.MRW_used * 2;
  // the per-env state the last phase updates is requested now (it would otherwise be one more memory round trip at the very end)
  int tr_prev = 0;
  float ret_prev = 0.f;
  if (threadIdx.x < kEpiEnvs && env < p.B && p.is_step) {
    tr_prev = p.t_render[env];
    if (p.ret_acc) ret_prev = p.ret_acc[env];
  }
  // the small coefficient matrices go to LDS once (the output threads would otherwise chase them through L2 serially)
  for (int i = threadIdx.x; i < n_out * p.MRW_used * 2; i += blockDim.x) cfs[i] = p.wfs_coef[i];
  for (int i = threadIdx.x; i < p.MRS_used * 2; i += blockDim.x) cfsci[i] = p.sci_coef[i];
  for (int s = q; s < NS; s += 16) {
    double a[4] = {0, 0, 0, 0};
    int c = cq;
    if (p.partials_f32) {
      const float* src = reinterpret_cast<const float*>(p.partials) + (size_t)s * p.Bp + env;
      for (; c + 3 * kEpiGroups < p.n_chunks; c += 4 * kEpiGroups) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += (double)src[(size_t)(c + kEpiGroups * u) * cstride];
      }
      for (; c < p.n_chunks; c += kEpiGroups) a[0] += (double)src[(size_t)c * cstride];
    } else {
      const double* src = p.partials + (size_t)s * p.Bp + env;
      for (; c + 3 * kEpiGroups < p.n_chunks; c += 4 * kEpiGroups) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += src[(size_t)(c + kEpiGroups * u) * cstride];
      }
      for (; c < p.n_chunks; c += kEpiGroups) a[0] += src[(size_t)c * cstride];
    }
    part[((size_t)cq * NS + s) * kEpiEnvs + e] = (a[0] + a[1]) + (a[2] + a[3]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NS * kEpiEnvs; i += blockDim.x) {
    double v = 0;
#pragma unroll
    for (int g = 0; g < kEpiGroups; ++g) v += part[(size_t)g * NS * kEpiEnvs + i];
    U[i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kEpiEnvs * kEpiOutSlots) {
    const int oe = threadIdx.x & (kEpiEnvs - 1), slot = threadIdx.x / kEpiEnvs;
    const int oenv = block * kEpiEnvs + oe;
    for (int j = slot; j <= n_out; j += kEpiOutSlots) {
      double zr = 0, zi = 0;
      if (j < n_out) {
        const double* cf = cfs + (size_t)j * p.MRW_used * 2;
        for (int m = 0; m < p.MRW_used; ++m) {
          const double u = U[(2 * m) * kEpiEnvs + oe], v = U[(2 * m + 1) * kEpiEnvs + oe];
          zr += cf[2 * m] * u - cf[2 * m + 1] * v;
          zi += cf[2 * m] * v + cf[2 * m + 1] * u;
        }
      } else {
        for (int m = 0; m < p.MRS_used; ++m) {
          const double u = U[(2 * (p.MRW + m)) * kEpiEnvs + oe], v = U[(2 * (p.MRW + m) + 1) * kEpiEnvs + oe];
          zr += cfsci[2 * m] * u - cfsci[2 * m + 1] * v;
          zi += cfsci[2 * m] * v + cfsci[2 * m + 1] * u;
        }
      }
      const double w = zr * zr + zi * zi;
      pw[(size_t)j * kEpiEnvs + oe] = w;
      if (j < p.n_obs && oenv < p.B) {
        if (p.obs_raw) p.obs_raw[(size_t)oenv * p.n_obs + j] = (float)w;
        if (p.obs) {
          const _Float16 hv = (_Float16)w;  // round-to-nearest-even from float64, like np.array(x, float16)
          p.obs[(size_t)oenv * p.n_obs + j] = *reinterpret_cast<const uint16_t*>(&hv);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x >= kEpiEnvs || env >= p.B || !p.is_step) return;
  double power = 0;
  for (int j = p.n_obs; j < n_out; ++j) power += pw[(size_t)j * kEpiEnvs + e];
  const double strehl = pw[(size_t)n_out * kEpiEnvs + e];
  double reward;
  if (p.reward_type == 0) {
    reward = -(100.0 - strehl * 100.0);
  } else {
    const double ssim = ssim_1d_delta_ref(pw + e, kEpiEnvs, p.n_obs, p.ssim_peak, p.n_obs / 2);
    reward = p.ssim_alpha * power + (1.0 - p.ssim_alpha) * ssim;
  }
  if (p.has_thr && reward < p.thr) reward = -1.0;
  const int tr = tr_prev + 1;
  p.t_render[env] = tr;
  if (p.reward) p.reward[env] = (float)reward;
  if (p.ret_acc) p.ret_acc[env] = ret_prev + (float)reward;
  if (p.done) p.done[env] = (tr == p.max_steps) ? 1 : 0;
  if (p.power) p.power[env] = (float)power;
  if (p.strehl) p.strehl[env] = (float)strehl;
}
__global__ __launch_bounds__(1024) void k_epilogue(EpilogueArgs p) {
  extern __shared__ double sm[];
  epilogue_body(p, (int)blockIdx.x, sm);
}
// Pipelined stepping (aog_step_pipelined): the epilogue of step t and the prologue of step t + 1 — which share nothing — in ONE launch:
// workgroups [0, n_epi) run the epilogue, the rest the prologue with one env per wave, 16 per workgroup (same arithmetic in the same order
// as the standalone kernel's four: bit-identical).  One launch and one dispatch gap less per step.
constexpr int kEpiProEnvs = 16;
struct PrologueArgs {
  const float* action;
  const double* gram;
  double* act_dm;
  float* act_rev;
  _Float16* act16;
  int B, A, A_pad, Bp, sh_operation;
  double target, two_over_lambda;
};
__global__ __launch_bounds__(1024) void k_epilogue_prologue(EpilogueArgs p, PrologueArgs q, int n_epi) {
  extern __shared__ double sm[];
  if ((int)blockIdx.x < n_epi) {
    epilogue_body(p, (int)blockIdx.x, sm);
    return;
  }
  prologue_body<true, kEpiProEnvs>(q.action, q.gram, q.act_dm, q.act_rev, q.act16, q.B, q.A, q.A_pad, q.Bp, q.sh_operation, q.target, q.two_over_lambda,
                                   (int)blockIdx.x - n_epi);
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K7  dynamic atmosphere: hcipy InfiniteAtmosphericLayer.evolve_until / _extrude (AO_env.py:125).
// One workgroup per env.  The float64 master screen is a toroidal ring buffer, so an extrusion writes N values
// instead of moving N^2:  'left'/'bottom' decrement the origin and fill logical column/row 0; 'right'/'top' (hcipy
// works on the 180-degree rotated screen) increment it and fill logical column/row N-1 in reversed order.
//   new = A z + sqrt(Cn^2) B n,   z = screen[stencil] (flat-index order, on the rotated screen when flipped),
//   n = N standard normals: caller-supplied (parity mode: numpy's stream) or Philox4x32-10 + Box-Muller.
// Matrices are stored transposed (At [nz][N], Bt [N][N]) so the N threads of a row read contiguous memory.
// ------------------------------------------------------------------------------------------------
struct ExtrudeArgs {
  float* ring;               // nullable: fp32 ring copy the fused kernel reads ([B][N][N + 4], see DynPsi); kept in step with master
  const double* ring_ref;    // [B] reference piston of the ring copy (hcipy units)
  double ring_inv;           // 1 / (2 pi lambda_wfs)
  double* master;            // [B][N*N]
  int32_t* origin;           // [B][2] (ox, oy)
  uint32_t* ext_counter;     // [B] extrusions done so far (RNG stream position)
  const double* velocity;    // [B][2] m/s
  const int32_t* stencil_v;  // [nz_v] flat logical indices
  const int32_t* stencil_h;  // [nz_h]
  const int32_t* stencil_v_yx;  // [nz_v] (sy << 16 | sx)
  const int32_t* stencil_h_yx;  // [nz_h]
  const double* At_v;        // [nz_v][N]
  const double* Bt_v;        // [N][N]
  const double* At_h;
  const double* Bt_h;
  const double* Wa_v;        // the same matrices blocked for the f64 MFMA A operand: [row block][k/8][lane][2]
  const double* Wb_v;
  const double* Wa_h;
  const double* Wb_h;
  const double* noise;       // nullable: [B][max_ext][N]
  int N, nz_v, nz_h, max_ext;
  int near_v, near_h;        // the stencils' first near_* samples lie in the two newest slices (rows / columns 0, 1), the rest further in
  double t_prev, t_new, pitch, sqrt_cn2;
  unsigned long long seed;
  int env_base;              // global id of env 0 of this handle: the Philox streams are keyed by env_base + env
};

// one new sample of env's master screen at physical (py, px): the float64 master and, when present, the fp32 ring copy (+ its duplicate
// of columns 0..3 beyond the row end)
__device__ __forceinline__ void store_master(const ExtrudeArgs& p, int env, int py, int px, double v) {
  p.master[(size_t)env * p.N * p.N + (size_t)py * p.N + px] = v;
  if (p.ring) {
    const int RS = p.N + 4;
    const float f = (float)((v - p.ring_ref[env]) * p.ring_inv);
    float* row = p.ring + ((size_t)env * p.N + py) * RS;
    row[px] = f;
    if (px < 4) row[p.N + px] = f;
  }
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// standard normal numbers 4 idx4 .. 4 idx4 + 3 of stream (seed, env, extrusion): one Philox4x32-10 call = four 32-bit words = two
// Box-Muller pairs, both the cosine and the sine branch of each used.  Hardware log/sin/cos (fp32 accuracy is ample for a noise
// sample; parity runs supply their normals from the host instead).
__device__ inline void philox_normal4(unsigned long long seed, uint32_t env, uint32_t ext, uint32_t idx4, double (&out)[4]) {
  uint32_t c[4] = {idx4, ext, env, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int rr = 0; rr < 10; ++rr) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const float u1 = ((float)(c[2 * pr] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
    const float u2 = (float)(c[2 * pr + 1] >> 8) * (1.0f / 16777216.0f);        // [0, 1) revolutions
    const float r = sqrtf(-2.0f * __logf(u1));
    out[2 * pr] = (double)(r * __builtin_amdgcn_cosf(u2));
    out[2 * pr + 1] = (double)(r * __builtin_amdgcn_sinf(u2));
  }
}
// standard normal number `idx` of the same stream (element idx & 3 of call idx >> 2)
__device__ inline double philox_normal(unsigned long long seed, uint32_t env, uint32_t ext, uint32_t idx) {
  double v[4];
  philox_normal4(seed, env, ext, idx >> 2, v);
  const double a = (idx & 2) ? v[2] : v[0], b = (idx & 2) ? v[3] : v[1];
  return (idx & 1) ? b : a;
}

#ifdef AOG_MAIN_TU
// One workgroup advances kExtG consecutive envs together.  The envs are independent, but they share the AR matrices, and
// those (2 MB per direction at N = 256) are what the kernel streams: in every round each matrix row is loaded ONCE per
// workgroup and used for all envs of the group that extrude in that direction (x shifts come first for every env, so the
// rounds of a group line up as horizontal ... horizontal, vertical ... vertical).
constexpr int kExtG = 4;
constexpr int kExtThreads = 512;  // N rows x 2 halves of the contraction index (more loads in flight per row)
__global__ __launch_bounds__(512) void k_extrude(ExtrudeArgs p, int B) {
  extern __shared__ double lds[];  // z [G][nzmax] | noise [G][N] | partial [G][N]
  const int N = p.N;
  const int nzmax = max(p.nz_v, p.nz_h);
  double* zb = lds;
  double* nb = lds + (size_t)kExtG * nzmax;
  double* pb = nb + (size_t)kExtG * N;
  __shared__ int s_ox[kExtG], s_oy[kExtG], s_dx[kExtG], s_dy[kExtG];
  const int env0 = blockIdx.x * kExtG;
  if (threadIdx.x < kExtG) {
    const int env = env0 + threadIdx.x;
    int dx = 0, dy = 0, ox = 0, oy = 0;
    if (env < B) {
      const double vx = p.velocity[2 * env], vy = p.velocity[2 * env + 1];
      // np.round(center / delta).astype(int) before and after (round-half-even = rint)
      dx = (int)rint(vx * p.t_new / p.pitch) - (int)rint(vx * p.t_prev / p.pitch);
      dy = (int)rint(vy * p.t_new / p.pitch) - (int)rint(vy * p.t_prev / p.pitch);
      ox = p.origin[2 * env];
      oy = p.origin[2 * env + 1];
    }
    s_dx[threadIdx.x] = dx; s_dy[threadIdx.x] = dy; s_ox[threadIdx.x] = ox; s_oy[threadIdx.x] = oy;
  }
  __syncthreads();
  int rounds = 0;
  for (int g = 0; g < kExtG; ++g) rounds = max(rounds, abs(s_dx[g]) + abs(s_dy[g]));
  for (int r = 0; r < rounds; ++r) {
    // class of env g this round: 1 horizontal, 2 vertical, 0 idle
    auto cls = [&](int g) { return r < abs(s_dx[g]) ? 1 : (r < abs(s_dx[g]) + abs(s_dy[g]) ? 2 : 0); };
    for (int g = 0; g < kExtG; ++g) {
      const int c = cls(g);
      if (!c) continue;
      const int env = env0 + g;
      const bool horizontal = c == 1;
      const bool flipped = horizontal ? s_dx[g] > 0 : s_dy[g] > 0;
      const int nz = horizontal ? p.nz_h : p.nz_v;
      const int32_t* st = horizontal ? p.stencil_h : p.stencil_v;
      const double* master = p.master + (size_t)env * N * N;
      const int ox = s_ox[g], oy = s_oy[g];
      for (int k = threadIdx.x; k < nz; k += blockDim.x) {
        int sy = st[k] / N, sx = st[k] - sy * N;
        if (flipped) { sy = N - 1 - sy; sx = N - 1 - sx; }
        int py = sy + oy, px = sx + ox;
        if (py >= N) py -= N;
        if (px >= N) px -= N;
        zb[(size_t)g * nzmax + k] = master[(size_t)py * N + px];
      }
      const uint32_t ext = p.ext_counter[env] + (uint32_t)r;
      for (int j = threadIdx.x; j < N; j += blockDim.x)
        nb[(size_t)g * N + j] = (p.noise && r < p.max_ext) ? p.noise[((size_t)env * p.max_ext + r) * N + j]
                                                           : philox_normal(p.seed, (uint32_t)(p.env_base + env), ext, (uint32_t)j);
    }
    __syncthreads();
    // thread (row i, half kh): rows i = tid % N (+ strides), kh = tid / N in {0, 1} sums one half of the stencil / noise
    // index; the halves meet in LDS.  (N <= 256 rows per pass; larger N loops.)
    const int half = blockDim.x >> 1;
    const int kh = threadIdx.x >= half ? 1 : 0;
    for (int i0 = 0; i0 < N; i0 += half) {
      const int i = i0 + (threadIdx.x - kh * half);
      double out[kExtG];
#pragma unroll
      for (int g = 0; g < kExtG; ++g) out[g] = 0.0;
      if (i < N) {
        for (int c = 1; c <= 2; ++c) {
          bool any = false;
          for (int g = 0; g < kExtG; ++g) any |= cls(g) == c;
          if (!any) continue;
          const int nz = c == 1 ? p.nz_h : p.nz_v;
          const double* At = c == 1 ? p.At_h : p.At_v;
          const double* Bt = c == 1 ? p.Bt_h : p.Bt_v;
          double a[kExtG], b[kExtG];
#pragma unroll
          for (int g = 0; g < kExtG; ++g) { a[g] = 0.0; b[g] = 0.0; }
          const int k0 = kh ? (nz + 1) / 2 : 0, k1 = kh ? nz : (nz + 1) / 2;
#pragma unroll 8
          for (int k = k0; k < k1; ++k) {
            const double w = At[(size_t)k * N + i];
#pragma unroll
            for (int g = 0; g < kExtG; ++g) a[g] = fma(w, zb[(size_t)g * nzmax + k], a[g]);
          }
          const int j0 = kh ? (N + 1) / 2 : 0, j1 = kh ? N : (N + 1) / 2;
#pragma unroll 8
          for (int j = j0; j < j1; ++j) {
            const double w = Bt[(size_t)j * N + i];
#pragma unroll
            for (int g = 0; g < kExtG; ++g) b[g] = fma(w, nb[(size_t)g * N + j], b[g]);
          }
#pragma unroll
          for (int g = 0; g < kExtG; ++g)
            if (cls(g) == c) out[g] = a[g] + b[g] * p.sqrt_cn2;
        }
        if (kh) {
#pragma unroll
          for (int g = 0; g < kExtG; ++g) pb[(size_t)g * N + i] = out[g];
        }
      }
      __syncthreads();
      if (i < N && !kh) {
#pragma unroll
        for (int g = 0; g < kExtG; ++g) {
          const int c = cls(g);
          if (!c) continue;
          const double v = out[g] + pb[(size_t)g * N + i];
          const bool horizontal = c == 1;
          const bool flipped = horizontal ? s_dx[g] > 0 : s_dy[g] > 0;
          int nox = s_ox[g], noy = s_oy[g];
          if (horizontal) nox = flipped ? (nox + 1 == N ? 0 : nox + 1) : (nox == 0 ? N - 1 : nox - 1);
          else noy = flipped ? (noy + 1 == N ? 0 : noy + 1) : (noy == 0 ? N - 1 : noy - 1);
          int ly, lx;
          if (horizontal) { ly = flipped ? N - 1 - i : i; lx = flipped ? N - 1 : 0; }
          else { ly = flipped ? N - 1 : 0; lx = flipped ? N - 1 - i : i; }
          int py = ly + noy, px = lx + nox;
          if (py >= N) py -= N;
          if (px >= N) px -= N;
          store_master(p, env0 + g, py, px, v);
        }
      }
      __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x < kExtG) {
      const int g = threadIdx.x;
      const int c = cls(g);
      if (c == 1) s_ox[g] = s_dx[g] > 0 ? (s_ox[g] + 1 == N ? 0 : s_ox[g] + 1) : (s_ox[g] == 0 ? N - 1 : s_ox[g] - 1);
      else if (c == 2) s_oy[g] = s_dy[g] > 0 ? (s_oy[g] + 1 == N ? 0 : s_oy[g] + 1) : (s_oy[g] == 0 ? N - 1 : s_oy[g] - 1);
    }
    __syncthreads();
  }
  if (threadIdx.x < kExtG && env0 + threadIdx.x < B) {
    const int g = threadIdx.x, env = env0 + g;
    p.origin[2 * env] = s_ox[g];
    p.origin[2 * env + 1] = s_oy[g];
    p.ext_counter[env] += (uint32_t)(abs(s_dx[g]) + abs(s_dy[g]));
  }
}

// ---- float64 matrix-core form: 16 envs per workgroup ---------------------------------------------------------------------
// Same algorithm as k_extrude with the two contractions on v_mfma_f64_16x16x4_f64: D[16 rows][16 envs] += A[16 rows][4 k] B[4 k][16 envs],
// A = transposed AR matrix rows straight from L2 (lane (row l&15, k l>>4)), B = stencil values / normals from LDS (lane (env l&15, k l>>4)).
// C/D map of the f64 instruction: col = lane & 15, row = (lane >> 4) + 4 * reg.  Every matrix element is streamed once per 16 envs.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kExt16G = 16;
#ifdef AOG_MAIN_TU
__global__ __launch_bounds__(512) void k_extrude16(ExtrudeArgs p, int B) {
  extern __shared__ double lds[];  // z [16][zs] | noise [16][ns]
  constexpr int G = kExt16G;
  const int N = p.N;
  const int nzmax = max(p.nz_v, p.nz_h);
  const int zs = nzmax | 1, ns = N | 1;   // odd strides: the 16 env rows fall on different LDS banks
  double* zb = lds;
  double* nb = lds + (size_t)G * zs;
  __shared__ int s_ox[G], s_oy[G], s_dx[G], s_dy[G];
  const int env0 = blockIdx.x * G;
  if (threadIdx.x < G) {
    const int env = env0 + threadIdx.x;
    int dx = 0, dy = 0, ox = 0, oy = 0;
    if (env < B) {
      const double vx = p.velocity[2 * env], vy = p.velocity[2 * env + 1];
      dx = (int)rint(vx * p.t_new / p.pitch) - (int)rint(vx * p.t_prev / p.pitch);
      dy = (int)rint(vy * p.t_new / p.pitch) - (int)rint(vy * p.t_prev / p.pitch);
      ox = p.origin[2 * env];
      oy = p.origin[2 * env + 1];
    }
    s_dx[threadIdx.x] = dx; s_dy[threadIdx.x] = dy; s_ox[threadIdx.x] = ox; s_oy[threadIdx.x] = oy;
  }
  __syncthreads();
  int rounds = 0;
  for (int g = 0; g < G; ++g) rounds = max(rounds, abs(s_dx[g]) + abs(s_dy[g]));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  for (int r = 0; r < rounds; ++r) {
    auto cls = [&](int g) { return r < abs(s_dx[g]) ? 1 : (r < abs(s_dx[g]) + abs(s_dy[g]) ? 2 : 0); };
    // gather the stencil samples of all 16 envs in ONE flattened loop (env-major pairs, 4 independent loads in flight per
    // thread); stencil coordinates come pre-split (sy << 16 | sx) so no integer division sits in front of the loads
    for (int base = threadIdx.x; base < G * nzmax; base += 4 * blockDim.x) {
      double v[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * blockDim.x;
        const int g = idx / nzmax, k = idx - g * nzmax;
        dst[u] = -1;
        v[u] = 0.0;
        if (idx < G * nzmax) {
          const int c = cls(g);
          const bool horizontal = c == 1;
          const int nz = horizontal ? p.nz_h : p.nz_v;
          if (c && k < nz) {
            const uint32_t pk = (uint32_t)(horizontal ? p.stencil_h_yx : p.stencil_v_yx)[k];
            int sy = (int)(pk >> 16), sx = (int)(pk & 0xFFFFu);
            if (horizontal ? s_dx[g] > 0 : s_dy[g] > 0) { sy = N - 1 - sy; sx = N - 1 - sx; }
            int py = sy + s_oy[g], px = sx + s_ox[g];
            if (py >= N) py -= N;
            if (px >= N) px -= N;
            v[u] = p.master[(size_t)(env0 + g) * N * N + (size_t)py * N + px];
            dst[u] = g * zs + k;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) zb[dst[u]] = v[u];
    }
    for (int idx = threadIdx.x; idx < G * N; idx += blockDim.x) {
      const int g = idx / N, j = idx - g * N;
      if (!cls(g)) continue;
      const int env = env0 + g;
      nb[(size_t)g * ns + j] = (p.noise && r < p.max_ext) ? p.noise[((size_t)env * p.max_ext + r) * N + j]
                                                          : philox_normal(p.seed, (uint32_t)(p.env_base + env), p.ext_counter[env] + (uint32_t)r, (uint32_t)j);
    }
    __syncthreads();
    const int my_cls = cls(li);     // class of the env this lane feeds as the B operand / owns as the D column
    for (int c = 1; c <= 2; ++c) {
      bool any = false;
      for (int g = 0; g < G; ++g) any |= cls(g) == c;
      if (!any) continue;
      const bool horizontal = c == 1;
      const int nz = horizontal ? p.nz_h : p.nz_v;
      const double* At = horizontal ? p.At_h : p.At_v;
      const double* Bt = horizontal ? p.Bt_h : p.Bt_v;
      const bool feed = my_cls == c;
      for (int rb = wave; rb * 16 < N; rb += nwaves) {
        const int row = rb * 16 + li;
        const bool row_ok = row < N;
        f64x4 accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
        // software pipeline: the 8 matrix loads of a 32-deep chunk are issued unconditionally (clamped index, masked by a
        // multiplier) before any is consumed, so 8 L2 round trips overlap instead of serialising behind per-element branches
        const int rowc = row_ok ? row : 0;
        const double rmask = row_ok ? 1.0 : 0.0;
        const double* zrow = zb + (size_t)li * zs;
        const double* nrow = nb + (size_t)li * ns;
        // two register sets: the loads of chunk n+1 are in flight while the 8 matrix instructions of chunk n issue
        auto load_chunk = [&](const double* __restrict__ W, const double* __restrict__ vec, int K, int k0, double (&av)[8], double (&bv)[8]) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + lk;
            const int kc = min(k, K - 1);
            av[u] = W[(size_t)kc * N + rowc];
            bv[u] = (k < K && feed) ? vec[kc] : 0.0;   // beyond K the B operand is zero: surplus chunks add nothing
          }
        };
        auto run = [&](const double* __restrict__ W, const double* __restrict__ vec, int K, f64x4& acc) {
          double a0[8], b0[8], a1[8], b1[8];
          load_chunk(W, vec, K, 0, a0, b0);
          for (int k0 = 0; k0 < K; k0 += 64) {
            load_chunk(W, vec, K, k0 + 32, a1, b1);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u] * rmask, b0[u], acc, 0, 0, 0);
            load_chunk(W, vec, K, k0 + 64, a0, b0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u] * rmask, b1[u], acc, 0, 0, 0);
          }
        };
        run(At, zrow, nz, accA);
        run(Bt, nrow, N, accB);
        // this lane holds column (env) li, rows rb*16 + lk + 4*q
        if (feed) {
          const int g = li;
          const bool flipped = horizontal ? s_dx[g] > 0 : s_dy[g] > 0;
          int nox = s_ox[g], noy = s_oy[g];
          if (horizontal) nox = flipped ? (nox + 1 == N ? 0 : nox + 1) : (nox == 0 ? N - 1 : nox - 1);
          else noy = flipped ? (noy + 1 == N ? 0 : noy + 1) : (noy == 0 ? N - 1 : noy - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = rb * 16 + lk + 4 * q;
            if (i >= N) continue;
            const double v = accA[q] + accB[q] * p.sqrt_cn2;
            int ly, lx;
            if (horizontal) { ly = flipped ? N - 1 - i : i; lx = flipped ? N - 1 : 0; }
            else { ly = flipped ? N - 1 : 0; lx = flipped ? N - 1 - i : i; }
            int py = ly + noy, px = lx + nox;
            if (py >= N) py -= N;
            if (px >= N) px -= N;
            store_master(p, env0 + g, py, px, v);
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < G) {
      const int g = threadIdx.x;
      const int c = cls(g);
      if (c == 1) s_ox[g] = s_dx[g] > 0 ? (s_ox[g] + 1 == N ? 0 : s_ox[g] + 1) : (s_ox[g] == 0 ? N - 1 : s_ox[g] - 1);
      else if (c == 2) s_oy[g] = s_dy[g] > 0 ? (s_oy[g] + 1 == N ? 0 : s_oy[g] + 1) : (s_oy[g] == 0 ? N - 1 : s_oy[g] - 1);
    }
    __syncthreads();
  }
  if (threadIdx.x < G && env0 + threadIdx.x < B) {
    const int g = threadIdx.x, env = env0 + g;
    p.origin[2 * env] = s_ox[g];
    p.origin[2 * env + 1] = s_oy[g];
    p.ext_counter[env] += (uint32_t)(abs(s_dx[g]) + abs(s_dy[g]));
  }
}
#endif  // AOG_MAIN_TU

// ---- same matrix-core extrusion with the rows of a 16-env group split over FOUR workgroups (all 256 CUs at B = 1024) ------
// The four workgroups of a group gather the same stencil samples, each computes a quarter of the new slice's row blocks and
// writes it in place; before the next round reads those rows they meet at a group barrier: plain stores -> every wave
// s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: agent-scope release fence, ticket add on the group's counter, relaxed poll
// (bounded) until all four tickets of this round are in, agent-scope acquire fence -> __syncthreads (cdna_hip_programming.md
// Guideline 16, counter form; from the second round on the same-XCD short form when the group has measured that it may: see the barrier).  Two sets of counters alternate between launches; a launch zeroes the set of the next one.  Workgroup L sits on XCD L % 8; the map
// below keeps a group's four workgroups on one XCD (speed only).  A timed-out spin sets *status and *host_flag (pinned host memory the
// library polls without synchronising) and the kernel still terminates; aog_step / aog_reset then fail with AOG_ERR_STATE.
constexpr int kExtParts = 4;
__host__ __device__ inline int ext_split_stride(int n) { return ((n + 27) / 32) * 32 + 4; }   // smallest s >= n, s = 4 mod 32
constexpr int kExtKs = 2;   // slices of the contraction per row block (template parameter KS: 4 KS waves per workgroup; 1 and 4 measured slower)
#ifdef AOG_MAIN_TU
template <int KS>
__global__ __launch_bounds__(256 * KS) void k_extrude16_split(ExtrudeArgs p, int B, const int* __restrict__ perm, unsigned* __restrict__ bar, int* __restrict__ status,
                                                              int* __restrict__ host_flag, int group0, unsigned spin_limit, int absent_part,
                                                              unsigned* __restrict__ bar_next, int force_agent_scope) {
  // force_agent_scope: never take the same-XCD form of the group barrier (AOG_EXTRUDE_AGENT_SCOPE: tests, measurements).
  // group0: first group of this launch (a batch whose groups x 4 workgroups exceed what the chip holds at once is extruded in several
  // launches: barrier partners must be co-resident).  spin_limit / absent_part: see aog_selftest_barrier_timeout (product launches pass
  // 1 << 24 and -1).
  extern __shared__ double lds[];  // z [16][zs] | noise [16][ns] | partial sums [KS-1][4][256]
  constexpr int G = kExt16G;
  const int N = p.N;
  const int nzmax = max(p.nz_v, p.nz_h);
  // row strides = 4 mod 32 doubles: the B-operand read (lane = 16 k + env, 8 B) then spreads over all banks (an odd stride
  // puts env + k on the same bank pair: 4-way conflicts, as expensive as the matrix passes themselves)
  const int zs = ext_split_stride(nzmax), ns = ext_split_stride(N);
  double* zb = lds;
  double* nb = lds + (size_t)G * zs;
  double* pb = nb + (size_t)G * ns;
  int32_t* st_v = reinterpret_cast<int32_t*>(pb + (size_t)(KS - 1) * 4 * 256);   // stencil codes (sy << 16 | sx), staged once
  int32_t* st_h = st_v + p.nz_v;
  __shared__ int s_ox[G], s_oy[G], s_dx[G], s_dy[G], s_env[G];
  __shared__ int s_same_xcd;
  const int L = blockIdx.x;
  const int part = (L >> 3) & (kExtParts - 1);
  // groups are sorted by wind (aog_set_wind): an XCD takes a contiguous run of them, so its workgroups want the same class of
  // matrices at the same time
  const int groups_per_xcd = (int)gridDim.x >> 5;
  const int group = group0 + (L & 7) * groups_per_xcd + (L >> 5);
  const int env0 = group * G;
  if (env0 >= B) return;   // whole groups only: no barrier partner is left waiting
  // the tickets of the NEXT launch live in the other half of the ticket array: zeroed here, by one lane per group (no zero-fill launch per step)
  if (part == 0 && threadIdx.x == 0) bar_next[group] = 0u;
  if (part == absent_part) return;   // (self-test: this group's partners wait for a ticket that never comes)
  for (int i = threadIdx.x; i < p.nz_v; i += blockDim.x) st_v[i] = p.stencil_v_yx[i];
  for (int i = threadIdx.x; i < p.nz_h; i += blockDim.x) st_h[i] = p.stencil_h_yx[i];
  if (threadIdx.x < G) {
    const int env = perm[env0 + threadIdx.x];   // slot -> env id (envs of similar wind share a group); -1 = padding slot
    int dx = 0, dy = 0, ox = 0, oy = 0;
    s_env[threadIdx.x] = max(env, 0);
    if (env >= 0) {
      const double vx = p.velocity[2 * env], vy = p.velocity[2 * env + 1];
      dx = (int)rint(vx * p.t_new / p.pitch) - (int)rint(vx * p.t_prev / p.pitch);
      dy = (int)rint(vy * p.t_new / p.pitch) - (int)rint(vy * p.t_prev / p.pitch);
      ox = p.origin[2 * env];
      oy = p.origin[2 * env + 1];
    }
    s_dx[threadIdx.x] = dx; s_dy[threadIdx.x] = dy; s_ox[threadIdx.x] = ox; s_oy[threadIdx.x] = oy;
  }
  __syncthreads();
  int rounds = 0;
  for (int g = 0; g < G; ++g) rounds = max(rounds, abs(s_dx[g]) + abs(s_dy[g]));
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int rbl = wave & 3, ks = wave >> 2;   // 4 KS waves: 4 row blocks x KS slices of the contraction
  const int li = lane & 15, lk = lane >> 4;
  const bool dbg = status[1] != 0 && blockIdx.x == 0 && threadIdx.x == 0;
  long long tm[6] = {0, 0, 0, 0, 0, 0};
  int n_pass = 0;
  long long cyc = 0;
  constexpr int GD = 12;
  const int n_waves = (int)(blockDim.x >> 6);
  // samples [k_lo, k_hi) of env slot g's stencil (class c: 1 = 'left' stencil, x extrusion; 2 = 'bottom', y) at origin (ox, oy) -> zb
  auto gather_env = [&](int g, int c, int k_lo, int k_hi, int ox, int oy) {
    const bool horizontal = c == 1;
    const int32_t* st = horizontal ? st_h : st_v;
    const bool flipped = __builtin_amdgcn_readfirstlane(horizontal ? s_dx[g] : s_dy[g]) > 0;
    const double* __restrict__ src = p.master + (size_t)__builtin_amdgcn_readfirstlane(s_env[g]) * N * N;
    double* zrow_g = zb + (size_t)g * zs;
    for (int k0 = k_lo; k0 < k_hi; k0 += 64 * GD) {
      double v[GD];
#pragma unroll
      for (int u = 0; u < GD; ++u) {
        const int k = min(k0 + 64 * u + lane, k_hi - 1);   // branch-free: every lane loads from a valid address
        const uint32_t pk = (uint32_t)st[k];
        int sy = (int)(pk >> 16), sx = (int)(pk & 0xFFFFu);
        sy = flipped ? N - 1 - sy : sy;
        sx = flipped ? N - 1 - sx : sx;
        int py = sy + oy, px = sx + ox;
        py -= py >= N ? N : 0;
        px -= px >= N ? N : 0;
        v[u] = src[py * N + px];
      }
#pragma unroll
      for (int u = 0; u < GD; ++u) {
        const int k = k0 + 64 * u + lane;
        if (k < k_hi) zrow_g[k] = v[u];
      }
    }
  };
  // the normals of env slot g's extrusion number rr of this step -> nb
  auto noise_env = [&](int g, int rr) {
    const int env = __builtin_amdgcn_readfirstlane(s_env[g]);
    double* nrow_g = nb + (size_t)g * ns;
    if (p.noise && rr < p.max_ext) {
      const double* __restrict__ src = p.noise + ((size_t)env * p.max_ext + rr) * N;
      for (int jx = lane; jx < N; jx += 64) nrow_g[jx] = src[jx];
    } else {
      const uint32_t ctr = p.ext_counter[env] + (uint32_t)rr;
      for (int j4 = lane; 4 * j4 < N; j4 += 64) {   // four normals per Philox call
        double v[4];
        philox_normal4(p.seed, (uint32_t)(p.env_base + env), ctr, (uint32_t)j4, v);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (4 * j4 + u < N) nrow_g[4 * j4 + u] = v[u];
      }
    }
  };
  bool same_xcd = false;   // the group's four workgroups share an XCD (measured in round 0: see the barrier)
  int pf = 0;   // bit j: the far samples of this wave's j-th env are already in zb (fetched in the previous round's tail)
  for (int r = 0; r < rounds; ++r) {
    long long t0 = dbg ? wall_clock64() : 0;
    auto cls = [&](int g) { return r < abs(s_dx[g]) ? 1 : (r < abs(s_dx[g]) + abs(s_dy[g]) ? 2 : 0); };
    // One env per wave (wave w takes envs w, w + #waves, ...): class, shift sign, origin and screen base are wave-uniform (scalar
    // registers), a sample costs ~15 vector instructions instead of ~80 (index division, per-sample class lookup): the gather was bound
    // by vector issue as much as by memory (PMC: 22 vector instructions per matrix instruction over the launch).  Every load of a
    // batch is issued before any is consumed: the samples come from HBM / L2 (the master screens do not fit the caches), one memory
    // round trip per batch of 12.  (What is left is sector traffic: 8 bytes used of every 64 fetched.  A transposed copy of the master
    // screens for the column stencils was tried: its scattered writes cost more than the contiguous reads saved.)
    // The stencils arrive with their NEAR samples (the two newest slices: rows / columns 0 and 1) first and the FAR ones after them
    // (aog_upload_layer orders them so).  An env that extrudes in the same direction as in the round before had its far samples and its
    // normals fetched in that round's tail, AHEAD of the group barrier (they do not depend on the slice the partners were writing): here
    // it only gathers the near samples, which come out of the L2 the partners just wrote.
    for (int jg = 0, g = wave; g < G; g += n_waves, ++jg) {
      const int c = __builtin_amdgcn_readfirstlane(cls(g));
      if (!c) continue;   // (rows of envs outside both classes keep stale samples: their product columns are never stored)
      const int nz = c == 1 ? p.nz_h : p.nz_v, near = c == 1 ? p.near_h : p.near_v;
      gather_env(g, c, 0, ((pf >> jg) & 1) ? near : nz, __builtin_amdgcn_readfirstlane(s_ox[g]), __builtin_amdgcn_readfirstlane(s_oy[g]));
    }
    if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long t = wall_clock64(); tm[0] += t - t0; t0 = t; }
    if (r == 0) {   // (later rounds: drawn in the previous round's tail)
      for (int g = wave; g < G; g += n_waves)
        if (__builtin_amdgcn_readfirstlane(cls(g))) noise_env(g, r);
    }
    __syncthreads();
    if (dbg) { long long t = wall_clock64(); tm[1] += t - t0; t0 = t; }
    const int my_cls = cls(li);
    for (int c = 1; c <= 2; ++c) {
      bool any = false;
      for (int g = 0; g < G; ++g) any |= cls(g) == c;
      if (!any) continue;
      if (dbg) ++n_pass;
      const long long c0 = dbg ? clock64() : 0;
      const bool horizontal = c == 1;
      const int nz = horizontal ? p.nz_h : p.nz_v;
      const double2* WA = reinterpret_cast<const double2*>(horizontal ? p.Wa_h : p.Wa_v);
      const double2* WB = reinterpret_cast<const double2*>(horizontal ? p.Wb_h : p.Wb_v);
      const int nz8 = (nz + 7) >> 3, n8 = (N + 7) >> 3;
      const bool feed = my_cls == c;
      for (int rb0 = 0; rb0 * 16 < N; rb0 += 4 * kExtParts) {   // uniform trip count: the partial-sum exchange syncs inside
        const int rb = rb0 + part * 4 + rbl;
        const int rbc = rb * 16 < N ? rb : 0;   // waves past the last row block compute a dummy tile and store nothing
        f64x4 accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
        const double* zrow = zb + (size_t)li * zs;
        const double* nrow = nb + (size_t)li * ns;
        // weights arrive MFMA-ready: one 16-B load per lane = the A operands of two consecutive k-steps (blocked on the host)
        auto load_chunk = [&](const double2* __restrict__ W, int K8, const double* __restrict__ vec, int K, int k0, int kend, double (&av)[8], double (&bv)[8]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int blk = min((k0 >> 3) + u, K8 - 1);
            const double2 w = W[(size_t)blk * 64 + lane];
            av[2 * u] = w.x;
            av[2 * u + 1] = w.y;
            const int ka = k0 + 8 * u + lk, kb = ka + 4;
            // unconditional reads, masked bitwise: a select here is turned back into a branch around the read, which
            // serialises the chunk (every read then waits for its own lgkmcnt)
            const long long za = __double_as_longlong(vec[min(ka, K - 1)]), zb2 = __double_as_longlong(vec[min(kb, K - 1)]);
            bv[2 * u] = __longlong_as_double(za & -(long long)(ka < kend && feed));
            bv[2 * u + 1] = __longlong_as_double(zb2 & -(long long)(kb < kend && feed));
          }
        };
        auto run = [&](const double2* __restrict__ W, int K8, const double* __restrict__ vec, int K, int kbeg, int kend, f64x4& acc) {
          double a0[8], b0[8], a1[8], b1[8];
          load_chunk(W, K8, vec, K, kbeg, kend, a0, b0);
          for (int k0 = kbeg; k0 < kend; k0 += 64) {
            // the scheduling fences keep the next chunk's loads AHEAD of this chunk's matrix ops (left alone the compiler sinks
            // every load to just before its use and the prefetch distance is gone)
            load_chunk(W, K8, vec, K, k0 + 32, kend, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_chunk(W, K8, vec, K, k0 + 64, kend, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        // Fast form when every K slice is a whole number of 32-deep blocks (N a multiple of 64 KS, 3 N stencil samples): no index
        // clamps, no masks — a column of the product belongs to ONE env, so whatever a lane of an env outside this class feeds
        // (stale LDS) only reaches columns that are never stored.  The masked form spent ~12 vector instructions per matrix
        // instruction on clamps and 64-bit masks: as much issue time as the fp64 matrix pipe itself.  The stencil and the noise
        // passes run as ONE stream of blocks with three blocks of weights and one block of LDS operands in flight; two accumulation chains per pass.
        auto run_fast = [&](const double2* __restrict__ Wa, const double* __restrict__ va, int na, const double2* __restrict__ Wb,
                            const double* __restrict__ vb, int nb_blk, f64x4& accA_, f64x4& accB_) {
          f64x4 accA2 = {0.0, 0.0, 0.0, 0.0}, accB2 = {0.0, 0.0, 0.0, 0.0};
          const int nblk = na + nb_blk;   // blocks of 32 k-values: 4 16-byte weight loads per lane, 8 LDS reads, 8 matrix ops
          double2 w0[4], w1[4], w2[4], w3[4];
          double b0[8], b1[8];
          auto loadw = [&](int jb, double2 (&w)[4]) {
            jb = min(jb, nblk - 1);   // past the end: re-reads the last block (in range, never used).  NOT a branch around the loads: the
                                      // compiler then loses count of the loads in flight and waits for all of them before every matrix op
            const double2* __restrict__ src = jb < na ? Wa + (size_t)jb * 4 * 64 : Wb + (size_t)(jb - na) * 4 * 64;
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = src[(size_t)u * 64 + lane];
          };
          // the B operands of a block (LDS) are requested one block ahead as well: read -> wait -> two matrix ops -> read ... was what
          // the compiler made of reads placed next to their use: an LDS round trip in front of every pair of matrix ops, 260 cycles per
          // matrix op and wave against the 64 it occupies the pipe (measured: weights served from L1 changed nothing)
          auto loadb = [&](int jb, double (&bv)[8]) {
            jb = min(jb, nblk - 1);
            const double* v = (jb < na ? va + 32 * jb : vb + 32 * (jb - na)) + lk;
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = v[4 * u];
          };
          auto mma = [&](int jb, const double2 (&w)[4], const double (&bv)[8]) {
            if (jb >= nblk) return;
            if (jb < na) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                accA_ = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u].x, bv[2 * u], accA_, 0, 0, 0);
                accA2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u].y, bv[2 * u + 1], accA2, 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                accB_ = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u].x, bv[2 * u], accB_, 0, 0, 0);
                accB2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u].y, bv[2 * u + 1], accB2, 0, 0, 0);
              }
            }
          };
          // the scheduling fences keep the loads AHEAD of the matrix ops (left alone the compiler sinks every load to just before its use)
#define AOG_EXT_STEP(J, WLOAD, BLOAD, WCUR, BCUR)   \
  loadw((J) + 3, WLOAD);                            \
  loadb((J) + 1, BLOAD);                            \
  __builtin_amdgcn_sched_barrier(0);                \
  mma((J), WCUR, BCUR);                             \
  __builtin_amdgcn_sched_barrier(0);
          loadw(0, w0);
          loadw(1, w1);
          loadw(2, w2);
          loadb(0, b0);
          for (int jb = 0; jb < nblk; jb += 4) {
            AOG_EXT_STEP(jb, w3, b1, w0, b0)
            AOG_EXT_STEP(jb + 1, w0, b0, w1, b1)
            AOG_EXT_STEP(jb + 2, w1, b1, w2, b0)
            AOG_EXT_STEP(jb + 3, w2, b0, w3, b1)
          }
#undef AOG_EXT_STEP
#pragma unroll
          for (int q = 0; q < 4; ++q) { accA_[q] += accA2[q]; accB_[q] += accB2[q]; }
        };
        {
          const int ka = ((nz + 32 * KS - 1) / (32 * KS)) * 32, kb = ((N + 32 * KS - 1) / (32 * KS)) * 32;
          const int a0 = min(ks * ka, nz), a1 = min(a0 + ka, nz), b0 = min(ks * kb, N), b1 = min(b0 + kb, N);
          if (nz % (64 * KS) == 0 && N % (64 * KS) == 0) {
            run_fast(WA + ((size_t)rbc * nz8 + (a0 >> 3)) * 64, zrow + a0, (a1 - a0) >> 5, WB + ((size_t)rbc * n8 + (b0 >> 3)) * 64, nrow + b0,
                     (b1 - b0) >> 5, accA, accB);
          } else {
            if (a1 > a0) run(WA + (size_t)rbc * nz8 * 64, nz8, zrow, nz, a0, a1, accA);
            if (b1 > b0) run(WB + (size_t)rbc * n8 * 64, n8, nrow, N, b0, b1, accB);
          }
        }
        double part_v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) part_v[q] = accA[q] + accB[q] * p.sqrt_cn2;
        long long t1 = 0;
        if (dbg) { t1 = wall_clock64(); tm[4] += t1 - t0; cyc += clock64() - c0; }
        if (ks > 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pb[((size_t)(ks - 1) * 4 + rbl) * 256 + q * 64 + lane] = part_v[q];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            for (int t = 0; t < KS - 1; ++t) part_v[q] += pb[((size_t)t * 4 + rbl) * 256 + q * 64 + lane];
        }
        __syncthreads();
        if (dbg) tm[5] += wall_clock64() - t1;
        if (feed && ks == 0) {
          const int g = li;
          const bool flipped = horizontal ? s_dx[g] > 0 : s_dy[g] > 0;
          int nox = s_ox[g], noy = s_oy[g];
          if (horizontal) nox = flipped ? (nox + 1 == N ? 0 : nox + 1) : (nox == 0 ? N - 1 : nox - 1);
          else noy = flipped ? (noy + 1 == N ? 0 : noy + 1) : (noy == 0 ? N - 1 : noy - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = rb * 16 + lk + 4 * q;
            if (i >= N) continue;
            const double v = part_v[q];
            int ly, lx;
            if (horizontal) { ly = flipped ? N - 1 - i : i; lx = flipped ? N - 1 : 0; }
            else { ly = flipped ? N - 1 : 0; lx = flipped ? N - 1 - i : i; }
            int py = ly + noy, px = lx + nox;
            if (py >= N) py -= N;
            if (px >= N) px -= N;
            store_master(p, s_env[g], py, px, v);
          }
        }
      }
    }
    // ---- tail: what the next round needs and this round's slice does not touch, while the partners finish ----
    // (every wave is past its last read of zb / nb: the partial-sum exchange above ends in a workgroup barrier)
    pf = 0;
    if (r + 1 < rounds) {
      auto cls_next = [&](int g) { return r + 1 < abs(s_dx[g]) ? 1 : (r + 1 < abs(s_dx[g]) + abs(s_dy[g]) ? 2 : 0); };
      for (int jg = 0, g = wave; g < G; g += n_waves, ++jg) {
        const int c1 = __builtin_amdgcn_readfirstlane(cls_next(g));
        if (!c1) continue;
        if (c1 == __builtin_amdgcn_readfirstlane(cls(g))) {
          // same direction again: next round's frame is this one moved by one slice, its far samples (>= 2 slices in) are >= 1 slice
          // in now — written in earlier rounds, behind earlier barriers
          int nox = __builtin_amdgcn_readfirstlane(s_ox[g]), noy = __builtin_amdgcn_readfirstlane(s_oy[g]);
          if (c1 == 1) nox = __builtin_amdgcn_readfirstlane(s_dx[g]) > 0 ? (nox + 1 == N ? 0 : nox + 1) : (nox == 0 ? N - 1 : nox - 1);
          else noy = __builtin_amdgcn_readfirstlane(s_dy[g]) > 0 ? (noy + 1 == N ? 0 : noy + 1) : (noy == 0 ? N - 1 : noy - 1);
          gather_env(g, c1, c1 == 1 ? p.near_h : p.near_v, c1 == 1 ? p.nz_h : p.nz_v, nox, noy);
          pf |= 1 << jg;
        }
        noise_env(g, r + 1);
      }
    }
    // ---- group barrier: this round's rows of all four workgroups are visible before anyone gathers again ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (dbg) { long long t = wall_clock64(); tm[2] += t - t0; t0 = t; }
    // Visibility between the partners.  Placement-independent form: agent-scope release (on this chip: write the XCD's L2 back) before the
    // ticket, agent-scope acquire (drop L1 and the L2 lines of other XCDs) after the wait.  The eight XCDs have an L2 each, and partners that
    // sit on ONE XCD need less on the WRITING side: a store is in that shared L2 once vmcnt has counted it (the vector L1 writes through), so
    // the writer only drains its stores — no write-back of the whole L2 per round.  Which it is, the group MEASURES: every workgroup ORs the bit of the XCD it really runs on (HW_REG_XCC_ID)
    // into the high half of the group's ticket word ahead of its first ticket — under the full protocol — and whoever sees the four tickets
    // of round 0 sees the four bits; one bit set = one XCD, and the later rounds take the short form.  (The dispatcher deals workgroups b
    // and b + 8 to the same XCD, but nothing promises it: a different placement costs speed, never correctness.)  The full form cost
    // 30-45 us per step at B = 1024: the L2 write-back, and every round's weights re-fetched after the wider invalidate.
    if (threadIdx.x == 0) {
      if (!same_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (r == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&bar[group], 1u << (16 + (xcc & 7u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_fetch_add(&bar[group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)kExtParts * (unsigned)(r + 1);
      unsigned spins = 0, word;
      bool timed_out = false;
      while (((word = __hip_atomic_load(&bar[group], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffu) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > spin_limit) {   // ~seconds: a partner never arrived (not co-resident).  The launch still terminates, but its
          atomicExch(status, 1);      // screens are invalid: flag it on the device and in host-visible memory — the host refuses
          __hip_atomic_store(host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // every later call on the handle
          timed_out = true;
          break;
        }
      }
      if (r == 0) s_same_xcd = (!timed_out && !force_agent_scope && __builtin_popcount((word >> 16) & 0xffu) == 1) ? 1 : 0;
      // (the reader side keeps the agent-scope acquire in both forms: outside threadgroup-split mode a `buffer_inv sc0` does not reliably
      // drop this CU's L1 lines — a 4096-env run differed from its 1024-env twin in a few samples, once in three full test runs)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (r == 0) same_xcd = s_same_xcd != 0;
    if (dbg) { long long t = wall_clock64(); tm[3] += t - t0; t0 = t; }
    if (threadIdx.x < G) {
      const int g = threadIdx.x;
      const int c = cls(g);
      if (c == 1) s_ox[g] = s_dx[g] > 0 ? (s_ox[g] + 1 == N ? 0 : s_ox[g] + 1) : (s_ox[g] == 0 ? N - 1 : s_ox[g] - 1);
      else if (c == 2) s_oy[g] = s_dy[g] > 0 ? (s_oy[g] + 1 == N ? 0 : s_oy[g] + 1) : (s_oy[g] == 0 ? N - 1 : s_oy[g] - 1);
    }
    __syncthreads();
  }
  if (dbg) {
    for (int i = 0; i < 4; ++i) status[4 + i] += (int)tm[i];
    status[9] += (int)tm[4];
    status[10] += (int)tm[5];
    status[8] += rounds;
    status[11] += n_pass;
    status[12] += (int)(cyc >> 4);
  }
  if (part == 0 && threadIdx.x < G && perm[env0 + threadIdx.x] >= 0) {
    const int g = threadIdx.x, env = s_env[g];
    p.origin[2 * env] = s_ox[g];
    p.origin[2 * env + 1] = s_oy[g];
    p.ext_counter[env] += (uint32_t)(abs(s_dx[g]) + abs(s_dy[g]));
  }
}
#endif  // AOG_MAIN_TU

// caller screens -> float64 master (origin 0)
template <typename T>
__global__ void k_store_master(const T* __restrict__ psi, double* __restrict__ master, int32_t* __restrict__ origin,
                               uint32_t* __restrict__ ext_counter, int first, int count, int n_pix2) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)count * n_pix2) return;
  const int e = (int)(idx / n_pix2);
  master[(size_t)(first + e) * n_pix2 + (idx - (size_t)e * n_pix2)] = (double)psi[idx];
  if (idx - (size_t)e * n_pix2 == 0) {
    origin[2 * (first + e)] = 0;
    origin[2 * (first + e) + 1] = 0;
    ext_counter[first + e] = 0;
  }
}

// float64 master screens of envs [first, first + gridDim.x) -> the fp32 ring copy the fused kernel reads (DynPsi): one workgroup per
// env; keep_ref = 0: the env's reference piston becomes the aperture mean of the screen as it stands (installation), 1: the stored
// reference is kept (state restore: the copy must come out bit-identical to the one the extrusions maintained)
__global__ __launch_bounds__(256) void k_ring_from_master(const double* __restrict__ master, const int32_t* __restrict__ origin,
                                                          const int32_t* __restrict__ ap_index, double* __restrict__ ring_ref,
                                                          float* __restrict__ ring, int first, int N, int n_ap, double inv, int keep_ref) {
  __shared__ double sm[8];
  const int env = first + blockIdx.x;
  const double* src = master + (size_t)env * N * N;
  const int ox = origin[2 * env], oy = origin[2 * env + 1];
  double ref;
  if (keep_ref) {
    ref = ring_ref[env];
  } else {
    double acc = 0;
    for (int p = threadIdx.x; p < n_ap; p += blockDim.x) {
      const int flat = ap_index[p], iy = flat / N, ix = flat - iy * N;
      int py = iy + oy, px = ix + ox;
      if (py >= N) py -= N;
      if (px >= N) px -= N;
      acc += src[(size_t)py * N + px];
    }
    ref = block_reduce_sum(acc, sm) / (double)n_ap;
    if (threadIdx.x == 0) ring_ref[env] = ref;
  }
  const int RS = N + 4;
  float* dst = ring + (size_t)env * N * RS;
  for (int i = threadIdx.x; i < N * RS; i += blockDim.x) {
    const int py = i / RS, c = i - py * RS;
    const int px = c < N ? c : c - N;
    dst[i] = (float)((src[(size_t)py * N + px] - ref) * inv);
  }
}

// Per-step repack of the float64 ring-buffer screens into the MFMA kernel's tiled fp32 layout (dynamic atmosphere).
// One workgroup = one 32-env tile x 2 pixel tiles: each wave reads 64 consecutive packed pixels of one env at a time
// (coalesced along x), the block transposes through LDS and every wave then writes whole 1-KiB rows of psi_tile.
// The piston offset subtracted is the aperture mean measured by the PREVIOUS repack (outputs are invariant to a global
// phase; the offset only keeps the fp32 magnitudes small), and this pass accumulates the sums for the next one.
constexpr int kRepackIters = 8;   // pixel-tile pairs per workgroup: one float64 atomic per (wave, env) per 8 x 64 pixels
__global__ __launch_bounds__(256) void k_repack_master(const double* __restrict__ master, const int32_t* __restrict__ origin,
                                                        const int32_t* __restrict__ ap_index, const double* __restrict__ offset,
                                                        double* __restrict__ sum_next, float* __restrict__ psi_tile, int B, int N,
                                                        int n_ap, int n_ptiles, double inv_two_pi_lambda) {
  constexpr int LD = 68;  // padded row (floats) of the [32 envs][64 pixels] staging tile
  __shared__ float stage[32 * LD];
  const int et = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int oy[8], ox[8];
  double off[8], sum[8];
  const double* base[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int env = min(et * 32 + wave * 8 + q, B - 1);
    ox[q] = origin[2 * env];
    oy[q] = origin[2 * env + 1];
    off[q] = offset[env];
    sum[q] = 0.0;
    base[q] = master + (size_t)env * N * N;
  }
  for (int it = 0; it < kRepackIters; ++it) {
    const int pt0 = (blockIdx.x * kRepackIters + it) * 2;
    if (pt0 >= n_ptiles) break;   // uniform
    const int p = pt0 * 32 + lane;
    const bool valid_p = p < n_ap;
    const int flat = ap_index[valid_p ? p : n_ap - 1];   // logical pupil coordinates of this lane's packed pixel (same for every env)
    const int iy = flat / N, ix = flat - iy * N;
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // all eight loads in flight before anything consumes them
      int py = iy + oy[q], px = ix + ox[q];
      if (py >= N) py -= N;
      if (px >= N) px -= N;
      v[q] = base[q][(size_t)py * N + px];
    }
    if (it) __syncthreads();   // the previous iteration's rows have left the staging tile
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = wave * 8 + q;
      const bool ok = valid_p && et * 32 + e < B;
      sum[q] += ok ? v[q] : 0.0;
      stage[e * LD + lane] = ok ? (float)((v[q] - off[q]) * inv_two_pi_lambda) : 0.f;
    }
    __syncthreads();
    // rows of psi_tile: [et][pt][g][lane = 32 h + e][4]; this pass owns pt0, pt0 + 1 (8 rows); wave w writes rows 2w, 2w+1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr, tl = row >> 2, g = row & 3;
      const int pt = pt0 + tl;
      if (pt >= n_ptiles) continue;
      const int h = lane >> 5, e = lane & 31;
      const float* src = stage + e * LD + tl * 32 + 8 * g + 4 * h;
      float4 w = make_float4(src[0], src[1], src[2], src[3]);
      reinterpret_cast<float4*>(psi_tile)[(((size_t)et * n_ptiles + pt) * 4 + g) * 64 + lane] = w;
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    double t = sum[q];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    if (lane == 0 && et * 32 + wave * 8 + q < B) atomicAdd(&sum_next[et * 32 + wave * 8 + q], t);
  }
}

// offsets for the next repack: mean of the sums the last one accumulated
__global__ void k_refresh_offsets(double* __restrict__ offset, double* __restrict__ sum_next, int B, int n_ap) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  offset[env] = sum_next[env] / (double)n_ap;
  sum_next[env] = 0.0;
}

// ring buffer -> plain [B][N][N] (tests, checkpointing)
__global__ void k_unroll_master(const double* __restrict__ master, const int32_t* __restrict__ origin, double* __restrict__ out,
                                int first, int count, int N) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)count * N * N) return;
  const int env = first + (int)(idx / ((size_t)N * N));
  const int flat = (int)(idx - (size_t)(env - first) * N * N);
  const int iy = flat / N, ix = flat - iy * N;
  int py = iy + origin[2 * env + 1], px = ix + origin[2 * env];
  if (py >= N) py -= N;
  if (px >= N) px -= N;
  out[idx] = master[(size_t)env * N * N + (size_t)py * N + px];
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K4  focal-plane field of one env (propagator_fiber, AO_env.py:138), off the step() path.
//   E[y][x]  = exp(2 pi i (psi + M a))  on the aperture (amplitude folded into focal_m1), 0 outside
//   T[v][x]  = sum_y m1[v][y] E[y][x];     F[v][u] = sum_x T[v][x] m2[x][u]        (float64 accumulation)
// ------------------------------------------------------------------------------------------------
#ifdef AOG_MAIN_TU
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
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K8  screen synthesis (hcipy FiniteAtmosphericLayer / SpectralNoiseFactoryFFT; layer.reset(), AO_env.py:77)
//   spectrum[b][v][u] = a(u, v) (g1 + i g2),  a = sqrt(PSD_vK (2 pi)^2 / du^2)  on the UNSHIFTED (q N)^2 FFT grid,
//   then an un-normalised inverse FFT (hipFFT) and k_screen_crop takes Re of the centred N x N crop / (M delta^2) * sqrt(Cn^2).
// ------------------------------------------------------------------------------------------------
// Spectrum stream: sample (line v, column u = q a + bg, a = lane + LW r) is word r & 3 of the Philox call whose counter is the flat index of
// the call's first sample, v m + q (lane + LW (r & ~3)) + bg — the four samples of a call are the ones ONE lane of the pruned row pass
// feeds into one radix-R butterfly (k_screen_rows), so that pass draws a call per four samples and keeps nothing across its steps.  One
// 32-bit word makes one complex normal: 16 bits of radius uniform, 16 bits of angle (the screen is a sum of 8 M such terms per pixel:
// only their variance and independence reach it; E r^2 of the 16-bit form is 2 to 1e-4).  LW = 64 or 60 (pupils of 64 R / 60 R pixels).
constexpr uint32_t kSpectrumTag = 0x5C4EE7u;       // literal (q N)^2 draw and the high band of the two-band form
constexpr uint32_t kSpectrumTagLow = 0x5C4EE8u;    // low band of the two-band form
__device__ __forceinline__ void spectrum_words(size_t cidx, uint32_t generation, uint32_t env_global, unsigned long long seed, uint32_t (&w)[4],
                                               uint32_t tag = kSpectrumTag) {
  uint32_t c[4] = {(uint32_t)cidx, (uint32_t)(cidx >> 32) ^ (generation * 0x9E3779B9u), env_global, tag};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int rr = 0; rr < 10; ++rr) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3];
}
// a(u, v) (g1 + i g2) from one word; raw hardware transcendentals (v_log = log2, v_exp = 2^x, v_sqrt, v_sin / v_cos in revolutions):
// arguments are normal floats in range by construction (u1 in (0, 1), f2 + u0^2 > 0)
__device__ __forceinline__ float2 spectrum_sample(uint32_t word, int uu, int m, float fv, float du, float u0sq, float amp_scale) {
  const float fu = du * (float)(uu < m / 2 ? uu : uu - m);
  const float f2 = fu * fu + fv * fv;
  const float amp = f2 == 0.f ? 0.f : amp_scale * __builtin_amdgcn_exp2f((-11.0f / 12.0f) * __builtin_amdgcn_logf(f2 + u0sq));
  const float u1 = ((float)(word & 0xffffu) + 0.5f) * (1.0f / 65536.0f);
  const float u2 = (float)(word >> 16) * (1.0f / 65536.0f);
  const float rad = amp * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  return make_float2(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}
__host__ __device__ inline int spectrum_lane_width(int N) { return (N % 64 != 0 && N % 60 == 0) ? 60 : 64; }

// Two-band form of the same synthesis: the variance of every spectrum sample is split by a smooth radial
// window, w_low(f) + w_high(f) = 1, f = |u| / du_fine:  w_high = smootherstep((f^2 - f1^2) / (f2^2 - f1^2)).  The low band keeps
// hcipy's (q N)^2 grid (only |k| < f2 is non-zero there), the high band is drawn on the (2 N)^2 grid.
struct BandWindow {
  float inv_du2;     // 1 / du_fine^2
  float f1sq;        // f1^2  (f in units of du_fine)
  float inv_band;    // 1 / (f2^2 - f1^2)
};
__device__ __forceinline__ float band_high_weight(float f2, const BandWindow& w) {
  const float t = __builtin_amdgcn_fmed3f((f2 * w.inv_du2 - w.f1sq) * w.inv_band, 0.f, 1.f);
  return t * t * t * fmaf(t, fmaf(t, 6.f, -15.f), 10.f);
}
// BAND 0: whole spectrum (literal form), 1: high band, 2: low band.  fu, fv: the sample's frequencies (rad / m).
template <int BAND>
__device__ __forceinline__ float2 band_sample(uint32_t word, float fu, float fv, float u0sq, float amp_scale, const BandWindow& win) {
  const float f2 = fu * fu + fv * fv;
  float amp = f2 == 0.f ? 0.f : amp_scale * __builtin_amdgcn_exp2f((-11.0f / 12.0f) * __builtin_amdgcn_logf(f2 + u0sq));
  if constexpr (BAND == 1) amp *= __builtin_amdgcn_sqrtf(band_high_weight(f2, win));
  if constexpr (BAND == 2) amp *= __builtin_amdgcn_sqrtf(fmaxf(1.f - band_high_weight(f2, win), 0.f));
  const float u1 = ((float)(word & 0xffffu) + 0.5f) * (1.0f / 65536.0f);
  const float u2 = (float)(word >> 16) * (1.0f / 65536.0f);
  const float rad = amp * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  return make_float2(rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2));
}

#ifdef AOG_MAIN_TU
// full (q N)^2 spectrum for the hipFFT route (pupils the pruned passes do not cover, and the equivalence test): one thread per Philox call
__global__ void k_spectrum_fill(float2* __restrict__ spec, int m, int q, int first_local, int env_base, unsigned long long seed,
                                const uint32_t* __restrict__ gen, float du, float u0sq, float amp_scale, int high_band, BandWindow win) {
  const int N = m / q, LW = spectrum_lane_width(N), R = (N + LW - 1) / LW, RG = (R + 3) / 4;
  const size_t calls_per_line = (size_t)q * LW * RG;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (tid >= calls_per_line * m) return;
  const int v = (int)(tid / calls_per_line);
  const size_t rem = tid - (size_t)v * calls_per_line;
  const int rg = (int)(rem / ((size_t)q * LW)), rem2 = (int)(rem - (size_t)rg * q * LW), lane_a = rem2 / q, bg = rem2 - lane_a * q;
  float2* line = spec + ((size_t)b * m + v) * m;
  // Half-plane form (see k_screen_rows): lines v > m/2 stay zero, lines 0 < v < m/2 carry sqrt(2) x the amplitude
  const bool zero_line = 2 * v > m;
  uint32_t w[4] = {0, 0, 0, 0};
  if (!zero_line)
    spectrum_words((size_t)v * m + (size_t)q * (lane_a + LW * 4 * rg) + bg, gen[first_local + b] + 1u, (uint32_t)(env_base + first_local + b), seed, w);
  const float line_scale = (v == 0 || 2 * v == m) ? 1.f : 1.41421356237f;
  const float fv = du * (float)v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int a = lane_a + LW * (4 * rg + j);
    if (a >= N) continue;
    const int uu = q * a + bg;
    float2 o = make_float2(0.f, 0.f);
    if (!zero_line) {
      o = high_band ? band_sample<1>(w[j], du * (float)(uu < m / 2 ? uu : uu - m), fv, u0sq, amp_scale, win)
                    : spectrum_sample(w[j], uu, m, fv, du, u0sq, amp_scale);
      o.x *= line_scale;
      o.y *= line_scale;
    }
    line[uu] = o;
  }
}

// Zero-fill on the caller's stream as a kernel of the library (the per-step paths zero a few KB .. MB: barrier tickets, lenslet sums,
// focal work buffers): hipMemsetAsync goes through the runtime's blit kernels, ~10 us per call in the profiles against ~3 here.
__global__ void k_zero_words(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

// after a synthesis launch: the envs it served have drawn one more screen
__global__ void k_bump_generation(uint32_t* __restrict__ gen, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) gen[i] += 1u;
}

__global__ void k_screen_crop(const float2* __restrict__ field, float* __restrict__ out, int m, int N, float scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= N * N) return;
  const int iy = idx / N, ix = idx - iy * N;
  // centred crop index i in [m/2 - N/2, m/2 + N/2) <-> unshifted (i - m/2) mod m
  const int jy = (iy - N / 2 + m) % m, jx = (ix - N / 2 + m) % m;
  out[(size_t)b * N * N + idx] = field[(size_t)b * m * m + (size_t)jy * m + jx].x * scale;
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// R1  policy query of the rollout (network.py:48-69): a 3-hidden-layer ReLU MLP with active dropout, Gaussian action sampling
// and its log-probability, one launch.  Workgroup = 16 envs (the 16 columns of v_mfma_f32_16x16x4_f32, exact fp32), 4 waves
// share the output-unit tiles of a layer; activations [unit][16 envs] ping-pong in LDS; nn.Linear weights [out][in] are read
// straight from global memory as the A operand (lane = unit % 16 + 16 (k % 4)).
// ------------------------------------------------------------------------------------------------
struct ActorArgs {
  const void* obs;
  const float *w1, *b1, *w2, *b2, *w3, *b3, *wo, *bo;
  float *mean, *action, *log_prob;
  int B, S, H, A, obs_f16, kpad;
  float p_drop, keep_scale, std, logp_const;
  unsigned long long seed;
  uint32_t call_lo, call_hi;
  int env_base;   // global id of obs row 0
};
#ifdef AOG_MAIN_TU
__device__ __forceinline__ void actor_philox(uint32_t (&c)[4], unsigned long long seed) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// out[m][e] = act(sum_k W[m][k] xin[k][e] + b[m]) for this workgroup's 16 envs; LAYER 1..3 hidden (relu + dropout), 4 output.
// The weights cross HBM/L2 -> registers -> LDS one row chunk at a time as a linear 16-byte-per-lane copy (every load of the chunk in
// flight at once: one memory round trip), and the NEXT chunk — of this layer or the first of the following layer — is requested
// before the matrix work on the current one starts, so the round trips of the four layers hide behind each other's arithmetic
// (the kernel is one latency chain: 64 workgroups at B = 1024, 220 KB of weights each).
constexpr int kActorWFloats = 24576;   // LDS floats for a weight chunk (96 KB)
constexpr int kActorThreads = 640;     // 10 waves: one 16-unit tile each for the reference's 150 hidden units
constexpr int kActorPre = 10;          // float4 registers per thread holding a chunk in flight (640 x 10 x 4 >= kActorWFloats)
__device__ __forceinline__ int actor_rows_max(int K) { return max(16, ((kActorWFloats / K) >> 4) << 4); }
// request rows [r0, r0 + rows_max) of W ([M][K], 16-byte aligned base; r0 is a multiple of 16)
__device__ __forceinline__ void actor_issue(f32x4 (&pre)[kActorPre], const float* __restrict__ W, int K, int M, int r0) {
  const int rc = min(actor_rows_max(K), M - r0);
  const int n4 = (rc * K) >> 2;
  const f32x4* src = reinterpret_cast<const f32x4*>(W + (size_t)r0 * K);
#pragma unroll
  for (int u = 0; u < kActorPre; ++u) pre[u] = src[min((int)threadIdx.x + kActorThreads * u, max(n4 - 1, 0))];
}
template <int LAYER>
__device__ __forceinline__ void actor_layer(const ActorArgs& p, const float* __restrict__ W, const float* __restrict__ bias, int K, int M,
                                            const float* xin, float* xout, float* lp_sum, float* wl, int env0, f32x4 (&pre)[kActorPre],
                                            const float* __restrict__ Wnext, int Knext, int Mnext) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el = lane & 15, kq = lane >> 4;
  const int n_steps = (K + 3) >> 2;
  const int rows_max = actor_rows_max(K);
  for (int r0 = 0; r0 < M; r0 += rows_max) {
    const int rc = min(rows_max, M - r0);
    const int n_fl = rc * K;
    const float* src = W + (size_t)r0 * K;   // 16-byte aligned: r0 is a multiple of 16 and the base pointer is (checked on the host)
    {
      // `pre` holds this chunk (requested during the previous chunk's matrix work, or at kernel start)
      const int n4 = n_fl >> 2;
#pragma unroll
      for (int u = 0; u < kActorPre; ++u)
        if ((int)threadIdx.x + kActorThreads * u < n4) reinterpret_cast<f32x4*>(wl)[threadIdx.x + kActorThreads * u] = pre[u];
      for (int i = (n4 << 2) + threadIdx.x; i < n_fl; i += kActorThreads) wl[i] = src[i];
    }
    __syncthreads();
    if (r0 + rows_max < M) actor_issue(pre, W, K, M, r0 + rows_max);
    else if (Wnext != nullptr) actor_issue(pre, Wnext, Knext, Mnext, 0);
    __builtin_amdgcn_sched_barrier(0);   // (keep the requests ahead of the matrix work)
    const int n_tiles = (rc + 15) >> 4;
    for (int tile = wave; tile < n_tiles; tile += kActorThreads / 64) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int lrow = tile * 16 + el;                 // row inside the chunk
      const float row_ok = lrow < rc ? 1.f : 0.f;
      const float* wrow = wl + (size_t)min(lrow, rc - 1) * K;
      // D: column = env (lane & 15), rows 4 (lane >> 4) + r.  The biases (a global-memory round trip) and the random words of this
      // tile do not depend on the products: both are started before the matrix loop
      const int m0 = r0 + tile * 16 + 4 * kq;
      const int env = env0 + el;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[min(m0 + r, M - 1)];
      uint32_t c[4] = {(uint32_t)m0 | ((uint32_t)LAYER << 24), (uint32_t)(p.env_base + env), p.call_lo, p.call_hi ^ 0xAC70u};
      actor_philox(c, p.seed);
      // Eight k-steps at a time: their 16 LDS reads are in flight together and the matrix ops follow back to back.  Whole groups
      // below K need no clamps or masks (rows past the chunk are clamped to a valid row and dropped at the output), so their reads
      // are base + immediate offset and the loop is the matrix pipe's: three waves share a SIMD's, and with ~15 address/mask
      // instructions per step the vector unit, not the matrix pipe, set the pace (timeline: 5.9 us per 150 x 150 layer).
      const int n_full = (K >> 2) & ~7;   // k-steps in whole unmasked groups
      {
        const float* wa = wrow + kq;
        const float* xb = xin + kq * 16 + el;
        for (int s0 = 0; s0 < n_full; s0 += 8, wa += 32, xb += 8 * 64) {
          float av[8], xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            av[u] = wa[4 * u];
            xv[u] = xb[64 * u];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], xv[u], acc, 0, 0, 0);
        }
      }
      for (int s0 = n_full; s0 < n_steps; s0 += 8) {
        float av[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = 4 * (s0 + u) + kq;
          av[u] = wrow[min(k, K - 1)] * (k < K ? row_ok : 0.f);
          xv[u] = xin[min(k, p.kpad - 1) * 16 + el];   // rows K .. kpad-1 of xin are zero; steps past the end multiply by a = 0
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], xv[u], acc, 0, 0, 0);
      }
      if constexpr (LAYER < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + r;
          float v = 0.f;
          if (m < M) {
            v = fmaxf(acc[r] + bv[r], 0.f);
            const float u = (float)(c[r] >> 8) * (1.0f / 16777216.0f);   // [0, 1): keep with probability 1 - p
            v = u >= p.p_drop ? v * p.keep_scale : 0.f;
          }
          if (m < p.kpad) xout[m * 16 + el] = v;   // units M .. are written as zero: the next layer's K padding
        }
      } else {
        float ssq = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float u1 = ((float)c[2 * h] + 0.5f) * (1.0f / 4294967296.0f), u2 = ((float)c[2 * h + 1] + 0.5f) * (1.0f / 4294967296.0f);
          const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
          const float eps[2] = {rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2)};
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int m = m0 + 2 * h + t;
            if (m < M && env < p.B) {
              const float mu = acc[2 * h + t] + bv[2 * h + t];
              if (p.mean) p.mean[(size_t)env * M + m] = mu;
              if (p.action) p.action[(size_t)env * M + m] = mu + p.std * eps[t];
              ssq += eps[t] * eps[t];
            }
          }
        }
        atomicAdd(&lp_sum[el], ssq);
      }
    }
    __syncthreads();   // the chunk is consumed before the next one (or the next layer's) overwrites wl
  }
}

__global__ __launch_bounds__(kActorThreads) void k_actor_act(ActorArgs p) {
  extern __shared__ float lds_act[];   // xa [kpad][16] | xb [kpad][16] | lp [16] | weight chunk [kActorWFloats]
  float* xa = lds_act;
  float* xb = xa + (size_t)p.kpad * 16;
  float* lp = xb + (size_t)p.kpad * 16;
  float* wt = lp + 16;
  const int env0 = blockIdx.x * 16;
  f32x4 pre[kActorPre];
  actor_issue(pre, p.w1, p.S, p.H, 0);   // the first weight chunk travels while the observations are staged
  for (int i = threadIdx.x; i < 2 * p.kpad * 16 + 16; i += kActorThreads) lds_act[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < p.S * 16; i += kActorThreads) {
    const int k = i >> 4, e = i & 15, env = min(env0 + e, p.B - 1);
    xa[i] = p.obs_f16 ? (float)reinterpret_cast<const _Float16*>(p.obs)[(size_t)env * p.S + k]
                      : reinterpret_cast<const float*>(p.obs)[(size_t)env * p.S + k];
  }
  __syncthreads();
  actor_layer<1>(p, p.w1, p.b1, p.S, p.H, xa, xb, lp, wt, env0, pre, p.w2, p.H, p.H);
  __syncthreads();
  actor_layer<2>(p, p.w2, p.b2, p.H, p.H, xb, xa, lp, wt, env0, pre, p.w3, p.H, p.H);
  __syncthreads();
  actor_layer<3>(p, p.w3, p.b3, p.H, p.H, xa, xb, lp, wt, env0, pre, p.wo, p.H, p.A);
  __syncthreads();
  actor_layer<4>(p, p.wo, p.bo, p.H, p.A, xb, nullptr, lp, wt, env0, pre, nullptr, 0, 0);
  __syncthreads();
  if (threadIdx.x < 16 && env0 + threadIdx.x < p.B && p.log_prob) p.log_prob[env0 + threadIdx.x] = -0.5f * lp[threadIdx.x] - p.logp_const;
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K8 (pruned form)  The centred N x N crop of the (qN)^2 inverse transform never needs the (qN)^2 array in memory:
//   out[i - N/2] = sum_{k < m} S[k] e^{2 pi i k (i - N/2) / m},  m = q N,  i < N.   With k = q a + b:
//   out = sum_b e^{2 pi i b (i - N/2) / m} F_b[i],   F_b = length-N inverse DFT over a of  (-1)^a S[q a + b].
// One wave = one line of length m.  Lane l holds a = l + 64 r (r < R = N / 64) for a group of b's; radix-R butterflies over r in
// registers, twiddle, an LDS transpose so that every lane owns one 64-point sequence, a 64-point transform entirely in registers,
// the b-twiddles by recurrence, a second LDS transpose and the sum over b.  Pass A (k_screen_rows) draws the spectrum line from
// Philox on the fly (same counter -> sample mapping as k_spectrum_fill) and writes T[v][i]; pass B (k_screen_cols) runs the same
// transform down the columns of T and writes Re(.) * scale.  Per env: 8 MB written + read instead of ~1 GB at N = 256, q = 16.
// ------------------------------------------------------------------------------------------------
struct cf32 { float x, y; };
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return {a.x - b.x, a.y - b.y}; }
struct Tw64 { float c[32][2]; };
__device__ constexpr Tw64 kTw64 = {{{1.000000000e+00f, 0.000000000e+00f}, {9.951847267e-01f, 9.801714033e-02f}, {9.807852804e-01f, 1.950903220e-01f}, {9.569403357e-01f, 2.902846773e-01f}, {9.238795325e-01f, 3.826834324e-01f}, {8.819212643e-01f, 4.713967368e-01f}, {8.314696123e-01f, 5.555702330e-01f}, {7.730104534e-01f, 6.343932842e-01f}, {7.071067812e-01f, 7.071067812e-01f}, {6.343932842e-01f, 7.730104534e-01f}, {5.555702330e-01f, 8.314696123e-01f}, {4.713967368e-01f, 8.819212643e-01f}, {3.826834324e-01f, 9.238795325e-01f}, {2.902846773e-01f, 9.569403357e-01f}, {1.950903220e-01f, 9.807852804e-01f}, {9.801714033e-02f, 9.951847267e-01f}, {6.123233996e-17f, 1.000000000e+00f}, {-9.801714033e-02f, 9.951847267e-01f}, {-1.950903220e-01f, 9.807852804e-01f}, {-2.902846773e-01f, 9.569403357e-01f}, {-3.826834324e-01f, 9.238795325e-01f}, {-4.713967368e-01f, 8.819212643e-01f}, {-5.555702330e-01f, 8.314696123e-01f}, {-6.343932842e-01f, 7.730104534e-01f}, {-7.071067812e-01f, 7.071067812e-01f}, {-7.730104534e-01f, 6.343932842e-01f}, {-8.314696123e-01f, 5.555702330e-01f}, {-8.819212643e-01f, 4.713967368e-01f}, {-9.238795325e-01f, 3.826834324e-01f}, {-9.569403357e-01f, 2.902846773e-01f}, {-9.807852804e-01f, 1.950903220e-01f}, {-9.951847267e-01f, 9.801714033e-02f}}};
constexpr int bitrev_c(int i, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((i >> b) & 1) << (bits - 1 - b);
  return r;
}
constexpr int log2_c(int n) { return n <= 1 ? 0 : 1 + log2_c(n / 2); }
// in-register inverse DFT (e^{+}) of NP points, decimation in frequency: X[i] ends up in x[bitrev(i)].  All indices are compile-time.
template <int NP>
__device__ __forceinline__ void dft_reg(cf32 (&x)[NP]) {
  static_for<log2_c(NP)>([&](auto sc) {
    constexpr int half = NP >> (decltype(sc)::v + 1);
    static_for<NP>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      if constexpr ((i & half) == 0) {
        constexpr int j = i | half;
        constexpr int k = (i & (half - 1)) * (32 / half);   // W_64^{k 64/(2 half)} = e^{2 pi i (i mod half) / (2 half)}
        const cf32 a = x[i], b = x[j];
        x[i] = cadd(a, b);
        const cf32 t = csub(a, b);
        if constexpr (k == 0) x[j] = t;
        else if constexpr (k == 16) x[j] = cf32{-t.y, t.x};
        else x[j] = cmul(t, cf32{kTw64.c[k][0], kTw64.c[k][1]});
      }
    });
  });
}

// 60-point variant (pupils of 60, 120, 240 (the reference's size), 480 pixels): mixed radix 2 x 2 x 3 x 5, recursive decimation in
// time over the smallest prime factor, every index compile-time; output in natural order.
struct Tw60 { float c[60][2]; };
__device__ constexpr Tw60 kTw60 = {{{1.000000000e+00f, 0.000000000e+00f}, {9.945218954e-01f, 1.045284633e-01f}, {9.781476007e-01f, 2.079116908e-01f}, {9.510565163e-01f, 3.090169944e-01f}, {9.135454576e-01f, 4.067366431e-01f}, {8.660254038e-01f, 5.000000000e-01f}, {8.090169944e-01f, 5.877852523e-01f}, {7.431448255e-01f, 6.691306064e-01f}, {6.691306064e-01f, 7.431448255e-01f}, {5.877852523e-01f, 8.090169944e-01f}, {5.000000000e-01f, 8.660254038e-01f}, {4.067366431e-01f, 9.135454576e-01f}, {3.090169944e-01f, 9.510565163e-01f}, {2.079116908e-01f, 9.781476007e-01f}, {1.045284633e-01f, 9.945218954e-01f}, {2.832769449e-16f, 1.000000000e+00f}, {-1.045284633e-01f, 9.945218954e-01f}, {-2.079116908e-01f, 9.781476007e-01f}, {-3.090169944e-01f, 9.510565163e-01f}, {-4.067366431e-01f, 9.135454576e-01f}, {-5.000000000e-01f, 8.660254038e-01f}, {-5.877852523e-01f, 8.090169944e-01f}, {-6.691306064e-01f, 7.431448255e-01f}, {-7.431448255e-01f, 6.691306064e-01f}, {-8.090169944e-01f, 5.877852523e-01f}, {-8.660254038e-01f, 5.000000000e-01f}, {-9.135454576e-01f, 4.067366431e-01f}, {-9.510565163e-01f, 3.090169944e-01f}, {-9.781476007e-01f, 2.079116908e-01f}, {-9.945218954e-01f, 1.045284633e-01f}, {-1.000000000e+00f, 5.665538898e-16f}, {-9.945218954e-01f, -1.045284633e-01f}, {-9.781476007e-01f, -2.079116908e-01f}, {-9.510565163e-01f, -3.090169944e-01f}, {-9.135454576e-01f, -4.067366431e-01f}, {-8.660254038e-01f, -5.000000000e-01f}, {-8.090169944e-01f, -5.877852523e-01f}, {-7.431448255e-01f, -6.691306064e-01f}, {-6.691306064e-01f, -7.431448255e-01f}, {-5.877852523e-01f, -8.090169944e-01f}, {-5.000000000e-01f, -8.660254038e-01f}, {-4.067366431e-01f, -9.135454576e-01f}, {-3.090169944e-01f, -9.510565163e-01f}, {-2.079116908e-01f, -9.781476007e-01f}, {-1.045284633e-01f, -9.945218954e-01f}, {-1.836970199e-16f, -1.000000000e+00f}, {1.045284633e-01f, -9.945218954e-01f}, {2.079116908e-01f, -9.781476007e-01f}, {3.090169944e-01f, -9.510565163e-01f}, {4.067366431e-01f, -9.135454576e-01f}, {5.000000000e-01f, -8.660254038e-01f}, {5.877852523e-01f, -8.090169944e-01f}, {6.691306064e-01f, -7.431448255e-01f}, {7.431448255e-01f, -6.691306064e-01f}, {8.090169944e-01f, -5.877852523e-01f}, {8.660254038e-01f, -5.000000000e-01f}, {9.135454576e-01f, -4.067366431e-01f}, {9.510565163e-01f, -3.090169944e-01f}, {9.781476007e-01f, -2.079116908e-01f}, {9.945218954e-01f, -1.045284633e-01f}}};
template <int N> struct smallest_factor { static constexpr int v = (N % 2 == 0) ? 2 : (N % 3 == 0) ? 3 : (N % 5 == 0) ? 5 : N; };
// in: element j of this sub-problem is src[OFF + STRIDE * j]; out: dst[0..N) natural order
template <int N, int NTOP, int OFF, int STRIDE, class C, class TWF>
__device__ __forceinline__ void dft_rec(const C* src, C* dst, TWF&& tw) {
  if constexpr (N == 1) {
    dst[0] = src[OFF];
  } else {
    constexpr int P = smallest_factor<N>::v, M = N / P;
    C sub[P][M];
    static_for<P>([&](auto sc) {
      constexpr int s = decltype(sc)::v;
      dft_rec<M, NTOP, OFF + STRIDE * s, STRIDE * P>(src, sub[s], tw);
    });
    static_for<M>([&](auto kc) {
      constexpr int k = decltype(kc)::v;
      C t[P];
      static_for<P>([&](auto sc) {
        constexpr int s = decltype(sc)::v;
        constexpr int e = (s * k * (NTOP / N)) % NTOP;          // W_N^{s k}
        t[s] = e == 0 ? sub[s][k] : cmul(sub[s][k], tw(e));
      });
      static_for<P>([&](auto jc) {
        constexpr int j = decltype(jc)::v;
        C acc = t[0];
        static_for<P - 1>([&](auto sc) {
          constexpr int s = decltype(sc)::v + 1;
          constexpr int e = ((s * j) % P) * (NTOP / P);           // W_P^{s j}
          acc = cadd(acc, e == 0 ? t[s] : cmul(t[s], tw(e)));
        });
        dst[k + M * j] = acc;
      });
    });
  }
}

template <int LW>
__device__ __forceinline__ void dft_lanes(cf32 (&z)[64]) {   // inverse DFT of z[0..LW) in place, natural order out
  if constexpr (LW == 64) {
    dft_reg<64>(z);
    cf32 t[64];
    static_for<64>([&](auto ic) { t[decltype(ic)::v] = z[bitrev_c(decltype(ic)::v, 6)]; });
    static_for<64>([&](auto ic) { z[decltype(ic)::v] = t[decltype(ic)::v]; });
  } else {
    cf32 t[LW];
    dft_rec<LW, LW, 0, 1>(z, t, [](int e) { return cf32{kTw60.c[e][0], kTw60.c[e][1]}; });
    static_for<LW>([&](auto ic) { z[decltype(ic)::v] = t[decltype(ic)::v]; });
  }
}

struct ScreenSynthArgs {
  float2* T;                 // [env in batch][m/2 + 1][N] complex64
  float* out;                // [env in batch][N][N]
  int N, q, first_local, env_base;   // envs [first_local, ...) of the handle; global id = env_base + local index
  unsigned long long seed;
  const uint32_t* gen;               // [B] screens drawn so far per env (see k_spectrum_fill)
  float du, u0sq, amp_scale, crop_scale;
};

// the shared transform: `load(bb_global, x)` supplies the samples x[r] = (a = lane + LW r, b = bb_global), r < R, already multiplied by (-1)^a;
// on return acc[p] (p < R) holds out[(p + R * lane) - N/2 ... i.e. output index i = p + R * lane of the centred crop.
template <int R, int LW, class Load>
__device__ __forceinline__ void pruned_line(Load&& load, int q, int N, float* __restrict__ lbuf, cf32 (&acc)[R]) {
  // lbuf: this wave's private [64][65] float plane; real and imaginary parts cross it one after the other (half the LDS of a
  // complex plane: two workgroups fit a CU)
  constexpr int BCmax = 64 / R;
  const int lane = threadIdx.x & 63;   // lanes LW .. 63 idle in the per-sample phases (LW = 60 for N = 60 R)
  const int m = q * N;
  const int BC = min(q, BCmax);
  auto lds_fence = [] {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS traffic has landed (the plane is private to the wave)
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int p = 0; p < R; ++p) acc[p] = cf32{0.f, 0.f};
  // W_N^{lane p}, p < R
  cf32 wl[R];
#pragma unroll
  for (int p = 0; p < R; ++p) {
    float sn, cs;
    __sincosf(6.2831853071795865f * (float)(lane * p) / (float)N, &sn, &cs);
    wl[p] = cf32{cs, sn};
  }
  for (int b0 = 0; b0 < q; b0 += BC) {
    // 1) radix-R over r, twiddle, first transpose: sequence s = p * BC + bb holds one point per lane
    cf32 z[64];
    const bool active = lane < R * BC;
    const float* row = lbuf + (size_t)(active ? lane : 0) * 65;
    // imaginary parts wait in z[].y (not live yet) while the real parts cross; compile-time indices keep them in registers
    static_for<BCmax>([&](auto bc) {
      constexpr int bb = decltype(bc)::v;
      if (bb < BC) {
        cf32 x[R];
        load(b0 + bb, x);   // the R samples a = lane + LW r of column group b0 + bb
        dft_reg<R>(x);
#pragma unroll
        for (int p = 0; p < R; ++p) {
          const cf32 y = cmul(x[bitrev_c(p, log2_c(R))], wl[p]);
          if (LW == 64 || lane < LW) lbuf[(p * BC + bb) * 65 + lane] = y.x;
          z[bb * R + p].y = y.y;
        }
      }
    });
    lds_fence();
#pragma unroll
    for (int t = 0; t < LW; ++t) z[t].x = row[t];
    lds_fence();
    static_for<BCmax>([&](auto bc) {
      constexpr int bb = decltype(bc)::v;
      if (bb < BC) {
#pragma unroll
        for (int p = 0; p < R; ++p)
          if (LW == 64 || lane < LW) lbuf[(p * BC + bb) * 65 + lane] = z[bb * R + p].y;
      }
    });
    lds_fence();
#pragma unroll
    for (int t = 0; t < LW; ++t) z[t].y = row[t];
    // 2) every lane s < R * BC owns an LW-point sequence
    dft_lanes<LW>(z);
    // 3) b-twiddles e^{2 pi i b (i - N/2) / m}, i = p + R i2, by recurrence over i2 (exact restart every 16 steps)
    const int p_of = lane / BC, b_of = b0 + (lane - p_of * BC);
    cf32 tw, step;
    {
      float sn, cs;
      __sincosf(6.2831853071795865f * (float)b_of * ((float)(p_of - N / 2) / (float)m), &sn, &cs);
      tw = cf32{cs, sn};
      __sincosf(6.2831853071795865f * (float)b_of * ((float)R / (float)m), &sn, &cs);
      step = cf32{cs, sn};
    }
    static_for<LW>([&](auto tc) {
      constexpr int i2 = decltype(tc)::v;
      z[i2] = cmul(z[i2], tw);
      if constexpr ((i2 & 15) == 15 && i2 != LW - 1) {
        float sn, cs;
        __sincosf(6.2831853071795865f * (float)b_of * ((float)(p_of + R * (i2 + 1) - N / 2) / (float)m), &sn, &cs);
        tw = cf32{cs, sn};
      } else {
        tw = cmul(tw, step);
      }
    });
    // 4) second transpose (real plane, then imaginary): lane j sums over the b's of this group for its R outputs i = p + R j
    lds_fence();
    static_for<LW>([&](auto tc) {
      constexpr int i2 = decltype(tc)::v;
      if (active) lbuf[lane * 65 + i2] = z[i2].x;
    });
    lds_fence();
#pragma unroll
    for (int p = 0; p < R; ++p) {
      float sum = acc[p].x;
      for (int bb = 0; bb < BC; ++bb) sum += lbuf[(p * BC + bb) * 65 + lane];
      acc[p].x = sum;
    }
    lds_fence();
    static_for<LW>([&](auto tc) {
      constexpr int i2 = decltype(tc)::v;
      if (active) lbuf[lane * 65 + i2] = z[i2].y;
    });
    lds_fence();
#pragma unroll
    for (int p = 0; p < R; ++p) {
      float sum = acc[p].y;
      for (int bb = 0; bb < BC; ++bb) sum += lbuf[(p * BC + bb) * 65 + lane];
      acc[p].y = sum;
    }
    lds_fence();
  }
}

// Pass A: grid ((m/2 + 1) / 4, envs in batch), 4 waves, one spectrum line v per wave.  LDS 4 x 64 x 65 x 4 B.  Two workgroups per CU
// (launch bound: 256 registers; left free the compiler takes 313 and one wave per SIMD runs 15 % slower than two with ~50 spills
// outside the hot code).
// Only the lines 0 <= v <= m/2 are drawn.  The screen is the REAL part of the transform of independent complex normals a(k) g_k, and
// a(k) = a(-k): the pair (k, -k) contributes a(k) [(g_k.re + g_-k.re) cos - (g_k.im - g_-k.im) sin], in which the two bracketed
// sums are independent N(0, 2) — the same law as sqrt(2) a(k) g_k alone.  So lines 0 < v < m/2 carry sqrt(2) a and stand for their
// conjugate lines m - v as well; lines 0 and m/2 pair within themselves and are drawn in full as before.  Half the Philox draws,
// half the row transforms, same distribution of the screens (not the same sample stream as the full-plane form).
template <int R, int LW>
__global__ __launch_bounds__(256, 2) void k_screen_rows(ScreenSynthArgs p) {
  extern __shared__ float lds_syn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = p.N, q = p.q, m = q * N;
  const int v = blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  const int lines = m / 2 + 1;
  if (v >= lines) return;
  const float fv = p.du * (float)v;
  // (-1)^a, a = lane + LW r (LW is even), times the half-plane weight of this line
  const float sign = ((lane & 1) ? -1.f : 1.f) * ((v == 0 || 2 * v == m) ? 1.f : 1.41421356237f);
  const uint32_t generation = p.gen[p.first_local + b] + 1u;
  const uint32_t env_global = (uint32_t)(p.env_base + p.first_local + b);
  const int a0 = min(lane, LW - 1);   // (idle lanes compute a duplicate that is never stored)
  auto load = [&](int bg, cf32 (&x)[R]) {
    // one Philox call per four samples of this lane (see spectrum_words)
    static_for<(R + 3) / 4>([&](auto gc) {
      constexpr int rg = decltype(gc)::v;
      uint32_t w[4];
      spectrum_words((size_t)v * m + (size_t)q * (a0 + LW * 4 * rg) + bg, generation, env_global, p.seed, w);
      static_for<4>([&](auto jc) {
        constexpr int r = 4 * rg + decltype(jc)::v;
        if constexpr (r < R) {
          const float2 o = spectrum_sample(w[decltype(jc)::v], q * (a0 + LW * r) + bg, m, fv, p.du, p.u0sq, p.amp_scale);
          x[r] = cf32{sign * o.x, sign * o.y};
        }
      });
    });
  };
  cf32 acc[R];
  pruned_line<R, LW>(load, q, N, lds_syn + (size_t)wave * 64 * 65, acc);
  float2* dst = p.T + ((size_t)b * lines + v) * N + (size_t)R * lane;
  if (lane < LW) {
#pragma unroll
    for (int pp = 0; pp < R; ++pp) dst[pp] = make_float2(acc[pp].x, acc[pp].y);
  }
}

// Pass B: grid (N / 8, envs in batch), one column ix per wave; reads T[v][ix]: the 8 waves of a workgroup take 8 adjacent columns,
// 64 B of every line of T (T does not fit the L2: with 4 columns per workgroup half of every fetched sector was unused).
constexpr int kColsWaves = 8;
template <int R, int LW>
__global__ __launch_bounds__(64 * kColsWaves) void k_screen_cols(ScreenSynthArgs p) {
  extern __shared__ float lds_syn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = p.N, q = p.q, m = q * N;
  const int ix = blockIdx.x * kColsWaves + wave;
  const int b = blockIdx.y;
  if (ix >= N) return;
  const float sign = (lane & 1) ? -1.f : 1.f;
  const int lines = m / 2 + 1;
  const float2* src = p.T + (size_t)b * lines * N + ix;
  auto load = [&](int bg, cf32 (&x)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int vv = q * (min(lane, LW - 1) + LW * r) + bg;
      float2 t = make_float2(0.f, 0.f);   // the conjugate half plane is folded into the lines below m/2 (see k_screen_rows)
      if (vv < lines) t = src[(size_t)vv * N];
      x[r] = cf32{sign * t.x, sign * t.y};
    }
  };
  cf32 acc[R];
  pruned_line<R, LW>(load, q, N, lds_syn + (size_t)wave * 64 * 65, acc);
  float* dst = p.out + (size_t)b * N * N + ix;
  if (lane < LW) {
#pragma unroll
    for (int pp = 0; pp < R; ++pp) dst[(size_t)(pp + R * lane) * N] = acc[pp].x * p.crop_scale;
  }
}

// ------------------------------------------------------------------------------------------------
// K8 (two-band form, the default of aog_generate_screens)  The literal method spends 16^2 = 256 spectrum samples per output pixel
// because ONE grid has to be fine enough for the outer scale (period 16 D) and reach the pixel Nyquist frequency.  The screen is a
// stationary Gaussian field, so it is the sum of two INDEPENDENT stationary Gaussian fields whose spectra add up to the literal one:
//   low band   hcipy's own (q N)^2 grid, variance a^2(k) w_low(|k|):  non-zero only for |k| < f2 = 2 q  (2 cycles per pupil diameter),
//              KL = 2 q half-plane lines of 2 KL samples, evaluated directly (a few thousand terms per line);
//   high band  the (2 N)^2 grid (period 2 D), variance PSD du_H^2 / (2 pi)^2 w_high(|u|): the high-passed covariance has decayed to
//              ~1e-5 of the variance at the wrap-around lag N + 1, so the coarser frequency grid changes the covariance on the N x N
//              crop by < 1e-4 C(0) (tests/test_screen_twoband.py evaluates both covariances exactly, in float64, at every lag).
// 4 N^2 + 4 KL^2 samples instead of 256 N^2.  The high band runs through the same pruned two-pass transform with q = 2; a line then
// fills only 2 R of the 64 lanes of the 64-point stage, so one wave carries NL = 32 / R lines (pass A) or columns (pass B) at once.
// ------------------------------------------------------------------------------------------------
// T is stored in column tiles of NL = 32 / R outputs: element (line v, output i) of an env at ((i / NL) * linesT + v) * NL + i % NL, so
// that pass B's wave (NL adjacent columns, a line per lane) reads ONE contiguous run of linesT x NL x 8 bytes instead of 64 bytes out of
// every 2 KB (PMC of the row-major form: two thirds of pass B's wave-cycles waiting for those loads)
__host__ __device__ inline size_t screen2_T_elems(int N, int KL, int NL) { return (size_t)((N + NL - 1) / NL) * (size_t)(N + 1 + KL) * NL; }
struct Screen2Args {
  float2* T;                 // [env in batch][column tile][N + 1 + KL][NL] complex64: high-band lines 0 .. N after pass A, then the KL low-band lines
  float* out;                // [env in batch][N][N]
  int N, qf, KL, first_local, env_base;
  unsigned long long seed;
  const uint32_t* gen;       // [B] screens drawn so far per env
  float duH, duL, u0sq, ampH, ampL;   // frequency steps of the two grids; amplitudes in the screen's final unit (sqrt(PSD) du / 2 pi sqrt(Cn^2))
  BandWindow win;
};

// NL = 64 / (R Q) lines of length m = Q N at once: virtual column group g = b * NL + line (b < Q), sequence s = p * BC + g, BC = 64 / R.
// `load(IC<g>, x)` supplies the samples x[r] = (a = lane + LW r, column group b) of line g % NL, already multiplied by (-1)^a.
// On return acc[line][p] holds output i = p + R * lane of the centred crop (REAL_ONLY: only .x).
template <int R, int LW, int Q, bool REAL_ONLY, class Load>
__device__ __forceinline__ void pruned_lines_multi(Load&& load, int N, float* __restrict__ lbuf, cf32 (&acc)[64 / (R * Q)][R]) {
  constexpr int BC = 64 / R, NL = BC / Q;
  static_assert(R * BC == 64 && NL * Q == BC, "every lane of the 64-point stage owns one sequence");
  const int lane = threadIdx.x & 63;
  const int m = Q * N;
  auto lds_fence = [] {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the plane is private to the wave
    __builtin_amdgcn_wave_barrier();
  };
  cf32 wl[R];
#pragma unroll
  for (int p = 0; p < R; ++p) {
    float sn, cs;
    __sincosf(6.2831853071795865f * (float)(lane * p) / (float)N, &sn, &cs);
    wl[p] = cf32{cs, sn};
  }
  cf32 z[64];
  const float* row = lbuf + (size_t)lane * 65;
  static_for<BC>([&](auto gc) {
    constexpr int g = decltype(gc)::v;
    cf32 x[R];
    load(gc, x);
    dft_reg<R>(x);
#pragma unroll
    for (int p = 0; p < R; ++p) {
      const cf32 y = cmul(x[bitrev_c(p, log2_c(R))], wl[p]);
      if (LW == 64 || lane < LW) lbuf[(p * BC + g) * 65 + lane] = y.x;
      z[g * R + p].y = y.y;
    }
  });
  lds_fence();
#pragma unroll
  for (int t = 0; t < LW; ++t) z[t].x = row[t];
  lds_fence();
  static_for<BC>([&](auto gc) {
    constexpr int g = decltype(gc)::v;
#pragma unroll
    for (int p = 0; p < R; ++p)
      if (LW == 64 || lane < LW) lbuf[(p * BC + g) * 65 + lane] = z[g * R + p].y;
  });
  lds_fence();
#pragma unroll
  for (int t = 0; t < LW; ++t) z[t].y = row[t];
  dft_lanes<LW>(z);
  // b-twiddles e^{2 pi i b (i - N/2) / m}, i = p + R i2 (b = 0: identity)
  const int p_of = lane / BC, b_of = (lane - p_of * BC) / NL;
  cf32 tw, step;
  {
    float sn, cs;
    __sincosf(6.2831853071795865f * (float)b_of * ((float)(p_of - N / 2) / (float)m), &sn, &cs);
    tw = cf32{cs, sn};
    __sincosf(6.2831853071795865f * (float)b_of * ((float)R / (float)m), &sn, &cs);
    step = cf32{cs, sn};
  }
  static_for<LW>([&](auto tc) {
    constexpr int i2 = decltype(tc)::v;
    if constexpr (REAL_ONLY) z[i2].x = z[i2].x * tw.x - z[i2].y * tw.y;
    else z[i2] = cmul(z[i2], tw);
    if constexpr ((i2 & 15) == 15 && i2 != LW - 1) {
      float sn, cs;
      __sincosf(6.2831853071795865f * (float)b_of * ((float)(p_of + R * (i2 + 1) - N / 2) / (float)m), &sn, &cs);
      tw = cf32{cs, sn};
    } else {
      tw = cmul(tw, step);
    }
  });
  // second transpose: lane j sums the Q column groups of every line for its R outputs i = p + R j
  lds_fence();
  static_for<LW>([&](auto tc) { lbuf[lane * 65 + decltype(tc)::v] = z[decltype(tc)::v].x; });
  lds_fence();
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int p = 0; p < R; ++p) {
      float sum = 0.f;
#pragma unroll
      for (int bb = 0; bb < Q; ++bb) sum += lbuf[(p * BC + bb * NL + l) * 65 + lane];
      acc[l][p].x = sum;
    }
  lds_fence();
  if constexpr (!REAL_ONLY) {
    static_for<LW>([&](auto tc) { lbuf[lane * 65 + decltype(tc)::v] = z[decltype(tc)::v].y; });
    lds_fence();
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int p = 0; p < R; ++p) {
        float sum = 0.f;
#pragma unroll
        for (int bb = 0; bb < Q; ++bb) sum += lbuf[(p * BC + bb * NL + l) * 65 + lane];
        acc[l][p].y = sum;
      }
    lds_fence();
  }
}

// e^{2 pi i (k x mod M) / M} with the product reduced exactly in integers
__device__ __forceinline__ cf32 unit_root(int k, int x, int M) {
  float sn, cs;
  __sincosf(6.2831853071795865f * ((float)((k * x) % M) / (float)M), &sn, &cs);
  return cf32{cs, sn};
}

// Pass A: grid (high-band blocks + low-band blocks, envs in batch), 4 waves.  A high-band wave draws NL spectrum lines of the (2N)^2 grid
// (half plane v <= N, the lines 0 < v < N carry sqrt(2) x the amplitude like the literal form) and writes T[v][i]; a low-band wave draws
// one line ky of the fine grid, kx in [-KL, KL), and sums it directly: T[N + 1 + ky][i] = sum_kx c e^{2 pi i kx (i - N/2) / (q N)}.
template <int R, int LW>
__global__ __launch_bounds__(256, 2) void k_screen2_rows(Screen2Args p) {
  extern __shared__ float lds_syn[];
  constexpr int Q = 2, BC = 64 / R, NL = BC / Q;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = p.N, m = Q * N, linesH = N + 1;
  const int b = blockIdx.y;
  const int nHgroups = (linesH + NL - 1) / NL, nHblocks = (nHgroups + 3) / 4;
  const uint32_t generation = p.gen[p.first_local + b] + 1u;
  const uint32_t env_global = (uint32_t)(p.env_base + p.first_local + b);
  const int linesT = linesH + p.KL;
  float2* Tenv = p.T + (size_t)b * screen2_T_elems(N, p.KL, NL);
  auto t_at = [&](int v, int i) { return Tenv + ((size_t)(i / NL) * linesT + v) * NL + (i % NL); };
  if ((int)blockIdx.x < nHblocks) {
    const int grp = blockIdx.x * 4 + wave;
    if (grp >= nHgroups) return;
    const int v0 = grp * NL;
    const int a0 = min(lane, LW - 1);   // (idle lanes compute a duplicate that is never stored)
    const float sgn = (lane & 1) ? -1.f : 1.f;
    // lines with |fv| >= f2 lie wholly in the pass band of the high window (v du_H = v (q/2) du_fine >= KL du_fine)
    const bool windowed = v0 * (p.qf / 2) < p.KL;
    auto load = [&](auto gc, cf32 (&x)[R]) {
      constexpr int g = decltype(gc)::v, bg = g / NL, l = g % NL;
      const int v = v0 + l;
      if (v >= linesH) {
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = cf32{0.f, 0.f};
        return;
      }
      const float fv = p.duH * (float)v;
      const float ls = sgn * ((v == 0 || 2 * v == m) ? 1.f : 1.41421356237f);
      static_for<(R + 3) / 4>([&](auto rgc) {
        constexpr int rg = decltype(rgc)::v;
        uint32_t w[4];
        spectrum_words((size_t)v * m + (size_t)Q * (a0 + LW * 4 * rg) + bg, generation, env_global, p.seed, w);
        static_for<4>([&](auto jc) {
          constexpr int r = 4 * rg + decltype(jc)::v;
          if constexpr (r < R) {
            const int uu = Q * (a0 + LW * r) + bg;
            const float fu = p.duH * (float)(uu < m / 2 ? uu : uu - m);
            const float2 o = windowed ? band_sample<1>(w[decltype(jc)::v], fu, fv, p.u0sq, p.ampH, p.win)
                                      : band_sample<0>(w[decltype(jc)::v], fu, fv, p.u0sq, p.ampH, p.win);
            x[r] = cf32{ls * o.x, ls * o.y};
          }
        });
      });
    };
    cf32 acc[NL][R];
    pruned_lines_multi<R, LW, Q, false>(load, N, lds_syn + (size_t)wave * 64 * 65, acc);
    if (lane < LW) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        if (v0 + l < linesH) {
#pragma unroll
          for (int pp = 0; pp < R; ++pp) *t_at(v0 + l, R * lane + pp) = make_float2(acc[l][pp].x, acc[l][pp].y);
        }
      }
    }
    return;
  }
  const int ky = ((int)blockIdx.x - nHblocks) * 4 + wave;
  const int KL = p.KL, Mf = p.qf * N;
  if (ky >= KL) return;
  const float fv = p.duL * (float)ky;
  const float ls = ky == 0 ? 1.f : 1.41421356237f;
  cf32 acc[R], st[R], tw[R];
  int xp[R];
#pragma unroll
  for (int pp = 0; pp < R; ++pp) {
    acc[pp] = cf32{0.f, 0.f};
    xp[pp] = pp + R * min(lane, LW - 1) - N / 2;
    st[pp] = unit_root(1, xp[pp], Mf);
    tw[pp] = st[pp];
  }
  for (int c0 = 0; c0 < 2 * KL; c0 += 64) {
    const int kxi = c0 + lane;   // kx = kxi - KL
    float2 smp = make_float2(0.f, 0.f);
    if (kxi < 2 * KL) {
      uint32_t w[4];
      spectrum_words((size_t)ky * (size_t)(2 * KL) + (size_t)kxi, generation, env_global, p.seed, w, kSpectrumTagLow);
      smp = band_sample<2>(w[0], p.duL * (float)(kxi - KL), fv, p.u0sq, p.ampL, p.win);
      smp.x *= ls;
      smp.y *= ls;
    }
    const int nk = min(64, 2 * KL - c0);
    for (int k = 0; k < nk; ++k) {
      if ((k & 15) == 0) {   // exact restart of the recurrence
#pragma unroll
        for (int pp = 0; pp < R; ++pp) tw[pp] = unit_root(c0 + k - KL, xp[pp], Mf);
      }
      const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, smp.x), k));
      const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, smp.y), k));
#pragma unroll
      for (int pp = 0; pp < R; ++pp) {
        acc[pp].x = fmaf(cx, tw[pp].x, fmaf(-cy, tw[pp].y, acc[pp].x));
        acc[pp].y = fmaf(cx, tw[pp].y, fmaf(cy, tw[pp].x, acc[pp].y));
        tw[pp] = cmul(tw[pp], st[pp]);
      }
    }
  }
  if (lane < LW) {
#pragma unroll
    for (int pp = 0; pp < R; ++pp) *t_at(linesH + ky, R * lane + pp) = make_float2(acc[pp].x, acc[pp].y);
  }
}

// Pass B: grid (ceil(N / (8 NL)), envs in batch), 8 waves, NL adjacent columns per wave: the transform down the columns of the high-band
// lines (two columns per 16-byte load), then the low band's KL lines summed directly (their values are wave-uniform), Re(.) written.
template <int R, int LW>
__global__ __launch_bounds__(64 * kColsWaves) void k_screen2_cols(Screen2Args p) {
  extern __shared__ float lds_syn[];
  constexpr int Q = 2, BC = 64 / R, NL = BC / Q;
  static_assert(NL >= 4 && NL % 4 == 0, "columns are loaded in pairs and stored in fours");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = p.N, linesH = N + 1, linesT = linesH + p.KL;
  const int ix0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kColsWaves + wave) * NL);
  const int b = blockIdx.y;
  if (ix0 >= N) return;
  const float sgn = (lane & 1) ? -1.f : 1.f;
  const float2* __restrict__ Tenv = p.T + (size_t)b * screen2_T_elems(N, p.KL, NL) + (size_t)(ix0 / NL) * linesT * NL;   // this wave's column tile
  const int a0 = min(lane, LW - 1);
  cf32 pend[R];
  auto load = [&](auto gc, cf32 (&x)[R]) {
    constexpr int g = decltype(gc)::v, bg = g / NL, l = g % NL;
    if constexpr ((l & 1) == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int vv = Q * (a0 + LW * r) + bg;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);   // the conjugate half plane is folded into the lines 0 .. N
        if (vv < linesH && ix0 + l < N) t = *reinterpret_cast<const float4*>(Tenv + (size_t)vv * NL + l);
        x[r] = cf32{sgn * t.x, sgn * t.y};
        pend[r] = cf32{sgn * t.z, sgn * t.w};
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = pend[r];
    }
  };
  cf32 acc[NL][R];
  pruned_lines_multi<R, LW, Q, true>(load, N, lds_syn + (size_t)wave * 64 * 65, acc);
  // low band: out[y][ix] += Re sum_ky T_L[ky][ix] e^{2 pi i ky (y - N/2) / (q N)},  y = pp + R lane
  const int Mf = p.qf * N;
  cf32 st[R], tw[R];
  int yp[R];
#pragma unroll
  for (int pp = 0; pp < R; ++pp) {
    yp[pp] = pp + R * a0 - N / 2;
    st[pp] = unit_root(1, yp[pp], Mf);
    tw[pp] = cf32{1.f, 0.f};
  }
  const float2* __restrict__ TL = Tenv + (size_t)linesH * NL;
  // the low-band values of a line are wave-uniform (scalar loads): line ky + 1 is requested before line ky is used, otherwise every
  // iteration starts with a scalar-memory round trip that two waves per SIMD cannot cover (PMC: 66 % of the wave-cycles waiting)
  float2 cn[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) cn[l] = TL[l];   // (columns past N of a last partial tile: in the workspace, never used)
  for (int ky = 0; ky < p.KL; ++ky) {
    if ((ky & 15) == 0 && ky) {
#pragma unroll
      for (int pp = 0; pp < R; ++pp) tw[pp] = unit_root(ky, yp[pp], Mf);
    }
    float2 cc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) cc[l] = cn[l];
    const int kn = min(ky + 1, p.KL - 1);
#pragma unroll
    for (int l = 0; l < NL; ++l) cn[l] = TL[(size_t)kn * NL + l];
#pragma unroll
    for (int l = 0; l < NL; ++l)   // (columns past N of a last partial tile carry workspace garbage through: never stored)
#pragma unroll
      for (int pp = 0; pp < R; ++pp) acc[l][pp].x = fmaf(cc[l].x, tw[pp].x, fmaf(-cc[l].y, tw[pp].y, acc[l][pp].x));
#pragma unroll
    for (int pp = 0; pp < R; ++pp) tw[pp] = cmul(tw[pp], st[pp]);
  }
  if (lane < LW) {
#pragma unroll
    for (int pp = 0; pp < R; ++pp) {
      float* dst = p.out + (size_t)b * N * N + (size_t)(pp + R * lane) * N + ix0;
#pragma unroll
      for (int l = 0; l < NL; l += 4)
        if (ix0 + l < N) *reinterpret_cast<float4*>(dst + l) = make_float4(acc[l][pp].x, acc[l + 1][pp].x, acc[l + 2][pp].x, acc[l + 3][pp].x);
    }
  }
}

#ifdef AOG_MAIN_TU
// Low band on the general route (pupils the pruned passes do not cover, and the equivalence test): spectrum, lines, sum — one thread per
// output, every twiddle from an exactly reduced integer product.
__global__ void k_lowband_spectrum(float2* __restrict__ c, Screen2Args p) {   // c: [env in batch][KL][2 KL]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  const int KL = p.KL;
  if (idx >= KL * 2 * KL) return;
  const int ky = idx / (2 * KL), kxi = idx - ky * 2 * KL;
  uint32_t w[4];
  spectrum_words((size_t)ky * (size_t)(2 * KL) + (size_t)kxi, p.gen[p.first_local + b] + 1u, (uint32_t)(p.env_base + p.first_local + b), p.seed, w,
                 kSpectrumTagLow);
  float2 s = band_sample<2>(w[0], p.duL * (float)(kxi - KL), p.duL * (float)ky, p.u0sq, p.ampL, p.win);
  const float ls = ky == 0 ? 1.f : 1.41421356237f;
  c[(size_t)b * KL * 2 * KL + idx] = make_float2(ls * s.x, ls * s.y);
}
__global__ void k_lowband_lines(const float2* __restrict__ c, float2* __restrict__ TL, int N, int KL, int Mf) {   // TL: [env in batch][KL][N]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (idx >= KL * N) return;
  const int ky = idx / N, i = idx - ky * N;
  const float2* line = c + ((size_t)b * KL + ky) * 2 * KL;
  float re = 0.f, im = 0.f;
  for (int kxi = 0; kxi < 2 * KL; ++kxi) {
    const cf32 t = unit_root(kxi - KL, i - N / 2, Mf);
    const float2 v = line[kxi];
    re = fmaf(v.x, t.x, fmaf(-v.y, t.y, re));
    im = fmaf(v.x, t.y, fmaf(v.y, t.x, im));
  }
  TL[(size_t)b * KL * N + idx] = make_float2(re, im);
}
__global__ void k_lowband_add(const float2* __restrict__ TL, float* __restrict__ out, int N, int KL, int Mf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (idx >= N * N) return;
  const int iy = idx / N, ix = idx - iy * N;
  float re = 0.f;
  for (int ky = 0; ky < KL; ++ky) {
    const cf32 t = unit_root(ky, iy - N / 2, Mf);
    const float2 v = TL[((size_t)b * KL + ky) * N + ix];
    re = fmaf(v.x, t.x, fmaf(-v.y, t.y, re));
  }
  out[(size_t)b * N * N + idx] += re;
}
#endif  // AOG_MAIN_TU

// ------------------------------------------------------------------------------------------------
// K10  Shack-Hartmann chain (AO_env.py:254-290): field on the magnified pupil x micro-lens phase -> angular-spectrum
// Fresnel propagation over one lenslet focal length (2x zero-padded hipFFT, float64) -> detector image -> photon noise ->
// centre of gravity per selected lenslet -> reconstructor GEMV + leaky integrator.
// ------------------------------------------------------------------------------------------------
#ifdef AOG_MAIN_TU
// CT = double2 (complex128 transforms) or float2 (complex64: the default — the detector image is photon-noise limited at 1e-3, see aog_sh_tables)
template <typename CT>
__global__ void k_sh_field(const float* __restrict__ phase_tile, const int32_t* __restrict__ ap_index, const double2* __restrict__ mla_phase,
                           CT* __restrict__ pad, int n_ap, int n_ptiles, int N, double amplitude, size_t env_stride, int row_stride) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = blockIdx.y;
  if (p >= n_ap) return;
  const double rev = (double)phase_tile[psi_tile_index(env, p, n_ptiles)];   // revolutions at lambda_wfs (k_phase_mfma)
  double sn, cs;
  sincospi(2.0 * (rev - rint(rev)), &sn, &cs);
  const int flat = ap_index[p];
  const int iy = flat / N, ix = flat - iy * N;
  const double2 m = mla_phase[flat];
  // E * mla: (cs + i sn) * (m.x + i m.y)
  CT v;
  v.x = (decltype(v.x))(amplitude * (cs * m.x - sn * m.y));
  v.y = (decltype(v.y))(amplitude * (cs * m.y + sn * m.x));
  pad[(size_t)env * env_stride + (size_t)iy * row_stride + ix] = v;   // zero-padded 2N x 2N (2-D transforms) or compact N x N (pruned passes)
}

// deformable_mirror_shack.actuators (metres, float64) -> the f16 hi/lo B-operand layout
__global__ void k_sh_act16(const double* __restrict__ sh_act, _Float16* __restrict__ act16, int B, int A, int A_pad, double two_over_lambda) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A_pad) return;
  const int env = idx / A_pad, i = idx % A_pad;
  store_act16(act16, env, i, A_pad, (i < A) ? (float)(sh_act[(size_t)env * A + i] * two_over_lambda) : 0.f);
}

template <typename CT>
__global__ void k_sh_transfer(CT* __restrict__ f, const CT* __restrict__ tf, size_t per_env) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_env) return;
  CT* v = f + (size_t)blockIdx.y * per_env + idx;
  const CT a = *v, b = tf[idx];
  CT o;
  o.x = a.x * b.x - a.y * b.y;
  o.y = a.x * b.y + a.y * b.x;
  *v = o;
}

template <typename CT>
__global__ void k_sh_intensity(const CT* __restrict__ f, double* __restrict__ image, int N, double scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = blockIdx.y;
  if (idx >= N * N) return;
  const int iy = idx / N, ix = idx - iy * N;
  const CT v = f[(size_t)env * 4 * N * N + (size_t)iy * 2 * N + ix];
  image[(size_t)env * N * N + idx] = ((double)v.x * (double)v.x + (double)v.y * (double)v.y) * scale;
}

// hcipy.util.large_poisson with the handle's Philox stream: exact inversion for lambda < 12, above it the rounded normal approximation with the
// Cornish-Fisher skewness term (hcipy switches to a plain rounded normal at 1e6; the sensor's controller reads flux-weighted centroids
// of ~1e3 pixels per lenslet: mean, variance and third moment of every pixel's count are those of the Poisson law).
// Stream layout: with x = l + LW r (LW = 64, or 60 for pupils of 60 R pixels: spectrum_lane_width), pixel (global env ge, row y, column x)
// takes word r & 3 of the Philox call with counter ((ge N + y) 64 + l, group r >> 2, call) — and, when it is bright, the same word of a second call for the Box-Muller angle.  The
// lane of the fused row pass that holds columns x, x + 64, x + 128, ... therefore draws ONE call per four of its pixels (a call per pixel
// with a float64 exp and a float64 inversion was ~350 instructions per pixel: two thirds of that pass); results do not depend on the
// batch split, nor on which kernel draws them.
__device__ __forceinline__ void sh_noise_words(size_t line, uint32_t group, bool second, unsigned long long seed, uint32_t call, uint32_t (&w)[4]) {
  uint32_t c[4] = {(uint32_t)line, (uint32_t)(line >> 32) ^ (group << 20) ^ (second ? 0x80000000u : 0u), call, 0x50155u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  w[0] = c[0]; w[1] = c[1]; w[2] = c[2]; w[3] = c[3];
}
constexpr double kShPoissonSwitch = 12.0;
// Poisson(lam), lam < 12, by inversion on a 32-bit uniform: k = number of partial sums of the pmf that stay below u.  The wave walks the
// terms in lockstep (k is wave-uniform, 1 / k is an immediate), FOUR terms per round of the "is any lane still below its u" vote, in fp32:
// the pmf recurrence p_k = p_{k-1} lam / k and its running sum carry ~1e-6 relative error, i.e. the sampled law differs from Poisson(lam)
// by ~1e-6 in total variation (the uniform is shrunk by 4e-6 so that the accumulated distribution always reaches it) — three orders
// below what a chi-square test on 1e6 draws resolves (tests: test_device_poisson_sampler_matches_scipy).  Round 2's form (float64 terms,
// one vote per term) spent ~70 cycles per term and was half of the fused row pass; this one spends ~25.  At most 48 terms: P(k > 47 | 12) < 1e-14.
template <int K0>
__device__ __forceinline__ void sh_poisson_terms(float lam, float u, float& pk, float& cdf, int& kres) {
  if (!__any(u > cdf ? 1 : 0)) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    kres += u > cdf ? 1 : 0;                   // (the sum only grows: once u <= cdf the lane stops counting)
    pk *= lam * (1.0f / (float)(K0 + j));      // compile-time reciprocal
    cdf += pk;
  }
  if constexpr (K0 + 4 < 48) sh_poisson_terms<K0 + 4>(lam, u, pk, cdf, kres);
}
__device__ __forceinline__ double sh_poisson_small(double lam_d, uint32_t word, bool active) {
  const float lam = (float)lam_d;
  const float u = active ? ((float)(word >> 8) + 0.5f) * (1.0f / 16777216.0f) * (1.0f - 4e-6f) : 0.0f;
  float pk = __expf(-lam), cdf = pk;
  int kres = 0;
  sh_poisson_terms<1>(lam, u, pk, cdf, kres);
  return (double)kres;
}
// rounded normal approximation with the Cornish-Fisher skewness term (matches mean, variance and third moment of Poisson(lam))
__device__ __forceinline__ double sh_poisson_large(double lam, uint32_t word_r, uint32_t word_a) {
  const float u1 = ((float)(word_r >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = (float)(word_a >> 8) * (1.0f / 16777216.0f);   // revolutions
  const float g = sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
  return fmax(0.0, rint(lam + (double)(g * sqrtf((float)lam) + (g * g - 1.0f) * (1.0f / 6.0f))));
}
// one pixel on its own (k_sh_noise: pupils the pruned passes do not cover, caller-visible images)
// sep_rl > 0 (handles on the separable two-pass propagation, whose last pass holds 32 rows y = p + RL k2 of ONE column per lane): the same
// scheme with the roles of the axes exchanged — pixel (ge, y, x) takes word k2 & 3 of the call with counter ((ge N + x) 64 + p, group k2 >> 2)
__device__ __forceinline__ double sh_noisy_value(double lam, size_t ge, int y, int x, int N, unsigned long long seed, uint32_t call, int sep_rl = 0) {
  const int lw = spectrum_lane_width(N);
  const size_t line = sep_rl ? (ge * N + x) * 64 + (size_t)(y % sep_rl) : (ge * N + y) * 64 + (x % lw);
  const uint32_t r = sep_rl ? (uint32_t)(y / sep_rl) : (uint32_t)(x / lw);
  uint32_t w[4];
  sh_noise_words(line, r >> 2, false, seed, call, w);
  const bool small = lam < kShPoissonSwitch;
  const double ks = sh_poisson_small(small ? lam : 0.0, w[r & 3], small);   // (every lane walks the wave's loop: no divergent call)
  if (small) return ks;
  uint32_t w2[4];
  sh_noise_words(line, r >> 2, true, seed, call, w2);
  return sh_poisson_large(lam, w[r & 3], w2[r & 3]);
}

// ---- pruned Fresnel propagation for pupils of N = 128, 256, 512 pixels (complex64) ------------------------------------------------------
// The 2-D route (zero-padded 2N x 2N field -> forward FFT -> x transfer function -> inverse FFT -> crop N x N) moves four full passes over
// the padded array per transform (rocFFT: 46 of the 76 ms of a config-5 iteration).  Three quarters of the forward input are zeros and
// three quarters of the inverse output are dropped, so the same arithmetic runs as three passes over HALF-size intermediates:
//   rows    field[iy][ix < N]  -> forward transform over x (length L = 2N, upper half of the input zero)  -> F1T (kx, iy < N), tiled
//   columns F1T                -> forward over y, x transfer[ky][kx], inverse over y, keep y < N          -> GT (kx, y < N), tiled
//   rows    GT                 -> inverse over kx, keep x < N, |.|^2 x scale                              -> image[y][x]     (float64)
// One wave transforms BC = 64 / RL lines of length L = 64 RL at a time (RL = 4, 8, 16), entirely in registers + one private LDS plane:
//   layout A: lane l holds elements l + 64 r (r < RL) of each of its BC lines             (contiguous in memory: coalesced rows)
//   layout B: lane (p, bb) = p BC + bb holds elements p + RL k2 (k2 < 64) of line bb
//   A -> B:  radix-RL over r in registers, twiddle W_L^{l p}, LDS transpose, 64-point transform in registers
//   B -> A:  64-point transform, LDS transpose, twiddle, radix-RL
// so a forward / inverse pair with the transfer function in between (the column pass) never leaves the registers, and the transposition
// between the passes happens in the layout of the intermediates: 512-byte tiles of RL columns x 64 / RL rows (see k_sh_rows_fwd).  Twiddles come from a table computed in float64 on the host.
// LW = 64 (lines of 64 RL) or 60 (lines of 60 RL: the reference's 240-pixel pupil): lanes LW .. 63 idle in the per-element phases and the
// in-register transform has LW points (mixed radix 2 x 2 x 3 x 5 for 60)
template <int RL, bool FWD, int LW = 64>
__device__ __forceinline__ void sh_fft_a2b(cf32 (&v)[64], float* __restrict__ lbuf, const float2* __restrict__ tw) {
  constexpr int BC = 64 / RL, LG = log2_c(RL);
  const int lane = threadIdx.x & 63;
  auto fence = [] {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the plane is private to the wave
    __builtin_amdgcn_wave_barrier();
  };
  // forward transform = swap(re, im) o inverse transform o swap(re, im): dft_reg is the e^{+} kernel
  if constexpr (FWD) static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; const float t = v[i].x; v[i].x = v[i].y; v[i].y = t; });
  cf32 wl[RL];
  static_for<RL>([&](auto pc) {
    constexpr int pp = decltype(pc)::v;
    if constexpr (pp > 0 && RL < 16) { const float2 t = tw[min(lane, LW - 1) * pp]; wl[pp] = cf32{t.x, t.y}; }
  });
  static_for<BC>([&](auto bc) {
    constexpr int bb = decltype(bc)::v;
    cf32 x[RL];
    static_for<RL>([&](auto rc) { x[decltype(rc)::v] = v[bb * RL + decltype(rc)::v]; });
    dft_reg<RL>(x);
    static_for<RL>([&](auto pc) {
      constexpr int pp = decltype(pc)::v;
      const cf32 y = x[bitrev_c(pp, LG)];
      if constexpr (pp == 0) v[bb * RL] = y;
      else if constexpr (RL >= 16) {   // (radix 16: the twiddles are re-read from the L1 per line instead of 32 registers held throughout)
        const float2 t = tw[min(lane, LW - 1) * pp];
        v[bb * RL + pp] = cmul(y, cf32{t.x, t.y});
      } else v[bb * RL + pp] = cmul(y, wl[pp]);
    });
  });
  float zx[LW];
  const bool owner = LW == 64 || lane < LW;
  static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; constexpr int bb = i / RL, pp = i % RL; if (owner) lbuf[(pp * BC + bb) * 65 + lane] = v[i].x; });
  fence();
  static_for<LW>([&](auto tc) { zx[decltype(tc)::v] = lbuf[lane * 65 + decltype(tc)::v]; });
  fence();
  static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; constexpr int bb = i / RL, pp = i % RL; if (owner) lbuf[(pp * BC + bb) * 65 + lane] = v[i].y; });
  fence();
  static_for<LW>([&](auto tc) { constexpr int t = decltype(tc)::v; v[t] = cf32{zx[t], lbuf[lane * 65 + t]}; });
  fence();
  cf32 o[64];
  if constexpr (LW == 64) {
    dft_reg<64>(v);
    static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; o[i] = v[bitrev_c(i, 6)]; });
  } else {
    cf32 zin[LW], zout[LW];
    static_for<LW>([&](auto ic) { zin[decltype(ic)::v] = v[decltype(ic)::v]; });
    dft_rec<LW, LW, 0, 1>(zin, zout, [](int e) { return cf32{kTw60.c[e][0], kTw60.c[e][1]}; });
    static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; if constexpr (i < LW) o[i] = zout[i]; else o[i] = cf32{0.f, 0.f}; });
  }
  static_for<64>([&](auto ic) {
    constexpr int i = decltype(ic)::v;
    if constexpr (FWD) v[i] = cf32{o[i].y, o[i].x};
    else v[i] = o[i];
  });
}
template <int RL, bool FWD, int LW = 64>
__device__ __forceinline__ void sh_fft_b2a(cf32 (&v)[64], float* __restrict__ lbuf, const float2* __restrict__ tw) {
  constexpr int BC = 64 / RL, LG = log2_c(RL);
  const int lane = threadIdx.x & 63;
  auto fence = [] {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  if constexpr (FWD) static_for<64>([&](auto ic) { constexpr int i = decltype(ic)::v; const float t = v[i].x; v[i].x = v[i].y; v[i].y = t; });
  cf32 T[LW];   // the LW-point transform of this lane's sequence, natural order
  if constexpr (LW == 64) {
    dft_reg<64>(v);
    static_for<64>([&](auto lc) { constexpr int l = decltype(lc)::v; T[l] = v[bitrev_c(l, 6)]; });
  } else {
    cf32 zin[LW];
    static_for<LW>([&](auto ic) { zin[decltype(ic)::v] = v[decltype(ic)::v]; });
    dft_rec<LW, LW, 0, 1>(zin, T, [](int e) { return cf32{kTw60.c[e][0], kTw60.c[e][1]}; });
  }
  float ux[64];
  static_for<LW>([&](auto lc) { constexpr int l = decltype(lc)::v; lbuf[l * 65 + lane] = T[l].x; });
  fence();
  static_for<64>([&](auto sc) { ux[decltype(sc)::v] = lbuf[min(lane, LW - 1) * 65 + decltype(sc)::v]; });
  fence();
  static_for<LW>([&](auto lc) { constexpr int l = decltype(lc)::v; lbuf[l * 65 + lane] = T[l].y; });
  fence();
  static_for<64>([&](auto sc) { constexpr int ss = decltype(sc)::v; v[ss] = cf32{ux[ss], lbuf[min(lane, LW - 1) * 65 + ss]}; });   // v[p BC + bb]
  fence();
  cf32 wl[RL];
  static_for<RL>([&](auto pc) {
    constexpr int pp = decltype(pc)::v;
    if constexpr (pp > 0 && RL < 16) { const float2 t = tw[min(lane, LW - 1) * pp]; wl[pp] = cf32{t.x, t.y}; }
  });
  cf32 o[64];
  static_for<BC>([&](auto bc) {
    constexpr int bb = decltype(bc)::v;
    cf32 x[RL];
    static_for<RL>([&](auto pc) {
      constexpr int pp = decltype(pc)::v;
      if constexpr (pp == 0) x[0] = v[bb];
      else if constexpr (RL >= 16) {
        const float2 t = tw[min(lane, LW - 1) * pp];
        x[pp] = cmul(v[pp * BC + bb], cf32{t.x, t.y});
      } else x[pp] = cmul(v[pp * BC + bb], wl[pp]);
    });
    dft_reg<RL>(x);
    static_for<RL>([&](auto rc) { constexpr int r = decltype(rc)::v; o[bb * RL + r] = x[bitrev_c(r, LG)]; });
  });
  static_for<64>([&](auto ic) {
    constexpr int i = decltype(ic)::v;
    if constexpr (FWD) v[i] = cf32{o[i].y, o[i].x};
    else v[i] = o[i];
  });
}

constexpr int kShFftWaves = 4;
// rows, forward over x:  field [B][N][N] -> F1T [B][L][N]
// GRID: `field` holds one float per pixel, the phase (atmosphere + mirror + micro-lens) in revolutions reduced to [-1/2, 1/2], or kShOutside
// (k_phase_mfma<.., true, true>): the field amplitude e^{2 pi i w} is formed here, in registers
template <int RL, int LW, bool GRID = false>
__global__ __launch_bounds__(64 * kShFftWaves, 2) void k_sh_rows_fwd(const float2* __restrict__ field, float2* __restrict__ F1T, const float2* __restrict__ tw,
                                                                     float amplitude = 0.f) {
  extern __shared__ float lds_shfft[];
  constexpr int L = LW * RL, N = L / 2, BC = 64 / RL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int iy0 = (blockIdx.x * kShFftWaves + wave) * BC;
  if (iy0 >= N) return;
  cf32 v[64];
  if constexpr (GRID) {
    const float* ph = reinterpret_cast<const float*>(field) + ((size_t)blockIdx.y * N + iy0) * N + min(lane, LW - 1);
    static_for<64>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int bb = i / RL, r = i % RL;
      if constexpr (r < RL / 2) {
        const float u = ph[(size_t)bb * N + LW * r];
        const float a = u > 1.0f ? 0.f : amplitude;
        float sn, cs;
        sincospif(2.0f * u, &sn, &cs);
        v[i] = cf32{a * cs, a * sn};
      } else {
        v[i] = cf32{0.f, 0.f};   // the zero padding
      }
    });
  } else {
    const float2* src = field + ((size_t)blockIdx.y * N + iy0) * N + min(lane, LW - 1);
    static_for<64>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int bb = i / RL, r = i % RL;
      if constexpr (r < RL / 2) { const float2 t = src[(size_t)bb * N + LW * r]; v[i] = cf32{t.x, t.y}; }
      else v[i] = cf32{0.f, 0.f};   // the zero padding
    });
  }
  sh_fft_a2b<RL, true, LW>(v, lds_shfft + (size_t)wave * 64 * 65, tw);
  // tiled intermediate: element (row y, column kx) lives in tile (kx / RL, y / BC) at [kx % RL][y % BC] — 64 elements = 512 bytes = exactly
  // what the 64 lanes (p, bb) of layout B hold for one k2: one fully coalesced store per k2 (a plain [kx][y] array took eight 64-byte
  // pieces in eight different rows per instruction: 512 scattered pieces per wave, and the pass fell to 40 % of its speed whenever the
  // allocation came back from the driver in small physical fragments)
  float2* dst = F1T + (size_t)blockIdx.y * L * N + (size_t)(iy0 / BC) * 64 + lane;
  static_for<LW>([&](auto kc) { constexpr int k2 = decltype(kc)::v; dst[(size_t)k2 * (N / BC) * 64] = make_float2(v[k2].x, v[k2].y); });
}
// columns: forward over y, transfer function, inverse over y:  F1T -> GT (both tiled, see k_sh_rows_fwd)
// The wave takes its BC columns in layout B of the y transform (lane (p, bb) holds rows y = p + RL k2 of column bb: for one k2 that is a
// whole 512-byte tile of the intermediates, or 64 / RL aligned pieces of neighbouring tiles), runs B -> A forward, multiplies by the
// transfer function in layout A, runs A -> B inverse and stores rows y < N the same way.
// tfq: [L / BC][64][64] = transfer[ky = lane + 64 r][kx = group BC + bb] for register bb RL + r (arranged on the host)
template <int RL, int LW>
__global__ __launch_bounds__(64 * kShFftWaves, 2) void k_sh_cols(const float2* __restrict__ F1T, float2* __restrict__ GT, const float2* __restrict__ tfq,
                                                                 const float2* __restrict__ tw) {
  extern __shared__ float lds_shfft[];
  constexpr int L = LW * RL, N = L / 2, BC = 64 / RL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cg = blockIdx.x * kShFftWaves + wave;
  if (cg * BC >= L) return;
  const int pp = lane / BC, bb = lane - pp * BC;
  const int kx = cg * BC + bb;
  // element (y = pp + RL k2, kx): tile (kx / RL, y / BC) at [kx % RL][y % BC]
  const size_t tile0 = ((size_t)(kx / RL) * (N / BC) + pp / BC) * 64 + (kx % RL) * BC + (pp % BC);   // the element of k2 = 0
  auto tiled = [&](int k2) -> size_t {
    if constexpr (RL % BC == 0) return tile0 + (size_t)k2 * (RL / BC) * 64;   // RL k2 rows further: RL / BC whole tiles (one base, fixed strides)
    const int y = pp + RL * k2;
    return ((size_t)(kx / RL) * (N / BC) + y / BC) * 64 + (kx % RL) * BC + (y % BC);
  };
  const float2* src = F1T + (size_t)blockIdx.y * L * N;
  cf32 v[64];
  static_for<64>([&](auto kc) {
    constexpr int k2 = decltype(kc)::v;
    if constexpr (k2 < N / RL) { const float2 t = src[tiled(k2)]; v[k2] = cf32{t.x, t.y}; }
    else v[k2] = cf32{0.f, 0.f};   // the zero padding (y >= N)
  });
  float* lbuf = lds_shfft + (size_t)wave * 64 * 65;
  sh_fft_b2a<RL, true, LW>(v, lbuf, tw);
  const float2* tf = tfq + (size_t)cg * 64 * 64 + lane;
  // (eight table loads at a time: left alone the compiler requests all 64 first — 128 more live registers, spills at RL = 16)
  static_for<8>([&](auto gc) {
    constexpr int g8 = decltype(gc)::v;
    float2 t8[8];
    static_for<8>([&](auto jc) { t8[decltype(jc)::v] = tf[(8 * g8 + decltype(jc)::v) * 64]; });
    static_for<8>([&](auto jc) { constexpr int i = 8 * g8 + decltype(jc)::v; v[i] = cmul(v[i], cf32{t8[decltype(jc)::v].x, t8[decltype(jc)::v].y}); });
    __builtin_amdgcn_sched_barrier(0);
  });
  sh_fft_a2b<RL, false, LW>(v, lbuf, tw);
  float2* dst = GT + (size_t)blockIdx.y * L * N;
  static_for<64>([&](auto kc) {
    constexpr int k2 = decltype(kc)::v;
    if constexpr (k2 < N / RL) dst[tiled(k2)] = make_float2(v[k2].x, v[k2].y);
  });
}
// rows, inverse over kx, intensity:  GT [B][L][N] -> image [B][N][N] float64
// FUSED (the image itself is not asked for: SH_step): photon noise and the estimator's per-lenslet sums (flux, flux-weighted x and y:
// k_sh_estimate's pixel loop) are taken here, while the intensities are in registers — the image is neither written nor re-read twice
// (k_sh_noise 0.71 ms + k_sh_estimate 0.73 ms per 1024 envs at N = 256 against 0.33 ms for this pass).  A lane keeps running sums per
// column while consecutive rows stay in the same lenslet, adds them to the wave's table in LDS when the lenslet changes, and the wave
// adds its table to the env's sums in global memory.
struct ShFuseArgs {
  const int32_t* sub_slot;   // [N*N]
  const double* x_det;       // [N]
  double* sums;              // [B][n_sub][3], zeroed before the launch
  int n_sub;
  size_t env_base;           // aog_config.env_id_base: the noise stream is keyed by the GLOBAL env id
  unsigned long long seed;
  uint32_t call;
};
template <int RL, int LW, bool FUSED>
__global__ __launch_bounds__(64 * kShFftWaves, 2) void k_sh_rows_inv(const float2* __restrict__ GT, double* __restrict__ image, const float2* __restrict__ tw,
                                                                     double scale, ShFuseArgs f) {
  extern __shared__ float lds_shfft[];
  constexpr int L = LW * RL, N = L / 2, BC = 64 / RL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y0 = (blockIdx.x * kShFftWaves + wave) * BC;
  if (y0 >= N) return;
  double* tab = reinterpret_cast<double*>(lds_shfft + (size_t)kShFftWaves * 64 * 65) + (size_t)wave * 3 * f.n_sub;   // [n_sub][3], this wave's
  if constexpr (FUSED) {
    for (int i = lane; i < 3 * f.n_sub; i += 64) tab[i] = 0.0;
  }
  const float2* src = GT + (size_t)blockIdx.y * L * N + (size_t)(y0 / BC) * 64 + lane;   // tiled layout: one coalesced 512-byte load per k2
  cf32 v[64];
  static_for<64>([&](auto kc) {
    constexpr int k2 = decltype(kc)::v;
    if constexpr (k2 < LW) { const float2 t = src[(size_t)k2 * (N / BC) * 64]; v[k2] = cf32{t.x, t.y}; }
    else v[k2] = cf32{0.f, 0.f};
  });
  sh_fft_b2a<RL, false, LW>(v, lds_shfft + (size_t)wave * 64 * 65, tw);
  const bool owner = LW == 64 || lane < LW;   // lanes LW .. 63 hold no pixels
  if constexpr (!FUSED) {
    double* dst = image + ((size_t)blockIdx.y * N + y0) * N + lane;
    static_for<64>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int b2 = i / RL, r = i % RL;
      if constexpr (r < RL / 2) {
        if (owner) dst[(size_t)b2 * N + LW * r] = ((double)v[i].x * (double)v[i].x + (double)v[i].y * (double)v[i].y) * scale;
      }
    });
  } else {
    constexpr int NX = RL / 2;
    int cur[NX];
    double s0[NX], sy[NX], xd[NX];
    static_for<NX>([&](auto rc) { constexpr int r = decltype(rc)::v; cur[r] = -1; s0[r] = 0.0; sy[r] = 0.0; xd[r] = f.x_det[min(lane, LW - 1) + LW * r]; });
    auto flush = [&](int slot, double a0, double ay, double xdet) {
      if (slot >= 0) {
        atomicAdd(&tab[3 * slot], a0);
        atomicAdd(&tab[3 * slot + 1], a0 * xdet);
        atomicAdd(&tab[3 * slot + 2], ay);
      }
    };
    static_for<BC>([&](auto bc) {
      constexpr int b2 = decltype(bc)::v;
      const int y = y0 + b2;
      const double yd = f.x_det[y];
      const size_t line = ((f.env_base + blockIdx.y) * N + y) * 64 + lane;   // (= x % LW: sh_noisy_value's key)
      uint32_t wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
      bool have_b = false;
      static_for<NX>([&](auto rc) {
        constexpr int r = decltype(rc)::v, i = b2 * RL + r;
        const int x = min(lane, LW - 1) + LW * r;
        if constexpr ((r & 3) == 0) {
          sh_noise_words(line, r >> 2, false, f.seed, f.call, wa);
          have_b = false;
        }
        const double lam = ((double)v[i].x * (double)v[i].x + (double)v[i].y * (double)v[i].y) * scale;
        const int slot = owner ? f.sub_slot[y * N + x] : -1;
        if (slot != cur[r]) {
          flush(cur[r], s0[r], sy[r], xd[r]);
          cur[r] = slot; s0[r] = 0.0; sy[r] = 0.0;
        }
        const bool small = slot >= 0 && lam < kShPoissonSwitch;
        double out = sh_poisson_small(small ? lam : 0.0, wa[r & 3], small);   // (the wave's loop: every lane takes part)
        if (slot >= 0) {
          if (!small) {
            if (!have_b) { sh_noise_words(line, r >> 2, true, f.seed, f.call, wb); have_b = true; }
            out = sh_poisson_large(lam, wa[r & 3], wb[r & 3]);
          }
          const double w = out + 1e-10;   // estimate([image + 1e-10]) (AO_env.py:277)
          s0[r] += w;
          sy[r] = fma(w, yd, sy[r]);
        }
      });
    });
    static_for<NX>([&](auto rc) { constexpr int r = decltype(rc)::v; flush(cur[r], s0[r], sy[r], xd[r]); });
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    double* out = f.sums + (size_t)blockIdx.y * 3 * f.n_sub;
    for (int i = lane; i < 3 * f.n_sub; i += 64) {
      const double t = tab[i];
      if (t != 0.0) atomicAdd(&out[i], t);
    }
  }
}

__global__ void k_sh_noise(const double* __restrict__ image, double* __restrict__ noisy, int N, size_t env_base, unsigned long long seed, uint32_t call,
                           int sep_rl) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * N) return;
  const size_t il = (size_t)blockIdx.y * N * N + idx;
  const int y = idx / N;
  noisy[il] = sh_noisy_value(image[il], env_base + blockIdx.y, y, idx - y * N, N, seed, call, sep_rl);
}

// ---- separable form of the same propagation (the default whenever the transfer function factorises, which the paraxial Fresnel one does:
// exp(-i z (kx^2 + ky^2) / 2k) = hx(kx) hy(ky)) ------------------------------------------------------------------------------------------
// pad -> FFT2 -> x H -> IFFT2 -> crop  ==  [rows: pad, FFT_x, x hx, IFFT_x, keep x < N]  then  [columns: pad, FFT_y, x hy, IFFT_y, keep y < N]:
// the x operation acts per row (rows y >= N of the padded field are zero and stay zero), the y operation per column (columns x >= N are
// dropped at the end, so they are dropped before it).  TWO passes over an N x N complex64 intermediate instead of three over 2N x N ones:
// 20 N^2 bytes per env instead of 68 N^2, and 4 N line transforms instead of 6 N (the three-pass column kernel transformed all 2N columns).
//   pass 1  k_sh_rows_sep: phase row (layout A) -> forward -> x hx (layout B) -> inverse -> layout A, x < N kept
//           -> G1[x / BC][y][x % BC]  (column groups of BC = 64 / RL: what a wave of pass 2 reads is one contiguous N x BC block)
//   pass 2  k_sh_cols_sep: its BC columns in layout B (lane (p, bb): rows y = p + RL k2 of column bb: 512 contiguous bytes per k2)
//           -> forward -> x hy (layout A) -> inverse -> layout B, y < N kept -> |.|^2 x scale -> image, or (FUSED) photon noise + lenslet sums
// hxq: [LW][64] = hx[lane / BC + RL k2];  hyq: [RL][64] = hy[lane + LW r]  (arranged on the host)
template <int RL, int LW>
__global__ __launch_bounds__(64 * kShFftWaves, 2) void k_sh_rows_sep(const float* __restrict__ phase, float2* __restrict__ G1, const float2* __restrict__ tw,
                                                                     const float2* __restrict__ hxq, float amplitude) {
  extern __shared__ float lds_shfft[];
  constexpr int L = LW * RL, N = L / 2, BC = 64 / RL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int iy0 = (blockIdx.x * kShFftWaves + wave) * BC;
  if (iy0 >= N) return;
  const int la = min(lane, LW - 1);
  cf32 v[64];
  const float* ph = phase + ((size_t)blockIdx.y * N + iy0) * N + la;
  static_for<64>([&](auto ic) {
    constexpr int i = decltype(ic)::v;
    constexpr int bb = i / RL, r = i % RL;
    if constexpr (r < RL / 2) {
      const float u = ph[(size_t)bb * N + LW * r];
      const float a = u > 1.0f ? 0.f : amplitude;
      float sn, cs;
      sincospif(2.0f * u, &sn, &cs);
      v[i] = cf32{a * cs, a * sn};
    } else {
      v[i] = cf32{0.f, 0.f};   // the zero padding
    }
  });
  float* lbuf = lds_shfft + (size_t)wave * 64 * 65;
  sh_fft_a2b<RL, true, LW>(v, lbuf, tw);
  const float2* hq = hxq + lane;
  static_for<8>([&](auto gc) {   // (eight table loads at a time, as in k_sh_cols)
    constexpr int g8 = decltype(gc)::v;
    float2 t8[8];
    static_for<8>([&](auto jc) { constexpr int k2 = 8 * g8 + decltype(jc)::v; if constexpr (k2 < LW) t8[decltype(jc)::v] = hq[k2 * 64]; });
    static_for<8>([&](auto jc) {
      constexpr int k2 = 8 * g8 + decltype(jc)::v;
      if constexpr (k2 < LW) v[k2] = cmul(v[k2], cf32{t8[decltype(jc)::v].x, t8[decltype(jc)::v].y});
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  sh_fft_b2a<RL, false, LW>(v, lbuf, tw);
  if (LW == 64 || lane < LW) {
    float2* dst = G1 + (size_t)blockIdx.y * N * N;
    static_for<64>([&](auto ic) {
      constexpr int i = decltype(ic)::v;
      constexpr int bb = i / RL, r = i % RL;
      if constexpr (r < RL / 2) {
        const int x = lane + LW * r;
        dst[((size_t)(x / BC) * N + (iy0 + bb)) * BC + (x % BC)] = make_float2(v[i].x, v[i].y);
      }
    });
  }
}

template <int RL, int LW, bool FUSED>
__global__ __launch_bounds__(64 * kShFftWaves, 2) void k_sh_cols_sep(const float2* __restrict__ G1, double* __restrict__ image, const float2* __restrict__ tw,
                                                                     const float2* __restrict__ hyq, double scale, ShFuseArgs f) {
  extern __shared__ float lds_shfft[];
  constexpr int L = LW * RL, N = L / 2, BC = 64 / RL, NK = N / RL;   // NK rows y = p + RL k2 per lane
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cg = blockIdx.x * kShFftWaves + wave;
  if (cg * BC >= N) return;
  const int pp = lane / BC, bb = lane - pp * BC;
  const int x = cg * BC + bb;
  double* tab = reinterpret_cast<double*>(lds_shfft + (size_t)kShFftWaves * 64 * 65) + (size_t)wave * 3 * f.n_sub;   // [n_sub][3], this wave's
  if constexpr (FUSED) {
    for (int i = lane; i < 3 * f.n_sub; i += 64) tab[i] = 0.0;
  }
  const float2* src = G1 + (size_t)blockIdx.y * N * N + (size_t)cg * N * BC + lane;   // element (y = pp + RL k2, bb) at (y BC + bb) = lane + 64 k2
  cf32 v[64];
  static_for<64>([&](auto kc) {
    constexpr int k2 = decltype(kc)::v;
    if constexpr (k2 < NK) { const float2 t = src[(size_t)k2 * 64]; v[k2] = cf32{t.x, t.y}; }
    else v[k2] = cf32{0.f, 0.f};   // the zero padding (y >= N)
  });
  float* lbuf = lds_shfft + (size_t)wave * 64 * 65;
  sh_fft_b2a<RL, true, LW>(v, lbuf, tw);
  static_for<RL>([&](auto rc) {
    constexpr int r = decltype(rc)::v;
    const float2 h = hyq[r * 64 + lane];
    static_for<BC>([&](auto bc) { constexpr int i = decltype(bc)::v * RL + r; v[i] = cmul(v[i], cf32{h.x, h.y}); });
  });
  sh_fft_a2b<RL, false, LW>(v, lbuf, tw);
  if constexpr (!FUSED) {
    double* dst = image + (size_t)blockIdx.y * N * N + x;
    static_for<NK>([&](auto kc) {
      constexpr int k2 = decltype(kc)::v;
      dst[(size_t)(pp + RL * k2) * N] = ((double)v[k2].x * (double)v[k2].x + (double)v[k2].y * (double)v[k2].y) * scale;
    });
  } else {
    // this lane: NK rows of ONE column; running sums while consecutive rows of the lane (RL apart) stay in the same lenslet.
    // The intensities go through the wave's LDS plane (each lane its own NK doubles) and the pixels are walked by a REAL loop, four per
    // Philox call: fully unrolled, the 32 pixels x 12-level Poisson chain made a 20 000-line kernel (more code than the instruction cache
    // holds, fetched once per wave) whose register allocation spilled 76 - 360 registers.
    static_assert((size_t)NK * 64 * sizeof(double) <= (size_t)64 * 65 * sizeof(float), "the intensities of a wave fit its transform plane");
    double* lamp = reinterpret_cast<double*>(lbuf) + lane;
    static_for<NK>([&](auto kc) {
      constexpr int k2 = decltype(kc)::v;
      lamp[k2 * 64] = ((double)v[k2].x * (double)v[k2].x + (double)v[k2].y * (double)v[k2].y) * scale;
    });
    const double xd = f.x_det[x];
    int cur = -1;
    double s0 = 0.0, sy = 0.0;
    auto flush = [&](int slot, double a0, double ay) {
      if (slot >= 0) {
        atomicAdd(&tab[3 * slot], a0);
        atomicAdd(&tab[3 * slot + 1], a0 * xd);
        atomicAdd(&tab[3 * slot + 2], ay);
      }
    };
    const size_t line = ((f.env_base + blockIdx.y) * N + x) * 64 + pp;   // (sh_noisy_value's key, sep_rl form)
    const int32_t* slot_col = f.sub_slot + x;
#pragma unroll 1
    for (int k4 = 0; k4 < (NK + 3) / 4; ++k4) {
      uint32_t wa[4], wb[4] = {0, 0, 0, 0};
      sh_noise_words(line, (uint32_t)k4, false, f.seed, f.call, wa);
      bool have_b = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k2 = 4 * k4 + j;
        if (NK % 4 != 0 && k2 >= NK) break;   // (lines of 60 RL: NK = 30)
        const int y = pp + RL * k2;
        const double lam = lamp[k2 * 64];
        const int slot = slot_col[(size_t)y * N];
        if (slot != cur) {
          flush(cur, s0, sy);
          cur = slot; s0 = 0.0; sy = 0.0;
        }
        const bool small = slot >= 0 && lam < kShPoissonSwitch;
        double out = sh_poisson_small(small ? lam : 0.0, wa[j], small);   // (the wave's loop: every lane takes part)
        if (slot >= 0) {
          if (!small) {
            if (!have_b) { sh_noise_words(line, (uint32_t)k4, true, f.seed, f.call, wb); have_b = true; }
            out = sh_poisson_large(lam, wa[j], wb[j]);
          }
          const double w = out + 1e-10;   // estimate([image + 1e-10]) (AO_env.py:277)
          s0 += w;
          sy = fma(w, f.x_det[y], sy);
        }
      }
    }
    flush(cur, s0, sy);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    double* out = f.sums + (size_t)blockIdx.y * 3 * f.n_sub;
    for (int i = lane; i < 3 * f.n_sub; i += 64) {
      const double t = tab[i];
      if (t != 0.0) atomicAdd(&out[i], t);
    }
  }
}

struct ShEstimateArgs {
  const double* image;          // [B][N*N] (already noisy)
  const double* sums_in;        // [B][n_sub][3] from the fused row pass (then `image` is not read), or null
  const int32_t* sub_slot;      // [N*N]
  const double* x_det;          // [N]
  const double* centres;        // [n_sub][2]
  const double* slopes_ref;     // [2 n_sub]
  const double* recon;          // [A][2 n_sub]
  double* sh_act;               // [B][A]
  double* action_out;           // [B][A]
  int N, n_sub, A;
  double gain, leakage;
};

// one workgroup per env; dynamic LDS: sums [3 n_sub] + slopes [2 n_sub] doubles
__global__ __launch_bounds__(256) void k_sh_estimate(ShEstimateArgs p) {
  extern __shared__ double sh[];
  double* sums = sh;                       // [n_sub][3]: flux, sum x, sum y
  double* slopes = sh + 3 * (size_t)p.n_sub;
  const int env = blockIdx.x;
  for (int i = threadIdx.x; i < 3 * p.n_sub; i += blockDim.x) sums[i] = 0.0;
  __syncthreads();
  if (p.sums_in) {
    for (int i = threadIdx.x; i < 3 * p.n_sub; i += blockDim.x) sums[i] = p.sums_in[(size_t)env * 3 * p.n_sub + i];
  }
  const double* img = p.image + (size_t)env * p.N * p.N;
  for (int idx = threadIdx.x; idx < (p.sums_in ? 0 : p.N * p.N); idx += blockDim.x) {
    const int slot = p.sub_slot[idx];
    if (slot < 0) continue;
    const int iy = idx / p.N, ix = idx - iy * p.N;
    const double w = img[idx] + 1e-10;      // estimate([wfs_image + 1e-10]) (AO_env.py:277)
    atomicAdd(&sums[3 * slot], w);
    atomicAdd(&sums[3 * slot + 1], w * p.x_det[ix]);
    atomicAdd(&sums[3 * slot + 2], w * p.x_det[iy]);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < p.n_sub; s += blockDim.x) {
    const double fl = sums[3 * s];
    slopes[s] = sums[3 * s + 1] / fl - p.centres[2 * s] - p.slopes_ref[s];
    slopes[p.n_sub + s] = sums[3 * s + 2] / fl - p.centres[2 * s + 1] - p.slopes_ref[p.n_sub + s];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < p.A; k += blockDim.x) {
    const double* r = p.recon + (size_t)k * 2 * p.n_sub;
    double acc = 0;
    for (int j = 0; j < 2 * p.n_sub; ++j) acc = fma(r[j], slopes[j], acc);
    const double a = (1.0 - p.leakage) * p.sh_act[(size_t)env * p.A + k] - p.gain * acc;
    p.sh_act[(size_t)env * p.A + k] = a;
    p.action_out[(size_t)env * p.A + k] = a;
  }
}
#endif  // AOG_MAIN_TU

#ifdef AOG_MAIN_TU
// stored screens of envs [first, first + count) of a quasi_static / semi_dynamic handle as achromatic float64 [count][N*N] (hcipy's
// unit: phase * lambda), exactly the values the fused kernel reads (fp32 revolutions widened: no rounding), 0 outside the aperture
__global__ void k_screens_from_store(const float* __restrict__ psi_tile, const double* __restrict__ psi64, const int32_t* __restrict__ ap_index,
                                     double* __restrict__ out, int first, int n_ap, int n_ptiles, int N2, double two_pi_lambda) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = first + blockIdx.y;
  if (p >= n_ap) return;
  const double v = psi64 ? psi64[(size_t)env * n_ap + p] : (double)psi_tile[psi_tile_index(env, p, n_ptiles)] * two_pi_lambda;
  out[(size_t)blockIdx.y * N2 + ap_index[p]] = v;
}

// atmosphere phase (radians at lambda_wfs) of one env on the full grid, from the tiled fp32 screens
__global__ void k_phase_screen(const float* __restrict__ psi_tile, const int32_t* __restrict__ ap_index, float* __restrict__ out, int env,
                               int n_ap, int n_ptiles) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_ap) return;
  out[ap_index[p]] = 6.2831853071795865f * psi_tile[psi_tile_index(env, p, n_ptiles)];
}
#endif  // AOG_MAIN_TU

// self-test hook: the three sin/cos flavours of the fused kernels on caller-supplied revolutions
#ifdef AOG_MAIN_TU
__global__ void k_selftest_sincos(const float* __restrict__ u, float* __restrict__ s, float* __restrict__ c, int n, int flavour) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sv, cv;
  if (flavour == 0) sincos_rev<0>(u[i], sv, cv);
  else if (flavour == 1) sincos_rev<1>(u[i], sv, cv);
  else { sv = __builtin_amdgcn_sinf(u[i]); cv = __builtin_amdgcn_cosf(u[i]); }
  s[i] = sv;
  c[i] = cv;
}
#endif  // AOG_MAIN_TU

}  // namespace aog
