// Screen installation kernels: caller / synthesised screens -> the internal fp32 layouts.
#pragma once
#include "k_common.h"

namespace aog {

// ------------------------------------------------------------------------------------------------
// K0  pack_screens: achromatic screens [count][N][N] (T = double|float) -> internal layouts.
//   psi_rev   fp32, revolutions at lambda_wfs, aperture mean removed, layout [quad q][env][4]
//             (lane = env reads one float4 = 4 consecutive packed pixels; 1 KiB per wave instruction)
//   psi_tile  fp32, same values in MFMA accumulator order (see k_fused_tab)
//   psi64     (validation mode) float64 [env][n_ap], aperture mean removed, hcipy units
// One workgroup per env.
// ------------------------------------------------------------------------------------------------

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

}  // namespace aog
