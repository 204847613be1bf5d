// K10: Shack-Hartmann chain.
#pragma once
#include "k_common.h"
#include "k_fft.h"

namespace aog {

// ------------------------------------------------------------------------------------------------
// K10  Shack-Hartmann chain (AO_env.py:254-290): field on the magnified pupil x micro-lens phase -> angular-spectrum
// Fresnel propagation over one lenslet focal length (2x zero-padded hipFFT, float64) -> detector image -> photon noise ->
// centre of gravity per selected lenslet -> reconstructor GEMV + leaky integrator.
// ------------------------------------------------------------------------------------------------
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
// the row pass of the separable form keeps the transfer function's L distinct entries in LDS behind the waves' planes (lines of 1024: 8 KB;
// two workgroups of 66.5 + 8 KB still share a CU)
template <int RL, int LW> constexpr bool kShHxInLds = RL == 16 && LW == 64;
template <int RL, int LW> constexpr size_t kShHxLdsBytes = kShHxInLds<RL, LW> ? (size_t)RL * LW * 8 : 0;
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
  // The transfer function's table.  A lane needs 64 entries, hx[lane / BC + RL k2]: fetched where they are used (eight batches of eight, each
  // awaited in turn) they were eight L2 round trips in the wave's in-order stream — 17 k of its 59 k cycles — and 4 GB of L2 reads per
  // launch, every wave re-reading the same 32 KB.  The workgroup's copy of the L distinct entries in LDS (kShHxLdsBytes behind the planes,
  // requested first thing, needed after the forward transform) serves them instead.
  constexpr bool HX_LDS = kShHxInLds<RL, LW>;
  cf32* hx_lds = reinterpret_cast<cf32*>(lds_shfft + (size_t)kShFftWaves * 64 * 65);
  if constexpr (HX_LDS) {
    static_for<L / (64 * kShFftWaves)>([&](auto uc) {
      const int en = (int)threadIdx.x + 64 * kShFftWaves * decltype(uc)::v;   // entry p + RL k2 lives at hxq[k2][BC p]
      const float2 t = hxq[(en / RL) * 64 + BC * (en % RL)];
      hx_lds[en] = cf32{t.x, t.y};
    });
    __syncthreads();
  }
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
  if constexpr (HX_LDS) {
    const cf32* hl = hx_lds + lane / BC;
    static_for<64>([&](auto kc) { constexpr int k2 = decltype(kc)::v; v[k2] = cmul(v[k2], hl[RL * k2]); });
  } else {
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
  }
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
    // the lenslet of a pixel and its row coordinate come from memory: the four pixels of the NEXT round are requested while this round's
    // noise is drawn (they were loaded where they were used: 8 rounds x a dependent L2 round trip per wave, with two waves per SIMD to hide it)
    int slot_n[4];
    double yd_n[4];
    auto request = [&](int k4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = pp + RL * min(4 * k4 + j, NK - 1);
        slot_n[j] = slot_col[(size_t)y * N];
        yd_n[j] = f.x_det[y];
      }
    };
    request(0);
#pragma unroll 1
    for (int k4 = 0; k4 < (NK + 3) / 4; ++k4) {
      int slot_c[4];
      double yd_c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { slot_c[j] = slot_n[j]; yd_c[j] = yd_n[j]; }
      request(min(k4 + 1, (NK + 3) / 4 - 1));
      uint32_t wa[4], wb[4] = {0, 0, 0, 0};
      sh_noise_words(line, (uint32_t)k4, false, f.seed, f.call, wa);
      bool have_b = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k2 = 4 * k4 + j;
        if (NK % 4 != 0 && k2 >= NK) break;   // (lines of 60 RL: NK = 30)
        const double lam = lamp[k2 * 64];
        const int slot = slot_c[j];
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
          sy = fma(w, yd_c[j], sy);
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

}  // namespace aog
