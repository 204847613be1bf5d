// K8: von Karman screen synthesis (literal and two-band forms).
#pragma once
#include "k_common.h"
#include "k_fft.h"

namespace aog {

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
  const uint32_t* ap_bits;   // nullable: [N][ceil(N / 32)] aperture bit mask — pass B then leaves each column tile's aperture sum in `part`
  double* part;              // [env in batch][column tiles]
  int n_tiles;
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
  if (p.ap_bits) {
    // the aperture mean the conversion to the kernels' layouts subtracts: summed here, where the screen is in registers, instead of by a
    // pass of its own over the stored screens (k_screen_means: 0.4 ms per 4096 envs at N = 256).  Fixed order: this lane's rows, then its
    // columns, then the wave's lanes by a butterfly; k_mean_from_parts adds the column tiles in order.
    double sum = 0.0;
    if (lane < LW) {
      const int nw = (N + 31) >> 5;
#pragma unroll
      for (int pp = 0; pp < R; ++pp) {
        const uint32_t m = p.ap_bits[(size_t)(pp + R * lane) * nw + (ix0 >> 5)] >> (ix0 & 31);   // (a tile lies inside one word: NL divides 32)
#pragma unroll
        for (int l = 0; l < NL; ++l)
          if ((m >> l) & 1u) sum += (double)acc[l][pp].x;   // (columns past N carry no mask bit)
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) p.part[(size_t)b * p.n_tiles + ix0 / NL] = sum;
  }
}

// aperture means from pass B's per-tile sums (one thread per env of the batch; tiles added in order)
__global__ void k_mean_from_parts(const double* __restrict__ part, double* __restrict__ mean, int n_env, int n_tiles, int n_ap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_env) return;
  double s = 0.0;
  for (int t = 0; t < n_tiles; ++t) s += part[(size_t)b * n_tiles + t];
  mean[b] = s / (double)n_ap;
}

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

}  // namespace aog
