// In-register DFT building blocks shared by the screen synthesis and the Shack-Hartmann propagation.
#pragma once
#include "k_common.h"

namespace aog {

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

}  // namespace aog
