// Device helpers shared by every kernel family of libaogym.so (gfx950 only).  See DESIGN.md for the data layout and the roofline of each kernel.
//
// Notation: B envs (padded to Bp, a multiple of 64), n_ap aperture pixels packed row-major (padded to a
// multiple of 32 = one MFMA pixel tile), A modes (padded to A_PAD), MRW / MRS real pupil-plane tables at the
// wavefront-sensing / science wavelength, NS = 2*(MRW+MRS) real sums per env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace aog {

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


// compile-time loops (indices usable as template arguments)
template <int K>
struct IC { static constexpr int v = K; };
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

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


// The 8-table variant of k_fused_tab folds its fp32 table sums into float64 every kFlushTiles tiles (32 terms per tile and accumulator
// element); the many-table variants run fp32 over a chunk of at most kTabF32Tiles tiles.
constexpr int kFlushTiles = 4;
constexpr int kTabF32Tiles = 13;
constexpr int kSkewNops = 150;   // x 16 cycles: start-up skew of the second workgroup of a CU (about half a stage)

constexpr int kExt16G = 16;   // envs per group of the float64 matrix-core extrusion kernels (k_extrude.h)

constexpr float kShOutside = 2.0f;   // phase-grid value of a pixel outside the aperture (reduced phases lie in [-1/2, 1/2])
// arguments of k_phase_mfma<.., FIELD / GRID> (k_fused.h): where each packed aperture pixel goes on an env's own pupil grid
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
__host__ __device__ inline int spectrum_lane_width(int N) { return (N % 64 != 0 && N % 60 == 0) ? 60 : 64; }

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

}  // namespace aog
