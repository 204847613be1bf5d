// Dynamic atmosphere on the int8 matrix cores: hcipy InfiniteAtmosphericLayer.evolve_until / _extrude (AO_env.py:125) as ONE exact
// fixed-point matrix product per axis and step.
//
// What is computed.  k successive one-pixel shifts along an axis are one linear map of the screen as it stood before them and of the k N
// normals they draw (extrusion_host.compose_extrusions; uploaded through aog_upload_layer_composite):
//     [R_1; ...; R_k] = A_k z + sqrt(Cn^2) B_k n,      z = screen[union stencil], n = [n_1; ...; n_k]
// so a step is: plan -> (prepare, product) for the x shifts -> (prepare, product) for the y shifts: no chain of k dependent rounds,
// no inter-workgroup barrier, every row of the product independent of every other.
//
// How it keeps float64-grade accuracy on an 8-bit pipe.  The AR recursion amplifies a white error of e rad per new sample to ~13 e of smooth
// phase error (measured on the host: profiles/HISTORY.md), so new samples must be good to ~1e-8 rad of ~10: fp32 accumulation cannot do
// that.  Integer accumulation can: every operand is written in balanced base-128 digits (int8),
//     A_k = qa sum_s As 128^(4-s)   (5 digits, quantum qa = 2^(ea-34)),      z - c0 - c1 x = qz sum_t Zt 128^(4-t)   (5 digits, qz per env)
//     sqrt(Cn^2) B_k = qb sum_s Bs 128^(3-s)   (4 digits),                     n = qn sum_t Nt 128^(4-t)               (5 digits)
// and v_mfma_i32_32x32x32_i8 sums the digit products EXACTLY in int32 (|digit product| <= 2^12, <= 2^26.5 per accumulator over the whole
// contraction).  Products of equal weight 128^(8-l), l = s + t, share one accumulator; levels l <= 4 are kept (15 digit pairs for A z, 14
// for B n: what is dropped is below 2^-35 of |A||z|).  With qn = qa qz 128 / qb per env the noise product lands on the same levels.  The
// only rounding anywhere is the quantisation of the operands (A to 2^-34 of its largest coefficient, z to 2^-34 of the env's largest
// detrended stencil sample): ~1e-9 rad per new sample at N = 256.  Piston and tilt of the stencil never enter the fixed-point product:
// A z = c0 (A 1) + c1 (A x) + A (z - c0 - c1 x) with the two vectors A 1 and A x exact in float64 — a quantisation error of A multiplied by
// a 30-rad piston would otherwise be the largest term (measured: 7e-9 against 1e-9 rad).
// Results do not depend on how envs are grouped into tiles (integer sums are exact and per column), so the Philox streams keyed by the global
// env id keep "split == whole" bit for bit.
#pragma once
#include "k_common.h"

namespace aog {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kX8DigA = 5, kX8DigZ = 5, kX8DigB = 4, kX8DigN = 5;
constexpr int kX8Levels = 5;        // product levels l = s + t kept: 0 .. 4
constexpr int kX8MaxK = 8;          // composite operators are built for k = 1 .. kcap <= 8 shifts per axis and step

// one composite operator (axis, k) in device memory
struct X8Table {
  const int32_t* yx;      // [U_pad] union stencil (sy << 16 | sx) on the screen hcipy's _extrude sees (rotated for 'top' / 'right'); padding repeats entry 0
  const int8_t* A8;       // [RT][KsA][5][64][16] digits of A_k as MFMA A operands: lane (row & 31, k-half g), byte b <-> column 32 ks + 16 g + b
  const int8_t* B8;       // [RT][KsB][4][64][16] digits of sqrt(Cn^2) B_k
  const double* r1;       // [RT * 32] A_k 1   (float64, exact piston response)
  const double* r2;       // [RT * 32] A_k x   (x = along-coordinate of the stencil sample - (N - 1) / 2)
  int k, U, KsA, KsB, RT; // U union size; k-steps (of 32) of the stencil and of the normals; row tiles (of 32 rows) = k Np / 32
  int Np;                 // rows per shift block: N rounded up to 64 (row (j - 1) Np + i = sample i of the slice shift j creates; i >= N: zero rows)
  int log2_qa;            // qa = 2^log2_qa
  int log2_cn;            // qn = qz * 2^log2_cn  (= qa 128 / qb)
  int ez_floor;           // smallest exponent of an env's stencil range: qz = 2^(ez - 34), ez >= ez_floor keeps |n| / qn inside 5 digits
  double sx, sxx;         // sum x, sum x^2 over the U stencil samples (detrending)
};

struct X8Args {
  const X8Table* tables;     // [2][kX8MaxK + 1]: [axis 0 vertical | 1 horizontal][k]
  double* master;            // [B][N*N]
  float* ring;               // nullable: fp32 ring copy ([B][N][N + 4])
  const double* ring_ref;    // [B]
  double ring_inv;
  int32_t* origin;           // [B][2] (ox, oy)
  uint32_t* ext_counter;     // [B]
  const double* velocity;    // [B][2]
  const double* noise;       // nullable replay normals [B][max_ext][N]
  int max_ext;
  int N, B, kcap;
  double t_prev, t_new, pitch;
  unsigned long long seed;
  int env_base;
  // plan (written by k_x8_plan, read by the other kernels)
  int32_t* dxy;              // [B][2] signed whole-pixel shifts of this step
  int32_t* slot;             // [2][B] phase (0 = x shifts, 1 = y shifts) -> slot = tile32 * 32 + column, -1 = no shift in that phase
  int32_t* list;             // [2][slots_max] slot -> env, -1 = empty
  int32_t* tile_k;           // [2][tiles64_max] k of each 64-env tile, 0 = unused
  int tiles64_max, slots_max;
  // prepared operands
  int8_t* Z8;                // [tiles32_max][KsTot_max][5][64][16]
  int KsTot_max;
  double* rec;               // [slots_max][4]: scale (qa qz 128^4), c0, c1, unused
  int* status;               // sticky error word (bit 2: a stencil sample or a normal left its fixed-point range)
};

// ---- plan: shifts of every env this step, envs grouped by shift count into 64-env tiles (one wave) ---------------------------------
__global__ __launch_bounds__(64) void k_x8_plan(X8Args p) {
  const int lane = threadIdx.x;
  __shared__ int cnt[2][kX8MaxK + 1], base[2][kX8MaxK + 1];
  if (lane < 2 * (kX8MaxK + 1)) (&cnt[0][0])[lane] = 0;
  __syncthreads();
  // pass 1: shifts and class counts
  for (int e0 = 0; e0 < p.B; e0 += 64) {
    const int e = e0 + lane;
    int kx = 0, ky = 0;
    if (e < p.B) {
      const double vx = p.velocity[2 * e], vy = p.velocity[2 * e + 1];
      // np.round(center / delta).astype(int) before and after (round-half-even = rint)
      const int dx = (int)rint(vx * p.t_new / p.pitch) - (int)rint(vx * p.t_prev / p.pitch);
      const int dy = (int)rint(vy * p.t_new / p.pitch) - (int)rint(vy * p.t_prev / p.pitch);
      p.dxy[2 * e] = dx;
      p.dxy[2 * e + 1] = dy;
      kx = abs(dx);
      ky = abs(dy);
      if (kx > p.kcap || ky > p.kcap) {   // the host sizes kcap from the largest wind component: cannot happen unless the wind was changed behind it
        atomicOr(p.status, 4);
        kx = min(kx, p.kcap);
        ky = min(ky, p.kcap);
      }
    }
    for (int k = 1; k <= p.kcap; ++k) {
      const unsigned long long mx = __ballot(e < p.B && kx == k), my = __ballot(e < p.B && ky == k);
      if (lane == 0) {
        cnt[0][k] += __popcll(mx);
        cnt[1][k] += __popcll(my);
      }
    }
  }
  __syncthreads();
  if (lane < 2) {   // tile bases per class (in 64-env tiles), tile -> k
    int t = 0;
    for (int k = 1; k <= p.kcap; ++k) {
      base[lane][k] = t * 64;
      const int nt = (cnt[lane][k] + 63) / 64;
      for (int i = 0; i < nt; ++i) p.tile_k[lane * p.tiles64_max + t + i] = k;
      t += nt;
    }
    for (; t < p.tiles64_max; ++t) p.tile_k[lane * p.tiles64_max + t] = 0;
  }
  for (int i = lane; i < 2 * p.slots_max; i += 64) p.list[i] = -1;
  __syncthreads();
  // pass 2: slots in env order within each class (deterministic, though nothing depends on it)
  for (int e0 = 0; e0 < p.B; e0 += 64) {
    const int e = e0 + lane;
    const int kx = e < p.B ? min(abs(p.dxy[2 * e]), p.kcap) : 0, ky = e < p.B ? min(abs(p.dxy[2 * e + 1]), p.kcap) : 0;
    int sx = -1, sy = -1;
    for (int k = 1; k <= p.kcap; ++k) {
      const unsigned long long mx = __ballot(kx == k), my = __ballot(ky == k);
      const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
      if (kx == k) sx = base[0][k] + __popcll(mx & below);
      if (ky == k) sy = base[1][k] + __popcll(my & below);
      __syncthreads();
      if (lane == 0) {
        base[0][k] += __popcll(mx);
        base[1][k] += __popcll(my);
      }
      __syncthreads();
    }
    if (e < p.B) {
      p.slot[e] = sx;
      p.slot[p.B + e] = sy;
      if (sx >= 0) p.list[sx] = e;
      if (sy >= 0) p.list[p.slots_max + sy] = e;
    }
  }
}

// balanced base-128 digits of x (|x| < 2^34 + 2^27), most significant first; ND of them
template <int ND>
__device__ __forceinline__ void x8_digits(long long x, int (&d)[ND]) {
#pragma unroll
  for (int t = ND - 1; t > 0; --t) {
    const int dg = (int)((x + 64) & 127) - 64;
    d[t] = dg;
    x = (x - dg) >> 7;
  }
  d[0] = (int)x;   // whatever is left: within int8 for arguments in range (checked by the caller)
}

__device__ __forceinline__ double x8_wave_sum(double v) {   // same association order whatever the data: results are reproducible
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double x8_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- prepare: one wave per env.  Gather the union stencil, detrend, choose the env's quantum, write digits of z and of the normals in the
// product's B-operand order; phase 1 also commits the step's origin and stream position. -------------------------------------------------
constexpr int kX8PrepWaves = 4;
constexpr int kX8MaxChunks = 3;   // 16-sample chunks per lane: union stencils of up to 64 * 3 * 16 = 3072 samples
__global__ __launch_bounds__(64 * kX8PrepWaves) void k_x8_prepare(X8Args p, int phase) {
  const int lane = threadIdx.x & 63;
  const int env = blockIdx.x * kX8PrepWaves + (threadIdx.x >> 6);
  if (env >= p.B) return;
  const int N = p.N;
  const int dx = p.dxy[2 * env], dy = p.dxy[2 * env + 1];
  const int d = phase == 0 ? dx : dy, k = min(abs(d), p.kcap);
  int ox = p.origin[2 * env], oy = p.origin[2 * env + 1];
  if (phase == 1) {   // the x shifts of this step have been applied
    ox = ((ox + dx) % N + N) % N;
  }
  if (k > 0) {
    const X8Table& tb = p.tables[(phase == 0 ? 1 : 0) * (kX8MaxK + 1) + k];   // x shifts use the horizontal ('left') operator
    const bool vertical = phase == 1, flipped = d > 0;
    const int slot = p.slot[phase * p.B + env];
    const int tile = slot >> 5, col = slot & 31;
    const double* master = p.master + (size_t)env * N * N;
    const int nchunk = tb.KsA * 2;   // 16-sample chunks of the (padded) union stencil
    double v[kX8MaxChunks][16];
    double s0 = 0.0, s1 = 0.0;
    const double mid = 0.5 * (double)(N - 1);
#pragma unroll
    for (int cc = 0; cc < kX8MaxChunks; ++cc) {
      const int c = lane + 64 * cc;
#pragma unroll
      for (int b = 0; b < 16; ++b) v[cc][b] = 0.0;
      if (c < nchunk) {
        const i32x4* yq = reinterpret_cast<const i32x4*>(tb.yx + 16 * c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 y4 = yq[q];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int kk = 16 * c + 4 * q + b;
            int sy = y4[b] >> 16, sx = y4[b] & 0xffff;
            const double xa = (double)(vertical ? sx : sy) - mid;
            if (flipped) { sy = N - 1 - sy; sx = N - 1 - sx; }
            int py = sy + oy, px = sx + ox;
            if (py >= N) py -= N;
            if (px >= N) px -= N;
            const double val = kk < tb.U ? master[(size_t)py * N + px] : 0.0;
            v[cc][4 * q + b] = val;
            s0 += val;
            s1 += kk < tb.U ? val * xa : 0.0;
          }
        }
      }
    }
    s0 = x8_wave_sum(s0);
    s1 = x8_wave_sum(s1);
    const double U = (double)tb.U;
    const double c1 = (s1 - s0 * tb.sx / U) / (tb.sxx - tb.sx * tb.sx / U);
    const double c0 = s0 / U - c1 * tb.sx / U;
    double m = 0.0;
#pragma unroll
    for (int cc = 0; cc < kX8MaxChunks; ++cc) {
      const int c = lane + 64 * cc;
      if (c < nchunk) {
        const i32x4* yq = reinterpret_cast<const i32x4*>(tb.yx + 16 * c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 y4 = yq[q];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int kk = 16 * c + 4 * q + b;
            const double xa = (double)(vertical ? (y4[b] & 0xffff) : (y4[b] >> 16)) - mid;
            const double zp = kk < tb.U ? v[cc][4 * q + b] - c0 - c1 * xa : 0.0;
            v[cc][4 * q + b] = zp;
            m = fmax(m, fabs(zp));
          }
        }
      }
    }
    m = x8_wave_max(m);
    const int ez = max(tb.ez_floor, m > 0.0 ? ilogb(m) + 1 : tb.ez_floor);   // 2^ez > every |zp|
    const double inv_qz = ldexp(1.0, 34 - ez);
    const size_t tile_base = (size_t)tile * p.KsTot_max;
    bool range_ok = isfinite(m);
    // digits of the detrended stencil
#pragma unroll
    for (int cc = 0; cc < kX8MaxChunks; ++cc) {
      const int c = lane + 64 * cc;
      if (c < nchunk) {
        uint32_t w[kX8DigZ][4];
#pragma unroll
        for (int t = 0; t < kX8DigZ; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) w[t][q] = 0u;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          int dg[kX8DigZ];
          x8_digits<kX8DigZ>((long long)rint(v[cc][b] * inv_qz), dg);
#pragma unroll
          for (int t = 0; t < kX8DigZ; ++t) w[t][b >> 2] |= (uint32_t)(dg[t] & 0xff) << (8 * (b & 3));
        }
        const int ks = c >> 1, g = c & 1;
#pragma unroll
        for (int t = 0; t < kX8DigZ; ++t) {
          u32x4 o = {w[t][0], w[t][1], w[t][2], w[t][3]};
          *reinterpret_cast<u32x4*>(p.Z8 + ((((tile_base + ks) * kX8DigZ + t) * 64 + (g * 32 + col)) << 4)) = o;
        }
      }
    }
    // digits of the normals: shift j' = 1 .. k, sample i' < N at contraction index (j' - 1) Np + i'
    const double inv_qn = ldexp(1.0, 34 - ez - tb.log2_cn);
    const uint32_t ext0 = p.ext_counter[env] + (phase == 1 ? (uint32_t)min(abs(dx), p.kcap) : 0u);
    const int r0 = phase == 1 ? min(abs(dx), p.kcap) : 0;   // index of this phase's first shift among the step's shifts (replay buffer)
    const int nchunk_n = tb.KsB * 2;
    for (int c = lane; c < nchunk_n; c += 64) {
      const int jj = (16 * c) / tb.Np, i0 = 16 * c - jj * tb.Np;   // shift jj + 1, samples i0 .. i0 + 15 (Np is a multiple of 64: no straddle)
      uint32_t w[kX8DigN][4];
#pragma unroll
      for (int t = 0; t < kX8DigN; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[t][q] = 0u;
      if (i0 < N) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double n4[4];
          const bool replay = p.noise && (r0 + jj) < p.max_ext;
          if (replay) {
#pragma unroll
            for (int b = 0; b < 4; ++b) n4[b] = (i0 + 4 * q + b) < N ? p.noise[((size_t)env * p.max_ext + (r0 + jj)) * N + i0 + 4 * q + b] : 0.0;
          } else {
            philox_normal4(p.seed, (uint32_t)(p.env_base + env), ext0 + (uint32_t)jj, (uint32_t)((i0 >> 2) + q), n4);
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double nv = (i0 + 4 * q + b) < N ? n4[b] : 0.0;
            const double ni = rint(nv * inv_qn);
            range_ok = range_ok && fabs(ni) < 17314086912.0;   // 2^34 + 2^27
            int dg[kX8DigN];
            x8_digits<kX8DigN>((long long)ni, dg);
#pragma unroll
            for (int t = 0; t < kX8DigN; ++t) w[t][q] |= (uint32_t)(dg[t] & 0xff) << (8 * b);
          }
        }
      }
      const int ks = tb.KsA + (c >> 1), g = c & 1;
#pragma unroll
      for (int t = 0; t < kX8DigN; ++t) {
        u32x4 o = {w[t][0], w[t][1], w[t][2], w[t][3]};
        *reinterpret_cast<u32x4*>(p.Z8 + ((((tile_base + ks) * kX8DigN + t) * 64 + (g * 32 + col)) << 4)) = o;
      }
    }
    if (!__all(range_ok)) {
      if (lane == 0) atomicOr(p.status, 4);
    }
    if (lane == 0) {
      double* rc = p.rec + (size_t)slot * 4;
      rc[0] = ldexp(1.0, tb.log2_qa + (ez - 34) + 28);   // qa qz 128^4
      rc[1] = c0;
      rc[2] = c1;
      rc[3] = __hiloint2double(oy, ox);                  // origin of the screen this phase reads: (oy in the high word, ox in the low)
    }
  }
  if (phase == 1 && lane == 0) {   // commit the step: the products read origins from `rec`, nothing reads these until the step's last launch has run
    p.origin[2 * env] = ox;
    p.origin[2 * env + 1] = ((oy + dy) % N + N) % N;
    p.ext_counter[env] += (uint32_t)(min(abs(dx), p.kcap) + min(abs(dy), p.kcap));
  }
}

// ---- product: workgroup = 64 rows x 64 envs (four waves, 2 x 2 tiles of 32 x 32), digits of both operands staged through LDS ---------
constexpr int kX8Blocks = 20;   // 1-KiB operand blocks per k-step: 2 row tiles x 5 digits + 2 env tiles x 5 digits

__device__ __forceinline__ void x8_store(const X8Args& p, int env, int N, int py, int px, double v) {
  p.master[(size_t)env * N * N + (size_t)py * N + px] = v;
  if (p.ring) {
    const int RS = N + 4;
    const float f = (float)((v - p.ring_ref[env]) * p.ring_inv);
    float* row = p.ring + ((size_t)env * N + py) * RS;
    row[px] = f;
    if (px < 4) row[N + px] = f;
  }
}

__global__ __launch_bounds__(256, 2) void k_x8_product(X8Args p, int phase) {
  __shared__ __attribute__((aligned(16))) int8_t lds[2][kX8Blocks * 1024];
  const int tile64 = blockIdx.x;
  const int k = p.tile_k[phase * p.tiles64_max + tile64];
  if (k == 0) return;
  const X8Table& tb = p.tables[(phase == 0 ? 1 : 0) * (kX8MaxK + 1) + k];
  const int rt0 = 2 * blockIdx.y;
  if (rt0 >= tb.RT) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rtl = wave & 1, ctl = wave >> 1;
  const int N = p.N;
  const int shift = rt0 / (tb.Np / 32);                         // both row tiles lie in shift block `shift` + 1 (Np / 32 is even)
  const int ksB_end = (shift + 1) * (tb.Np / 32);               // normals of later shifts do not reach these rows
  const int n_steps = tb.KsA + ksB_end;
  const size_t zt0 = (size_t)(2 * tile64) * p.KsTot_max, zt1 = (size_t)(2 * tile64 + 1) * p.KsTot_max;

  // the 1-KiB block `blk` of step `st`: A-part st < KsA: blocks 0..9 = A digits of the two row tiles, 10..19 = Z digits of the two env tiles;
  // B-part: blocks 0..7 = B digits (4 per row tile), 10..19 = N digits
  auto block_ptr = [&](int st, int blk) -> const int8_t* {
    if (blk >= 10) {
      const int h = (blk - 10) / 5, dgt = (blk - 10) % 5;
      return p.Z8 + ((((h ? zt1 : zt0) + st) * 5 + dgt) << 10);
    }
    if (st < tb.KsA) {
      const int h = blk / 5, dgt = blk % 5;
      return tb.A8 + ((((size_t)(rt0 + h) * tb.KsA + st) * kX8DigA + dgt) << 10);
    }
    const int h = blk >> 2, dgt = blk & 3;   // blocks 8, 9 unused in the B-part
    return tb.B8 + ((((size_t)(rt0 + h) * tb.KsB + (st - tb.KsA)) * kX8DigB + dgt) << 10);
  };
  // wave w stages blocks w, w + 4, ... of a step (one contiguous KiB per wave instruction)
  i32x4 stage[5];
  auto load_step = [&](int st) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int blk = 4 * r + wave;
      const bool used = st < tb.KsA || blk < 8 || blk >= 10;
      stage[r] = used ? *reinterpret_cast<const i32x4*>(block_ptr(st, blk) + (lane << 4)) : i32x4{0, 0, 0, 0};
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 5; ++r) *reinterpret_cast<i32x4*>(&lds[buf][(4 * r + wave) * 1024 + (lane << 4)]) = stage[r];
  };

  i32x16 acc[kX8Levels];
#pragma unroll
  for (int l = 0; l < kX8Levels; ++l)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[l][r] = 0;

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int st = 0; st < n_steps; ++st) {
    const int buf = st & 1;
    if (st + 1 < n_steps) load_step(st + 1);
    i32x4 a[5], z[5];
    const bool apart = st < tb.KsA;
#pragma unroll
    for (int dgt = 0; dgt < 5; ++dgt) z[dgt] = *reinterpret_cast<const i32x4*>(&lds[buf][(10 + ctl * 5 + dgt) * 1024 + (lane << 4)]);
    if (apart) {
#pragma unroll
      for (int dgt = 0; dgt < 5; ++dgt) a[dgt] = *reinterpret_cast<const i32x4*>(&lds[buf][(rtl * 5 + dgt) * 1024 + (lane << 4)]);
#pragma unroll
      for (int s = 0; s < kX8DigA; ++s)
#pragma unroll
        for (int t = 0; t < kX8DigZ; ++t)
          if (s + t < kX8Levels) acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], z[t], acc[s + t], 0, 0, 0);
    } else {
#pragma unroll
      for (int dgt = 0; dgt < 4; ++dgt) a[dgt] = *reinterpret_cast<const i32x4*>(&lds[buf][(rtl * 4 + dgt) * 1024 + (lane << 4)]);
#pragma unroll
      for (int s = 0; s < kX8DigB; ++s)
#pragma unroll
        for (int t = 0; t < kX8DigN; ++t)
          if (s + t < kX8Levels) acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], z[t], acc[s + t], 0, 0, 0);
    }
    if (st + 1 < n_steps) store_step(buf ^ 1);
    __syncthreads();
  }

  // epilogue: fixed point -> float64, add the exact piston / tilt response, scatter into the toroidal master (and its fp32 ring copy)
  const int slot = (2 * tile64 + ctl) * 32 + (lane & 31);
  const int env = p.list[phase * p.slots_max + slot];
  if (env < 0) return;
  const double* rc = p.rec + (size_t)slot * 4;
  const double scale = rc[0], c0 = rc[1], c1 = rc[2];
  const int oy = __double2hiint(rc[3]), ox = __double2loint(rc[3]);
  const int d = p.dxy[2 * env + phase];
  const bool flipped = d > 0, vertical = phase == 1;
  const int j = shift + 1;   // these rows are the slice shift j creates
  const int row_base = (rt0 + rtl) * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int i = row - shift * tb.Np;
    if (i >= N) continue;
    // sum_l acc_l 128^(4 - l): exact in float64 up to its last bit (|acc| < 2^27)
    double f = (double)acc[0][r];
#pragma unroll
    for (int l = 1; l < kX8Levels; ++l) f = f * 128.0 + (double)acc[l][r];
    const double val = f * scale + c0 * tb.r1[row] + c1 * tb.r2[row];   // scale = qa qz 128^4: level l carries 128^(8 - l)
    int py, px;
    if (vertical) {
      py = flipped ? oy + j - 1 : oy - j;
      px = (flipped ? N - 1 - i : i) + ox;
    } else {
      px = flipped ? ox + j - 1 : ox - j;
      py = (flipped ? N - 1 - i : i) + oy;
    }
    py = ((py % N) + N) % N;
    px = ((px % N) + N) % N;
    x8_store(p, env, N, py, px, val);
  }
}

}  // namespace aog
