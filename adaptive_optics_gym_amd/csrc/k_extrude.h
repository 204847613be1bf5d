// Dynamic atmosphere: float64 AR extrusion kernels (validation forms) and the master / ring maintenance kernels.
#pragma once
#include "k_common.h"

namespace aog {

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

}  // namespace aog
