// Dynamic atmosphere on the int8 matrix cores: hcipy InfiniteAtmosphericLayer.evolve_until / _extrude (AO_env.py:125) as ONE exact
// fixed-point matrix product per axis and step.
//
// What is computed.  k successive one-pixel shifts along an axis are one linear map of the screen as it stood before them and of the k N
// normals they draw (extrusion_host.compose_extrusions; uploaded through aog_upload_layer_composite):
//     [R_1; ...; R_k] = A_k z + sqrt(Cn^2) B_k n,      z = screen[union stencil], n = [n_1; ...; n_k]
// so a step is: plan -> (prepare, product) for the x shifts -> (prepare, product) for the y shifts: no chain of k dependent rounds,
// no inter-workgroup barrier, every row of the product independent of every other.  Slice j depends on the old screen and on the normals
// of shifts 1 .. j only, so ONE table per axis (the k_max-shift operator, stencil columns ordered by first use) serves the envs of every
// shift count.  The x phase writes operands and staged columns only — the screens first change in the y phase's launches — so the
// host (x8_evolve, atmosphere.hip) runs the plan and the x phase of step t + 1 on a side stream beside step t's fused kernel.
//
// How it keeps float64-grade accuracy on an 8-bit pipe.  The AR recursion amplifies a white error of e rad per new sample to ~13 e of smooth
// phase error (measured on the host: profiles/HISTORY.md), so new samples must be good to ~1e-8 rad of ~10: fp32 accumulation cannot do
// that.  Integer accumulation can: every operand is written in balanced base-128 digits (int8),
//     A_k = qa sum_s As 128^(4-s)   (5 digits, quantum qa = 2^(ea-34)),      z - c0 - c1 x = qz sum_t Zt 128^(4-t)   (5 digits, qz per env)
//     sqrt(Cn^2) B_k = qb sum_s Bs 128^(4-s)   (5 digits),                     n = qn sum_t Nt 128^(4-t)               (5 digits)
// and v_mfma_i32_32x32x32_i8 sums the digit products EXACTLY in int32 (|digit product| <= 2^12, <= 2^26.5 per accumulator over the whole
// contraction).  Products of equal weight 128^(8-l), l = s + t, share one accumulator; levels l <= 5 are kept (19 of the 25 digit pairs: with
// l <= 4 the dropped level-5 terms, 5e-9 rad per new sample at N = 256, were ten times the quantisation floor; so was a 4-digit B: both
// measured on the host, profiles/HISTORY.md).  With qn = qa qz / qb per env the noise product lands on the same levels.  The
// only rounding anywhere is the quantisation of the operands (A to 2^-34 of its largest coefficient, z to 2^-34 of the env's largest
// detrended stencil sample): ~1e-9 rad per new sample at N = 256.  Piston and tilt of the stencil never enter the fixed-point product:
// A z = c0 (A 1) + c1 (A x) + A (z - c0 - c1 x) with the two vectors A 1 and A x exact in float64 — a quantisation error of A multiplied by
// a 30-rad piston would otherwise be the largest term (measured: 7e-9 against 1e-9 rad).
// Results do not depend on how envs are grouped into tiles (integer sums are exact and per column), so the Philox streams keyed by the global
// env id keep "split == whole" bit for bit.
#pragma once
#include "k_common.h"

// developer experiments (timing ablations that break the results) exist only in -DAOG_DEV builds
#ifdef AOG_DEV
#define AOG_X8_DEV(p) ((p).dev)
#else
#define AOG_X8_DEV(p) 0
#endif

namespace aog {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kX8Levels = 6;        // product levels l = s + t kept: 0 .. 5 (19 of the 25 digit pairs)
constexpr int kX8MaxK = 8;          // composite operators are built for k = 1 .. kcap <= 8 shifts per axis and step
constexpr int kX8PadSteps = 3;      // zero steps behind every table row tile (and behind Z8): the product runs its steps in fours

// composite operator of one axis in device memory, as seen by shift count k: the arrays are the K-shift operator's (shared by every k: the rows
// of shift j are the same in every operator of at least j shifts, and the stencil columns are ordered by first use), U / KsA / KsB / RT / sx /
// sxx say how much of them k shifts use
struct X8Table {
  const int32_t* yx;      // [U_pad] union stencil (sy << 16 | sx) on the screen hcipy's _extrude sees (rotated for 'top' / 'right'); padding repeats entry 0
  const int8_t* T8;       // [K Np / 32][KsT][5][64][16] digits of [A_K | sqrt(Cn^2) B_K | kX8PadSteps zero steps] as MFMA A operands: lane (row & 31, k-half g), byte b <-> column 32 ks + 16 g + b
  const double* r1;       // [RT * 32] A_k 1   (float64, exact piston response)
  const double* r2;       // [RT * 32] A_k x   (x = along-coordinate of the stencil sample - (N - 1) / 2)
  int k, U, KsA, KsB, RT; // U union size; k-steps (of 32) of the stencil and of the normals; row tiles (of 32 rows) = k Np / 32
  int KsAmax, KsT;        // the K-shift operator's stencil steps (the normals' steps start there); steps per table row tile: KsAmax + K Np / 32 + kX8PadSteps
  int Np;                 // rows per shift block: N rounded up to 64 (row (j - 1) Np + i = sample i of the slice shift j creates; i >= N: zero rows)
  int log2_qa;            // qa = 2^log2_qa
  int log2_cn;            // qn = qz * 2^log2_cn  (= qa / qb)
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
  int32_t* items;            // [2][items_max] workgroup -> (64-env tile | row pair << 16) of k_x8_product, heaviest first, -1 = none
  int tiles64_max, slots_max, items_max;
  // prepared operands
  int8_t* Z8;                // [tiles32_max][KsTot_max][5][64][16]
  int KsTot_max;
  double* rec;               // [slots_max][4]: scale (qa qz 128^3), c0, c1, origin of the screen the phase reads
  double* colbuf;            // [slots_max][kcap][Np] the new COLUMNS of phase 0, one contiguous run per (env slot, shift): k_x8_prepare of phase 1 puts them into the screens
  int dev;                   // AOG_DEV builds only (AOG_X8_DEV in the environment): 4 no result stores, prepare kernel: 16 no normals, 32 no digits, 64 no gather, 128 no column scatter, 1024 cycles per step read-out
  int* status;               // sticky error word (bit 2: a stencil sample or a normal left its fixed-point range)
};

// ---- plan: shifts of every env this step; envs packed into 64-env tiles, most shifts first (a tile runs the rows of its first env's shift
// count; an env stores the slices of its own shifts only); the product's workgroups dealt to the XCDs -----------------------------------------
constexpr int kX8PlanThreads = 1024;
constexpr int kX8MaxPlanTiles = 1024;   // 64-env tiles per axis the plan kernel's shared arrays hold (B <= 65536)
constexpr int kX8MaxPlanGroups = 8 * 64;
__host__ __device__ constexpr int x8_chunk_tiles(int tiles) { return tiles <= 128 ? 4 : (tiles + 31) / 32; }   // tiles per workgroup group: at most 32 groups per shift
__global__ __launch_bounds__(kX8PlanThreads) void k_x8_plan(X8Args p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kX8PlanThreads / 64, NC = kX8MaxK + 1;
  __shared__ int wcnt[2][NC][NW];      // per wave and class: envs of this pass
  __shared__ int base[2][NC];          // slots handed out so far per class
  __shared__ int total[2][NC];
  __shared__ int ntiles[2], qlen[2][8], ntj[2][kX8MaxK + 1];
  __shared__ short tk[2][kX8MaxPlanTiles];                      // tile -> shifts of its first (= every other's at most) env
  __shared__ short gx[2][kX8MaxPlanGroups];                     // group -> XCD
  __shared__ int gbase[2][kX8MaxPlanGroups];                    // group -> first position in that XCD's queue
  if (tid < 2 * NC) { (&base[0][0])[tid] = 0; (&total[0][0])[tid] = 0; }
  for (int i = tid; i < 2 * p.slots_max; i += kX8PlanThreads) p.list[i] = -1;
  for (int i = tid; i < 2 * p.items_max; i += kX8PlanThreads) p.items[i] = -1;
  __syncthreads();
  // pass A: shifts and class totals (slot bases need the totals of every class before any slot can be handed out)
  for (int e0 = 0; e0 < p.B; e0 += kX8PlanThreads) {
    const int e = e0 + tid;
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
      const unsigned long long mx = __ballot(kx == k), my = __ballot(ky == k);
      if (lane == 0) {
        if (mx) atomicAdd(&total[0][k], __popcll(mx));
        if (my) atomicAdd(&total[1][k], __popcll(my));
      }
    }
  }
  __syncthreads();
  if (tid < 2) {   // slot bases per class, most shifts first, no gaps
    int b = 0;
    for (int k = p.kcap; k >= 1; --k) {
      base[tid][k] = b;
      b += total[tid][k];
    }
    ntiles[tid] = min((b + 63) / 64, min(p.tiles64_max, kX8MaxPlanTiles));
    if ((b + 63) / 64 > ntiles[tid]) atomicOr(p.status, 4);   // (B beyond what the plan's arrays hold: aog_upload_layer_composite refuses it)
  }
  __syncthreads();
  for (int i = tid; i < 2 * p.tiles64_max; i += kX8PlanThreads) {
    const int ph = i / p.tiles64_max, t = i - ph * p.tiles64_max;
    int k = 0;
    if (t < ntiles[ph]) {
      k = p.kcap;
      while (k > 1 && base[ph][k] + total[ph][k] <= 64 * t) --k;   // the class of slot 64 t
    }
    p.tile_k[i] = k;
    if (t < kX8MaxPlanTiles) tk[ph][t] = (short)k;
  }
  __syncthreads();
  // the product's workgroups: one per (64-env tile, row pair of 64 table rows).  What bounds that kernel at B = 1024 is operand traffic over the
  // fabric into each XCD's L2 (measured: 2500 .. 3600 cycles per double step where every workgroup of an XCD streams its own table rows,
  // 1300 — the matrix pipe's pace — where they are shared), so workgroups go to XCDs (workgroup L runs on XCD L mod 8) in GROUPS of the 4 row
  // pairs of one shift x a chunk of tiles: 8 table row tiles and at most 2 x chunk env tiles feed up to 4 x chunk workgroups from one L2.
  // Groups are dealt, largest (shift 1: every tile) first, to the XCD with the shortest queue — no queue ends more than a group longer than
  // another — and each queue then runs backwards: last shifts, the longest products, first.
  if (tid < 2 * NC) {   // tiles of at least j shifts (tiles are sorted: they come first)
    const int ph = tid / NC, j = tid % NC;
    int n = 0;
    while (n < ntiles[ph] && tk[ph][n] >= max(j, 1)) ++n;
    ntj[ph][j] = n;
  }
  __syncthreads();
  if (tid == 0 || tid == 512) {
    const int ph = tid >> 9, ct = x8_chunk_tiles(ntiles[ph]), rps = p.tables[(ph == 0 ? 1 : 0) * (kX8MaxK + 1) + 1].Np / 64;
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (only ever indexed by unrolled loops: registers)
    int g = 0;
    for (int j = 1; j <= p.kcap; ++j) {
      const int n = ntj[ph][j];
      for (int c0 = 0; c0 < n; c0 += ct, ++g) {
        int x = 0, best = cnt[0];
#pragma unroll
        for (int q = 1; q < 8; ++q)
          if (cnt[q] < best) { best = cnt[q]; x = q; }
        gx[ph][g] = (short)x;
        gbase[ph][g] = best;
        const int add = rps * min(ct, n - c0);
#pragma unroll
        for (int q = 0; q < 8; ++q) cnt[q] += q == x ? add : 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) qlen[ph][q] = cnt[q];
  }
  __syncthreads();
  {
    const int ph = tid >> 9, t0 = tid & 511, ct = x8_chunk_tiles(ntiles[ph]), rps = p.tables[(ph == 0 ? 1 : 0) * (kX8MaxK + 1) + 1].Np / 64;
    int g0 = 0;
    for (int j = 1; j <= p.kcap; ++j) {
      const int n = ntj[ph][j];
      for (int i = t0; i < rps * n; i += 512) {   // workgroup i of shift j: tile i % n, row pair (j - 1) rps + i / n
        const int t = i % n, r = i / n, g = g0 + t / ct, tw = min(ct, n - (t / ct) * ct);
        const int x = gx[ph][g], fwd = gbase[ph][g] + r * tw + (t % ct), L = (qlen[ph][x] - 1 - fwd) * 8 + x;
        if (L < p.items_max) p.items[ph * p.items_max + L] = t | (((j - 1) * rps + r) << 16);
        else atomicOr(p.status, 4);   // (cannot happen: see x8_ensure_buffers)
      }
      g0 += (n + ct - 1) / ct;
    }
  }
  __syncthreads();
  // pass B: slots in env order within each class (deterministic, though no result depends on it)
  for (int e0 = 0; e0 < p.B; e0 += kX8PlanThreads) {
    const int e = e0 + tid;
    const int kx = e < p.B ? min(abs(p.dxy[2 * e]), p.kcap) : 0, ky = e < p.B ? min(abs(p.dxy[2 * e + 1]), p.kcap) : 0;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int rx = 0, ry = 0;
    for (int k = 1; k <= p.kcap; ++k) {
      const unsigned long long mx = __ballot(kx == k), my = __ballot(ky == k);
      if (kx == k) rx = __popcll(mx & below);
      if (ky == k) ry = __popcll(my & below);
      if (lane == 0) {
        wcnt[0][k][wave] = __popcll(mx);
        wcnt[1][k][wave] = __popcll(my);
      }
    }
    __syncthreads();
    int sx = -1, sy = -1;
    if (kx) {
      int b = base[0][kx];
      for (int w = 0; w < wave; ++w) b += wcnt[0][kx][w];
      sx = b + rx;
    }
    if (ky) {
      int b = base[1][ky];
      for (int w = 0; w < wave; ++w) b += wcnt[1][ky][w];
      sy = b + ry;
    }
    if (e < p.B) {
      p.slot[e] = sx;
      p.slot[p.B + e] = sy;
      if (sx >= 0) p.list[sx] = e;
      if (sy >= 0) p.list[p.slots_max + sy] = e;
    }
    __syncthreads();
    if (tid < 2 * NC) {
      const int ph = tid / NC, k = tid % NC;
      int add = 0;
      for (int w = 0; w < NW; ++w) add += wcnt[ph][k][w];
      base[ph][k] += add;
    }
    __syncthreads();
  }
}

// balanced base-128 digits of the integer-valued x (|x| <= 2^34), most significant first.  x = hi 2^21 + lo with |lo| <= 2^20, both int32:
// three float64 operations, the rest 32-bit integer arithmetic.  Any digits in [-128, 127] that sum to x are as good as any other (the
// product is exact); these stay within [-65, 64].
__device__ __forceinline__ void x8_digits5(double x, int (&d)[5]) {
  const double h = rint(x * (1.0 / 2097152.0));
  int lo = (int)fma(h, -2097152.0, x), hi = (int)h;
  d[4] = ((lo + 64) & 127) - 64;
  lo = (lo - d[4]) >> 7;
  d[3] = ((lo + 64) & 127) - 64;
  d[2] = (lo - d[3]) >> 7;
  d[1] = ((hi + 64) & 127) - 64;
  d[0] = (hi - d[1]) >> 7;
}

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

// ---- prepare: one workgroup per env, one 16-sample chunk per thread: the first threads gather the union stencil (detrended, quantised with
// the env's own quantum), the last ones draw the normals; both write base-128 digits in the product's B-operand order.  Phase 1 also commits
// the step's origin and stream position. ---------------------------------------------------------------------------------------------------
constexpr int kX8PrepMaxThreads = 512;   // >= stencil chunks + normal chunks of every operator built (aog_upload_layer_composite checks)
__device__ __forceinline__ double x8_block_sum(double v, double* sm) {   // fixed association order: reproducible
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
  return r;
}
__device__ __forceinline__ double x8_block_max(double v, double* sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, sm[w]);
  return r;
}

__global__ __launch_bounds__(kX8PrepMaxThreads) void k_x8_prepare(X8Args p, int phase) {
  __shared__ double sm[kX8PrepMaxThreads / 64];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int env = blockIdx.x;
  const int N = p.N;
  const int dx = p.dxy[2 * env], dy = p.dxy[2 * env + 1];
  const int d = phase == 0 ? dx : dy, k = min(abs(d), p.kcap);
  int ox = p.origin[2 * env], oy = p.origin[2 * env + 1];
  if (phase == 1) ox = ((ox + dx) % N + N) % N;   // the x shifts of this step have been applied
  const uint32_t ext_old = p.ext_counter[env];
  if (phase == 1 && dx != 0 && !(AOG_X8_DEV(p) & 128)) {
    // the columns phase 0 made go into the screen here, a row's (adjacent) columns by adjacent lanes: one partly written cache line per row
    // and array where the product's own stores would leave one per row AND shift (~2 M scattered 8-byte stores a step at B = 1024: their
    // write-backs cost the x phase ~25 us)
    const int kx = min(abs(dx), p.kcap), sl0 = p.slot[env], Np = p.tables[kX8MaxK + 2].Np, ox0 = p.origin[2 * env];
    const double* cb = p.colbuf + (size_t)sl0 * p.kcap * Np;
    for (int idx = tid; idx < N * kx; idx += nthr) {   // neighbouring lanes: neighbouring columns of one row (the memory pipeline merges them)
      const int i = idx / kx, j = idx - i * kx + 1;
      int py = (dx > 0 ? N - 1 - i : i) + oy;
      if (py >= N) py -= N;
      int px = dx > 0 ? ox0 + j - 1 : ox0 - j;
      px = ((px % N) + N) % N;
      x8_store(p, env, N, py, px, cb[(size_t)(j - 1) * Np + i]);
    }
  }
  __syncthreads();   // (every thread has read the origin and the stream position before thread 0 commits the step below; the new columns are in place)
  if (phase == 1 && tid == 0) {   // commit the step: the products read origins from `rec`, nothing reads these before the step's last launch has run
    p.origin[2 * env] = ox;
    p.origin[2 * env + 1] = ((oy + dy) % N + N) % N;
    p.ext_counter[env] = ext_old + (uint32_t)(min(abs(dx), p.kcap) + min(abs(dy), p.kcap));
  }
  if (k == 0) return;
  const X8Table& tb = p.tables[(phase == 0 ? 1 : 0) * (kX8MaxK + 1) + k];   // x shifts use the horizontal ('left') operator
  const bool vertical = phase == 1, flipped = d > 0;
  const int slot = p.slot[phase * p.B + env];
  const int tile = slot >> 5, col = slot & 31;
  const double* master = p.master + (size_t)env * N * N;
  const int nchunk = tb.KsA * 2, nchunk_n = tb.KsB * 2;   // 16-sample chunks of the (padded) union stencil / of the normals
  const double mid = 0.5 * (double)(N - 1);
  const int r0 = phase == 1 ? min(abs(dx), p.kcap) : 0;   // index of this phase's first shift among the step's shifts (stream position, replay buffer)
  const bool zjob = tid < nchunk;
  const int cn = nthr - 1 - tid;                          // normals' chunks are dealt from the last thread down: a thread has one job
  const bool njob = !zjob && cn < nchunk_n;

  double x[16];     // the job's 16 samples: stencil values (then detrended) or normals
  i32x4 y4[4];      // stencil jobs: the samples' positions
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int b = 0; b < 16; ++b) x[b] = 0.0;
  if (zjob) {
    const i32x4* yq = reinterpret_cast<const i32x4*>(tb.yx + 16 * tid);
#pragma unroll
    for (int q = 0; q < 4; ++q) y4[q] = yq[q];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int sy = y4[q][b] >> 16, sx = y4[q][b] & 0xffff;
        if (flipped) { sy = N - 1 - sy; sx = N - 1 - sx; }
        int py = sy + oy, px = sx + ox;
        if (py >= N) py -= N;
        if (px >= N) px -= N;
        const bool in = 16 * tid + 4 * q + b < tb.U;
        x[4 * q + b] = (in && !(AOG_X8_DEV(p) & 64)) ? master[(size_t)py * N + px] : 0.0;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double xa = (double)(vertical ? (y4[q][b] & 0xffff) : (y4[q][b] >> 16)) - mid;
        s0 += x[4 * q + b];
        s1 += 16 * tid + 4 * q + b < tb.U ? x[4 * q + b] * xa : 0.0;
      }
  } else if (njob && !(AOG_X8_DEV(p) & 16)) {
    const int jj = (16 * cn) / tb.Np, i0 = 16 * cn - jj * tb.Np;   // shift jj + 1, samples i0 .. i0 + 15 (Np is a multiple of 64: no straddle)
    if (i0 < N) {
      const bool replay = p.noise && (r0 + jj) < p.max_ext;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double n4[4];
        if (replay) {
#pragma unroll
          for (int b = 0; b < 4; ++b) n4[b] = (i0 + 4 * q + b) < N ? p.noise[((size_t)env * p.max_ext + (r0 + jj)) * N + i0 + 4 * q + b] : 0.0;
        } else {
          philox_normal4(p.seed, (uint32_t)(p.env_base + env), ext_old + (uint32_t)(r0 + jj), (uint32_t)((i0 >> 2) + q), n4);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) x[4 * q + b] = (i0 + 4 * q + b) < N ? n4[b] : 0.0;
      }
    }
  }
  s0 = x8_block_sum(s0, sm);
  s1 = x8_block_sum(s1, sm);
  const double U = (double)tb.U;
  const double c1 = (s1 - s0 * tb.sx / U) / (tb.sxx - tb.sx * tb.sx / U);
  const double c0 = s0 / U - c1 * tb.sx / U;
  double m = 0.0;
  if (zjob) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double xa = (double)(vertical ? (y4[q][b] & 0xffff) : (y4[q][b] >> 16)) - mid;
        const double zp = 16 * tid + 4 * q + b < tb.U ? x[4 * q + b] - c0 - c1 * xa : 0.0;
        x[4 * q + b] = zp;
        m = fmax(m, fabs(zp));
      }
  }
  m = x8_block_max(m, sm);
  const int ez = max(tb.ez_floor, m > 0.0 ? ilogb(m) + 1 : tb.ez_floor);   // 2^ez > every |zp|
  if ((zjob || njob) && !(AOG_X8_DEV(p) & 32)) {
    const double inv_q = zjob ? ldexp(1.0, 34 - ez) : ldexp(1.0, 34 - ez - tb.log2_cn);
    const int c = zjob ? tid : cn, ks = (zjob ? 0 : tb.KsAmax) + (c >> 1), g = c & 1;
    bool range_ok = isfinite(m);
    uint32_t w[5][4];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) w[t][q] = 0u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const double xi = rint(x[b] * inv_q);
      range_ok = range_ok && fabs(xi) <= 17179869184.0;   // 2^34
      int dg[5];
      x8_digits5(xi, dg);
#pragma unroll
      for (int t = 0; t < 5; ++t) w[t][b >> 2] |= (uint32_t)(dg[t] & 0xff) << (8 * (b & 3));
    }
    int8_t* dst = p.Z8 + ((((size_t)tile * p.KsTot_max + ks) * 5 * 64 + (g * 32 + col)) << 4);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      u32x4 o = {w[t][0], w[t][1], w[t][2], w[t][3]};
      *reinterpret_cast<u32x4*>(dst + t * 1024) = o;
    }
    if (!range_ok) atomicOr(p.status, 4);
  }
  if (tid == 0) {
    double* rc = p.rec + (size_t)slot * 4;
    rc[0] = ldexp(1.0, tb.log2_qa + (ez - 34) + 21);   // qa qz 128^3: level l of the product carries 128^(8 - l), l = 0 .. 5
    rc[1] = c0;
    rc[2] = c1;
    rc[3] = __hiloint2double(oy, ox);                  // origin of the screen this phase reads (oy in the high word, ox in the low)
  }
}


// ---- product: workgroup = 64 rows x 64 envs = 2 x 2 tiles of 32 x 32, EIGHT waves: wave = (tile, half); the two waves of a tile sit on the same
// SIMD and share the tile's 32-deep steps (even / odd), each into its own six int32 accumulators (19 digit products per step) — summed at the
// end, exactly, so the split changes no bit.  Every wave moves its share of its half's operand blocks (5 of the 20 one-KiB blocks of a step: 2
// row tiles x 5 digits + 2 env tiles x 5) global -> LDS by LDS-DMA, kX8Stages - 1 double steps ahead (counted vmcnt waits), reads the NEXT
// step's ten operand blocks from LDS and issues the loads of a later stage in between this step's matrix instructions; one raw s_barrier per
// double step.  Workgroups come from the plan's list (heaviest first, balanced over the XCDs).
// How it got here (B = 1024, N = 256, v = 10 m/s: ~250 workgroups per phase on 256 CUs — one tile per SIMD; profiles/HISTORY.md): a tile's
// steps are a serial chain on one SIMD (up to 81 x 608 cycles of matrix instructions) and with one wave per SIMD nothing hides a stall: four
// consumer + four loader waves ran 950 .. 1800 cycles per step (measured per workgroup) and the longest chain set the launch time (67 .. 78 us
// against ~18 us of matrix work).  Two waves per SIMD halve the chain and fill each other's stalls.
constexpr int kX8Blocks = 20;   // 1-KiB operand blocks per 32-deep step
constexpr int kX8Halves = 2;    // waves per tile = steps per stage
constexpr int kX8Stages = 3;    // an LDS-DMA lands ~1.1 us after its issue (MI355X_MICROARCH.md, ldsdma-fill): loads run two double steps ahead
constexpr int kX8ProductLds = kX8Stages * kX8Halves * kX8Blocks * 1024;   // (the epilogue's exchange / transposition areas live in the ring)

__global__ __launch_bounds__(512, 2) void k_x8_product(X8Args p, int phase) {
  extern __shared__ __attribute__((aligned(1024))) int8_t lds8[];   // kX8ProductLds bytes
  const int item = p.items[phase * p.items_max + (int)blockIdx.x];
  if (item < 0) return;
  const int tile64 = item & 0xffff, rt0 = 2 * (item >> 16);
  const int axis = phase == 0 ? 1 : 0;
  const int Np = p.tables[axis * (kX8MaxK + 1) + 1].Np;
  const int shift = rt0 / (Np / 32);                            // both row tiles lie in shift block `shift` + 1 (Np / 32 is even)
  const X8Table& tb = p.tables[axis * (kX8MaxK + 1) + shift + 1];   // (the shared arrays; KsA: the stencil steps the rows of this shift read)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (scalar: everything selected by it is wave-uniform)
  const int t_wg0 = (AOG_X8_DEV(p) & 1024) ? (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffff) : 0;   // 10 ns ticks
  const int tile = wave & 3, half = wave >> 2;
  const int rtl = tile & 1, ctl = tile >> 1;
  const int N = p.N;
  const int KsA = tb.KsA, gap = tb.KsAmax - KsA;                // stencil steps 0 .. KsA - 1, then the normals' steps from KsAmax on
  const int n_steps = KsA + (shift + 1) * (Np / 32);            // normals of later shifts do not reach these rows
  // double steps (this wave's step of double step q is 2 q + half), an even count: the steps past n_steps multiply by exact zeros — normals of
  // later shifts (their B blocks are zero for these rows) or the table's zero padding
  const int n_q = ((n_steps + 3) >> 2) << 1;
  const int st_last = tb.KsT - 1;

  // block `blk` of a step: 0..9 = digits of the two row tiles of [A_k | sqrt(Cn^2) B_k], 10..19 = digits of the two env tiles; this wave moves
  // blocks tile, tile + 4, ..., tile + 16 of its own half's step.  Every block advances 5 KiB per step.
  const int8_t* src0[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int blk = 4 * r + tile, h = blk >= 10 ? (blk - 10) / 5 : blk / 5, dgt = blk % 5;
    src0[r] = blk >= 10 ? p.Z8 + ((((size_t)(2 * tile64 + h) * p.KsTot_max) * 5 + dgt) << 10) : tb.T8 + ((((size_t)(rt0 + h) * tb.KsT) * 5 + dgt) << 10);
  }
  auto issue1 = [&](int stage, int buf, int r) {
    const int sq = stage * kX8Halves + half;
    const unsigned off = (unsigned)min(sq < KsA ? sq : sq + gap, st_last) * 5120u + ((unsigned)lane << 4);   // (past the end: a harmless re-load keeps the vmcnt arithmetic uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0[r] + off),
                                     (__attribute__((address_space(3))) void*)(&lds8[((buf * kX8Halves + half) * kX8Blocks + 4 * r + tile) * 1024]), 16, 0, 0);
  };

  i32x16 acc[kX8Levels];
#pragma unroll
  for (int l = 0; l < kX8Levels; ++l)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[l][r] = 0;
  i32x4 a[5], z[5], an[5], zn[5];
  // one step: the ten LDS reads of this wave's NEXT step's operands and its five loads of a later stage ride in between this step's matrix
  // instructions (an MFMA holds the issue port 8 of its 32 cycles); digit-major order: consecutive matrix instructions never write the same
  // accumulator
  auto step = [&](int read_buf, int load_stage, int load_buf, i32x4 (&ac)[5], i32x4 (&zc)[5], i32x4 (&ax)[5], i32x4 (&zx)[5]) {
    const int8_t* sb = &lds8[(read_buf * kX8Halves + half) * kX8Blocks * 1024];
    int q = 0;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (s + t < kX8Levels) {
          acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ac[s], zc[t], acc[s + t], 0, 0, 0);
          if (q < 5) ax[q] = *reinterpret_cast<const i32x4*>(sb + (rtl * 5 + q) * 1024 + (lane << 4));
          else if (q < 10) zx[q - 5] = *reinterpret_cast<const i32x4*>(sb + (10 + ctl * 5 + q - 5) * 1024 + (lane << 4));
          else if (q >= 11 && q < 16) issue1(load_stage, load_buf, q - 11);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
  };
  auto next = [](int b) { return b + 1 == kX8Stages ? 0 : b + 1; };

#pragma unroll
  for (int st = 0; st < kX8Stages; ++st)
#pragma unroll
    for (int r = 0; r < 5; ++r) issue1(st, st, r);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (kX8Stages - 1)) : "memory");   // this wave's blocks of stage 0 have landed
  __builtin_amdgcn_s_barrier();                                                  // ... and everybody's
  asm volatile("" ::: "memory");
  {
    const int8_t* sb = &lds8[half * kX8Blocks * 1024];
#pragma unroll
    for (int dgt = 0; dgt < 5; ++dgt) {
      a[dgt] = *reinterpret_cast<const i32x4*>(sb + (rtl * 5 + dgt) * 1024 + (lane << 4));
      z[dgt] = *reinterpret_cast<const i32x4*>(sb + (10 + ctl * 5 + dgt) * 1024 + (lane << 4));
    }
  }
  const long long t_loop0 = (AOG_X8_DEV(p) & 1024) ? (long long)__builtin_amdgcn_s_memtime() : 0;
  // double step q (operands of stage q in registers): behind the barrier every wave has read stage q (so its buffer takes stage q + kX8Stages)
  // and stage q + 1 has landed (so the next step's operands can be read)
  int buf = 0;
  for (int q = 0; q < n_q; q += 2) {
    __builtin_amdgcn_s_waitcnt(0x0070 | (5 * (kX8Stages - 2)));   // lgkmcnt(0), vmcnt(5 (kX8Stages - 2)) — as a builtin, so that the compiler's own wait insertion knows it
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    step(next(buf), q + kX8Stages, buf, a, z, an, zn);
    buf = next(buf);
    __builtin_amdgcn_s_waitcnt(0x0070 | (5 * (kX8Stages - 2)));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    step(next(buf), q + 1 + kX8Stages, buf, an, zn, a, z);
    buf = next(buf);
  }
  if ((AOG_X8_DEV(p) & 1024) && wave == 0 && lane == 0) {   // developer read-out: cycles per step of the slowest and of the fastest workgroup, steps of the longest, timeline
    const int cyc = (int)(((long long)__builtin_amdgcn_s_memtime() - t_loop0) / n_q);
    atomicMax(p.status + 8, cyc);
    atomicMin(p.status + 9, cyc);
    atomicMax(p.status + 10, n_steps);
    atomicAdd(p.status + 11, 1);
    atomicMin(p.status + 2, t_wg0);
    atomicMax(p.status + 3, t_wg0);
    atomicMax(p.status + 4, (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffff));
    atomicAdd(p.status + 6, (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffff) - t_wg0);
    const int w = atomicAdd(p.status + 12, 1);
    if (w < 2048) {   // per-workgroup records behind the 16 status words (aog_device_status writes them out)
      int* r = p.status + 16 + 4 * w;
      r[0] = n_steps | (p.tile_k[phase * p.tiles64_max + tile64] << 8) | (phase << 12) | ((__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 16) | (((int)blockIdx.x >> 3) << 20);   // (position in its XCD's queue)
      r[1] = cyc;
      r[2] = t_wg0;
      r[3] = (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffff);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the re-loads past the end)
  __syncthreads();   // every wave is done with the ring: its memory serves the epilogue

  // epilogue.  The two waves of a tile swap half of their accumulators (exact int32 sums) and each finishes 8 of the tile's 16 register rows:
  // fixed point -> float64 (sum_l acc_l 128^(5 - l): every term exact, the sum rounded at 2^-53), add the exact piston / tilt response, scatter
  // into the toroidal master (and its fp32 ring copy)
  int* const xch = reinterpret_cast<int*>(lds8);   // [tile][destination half][level][8][64] int32: 96 KB
  auto xat = [&](int dest, int l, int r8) { return xch + ((((tile * 2 + dest) * kX8Levels + l) * 8 + r8) << 6) + lane; };
  if (half == 0) {
#pragma unroll
    for (int l = 0; l < kX8Levels; ++l)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) *xat(1, l, r8) = acc[l][8 + r8];
  } else {
#pragma unroll
    for (int l = 0; l < kX8Levels; ++l)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) *xat(0, l, r8) = acc[l][r8];
  }
  __syncthreads();
  double f8[8];
  if (half == 0) {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      double f = (double)(acc[0][r8] + *xat(0, 0, r8));
#pragma unroll
      for (int l = 1; l < kX8Levels; ++l) f = f * 128.0 + (double)(acc[l][r8] + *xat(0, l, r8));
      f8[r8] = f;
    }
  } else {
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      double f = (double)(acc[0][8 + r8] + *xat(1, 0, r8));
#pragma unroll
      for (int l = 1; l < kX8Levels; ++l) f = f * 128.0 + (double)(acc[l][8 + r8] + *xat(1, l, r8));
      f8[r8] = f;
    }
  }
  const bool vertical = phase == 1;
  __syncthreads();   // (the transposition area overlaps the exchange area)

  const int slot = (2 * tile64 + ctl) * 32 + (lane & 31);
  const int j = shift + 1;   // these rows are the slice shift j creates
  int env = p.list[phase * p.slots_max + slot];
  if (env >= 0 && abs(p.dxy[2 * env + (vertical ? 1 : 0)]) < j) env = -1;   // (an env of fewer shifts than its tile's first: this slice is not its)
  const int row_base = (rt0 + rtl) * 32;
  const double* rc = p.rec + (size_t)slot * 4;
  const double scale = env >= 0 ? rc[0] : 0.0, c0 = env >= 0 ? rc[1] : 0.0, c1 = env >= 0 ? rc[2] : 0.0;
  // the 32 samples of a tile row-block are consecutive for each env — in a new ROW of the screen (phase 1), in the staging run of a new COLUMN
  // (phase 0: k_x8_prepare of phase 1 puts the columns into the screens, see there): they are transposed through LDS (XOR-swizzled 32 x 32
  // float64 image: conflict-free both ways) so that a store instruction writes two envs x 32 consecutive samples instead of 64 samples of 32
  // different envs
  double* mt = reinterpret_cast<double*>(lds8) + (size_t)tile * 32 * 32;
#pragma unroll
  for (int r8 = 0; r8 < 8; ++r8) {
    const int r = 8 * half + r8, rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), row = row_base + rr;
    mt[(lane & 31) * 32 + (rr ^ (lane & 31))] = f8[r8] * scale + c0 * tb.r1[row] + c1 * tb.r2[row];
  }
  if (!(AOG_X8_DEV(p) & 4)) {
    // lane c (and c + 32) holds what the stores of env column c need: phase 1: its screen row and the start and direction of the run along it
    int info = -1;
    if (env >= 0 && vertical) {
      const int oy = __double2hiint(rc[3]), ox = __double2loint(rc[3]);
      const bool flipped = p.dxy[2 * env + 1] > 0;
      int pyv = flipped ? oy + j - 1 : oy - j;
      pyv = ((pyv % N) + N) % N;
      info = pyv | (ox << 12) | ((flipped ? 1 : 0) << 24);   // (N <= 4096)
    }
    __syncthreads();   // both waves of the tile have written their rows of the image
    const int i = row_base + (lane & 31) - shift * Np;
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) {
      const int c = 2 * (8 * half + e8) + (lane >> 5);
      const int en = __shfl(env, c, 64), inf = __shfl(info, c, 64);
      if (en < 0 || i >= N) continue;
      const double v = mt[c * 32 + ((lane & 31) ^ c)];
      if (vertical) {
        int px = ((inf >> 24) ? N - 1 - i : i) + ((inf >> 12) & 4095);
        if (px >= N) px -= N;
        x8_store(p, en, N, inf & 4095, px, v);
      } else {
        p.colbuf[((size_t)((2 * tile64 + ctl) * 32 + c) * p.kcap + shift) * Np + i] = v;
      }
    }
  }
  if ((AOG_X8_DEV(p) & 1024) && wave == 0 && lane == 0) atomicMax(p.status + 5, (int)(__builtin_amdgcn_s_memrealtime() & 0x3fffffff));
}

}  // namespace aog
