// R1: policy query of the rollout as one kernel.
#pragma once
#include "k_common.h"

namespace aog {

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

}  // namespace aog
