// Per-step kernels around the fused pupil pass: prologue (action -> actuators), float64 validation pass, epilogue, small state kernels.
#pragma once
#include "k_common.h"

namespace aog {

// ------------------------------------------------------------------------------------------------
// K1  prologue: action -> actuators (AO_env.py:115-120).  One wave per env, four per workgroup.
//   a'_i = action_i / (i + 10);  var = a'^T G a'  (G = centred Gram, float64);  a'' = a' * target / sqrt(var)
//   act_dm  [B][A] float64 (metres)         — deformable_mirror.actuators
//   act_rev [A_PAD][Bp] float32             — 2 a''/lambda_wfs (revolutions of wfs phase per unit mode)
//   act16   (MFMA B-operand order, hi/lo f16 halves) [env/32][A_PAD/16][hi|lo][64 lanes][8]
// A zero action gives 0/0 = NaN exactly like numpy (documented in DESIGN.md).
// ------------------------------------------------------------------------------------------------
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

// actuators (metres, float64) -> the two fp32 operand layouts (used by reset / set_actuators)
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

// AOEnv.reset bookkeeping (AO_env.py:79-83)
__global__ void k_reset_state(const uint8_t* __restrict__ mask, double* __restrict__ act_dm, int32_t* __restrict__ t_render,
                              int B, int A, int flatten) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int env = idx / A, i = idx % A;
  if (mask && !mask[env]) return;
  if (flatten) act_dm[idx] = 0.0;
  if (i == 0) t_render[env] = 0;
}

// ------------------------------------------------------------------------------------------------
// K3c  float64 validation form (AOG_PRECISION_FP64): one workgroup per env, everything in float64 from
// float64 tables; also the general path for shapes the fast kernels are not instantiated for.
// ------------------------------------------------------------------------------------------------
constexpr int kRefMaxSums = 2 * 80;

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
    {
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

// Zero-fill on the caller's stream as a kernel of the library (the per-step paths zero a few KB .. MB: barrier tickets, lenslet sums,
// focal work buffers): hipMemsetAsync goes through the runtime's blit kernels, ~10 us per call in the profiles against ~3 here.
__global__ void k_zero_words(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

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

// self-test hook: the three sin/cos flavours of the fused kernels on caller-supplied revolutions
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

}  // namespace aog
