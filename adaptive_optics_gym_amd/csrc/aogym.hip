// libaogym.so — C-ABI (include/aogym.h) over the gfx950 kernels in aogym_kernels.h.
// Host side only: argument checking, table conversion/upload, launch geometry, stream-ordered launches.
#define AOG_MAIN_TU 1
#include "aogym_internal.h"
#include "aogym_kernels.h"

#include <hipfft/hipfft.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(AOG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

using aog_host::round_up;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace


#ifdef AOG_DEV
namespace aog_host { long long* dev_timeline = nullptr; }
extern "C" int aog_dev_read_timeline(void* dst, size_t nbytes) {   // developer builds only: not in include/aogym.h
  if (!aog_host::dev_timeline) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpy(dst, aog_host::dev_timeline, nbytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
namespace aog_host {
int ensure_dynamic_lds(const void* fn, size_t bytes, int device) {
  if (bytes <= 64 * 1024) return AOG_OK;   // the default limit
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[{fn, device}];
  if (bytes > have) {
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return AOG_OK;
}
}  // namespace aog_host

namespace {

template <typename T>
int dev_alloc(aog_env* e, T** out, size_t count, bool zero = true) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  if (zero) HIP_TRY(hipMemset(p, 0, bytes));
  e->allocs.push_back(p);
  e->alloc_bytes.push_back(bytes);
  e->dev_bytes += (int64_t)bytes;
  *out = static_cast<T*>(p);
  return AOG_OK;
}

int pick_pad(int v, const int* opts, int n) {
  for (int i = 0; i < n; ++i)
    if (v <= opts[i]) return opts[i];
  return -1;
}

const int kApadOpts[] = {16, 32, 64, 128};
const int kMrwOpts[] = {7, 12, 20, 28};

int launch_fast(aog_env* e, hipStream_t s) {
  switch (e->A_pad) {
    case 16: return aog_host::launch_fused_apad16(e, s);
    case 32: return aog_host::launch_fused_apad32(e, s);
    case 64: return aog_host::launch_fused_apad64(e, s);
    default: return aog_host::launch_fused_apad128(e, s);
  }
}

}  // namespace

namespace {
// zero `n_words` 32-bit words at p on stream s with a kernel of the library (see k_zero_words for why not hipMemsetAsync)
void zero_words(void* p, size_t n_words, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<size_t>((n_words + 255) / 256, 4096);
  if (n_words) hipLaunchKernelGGL(aog::k_zero_words, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(p), n_words);
}
}  // namespace

namespace aog_host {
template <int A_PAD>
static void launch_phase_t(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile) {
  hipLaunchKernelGGL((aog::k_phase_mfma<A_PAD>), dim3((e->n_ptiles + 3) / 4, e->n_etiles), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile),
                     reinterpret_cast<const aog::f16x8*>(act16), reinterpret_cast<aog::f32x4*>(out_tile), e->n_ptiles, e->n_etiles);
}
template <int A_PAD>
static void launch_phase_field_t(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, bool grid) {
  auto kern = grid ? aog::k_phase_mfma<A_PAD, true, true> : aog::k_phase_mfma<A_PAD, true, false>;
  hipLaunchKernelGGL(kern, dim3((e->n_ptiles + 3) / 4, e->n_etiles), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile),
                     reinterpret_cast<const aog::f16x8*>(act16), static_cast<aog::f32x4*>(nullptr), e->n_ptiles, e->n_etiles, fa);
}
// grid = true: one float per pixel (reduced phase) instead of the complex field: see k_phase_mfma<.., GRID>
void launch_phase_field(aog_env* e, hipStream_t s, const _Float16* act16, float* field, size_t env_stride, int row_stride, bool grid) {
  aog::PhaseFieldArgs fa{};
  fa.ap_yx = e->sh_ap_yx;
  fa.mla32 = reinterpret_cast<const float2*>(e->sh_mla32);
  fa.mla_rev = e->sh_ftab;
  fa.field = reinterpret_cast<float2*>(field);
  fa.env_stride = env_stride;
  fa.row_stride = row_stride;
  fa.n_ap = e->n_ap;
  fa.B = e->B;
  fa.N = e->cfg.n_pupil;
  fa.amplitude = (float)e->sh_amp;
  switch (e->A_pad) {
    case 16: launch_phase_field_t<16>(e, s, act16, fa, grid); break;
    case 32: launch_phase_field_t<32>(e, s, act16, fa, grid); break;
    case 64: launch_phase_field_t<64>(e, s, act16, fa, grid); break;
    default: launch_phase_field_t<128>(e, s, act16, fa, grid); break;
  }
}
// K4: reduced phases of env tiles [etile0, etile0 + n_et) as one float per pixel on a dense [env][rows][row_stride] grid (no micro-lens term)
template <int A_PAD>
static void launch_phase_grid_t(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, int etile0, int n_et) {
  hipLaunchKernelGGL((aog::k_phase_mfma<A_PAD, true, true>), dim3((e->n_ptiles + 3) / 4, n_et), dim3(256), 0, s,
                     reinterpret_cast<const aog::f16x8*>(e->modes16), reinterpret_cast<const aog::f32x4*>(e->psi_tile) + (size_t)etile0 * e->n_ptiles * 4 * 64,
                     reinterpret_cast<const aog::f16x8*>(act16) + (size_t)etile0 * (A_PAD / 16) * 2 * 64, static_cast<aog::f32x4*>(nullptr), e->n_ptiles,
                     n_et, fa);
}
void launch_phase_grid(aog_env* e, hipStream_t s, const _Float16* act16, float* grid, size_t env_stride, int row_stride, int etile0, int n_et) {
  aog::PhaseFieldArgs fa{};
  fa.ap_yx = e->focal_ap_yx;
  fa.mla_rev = nullptr;
  fa.act_ll = reinterpret_cast<const aog::f16x8*>(e->focal_act_ll) + (size_t)etile0 * (e->A_pad / 16) * 64;
  fa.field = reinterpret_cast<float2*>(grid);   // grid row 0 = env etile0 * 32
  fa.env_stride = env_stride;
  fa.row_stride = row_stride;
  fa.n_ap = e->n_ap;
  fa.B = e->B - etile0 * 32;
  fa.N = e->cfg.n_pupil;
  switch (e->A_pad) {
    case 16: launch_phase_grid_t<16>(e, s, act16, fa, etile0, n_et); break;
    case 32: launch_phase_grid_t<32>(e, s, act16, fa, etile0, n_et); break;
    case 64: launch_phase_grid_t<64>(e, s, act16, fa, etile0, n_et); break;
    default: launch_phase_grid_t<128>(e, s, act16, fa, etile0, n_et); break;
  }
}
void launch_phase(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile) {
  switch (e->A_pad) {
    case 16: launch_phase_t<16>(e, s, act16, out_tile); break;
    case 32: launch_phase_t<32>(e, s, act16, out_tile); break;
    case 64: launch_phase_t<64>(e, s, act16, out_tile); break;
    default: launch_phase_t<128>(e, s, act16, out_tile); break;
  }
}
}  // namespace aog_host

namespace {

// HIP-event bracket around the launches of one kernel id while profiling is on (aog_profile_read_kernel): the closing record is made by
// the destructor, on the same stream.
struct TimedRegion {
  aog_env* e;
  hipStream_t s;
  hipEvent_t ev1 = nullptr;
  TimedRegion(aog_env* env, hipStream_t stream, int kernel_id, bool on = true) : e(env), s(stream) {
    if (!e->profile || !on) return;
    hipEvent_t ev0 = nullptr;
    if (e->events_used == e->events.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      e->events.emplace_back(a, b);
    }
    ev0 = e->events[e->events_used].first;
    ev1 = e->events[e->events_used].second;
    if (e->event_kernel.size() <= e->events_used) e->event_kernel.resize(e->events_used + 1);
    e->event_kernel[e->events_used] = kernel_id;
    ++e->events_used;
    (void)hipEventRecord(ev0, s);
  }
  ~TimedRegion() {
    if (ev1) (void)hipEventRecord(ev1, s);
  }
  TimedRegion(const TimedRegion&) = delete;
  TimedRegion& operator=(const TimedRegion&) = delete;
};

int launch_fused(aog_env* e, hipStream_t s) {
  // event timing of one launch block in profile_every: the two records cost ~3 us each on the stream, so a throughput measurement that
  // also wants the kernel's duration samples instead of timing every launch
  // (blocks of 8 consecutive launches, one block in profile_every: a timed launch mostly sees the same neighbours as with every launch timed)
  // (the MIDDLE block of each period of profile_every blocks: a short window's first launches come right after a synchronisation and are its slowest)
  const bool timed = e->profile && ((e->profile_phase++ / (unsigned)e->profile_block) % (unsigned)e->profile_every) == (unsigned)e->profile_every / 2;
  TimedRegion tr(e, s, AOG_PROF_FUSED, timed);
  if (e->cfg.precision == AOG_PRECISION_FP64) {
    hipLaunchKernelGGL(aog::k_fused_ref, dim3(e->B), dim3(256), 0, s, e->modes64, e->tabs64, e->psi64, e->act_dm,
                       e->partials, e->n_ap, e->A, e->MRW_used, e->MRS_used, e->Bp, e->cfg.wavelength_wfs,
                       e->cfg.wavelength_sci);
  } else if (int rc = launch_fast(e, s)) {
    return rc;   // (the dynamic-LDS request of this shape was refused: the message names the size)
  }
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

// action_next (aog_step_pipelined): the prologue of the NEXT step rides in the same launch (k_epilogue_prologue)
int launch_epilogue(aog_env* e, bool is_step, float* obs_raw, uint16_t* obs, float* reward, uint8_t* done, float* power,
                    float* strehl, hipStream_t s, const float* action_next = nullptr) {
  aog::EpilogueArgs p{};
  p.partials = e->partials;
  p.wfs_coef = e->wfs_coef;
  p.sci_coef = e->sci_coef;
  p.obs_raw = obs_raw;
  p.obs = obs;
  p.reward = reward;
  p.done = done;
  p.power = power;
  p.strehl = strehl;
  p.t_render = e->t_render;
  p.B = e->B;
  p.Bp = e->Bp;
  const bool ref = e->cfg.precision == AOG_PRECISION_FP64;
  p.n_chunks = ref ? 1 : e->n_chunks;
  if (e->kernel == AOG_KERNEL_MFMA && e->MRW > 8 && e->cfg.precision == AOG_PRECISION_FAST) {
    // many short float chunks: fold them first with a fully coalesced pass, the epilogue then reads one float64 slab
    const int NSr = 2 * (e->MRW + e->MRS);
    hipLaunchKernelGGL(aog::k_reduce_slabs, dim3(e->Bp / 64, (NSr + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float*>(e->partials),
                       e->slab_reduced, e->n_chunks, NSr, e->Bp);
    p.partials = e->slab_reduced;
    p.n_chunks = 1;
    p.partials_f32 = 0;
  }
  p.MRW = ref ? e->MRW_used : e->MRW;
  p.MRS = ref ? e->MRS_used : e->MRS;
  p.MRW_used = e->MRW_used;
  p.MRS_used = e->MRS_used;
  p.n_obs = e->n_obs;
  p.n_fiber = e->cfg.n_fiber_modes;
  p.reward_type = e->cfg.reward_type;
  p.has_thr = e->cfg.has_rew_threshold;
  p.max_steps = e->cfg.max_steps;
  p.is_step = is_step ? 1 : 0;
  p.ret_acc = is_step ? e->ret_acc : nullptr;
  p.partials_f32 = 0;   // (float slabs are folded by k_reduce_slabs above; the epilogue's own float path is kept for reference)
  p.thr = e->cfg.rew_threshold;
  p.ssim_peak = e->cfg.ssim_ref_peak;
  p.ssim_alpha = e->cfg.ssim_alpha;
  const int NS = 2 * (p.MRW + p.MRS);
  const size_t lds = aog::epilogue_lds_bytes(NS, p.n_obs, p.n_fiber, p.MRW_used, p.MRS_used);
  const int n_epi = (e->Bp + aog::kEpiEnvs - 1) / aog::kEpiEnvs;
  if (action_next) {
    aog::PrologueArgs q{};
    const bool mfma_fast = e->kernel == AOG_KERNEL_MFMA && e->cfg.precision == AOG_PRECISION_FAST && !e->sh_ready;
    q.action = action_next;
    q.gram = e->gram;
    q.act_dm = e->act_dm;
    q.act_rev = mfma_fast ? nullptr : e->act_rev;
    q.act16 = e->act16;
    q.B = e->B; q.A = e->A; q.A_pad = e->A_pad; q.Bp = e->Bp;
    q.sh_operation = e->cfg.sh_operation;
    q.target = e->cfg.surface_rms_target;
    q.two_over_lambda = 2.0 / e->cfg.wavelength_wfs;
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_epilogue_prologue), lds, e->device)) return rc;
    hipLaunchKernelGGL(aog::k_epilogue_prologue, dim3(n_epi + (e->B + aog::kEpiProEnvs - 1) / aog::kEpiProEnvs), dim3(1024), lds, s, p, q, n_epi);
  } else {
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_epilogue), lds, e->device)) return rc;
    hipLaunchKernelGGL(aog::k_epilogue, dim3(n_epi), dim3(1024), lds, s, p);
  }
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

// float64 ring-buffer master screens of envs [first, first+count) -> the fused kernels' fp32 layouts
int pack_from_master(aog_env* e, int first, int count, hipStream_t s, bool per_step = false) {
  const double inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  const int N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  if (per_step && e->kernel == AOG_KERNEL_MFMA && e->cfg.precision == AOG_PRECISION_FAST && first == 0 && count == e->B) {
    // fast path: offsets = means measured by the previous repack, whole-row writes
    hipLaunchKernelGGL(aog::k_refresh_offsets, dim3((e->B + 255) / 256), dim3(256), 0, s, e->psi_offset, e->psi_sum, e->B, e->n_ap);
    dim3 grid(((e->n_ptiles + 1) / 2 + aog::kRepackIters - 1) / aog::kRepackIters, e->n_etiles);
    hipLaunchKernelGGL(aog::k_repack_master, grid, dim3(256), 0, s, e->psi_master, e->origin, e->ap_index, e->psi_offset, e->psi_sum,
                       e->psi_tile, e->B, e->cfg.n_pupil, e->n_ap, e->n_ptiles, inv);
    HIP_TRY(hipGetLastError());
    return AOG_OK;
  }
  // the MFMA kernel only reads psi_tile, the VALU kernel only psi_rev: write the one that is used
  float* rev = e->kernel == AOG_KERNEL_VALU ? e->psi_rev : nullptr;
  float* tile = (e->kernel == AOG_KERNEL_MFMA || e->sh_ready) ? e->psi_tile : nullptr;
  hipLaunchKernelGGL((aog::k_pack_screens<double>), dim3(count), dim3(256), 0, s, e->psi_master + (size_t)first * N2, e->ap_index,
                     rev, tile, e->psi64, first, N2, e->n_ap, e->n_ap_pad, e->Bp, inv, (const int32_t*)e->origin,
                     e->cfg.n_pupil, e->psi_offset, e->psi_sum);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

// layer.t = timestep * delta_t (AO_env.py:125): wind extrusion of every env, then refresh the fp32 layouts
constexpr size_t kLdsBytes = 160 * 1024;   // LDS per CU on gfx950
size_t ext16_lds(const aog_env* e) {
  return (size_t)aog::kExt16G * ((std::max(e->nz_v, e->nz_h) | 1) + (e->cfg.n_pupil | 1)) * sizeof(double);
}

size_t ext_split_lds(const aog_env* e) {
  return ((size_t)aog::kExt16G * (aog::ext_split_stride(std::max(e->nz_v, e->nz_h)) + aog::ext_split_stride(e->cfg.n_pupil)) +
          (size_t)(aog::kExtKs - 1) * 4 * 256) * sizeof(double) + (size_t)(e->nz_v + e->nz_h) * sizeof(int32_t);
}

// step_index: the AOEnv.timestep this extrusion brings the layer to (layer.t = step_index * delta_t)
int evolve_layer(aog_env* e, hipStream_t s, long long step_index) {
  if (!e->layer_ready) return fail(AOG_ERR_STATE, "dynamic atmosphere: aog_upload_layer / aog_set_wind not called");
  aog::ExtrudeArgs p{};
  p.master = e->psi_master;
  p.origin = e->origin;
  p.ext_counter = e->ext_counter;
  p.velocity = e->velocity;
  p.stencil_v = e->stencil_v;
  p.stencil_h = e->stencil_h;
  p.stencil_v_yx = e->stencil_v_yx;
  p.stencil_h_yx = e->stencil_h_yx;
  p.At_v = e->At_v;
  p.Bt_v = e->Bt_v;
  p.At_h = e->At_h;
  p.Bt_h = e->Bt_h;
  p.Wa_v = e->Wa_v;
  p.Wb_v = e->Wb_v;
  p.Wa_h = e->Wa_h;
  p.Wb_h = e->Wb_h;
  p.noise = e->next_noise;
  p.max_ext = e->next_noise_max_ext;
  p.N = e->cfg.n_pupil;
  p.nz_v = e->nz_v;
  p.nz_h = e->nz_h;
  p.near_v = e->near_v;
  p.near_h = e->near_h;
  p.t_prev = (double)(step_index - 1) * e->delta_t;
  p.t_new = (double)step_index * e->delta_t;
  p.pitch = e->pitch;
  p.sqrt_cn2 = e->sqrt_cn2;
  p.seed = e->rng_seed;
  p.env_base = e->cfg.env_id_base;
  p.ring = e->ring_direct ? e->psi_ring : nullptr;
  p.ring_ref = e->psi_offset;
  p.ring_inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  // (sampled like the fused kernel's launches — blocks of 8 steps, one block in profile_every: two event records cost ~6 us of a 250 us step)
  TimedRegion tr_ext(e, s, AOG_PROF_EXTRUDE, ((e->profile_phase / (unsigned)e->profile_block) % (unsigned)e->profile_every) == (unsigned)e->profile_every / 2);
  if (e->ext_bar && !getenv("AOG_EXTRUDE_SIMPLE") && !getenv("AOG_EXTRUDE_NOSPLIT") && ext_split_lds(e) <= kLdsBytes) {
    // float64 matrix-core form with each 16-env group's rows split over four workgroups + group barrier
    const size_t lds = ext_split_lds(e);
    auto kern = aog::k_extrude16_split<aog::kExtKs>;
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, e->device)) return rc;
    if (!e->ext_resident) {
      // The four workgroups of a group meet at a spin barrier: they must be resident together.  Ask once per handle how many of these
      // workgroups a CU holds (registers + this shape's LDS), keep one CU's worth of margin (the query over-reports by one block per CU for
      // some kernels: MI355X_MICROARCH.md, Residency), and never put more workgroups than that into one launch.
      int per_cu = 0, cus = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256 * aog::kExtKs, lds));
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
      if (per_cu < 1 || cus < 8) return fail(AOG_ERR_HIP, "k_extrude16_split does not fit a compute unit (occupancy query: %d)", per_cu);
      e->ext_resident = std::max(1, per_cu > 1 ? per_cu - 1 : 1) * cus;
      if (const char* v = getenv("AOG_EXTRUDE_RESIDENT")) e->ext_resident = std::max(8 * aog::kExtParts, atoi(v));   // (tests: force several launches)
    }
    p.origin = e->origin;
    const int groups8 = round_up(e->n_ext_groups, 8);
    // two ticket sets alternate between steps: this step's launches poll `bar` and zero `bar_next` (both start zeroed at creation)
    unsigned* bar = e->ext_bar + (size_t)(e->ext_bar_phase & 1) * groups8;
    unsigned* bar_next = e->ext_bar + (size_t)((e->ext_bar_phase ^ 1) & 1) * groups8;
    e->ext_bar_phase ^= 1;
    const int groups_per_launch = std::max(8, e->ext_resident / aog::kExtParts / 8 * 8);
    for (int g0 = 0; g0 < groups8; g0 += groups_per_launch) {
      const int ng = std::min(groups_per_launch, groups8 - g0);
      hipLaunchKernelGGL(kern, dim3(ng * aog::kExtParts), dim3(256 * aog::kExtKs), lds, s, p, e->B, e->ext_perm, bar, e->dev_status,
                         e->host_flag_dev, g0, e->ext_spin_limit, e->ext_absent_part, bar_next, getenv("AOG_EXTRUDE_AGENT_SCOPE") ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
  } else if (!getenv("AOG_EXTRUDE_SIMPLE") && ext16_lds(e) <= kLdsBytes) {
    // default: float64 matrix-core form, 16 envs per workgroup (a workgroup owns whole envs: no cross-workgroup hazard)
    const size_t lds = (size_t)aog::kExt16G * ((std::max(e->nz_v, e->nz_h) | 1) + (e->cfg.n_pupil | 1)) * sizeof(double);
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_extrude16), lds, e->device)) return rc;
    p.origin = e->origin;
    hipLaunchKernelGGL(aog::k_extrude16, dim3((e->B + aog::kExt16G - 1) / aog::kExt16G), dim3(512), lds, s, p, e->B);
    HIP_TRY(hipGetLastError());
  } else {
    const size_t lds = (size_t)aog::kExtG * (std::max(e->nz_v, e->nz_h) + 2 * e->cfg.n_pupil) * sizeof(double);
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_extrude), lds, e->device)) return rc;
    p.origin = e->origin;
    hipLaunchKernelGGL(aog::k_extrude, dim3((e->B + aog::kExtG - 1) / aog::kExtG), dim3(aog::kExtThreads), lds, s, p, e->B);
    HIP_TRY(hipGetLastError());
  }
  e->next_noise = nullptr;
  e->next_noise_max_ext = 0;
  if (e->ring_direct) {   // the extrusion kept the fp32 ring copy in step: nothing to repack
    e->tiles_stale = true;
    return AOG_OK;
  }
  return pack_from_master(e, 0, e->B, s, true);
}

// psi_tile of a ring-direct handle is only refreshed when something other than the step kernel needs it
int ensure_tiles(aog_env* e, hipStream_t s) {
  if (!e->ring_direct || !e->tiles_stale) return AOG_OK;
  const double inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  const int N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  hipLaunchKernelGGL((aog::k_pack_screens<double>), dim3(e->B), dim3(256), 0, s, e->psi_master, e->ap_index, (float*)nullptr, e->psi_tile,
                     (double*)nullptr, 0, N2, e->n_ap, e->n_ap_pad, e->Bp, inv, (const int32_t*)e->origin, e->cfg.n_pupil, (double*)nullptr,
                     (double*)nullptr);
  HIP_TRY(hipGetLastError());
  e->tiles_stale = false;
  return AOG_OK;
}

int ring_from_master(aog_env* e, int first, int count, int keep_ref, hipStream_t s) {
  hipLaunchKernelGGL(aog::k_ring_from_master, dim3(count), dim3(256), 0, s, e->psi_master, e->origin, e->ap_index, e->psi_offset, e->psi_ring, first,
                     e->cfg.n_pupil, e->n_ap, 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs), keep_ref);
  HIP_TRY(hipGetLastError());
  e->tiles_stale = true;
  return AOG_OK;
}

// A bounded inter-workgroup wait of an earlier launch timed out (k_extrude16_split): every screen that launch touched is suspect.
// The flag lives in pinned host memory, so this costs one load and no synchronisation; it is seen at the latest by the call after
// the one whose launch tripped it.  Installing fresh screens for the whole batch (aog_set_screens / aog_set_state) clears it.
int check_poisoned(const aog_env* e, const char* who) {
  if (e->host_flag && *static_cast<volatile const int*>(e->host_flag) != 0)
    return fail(AOG_ERR_STATE, "%s: an inter-workgroup wait of the dynamic-atmosphere kernel timed out in an earlier step; the screens of "
                "this handle are invalid (install new screens or restore a saved state)", who);
  return AOG_OK;
}

// give a work buffer of the handle back (workspaces that are re-sized when the caller changes the synthesis method or oversampling:
// without this every change would keep the old gigabytes until aog_destroy)
// With lookahead on, between aog_step(t) and aog_step(t + 1) the screens already stand at step t + 1: anything that reads or replaces
// them then would see (or break) a state the env is not in.  Episode boundaries are safe: the last step of an episode does not look ahead.
int refuse_pre_evolved(const aog_env* e, const char* who) {
  if (e->pro_pending)
    return fail(AOG_ERR_STATE, "%s: a pipelined step has already loaded the NEXT action into the mirror (aog_step_pipelined with action_next): finish "
                "the sequence with action_next = NULL (or reset the whole batch) first", who);
  if (e->pre_evolved)
    return fail(AOG_ERR_STATE, "%s: the atmosphere of this handle has been advanced to the next step already (aog_set_lookahead): call it at an "
                "episode boundary (after a step that returned done), or switch lookahead off and take one more step first", who);
  return AOG_OK;
}

template <typename T>
void dev_release(aog_env* e, T** ptr) {
  if (!*ptr) return;
  for (size_t i = 0; i < e->allocs.size(); ++i)
    if (e->allocs[i] == static_cast<void*>(*ptr)) {
      e->dev_bytes -= (int64_t)e->alloc_bytes[i];   // (aog_info.device_bytes stays what the handle owns)
      e->allocs.erase(e->allocs.begin() + (long)i);
      e->alloc_bytes.erase(e->alloc_bytes.begin() + (long)i);
      break;
    }
  (void)hipFree(*ptr);
  *ptr = nullptr;
}

template <typename T>
int set_screens(aog_env* e, const T* psi, int first, int count, hipStream_t s) {
  if (!e || !psi) return fail(AOG_ERR_INVALID, "aog_set_screens: null argument");
  if (!e->tables_ready) return fail(AOG_ERR_STATE, "aog_set_screens before aog_upload_tables");
  if (first < 0 || count < 0 || first + count > e->B)
    return fail(AOG_ERR_INVALID, "aog_set_screens: env range [%d,%d) outside [0,%d)", first, first + count, e->B);
  if (int rc = refuse_pre_evolved(e, "aog_set_screens")) return rc;
  if (count == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(e->device));
  const double inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  const int N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  if (e->cfg.atm_dynamic) {
    const size_t n = (size_t)count * N2;
    hipLaunchKernelGGL((aog::k_store_master<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, psi, e->psi_master, e->origin,
                       e->ext_counter, first, count, N2);
    HIP_TRY(hipGetLastError());
    int rc = e->ring_direct ? ring_from_master(e, first, count, 0, s) : pack_from_master(e, first, count, s);
    if (rc != AOG_OK) return rc;
  } else if (e->cfg.precision == AOG_PRECISION_FAST && count >= 8) {
    // batches: aperture means, then tiled conversion with whole-line stores (k_pack_tiles)
    if (!e->pack_mean) {
      int rc = dev_alloc(e, &e->pack_mean, (size_t)e->B, false);
      if (rc != AOG_OK) return rc;
    }
    TimedRegion tr(e, s, AOG_PROF_PACK);
    hipLaunchKernelGGL((aog::k_screen_means<T>), dim3(count), dim3(256), 0, s, psi, e->ap_index, e->pack_mean, N2, e->n_ap);
    const int et0 = first >> 5, et1 = (first + count - 1) >> 5;
    hipLaunchKernelGGL((aog::k_pack_tiles<T>), dim3((e->n_ptiles + aog::kPackTiles - 1) / aog::kPackTiles, et1 - et0 + 1), dim3(256), 0, s, psi,
                       e->ap_index, e->pack_mean, e->psi_rev, e->psi_tile, first, count, N2, e->n_ap, e->n_ptiles, e->Bp, inv);
    HIP_TRY(hipGetLastError());
  } else {
    TimedRegion tr(e, s, AOG_PROF_PACK);
    hipLaunchKernelGGL((aog::k_pack_screens<T>), dim3(count), dim3(256), 0, s, psi, e->ap_index, e->psi_rev, e->psi_tile,
                       e->psi64, first, N2, e->n_ap, e->n_ap_pad, e->Bp, inv, (const int32_t*)nullptr, e->cfg.n_pupil);
    HIP_TRY(hipGetLastError());
  }
  e->screens_ready = true;
  e->sh_sums_ready = false;   // lenslet sums of an earlier aog_sh_image(NULL) belong to the old screens
  return AOG_OK;
}

// New screens for the WHOLE batch make a handle whose extrusion kernel once timed out usable again (see check_poisoned).  Called by the
// public entry points with the range of the whole call (aog_generate_screens installs large batches in several chunks).  Dynamic handles
// drain the stream first: a timeout of a launch that is still running would otherwise poison the screens just installed.
int clear_poison_if_whole(aog_env* e, int first, int count, hipStream_t s) {
  if (first != 0 || count != e->B || !e->host_flag) return AOG_OK;
  if (e->cfg.atm_dynamic) HIP_TRY(hipStreamSynchronize(s));
  if (*static_cast<volatile int*>(e->host_flag)) {
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemset(e->dev_status, 0, sizeof(int)));
    *static_cast<volatile int*>(e->host_flag) = 0;
  }
  return AOG_OK;
}

}  // namespace

extern "C" {

int aog_abi_version(void) { return AOG_ABI_VERSION; }

const char* aog_last_error(void) { return g_last_error.c_str(); }

int64_t aog_struct_size(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(aog_config);
    case 1: return (int64_t)sizeof(aog_tables);
    case 2: return (int64_t)sizeof(aog_layer_tables);
    case 3: return (int64_t)sizeof(aog_sh_tables);
    case 4: return (int64_t)sizeof(aog_actor);
    case 5: return (int64_t)sizeof(aog_info);
    default: return -1;
  }
}

int aog_create(const aog_config* cfg, int device, aog_env** out) {
  if (!cfg || !out) return fail(AOG_ERR_INVALID, "aog_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != AOG_ABI_VERSION)
    return fail(AOG_ERR_INVALID, "aog_create: abi_version %d != %d", cfg->abi_version, AOG_ABI_VERSION);
  if (cfg->num_envs < 1 || cfg->n_pupil < 2 || cfg->n_modes < 1 || cfg->obs_dim < 1 || cfg->n_ap < 1 ||
      cfg->n_ap > cfg->n_pupil * cfg->n_pupil)
    return fail(AOG_ERR_INVALID, "aog_create: bad sizes (B=%d N=%d A=%d o=%d n_ap=%d)", cfg->num_envs, cfg->n_pupil,
                cfg->n_modes, cfg->obs_dim, cfg->n_ap);
  if (cfg->n_wfs_tables < 1 || cfg->n_sci_tables < 1 || cfg->n_fiber_modes < 0)
    return fail(AOG_ERR_INVALID, "aog_create: bad table counts");
  if (cfg->reward_type != AOG_REWARD_STREHL && cfg->reward_type != AOG_REWARD_SMF_SSIM)
    return fail(AOG_ERR_INVALID, "aog_create: reward_type must be 'strehl_ratio' or 'smf_ssim' (AO_env.py:476,487)");
  if (cfg->obs_dim * cfg->obs_dim > 64) return fail(AOG_ERR_UNSUPPORTED, "aog_create: obs_dim > 8 not built");
  if (cfg->n_modes > 256) return fail(AOG_ERR_UNSUPPORTED, "aog_create: act_dim > 256 not built");
  if (cfg->n_wfs_tables + cfg->n_sci_tables > 80) return fail(AOG_ERR_UNSUPPORTED, "aog_create: > 80 tables");
  if (cfg->env_id_base < 0) return fail(AOG_ERR_INVALID, "aog_create: env_id_base must be >= 0");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(AOG_ERR_HIP, "aog_create: device %d not present (%d HIP devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  aog_env* e = new aog_env();
  e->cfg = *cfg;
  e->device = device;
  e->B = cfg->num_envs;
  e->Bp = round_up(e->B, 64);
  e->A = cfg->n_modes;
  e->n_ap = cfg->n_ap;
  e->n_ap_pad = round_up(e->n_ap, 32);
  e->n_quads = e->n_ap_pad / 4;
  e->n_ptiles = e->n_ap_pad / 32;
  e->n_etiles = e->Bp / 32;
  e->MRW_used = cfg->n_wfs_tables;
  e->MRS_used = cfg->n_sci_tables;
  e->n_obs = cfg->obs_dim * cfg->obs_dim;
  e->n_out = e->n_obs + cfg->n_fiber_modes;
  // sin/cos flavour of the fast kernels: "hwraw" (default; v_sin_f32/v_cos_f32 on the revolutions, the instruction
  // reduces them itself), "hw" (same instructions after an explicit exact reduction), "poly" (degree-7/8 polynomial)
  e->sincos_hw = 2;
  if (const char* sc = getenv("AOG_SINCOS")) e->sincos_hw = strcmp(sc, "poly") == 0 ? 0 : (strcmp(sc, "hwraw") == 0 ? 2 : 1);

  if (cfg->precision == AOG_PRECISION_FAST) {
    e->A_pad = pick_pad(e->A, kApadOpts, 4);
    e->MRW = pick_pad(e->MRW_used, kMrwOpts, 4);
    e->MRS = 1;
    if (e->A_pad < 0 || e->MRW < 0 || e->MRS_used != 1) {
      delete e;
      return fail(AOG_ERR_UNSUPPORTED,
                  "aog_create: fast kernels are built for act_dim <= 128, <= 28 wfs tables and 1 science table; "
                  "use AOG_PRECISION_FP64 for this shape");
    }
    e->kernel = cfg->kernel == AOG_KERNEL_AUTO ? AOG_KERNEL_MFMA : cfg->kernel;
    // launch geometry: aim at ~3 (VALU) / ~2 (MFMA) waves per SIMD over 256 CUs
    const int n_groups = e->Bp / 64;
    int P = cfg->pixel_chunks > 0 ? cfg->pixel_chunks : std::max(1, (256 * 4 * 3 + n_groups - 1) / n_groups);
    int qpc = round_up((e->n_quads + P - 1) / P, 8);
    e->valu_qpc = qpc;
    e->valu_chunks = (e->n_quads + qpc - 1) / qpc;
    e->mfma_we = e->n_etiles >= 4 ? 4 : (e->n_etiles >= 2 ? 2 : 1);
    // Asymmetric wave pairs (see k_fused_tab): the float64-flush variant with at least 4 env tiles runs 8-wave workgroups, one per
    // CU, whose two pixel sub-chunks split a chunk about 2 : 1 with the priority on the larger share; the many-table variants keep the
    // 4-wave interleaved form (their chunks are short and come in many rounds, which balances itself).
    const bool asym = e->kernel == AOG_KERNEL_MFMA && e->MRW <= 8 && e->n_etiles >= 4;
    e->mfma_waves = asym ? 8 : 4;
    e->mfma_heavy = asym ? 672 : 0;
    const int wp = e->mfma_waves / e->mfma_we;
    const int wg_y = (e->n_etiles + e->mfma_we - 1) / e->mfma_we;
    // P pixel chunks (proportional split of the tiles), 8 waves per CU when the batch allows
    int Pm = cfg->pixel_chunks > 0 ? cfg->pixel_chunks : std::max(1, (asym ? 256 : 256 * 2) / wg_y);
    const int max_tpc = e->MRW > 8 ? aog::kTabF32Tiles * wp : 4096;   // fp32-only sums (many-table variants): bounded chunks
    Pm = std::max(Pm, (e->n_ptiles + max_tpc - 1) / max_tpc);
    Pm = std::min(Pm, e->n_ptiles);
    e->mfma_chunks_x = Pm;
    e->mfma_tpc = (e->n_ptiles + Pm - 1) / Pm;  // max tiles of any chunk: ceil(n/P)
    e->n_chunks = e->kernel == AOG_KERNEL_MFMA ? e->mfma_chunks_x * wp : e->valu_chunks;
  } else {
    e->A_pad = round_up(e->A, 8);
    e->MRW = e->MRW_used;
    e->MRS = e->MRS_used;
    e->kernel = 0;
    e->n_chunks = 1;
  }

  int rc = AOG_OK;
  const size_t NS = 2 * (size_t)(e->MRW + e->MRS);
  e->partial_elems = (size_t)e->n_chunks * NS * e->Bp;
#define TRY_ALLOC(x) if ((rc = (x)) != AOG_OK) { aog_destroy(e); return rc; }
  TRY_ALLOC(dev_alloc(e, &e->ap_index, e->n_ap));
  TRY_ALLOC(dev_alloc(e, &e->gram, (size_t)e->A * e->A));
  TRY_ALLOC(dev_alloc(e, &e->wfs_coef, (size_t)e->n_out * e->MRW_used * 2));
  TRY_ALLOC(dev_alloc(e, &e->sci_coef, (size_t)e->MRS_used * 2));
  TRY_ALLOC(dev_alloc(e, &e->act_dm, (size_t)e->B * e->A));
  TRY_ALLOC(dev_alloc(e, &e->act_rev, (size_t)e->A_pad * e->Bp));
  TRY_ALLOC(dev_alloc(e, &e->act16, (size_t)e->n_etiles * e->A_pad * 32 * 2));
  TRY_ALLOC(dev_alloc(e, &e->t_render, e->B));
  TRY_ALLOC(dev_alloc(e, &e->screen_gen, e->B));
  TRY_ALLOC(dev_alloc(e, &e->dev_status, 16));
  {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
      if (hp) (void)hipHostFree(hp);
      aog_destroy(e);
      return fail(AOG_ERR_HIP, "aog_create: pinned status word allocation failed");
    }
    memset(hp, 0, 64);
    e->host_flag = static_cast<int*>(hp);
    e->host_flag_dev = static_cast<int*>(dp);
  }
  TRY_ALLOC(dev_alloc(e, &e->partials, e->partial_elems));
  TRY_ALLOC(dev_alloc(e, &e->slab_reduced, (size_t)2 * 64 * e->Bp));   // [NS <= 58][Bp] float64 (k_reduce_slabs)
  if (cfg->atm_dynamic) {
    const size_t N2 = (size_t)cfg->n_pupil * cfg->n_pupil;
    TRY_ALLOC(dev_alloc(e, &e->psi_master, (size_t)e->B * N2));
    TRY_ALLOC(dev_alloc(e, &e->origin, (size_t)e->B * 2));
    e->n_ext_groups = (e->B + aog::kExt16G - 1) / aog::kExt16G;
    TRY_ALLOC(dev_alloc(e, &e->ext_bar, (size_t)2 * round_up(e->n_ext_groups, 8)));   // two ticket sets (see evolve_layer)
    TRY_ALLOC(dev_alloc(e, &e->ext_perm, (size_t)e->n_ext_groups * aog::kExt16G));
    {
      std::vector<int32_t> ident((size_t)e->n_ext_groups * aog::kExt16G, -1);
      for (int i = 0; i < e->B; ++i) ident[i] = i;
      if (hipMemcpy(e->ext_perm, ident.data(), sizeof(int32_t) * ident.size(), hipMemcpyHostToDevice) != hipSuccess) {
        aog_destroy(e);
        return fail(AOG_ERR_HIP, "aog_create: hipMemcpy failed");
      }
    }
    TRY_ALLOC(dev_alloc(e, &e->ext_counter, (size_t)e->B));
    TRY_ALLOC(dev_alloc(e, &e->velocity, (size_t)e->B * 2));
    TRY_ALLOC(dev_alloc(e, &e->psi_offset, (size_t)e->B));
    TRY_ALLOC(dev_alloc(e, &e->psi_sum, (size_t)e->B));
  }
  if (cfg->precision == AOG_PRECISION_FAST) {
    const int TROW = round_up(e->MRW + e->MRS, 4);
    TRY_ALLOC(dev_alloc(e, &e->modes_f32, (size_t)e->n_ap_pad * e->A_pad));
    TRY_ALLOC(dev_alloc(e, &e->modes16, (size_t)e->n_ap_pad * e->A_pad * 2));
    TRY_ALLOC(dev_alloc(e, &e->tabs_f32, (size_t)e->n_ap_pad * TROW));
    TRY_ALLOC(dev_alloc(e, &e->tab16, (size_t)e->n_ptiles * 2 * 2 * 64 * 8));
    TRY_ALLOC(dev_alloc(e, &e->sci_tile, (size_t)e->n_ptiles * 32));
    if (e->kernel == AOG_KERNEL_VALU) TRY_ALLOC(dev_alloc(e, &e->psi_rev, (size_t)e->n_quads * e->Bp * 4));   // only the VALU kernel reads this layout
    TRY_ALLOC(dev_alloc(e, &e->psi_tile, (size_t)e->n_etiles * e->n_ptiles * 1024));
  } else {
    TRY_ALLOC(dev_alloc(e, &e->modes64, (size_t)e->n_ap * e->A));
    TRY_ALLOC(dev_alloc(e, &e->tabs64, (size_t)e->n_ap * (e->MRW_used + e->MRS_used)));
    TRY_ALLOC(dev_alloc(e, &e->psi64, (size_t)e->B * e->n_ap));
  }
#undef TRY_ALLOC
  *out = e;
  return AOG_OK;
}

void aog_destroy(aog_env* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->host_flag) (void)hipHostFree(e->host_flag);
  if (e->ext_stream) {
    (void)hipStreamSynchronize(e->ext_stream);
    (void)hipStreamDestroy(e->ext_stream);
    (void)hipEventDestroy(e->ev_fused_done);
    (void)hipEventDestroy(e->ev_ext_done);
  }
  if (e->fft_plan) hipfftDestroy((hipfftHandle)(uintptr_t)e->fft_plan);
  if (e->sh_plan) hipfftDestroy((hipfftHandle)(uintptr_t)e->sh_plan);
  for (auto& ev : e->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  delete e;
}

int aog_get_info(const aog_env* e, aog_info* out) {
  if (!e || !out) return fail(AOG_ERR_INVALID, "aog_get_info: null argument");
  memset(out, 0, sizeof *out);
  out->abi_version = AOG_ABI_VERSION;
  out->num_envs = e->B;
  out->num_envs_padded = e->Bp;
  out->n_ap = e->n_ap;
  out->n_ap_padded = e->n_ap_pad;
  out->n_modes_padded = e->A_pad;
  out->pixel_chunks = e->n_chunks;
  out->kernel = e->kernel;
  out->n_sums = 2 * (e->MRW + e->MRS);
  out->reserved = e->ring_direct ? 1 : 0;
  out->device_bytes = e->dev_bytes;
  return AOG_OK;
}

int aog_upload_tables(aog_env* e, const aog_tables* t) {
  if (!e || !t) return fail(AOG_ERR_INVALID, "aog_upload_tables: null argument");
  if (!t->ap_index || !t->modes || !t->gram || !t->wfs_tables || !t->sci_tables || !t->wfs_coef || !t->sci_coef)
    return fail(AOG_ERR_INVALID, "aog_upload_tables: null table pointer");
  HIP_TRY(hipSetDevice(e->device));
  const int n_ap = e->n_ap, A = e->A, N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  for (int p = 0; p < n_ap; ++p) {
    if (t->ap_index[p] < 0 || t->ap_index[p] >= N2) return fail(AOG_ERR_INVALID, "aog_upload_tables: ap_index[%d] out of range", p);
    if (p && t->ap_index[p] <= t->ap_index[p - 1]) return fail(AOG_ERR_INVALID, "aog_upload_tables: ap_index must be strictly increasing");
  }
  HIP_TRY(hipMemcpy(e->ap_index, t->ap_index, sizeof(int32_t) * n_ap, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->gram, t->gram, sizeof(double) * A * A, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->wfs_coef, t->wfs_coef, sizeof(double) * e->n_out * e->MRW_used * 2, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->sci_coef, t->sci_coef, sizeof(double) * e->MRS_used * 2, hipMemcpyHostToDevice));
  if (e->cfg.precision == AOG_PRECISION_FP64) {
    HIP_TRY(hipMemcpy(e->modes64, t->modes, sizeof(double) * n_ap * A, hipMemcpyHostToDevice));
    const int MR = e->MRW_used + e->MRS_used;
    std::vector<double> tb((size_t)n_ap * MR);
    for (int p = 0; p < n_ap; ++p) {
      for (int m = 0; m < e->MRW_used; ++m) tb[(size_t)p * MR + m] = t->wfs_tables[(size_t)m * n_ap + p];
      for (int m = 0; m < e->MRS_used; ++m) tb[(size_t)p * MR + e->MRW_used + m] = t->sci_tables[(size_t)m * n_ap + p];
    }
    HIP_TRY(hipMemcpy(e->tabs64, tb.data(), sizeof(double) * tb.size(), hipMemcpyHostToDevice));
  } else {
    const int Ap = e->A_pad, MR = e->MRW + e->MRS, TROW = round_up(MR, 4);
    std::vector<float> mf((size_t)e->n_ap_pad * Ap, 0.f);
    std::vector<_Float16> m16((size_t)e->n_ap_pad * Ap * 2, (_Float16)0.f);
    const int nstep = Ap / 16;
    for (int p = 0; p < n_ap; ++p)
      for (int k = 0; k < A; ++k) {
        const float v = (float)t->modes[(size_t)p * A + k];
        mf[(size_t)p * Ap + k] = v;
        // modes16[pt][s][hi|lo][lane = 32*h + i][el], mode k = 16 s + 8 h + el
        _Float16 hi, lo;
        aog::split_f16(v * aog::kModeScale, hi, lo);
        const int pt = p >> 5, i = p & 31, sidx = k >> 4, h = (k >> 3) & 1, el = k & 7;
        const size_t base = (((size_t)pt * nstep + sidx) * 2) * 64 + (h * 32 + i);
        m16[base * 8 + el] = hi;
        m16[(base + 64) * 8 + el] = lo;
      }
    std::vector<float> tf((size_t)e->n_ap_pad * TROW, 0.f);
    auto tab = [&](int m, int p) -> float {
      if (m < e->MRW_used) return (float)t->wfs_tables[(size_t)m * n_ap + p];
      if (m >= e->MRW && m - e->MRW < e->MRS_used) return (float)t->sci_tables[(size_t)(m - e->MRW) * n_ap + p];
      return 0.f;
    };
    for (int p = 0; p < n_ap; ++p)
      for (int m = 0; m < MR; ++m) {
        tf[(size_t)p * TROW + m] = tab(m, p);
      }
    HIP_TRY(hipMemcpy(e->modes_f32, mf.data(), sizeof(float) * mf.size(), hipMemcpyHostToDevice));
    {
      // table-MFMA form: A operand of step s, lane (kg, m), element el <-> pixel i = (el & 3) + 16 s + 8 (el >> 2) + 4 kg of the tile
      std::vector<_Float16> t16((size_t)e->n_ptiles * 2 * 2 * 64 * 8, (_Float16)0.f);
      std::vector<float> st((size_t)e->n_ptiles * 32, 0.f);
      for (int pt = 0; pt < e->n_ptiles; ++pt)
        for (int sidx = 0; sidx < 2; ++sidx)
          for (int kg = 0; kg < 2; ++kg)
            for (int el = 0; el < 8; ++el) {
              const int i = (el & 3) + 16 * sidx + 8 * (el >> 2) + 4 * kg;
              const int p = pt * 32 + i;
              if (p >= n_ap) continue;
              for (int m = 0; m < e->MRW_used && m < 32; ++m) {
                const float v = (float)t->wfs_tables[(size_t)m * n_ap + p];
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const size_t base = ((((size_t)pt * 2 + sidx) * 2) * 64 + (kg * 32 + m)) * 8 + el;
                t16[base] = hi;
                t16[base + (size_t)64 * 8] = lo;
              }
              // science table: register a = 8 s + el of half-wave h = kg
              if (e->MRS_used > 0) st[((size_t)pt * 2 + kg) * 16 + 8 * sidx + el] = (float)t->sci_tables[p];
            }
      HIP_TRY(hipMemcpy(e->tab16, t16.data(), sizeof(_Float16) * t16.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(e->sci_tile, st.data(), sizeof(float) * st.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(e->modes16, m16.data(), sizeof(_Float16) * m16.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->tabs_f32, tf.data(), sizeof(float) * tf.size(), hipMemcpyHostToDevice));
  }
  if (e->cfg.atm_dynamic && e->cfg.precision == AOG_PRECISION_FAST && e->kernel == AOG_KERNEL_MFMA && !e->psi_ring && !getenv("AOG_DYNAMIC_REPACK")) {
    // ring-direct form: where every packed 4-pixel group of every tile starts on the pupil grid, and where it continues when the
    // aperture's row ends inside it.  A group that would touch three rows (pupils of a dozen pixels) keeps the repack form.
    const int N = e->cfg.n_pupil;
    std::vector<uint32_t> desc((size_t)e->n_ptiles * 8, 4u), cont((size_t)e->n_ptiles * 8, 0u);
    bool ok = N >= 8 && N < 16384;
    for (int pt = 0; pt < e->n_ptiles && ok; ++pt)
      for (int g = 0; g < 4; ++g) {
        bool straddle = false;
        for (int hh = 0; hh < 2; ++hh) {
          const int p0 = 32 * pt + 8 * g + 4 * hh;
          const size_t slot = ((size_t)pt * 2 + hh) * 4 + g;
          if (p0 >= n_ap) continue;   // padding group: reads logical (0, 0), its table rows are zero
          const int f0 = t->ap_index[p0], iy = f0 / N, ix = f0 % N;
          int k = 1;
          while (k < 4 && p0 + k < n_ap && t->ap_index[p0 + k] == f0 + k && ix + k < N) ++k;
          if (k < 4 && p0 + k >= n_ap) k = 4;   // the batch of pixels ends here: the rest of the group is padding
          desc[slot] = ((uint32_t)iy << 18) | ((uint32_t)ix << 4) | (uint32_t)k;
          if (k < 4) {
            const int f2 = t->ap_index[p0 + k], iy2 = f2 / N, ix2 = f2 % N;
            for (int q = k + 1; q < 4 && p0 + q < n_ap; ++q) ok = ok && t->ap_index[p0 + q] == f2 + (q - k) && ix2 + (q - k) < N;
            cont[slot] = ((uint32_t)iy2 << 18) | ((uint32_t)((ix2 - k + N) % N) << 4);
            straddle = true;
          }
        }
        if (straddle)
          for (int hh = 0; hh < 2; ++hh) desc[((size_t)pt * 2 + hh) * 4 + g] |= 8u;
      }
    if (ok) {
      int rc;
      if ((rc = dev_alloc(e, &e->quad_desc, desc.size(), false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->quad_cont, cont.size(), false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->psi_ring, (size_t)e->B * N * (N + 4), true)) != AOG_OK) return rc;
      HIP_TRY(hipMemcpy(e->quad_desc, desc.data(), sizeof(uint32_t) * desc.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(e->quad_cont, cont.data(), sizeof(uint32_t) * cont.size(), hipMemcpyHostToDevice));
      e->ring_direct = true;
    }
  }
  if (t->focal_m1 && t->focal_m2 && t->n_focal > 0 && !e->focal_m1) {
    const int N = e->cfg.n_pupil, nf = t->n_focal;
    int rc;
    if ((rc = dev_alloc(e, &e->focal_m1, (size_t)nf * N * 2, false)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->focal_m2, (size_t)nf * N * 2, false)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->focal_E, (size_t)N * N * 2)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->focal_T, (size_t)nf * N * 2)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(e->focal_m1, t->focal_m1, sizeof(double) * nf * N * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->focal_m2, t->focal_m2, sizeof(double) * nf * N * 2, hipMemcpyHostToDevice));
    if (e->cfg.precision == AOG_PRECISION_FAST) {   // split-f16 operand tables of the batched matrix-core path (k_focal_pass1 / k_focal_pass2)
      const int Nxp = round_up(N, 128), Nyp = round_up(N, 16), nfp = round_up(nf, 128);
      // power-of-two scales: the largest component of a table lands in [1/2, 1)
      auto scale_of = [](const double* v, size_t n) {
        double mx = 0.0;
        for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(v[i]));
        return mx > 0.0 ? std::ldexp(1.0, -(std::ilogb(mx) + 1)) : 1.0;
      };
      const double s1 = scale_of(t->focal_m1, (size_t)nf * N * 2), s2 = scale_of(t->focal_m2, (size_t)nf * N * 2);
      auto put = [](std::vector<_Float16>& tab, size_t tile, int lane, int slot, double re, double im) {
        const double c[2] = {re, im};
        for (int q = 0; q < 2; ++q) {
          const _Float16 hi = (_Float16)(float)c[q];   // round to nearest, like the kernels' split8
          tab[((tile * 4 + 2 * q) * 64 + lane) * 8 + slot] = hi;
          tab[((tile * 4 + 2 * q + 1) * 64 + lane) * 8 + slot] = (_Float16)(float)(c[q] - (double)(float)hi);
        }
      };
      // m1s [v block][k-step over y]: lane l = column v = 32 vb + (l & 31), slot j = y = 16 ks + 8 (l >> 5) + j
      std::vector<_Float16> m1s((size_t)(nfp / 32) * (Nyp / 16) * 4 * 64 * 8, (_Float16)0.f);
      for (int vb = 0; vb < nfp / 32; ++vb)
        for (int ks = 0; ks < Nyp / 16; ++ks)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int v = 32 * vb + (l & 31), y = 16 * ks + 8 * (l >> 5) + j;
              if (v < nf && y < N)
                put(m1s, (size_t)vb * (Nyp / 16) + ks, l, j, t->focal_m1[((size_t)v * N + y) * 2] * s1, t->focal_m1[((size_t)v * N + y) * 2 + 1] * s1);
            }
      // m2s [u block][x tile][s]: lane l = column u = 32 ub + (l & 31), slot j = x = 32 xt + (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 8 s + j
      // (the order in which pass 1's accumulator registers hold x)
      std::vector<_Float16> m2s((size_t)(nfp / 32) * (Nxp / 32) * 2 * 4 * 64 * 8, (_Float16)0.f);
      for (int ub = 0; ub < nfp / 32; ++ub)
        for (int xt = 0; xt < Nxp / 32; ++xt)
          for (int s2i = 0; s2i < 2; ++s2i)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int r = 8 * s2i + j, x = 32 * xt + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), u = 32 * ub + (l & 31);
                if (u < nf && x < N)
                  put(m2s, ((size_t)ub * (Nxp / 32) + xt) * 2 + s2i, l, j, t->focal_m2[((size_t)x * nf + u) * 2] * s2,
                      t->focal_m2[((size_t)x * nf + u) * 2 + 1] * s2);
              }
      if ((rc = dev_alloc(e, &e->focal_m1s, m1s.size(), false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->focal_m2s, m2s.size(), false)) != AOG_OK) return rc;
      HIP_TRY(hipMemcpy(e->focal_m1s, m1s.data(), sizeof(_Float16) * m1s.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(e->focal_m2s, m2s.data(), sizeof(_Float16) * m2s.size(), hipMemcpyHostToDevice));
      e->focal_unscale = (float)(1.0 / (s1 * s2));
      std::vector<int32_t> apidx((size_t)e->n_ap), yx((size_t)e->n_ap);
      HIP_TRY(hipMemcpy(apidx.data(), e->ap_index, sizeof(int32_t) * e->n_ap, hipMemcpyDeviceToHost));
      for (int i = 0; i < e->n_ap; ++i) yx[i] = ((apidx[i] / N) << 16) | (apidx[i] % N);
      if ((rc = dev_alloc(e, &e->focal_ap_yx, yx.size(), false)) != AOG_OK) return rc;
      HIP_TRY(hipMemcpy(e->focal_ap_yx, yx.data(), sizeof(int32_t) * yx.size(), hipMemcpyHostToDevice));
    }
    e->n_focal = nf;
  }
  e->tables_ready = true;
  return AOG_OK;
}

int aog_set_screens_f64(aog_env* e, const double* psi, int first, int count, void* stream) {
  if (int rc = set_screens<double>(e, psi, first, count, static_cast<hipStream_t>(stream))) return rc;
  return clear_poison_if_whole(e, first, count, static_cast<hipStream_t>(stream));
}

int aog_set_screens_f32(aog_env* e, const float* psi, int first, int count, void* stream) {
  if (int rc = set_screens<float>(e, psi, first, count, static_cast<hipStream_t>(stream))) return rc;
  return clear_poison_if_whole(e, first, count, static_cast<hipStream_t>(stream));
}

int aog_upload_layer(aog_env* e, const aog_layer_tables* t) {
  if (!e || !t) return fail(AOG_ERR_INVALID, "aog_upload_layer: null argument");
  if (!e->cfg.atm_dynamic) return fail(AOG_ERR_STATE, "aog_upload_layer: handle was not created with atm_dynamic = 1");
  if (!t->stencil_vertical || !t->stencil_horizontal || !t->A_vertical || !t->B_vertical || !t->A_horizontal || !t->B_horizontal)
    return fail(AOG_ERR_INVALID, "aog_upload_layer: null table pointer");
  const int N = e->cfg.n_pupil;
  if (t->nz_vertical < 1 || t->nz_horizontal < 1 || t->nz_vertical > 4 * N || t->nz_horizontal > 4 * N || !(t->pixel_pitch > 0) ||
      !(t->delta_t > 0))
    return fail(AOG_ERR_INVALID, "aog_upload_layer: bad sizes");
  for (int k = 0; k < t->nz_vertical; ++k)
    if (t->stencil_vertical[k] < 0 || t->stencil_vertical[k] >= N * N) return fail(AOG_ERR_INVALID, "aog_upload_layer: stencil index out of range");
  for (int k = 0; k < t->nz_horizontal; ++k)
    if (t->stencil_horizontal[k] < 0 || t->stencil_horizontal[k] >= N * N) return fail(AOG_ERR_INVALID, "aog_upload_layer: stencil index out of range");
  // the lock-step round kernel overwrites the row / column that drops out while other workgroups still gather stencil
  // samples: only legal if no stencil sample lies in the last logical row (vertical) / column (horizontal)
  bool safe = true;
  for (int k = 0; k < t->nz_vertical; ++k) safe &= t->stencil_vertical[k] / N != N - 1;
  for (int k = 0; k < t->nz_horizontal; ++k) safe &= t->stencil_horizontal[k] % N != N - 1;
  (void)safe;
  HIP_TRY(hipSetDevice(e->device));
  e->nz_v = t->nz_vertical;
  e->nz_h = t->nz_horizontal;
  e->sqrt_cn2 = t->sqrt_cn_squared;
  e->pitch = t->pixel_pitch;
  e->delta_t = t->delta_t;
  int rc;
  auto upload_t = [&](const double* src, int rows, int cols, double** dst) -> int {  // src [rows][cols] -> dst [cols][rows]
    std::vector<double> tr((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) tr[(size_t)c * rows + r] = src[(size_t)r * cols + c];
    if (!*dst && (rc = dev_alloc(e, dst, tr.size(), false)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(*dst, tr.data(), sizeof(double) * tr.size(), hipMemcpyHostToDevice));
    return AOG_OK;
  };
  // src [rows][cols] -> [row block][k / 8][lane = (k % 4) * 16 + row % 16][(k / 4) % 2], zero padded: one 16-B load per lane
  // feeds the A operands of two consecutive v_mfma_f64_16x16x4 k-steps
  auto upload_blocked = [&](const double* src, int rows, int cols, double** dst) -> int {
    const int nrb = (rows + 15) / 16, k8 = (cols + 7) / 8;
    std::vector<double> blk((size_t)nrb * k8 * 128, 0.0);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const int lane = (c & 3) * 16 + (r & 15);
        blk[(((size_t)(r >> 4) * k8 + (c >> 3)) * 64 + lane) * 2 + ((c >> 2) & 1)] = src[(size_t)r * cols + c];
      }
    if (!*dst && (rc = dev_alloc(e, dst, blk.size(), false)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(*dst, blk.data(), sizeof(double) * blk.size(), hipMemcpyHostToDevice));
    return AOG_OK;
  };
  if (e->layer_ready) return fail(AOG_ERR_STATE, "aog_upload_layer: already uploaded");
  // The device keeps the stencil samples (and the matching columns of A) with the NEAR ones first — the samples in the two newest slices
  // (rows 0, 1 of the 'bottom' stencil, columns 0, 1 of the 'left' one), which change with every extrusion — and the FAR ones after
  // them: k_extrude16_split fetches an env's far samples for the next round ahead of the inter-workgroup barrier.  A permutation of
  // the terms of A z: every kernel form reads the same arrays.
  auto near_first = [&](const int32_t* stencil, const double* A, int nz, bool vertical, std::vector<int32_t>& st, std::vector<double>& Ap) -> int {
    std::vector<int> order;
    for (int pass = 0; pass < 2; ++pass)
      for (int k = 0; k < nz; ++k) {
        const int slice = vertical ? stencil[k] / N : stencil[k] % N;
        if ((slice < 2) == (pass == 0)) order.push_back(k);
      }
    int n_near = 0;
    for (int k = 0; k < nz; ++k) n_near += (vertical ? stencil[k] / N : stencil[k] % N) < 2;
    st.resize(nz);
    Ap.resize((size_t)N * nz);
    for (int k = 0; k < nz; ++k) {
      st[k] = stencil[order[k]];
      for (int r = 0; r < N; ++r) Ap[(size_t)r * nz + k] = A[(size_t)r * nz + order[k]];
    }
    return n_near;
  };
  std::vector<int32_t> st_v, st_h;
  std::vector<double> Ap_v, Ap_h;
  e->near_v = near_first(t->stencil_vertical, t->A_vertical, e->nz_v, true, st_v, Ap_v);
  e->near_h = near_first(t->stencil_horizontal, t->A_horizontal, e->nz_h, false, st_h, Ap_h);
  if ((rc = upload_blocked(Ap_v.data(), N, e->nz_v, &e->Wa_v)) != AOG_OK) return rc;
  if ((rc = upload_blocked(t->B_vertical, N, N, &e->Wb_v)) != AOG_OK) return rc;
  if ((rc = upload_blocked(Ap_h.data(), N, e->nz_h, &e->Wa_h)) != AOG_OK) return rc;
  if ((rc = upload_blocked(t->B_horizontal, N, N, &e->Wb_h)) != AOG_OK) return rc;
  if ((rc = upload_t(Ap_v.data(), N, e->nz_v, &e->At_v)) != AOG_OK) return rc;
  if ((rc = upload_t(t->B_vertical, N, N, &e->Bt_v)) != AOG_OK) return rc;
  if ((rc = upload_t(Ap_h.data(), N, e->nz_h, &e->At_h)) != AOG_OK) return rc;
  if ((rc = upload_t(t->B_horizontal, N, N, &e->Bt_h)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->stencil_v, e->nz_v, false)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->stencil_h, e->nz_h, false)) != AOG_OK) return rc;
  HIP_TRY(hipMemcpy(e->stencil_v, st_v.data(), sizeof(int32_t) * e->nz_v, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->stencil_h, st_h.data(), sizeof(int32_t) * e->nz_h, hipMemcpyHostToDevice));
  {
    std::vector<int32_t> pv(e->nz_v), ph(e->nz_h);
    for (int k = 0; k < e->nz_v; ++k) pv[k] = (int32_t)(((uint32_t)(st_v[k] / N) << 16) | (uint32_t)(st_v[k] % N));
    for (int k = 0; k < e->nz_h; ++k) ph[k] = (int32_t)(((uint32_t)(st_h[k] / N) << 16) | (uint32_t)(st_h[k] % N));
    if ((rc = dev_alloc(e, &e->stencil_v_yx, e->nz_v, false)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->stencil_h_yx, e->nz_h, false)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(e->stencil_v_yx, pv.data(), sizeof(int32_t) * e->nz_v, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->stencil_h_yx, ph.data(), sizeof(int32_t) * e->nz_h, hipMemcpyHostToDevice));
  }
  e->layer_ready = true;
  return AOG_OK;
}

int aog_set_wind(aog_env* e, const double* velocity_dev, double max_abs_component, void* stream) {
  if (!e || !velocity_dev || !(max_abs_component >= 0)) return fail(AOG_ERR_INVALID, "aog_set_wind: bad argument");
  e->max_wind = max_abs_component;
  if (!e->cfg.atm_dynamic) return fail(AOG_ERR_STATE, "aog_set_wind: handle was not created with atm_dynamic = 1");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(e->velocity, velocity_dev, sizeof(double) * 2 * e->B, hipMemcpyDeviceToDevice, s));
  // Group envs of similar per-step shift (|dx|, |dy|) for k_extrude16_split: a 16-env group runs max(|dx| + |dy|) rounds and a
  // round whose envs are split between column and row extrusion costs two matrix passes.  Results do not depend on the grouping.
  std::vector<double> v((size_t)2 * e->B);
  HIP_TRY(hipMemcpyAsync(v.data(), velocity_dev, sizeof(double) * v.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<int32_t> perm((size_t)e->n_ext_groups * aog::kExt16G, -1);
  std::vector<int32_t> order(e->B);
  for (int i = 0; i < e->B; ++i) order[i] = i;
  const double per_step = e->pitch > 0 ? e->delta_t / e->pitch : 1.0;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const double ax = std::fabs(v[2 * a]) * per_step, bx = std::fabs(v[2 * b]) * per_step;
    const long qa = std::lround(ax * 2), qb = std::lround(bx * 2);   // half-pixel bins of |dx|, then by |dy|
    if (qa != qb) return qa < qb;
    return std::fabs(v[2 * a + 1]) < std::fabs(v[2 * b + 1]);
  });
  for (int i = 0; i < e->B; ++i) perm[i] = order[i];
  HIP_TRY(hipMemcpyAsync(e->ext_perm, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return AOG_OK;
}

int aog_set_extrusion_noise(aog_env* e, const double* noise_dev, int max_ext, void* stream) {
  (void)stream;
  if (!e || max_ext < 0) return fail(AOG_ERR_INVALID, "aog_set_extrusion_noise: bad argument");
  if (!e->cfg.atm_dynamic) return fail(AOG_ERR_STATE, "aog_set_extrusion_noise: handle was not created with atm_dynamic = 1");
  e->next_noise = noise_dev;
  e->next_noise_max_ext = noise_dev ? max_ext : 0;
  return AOG_OK;
}

int aog_set_rng_seed(aog_env* e, uint64_t seed) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_rng_seed: null handle");
  e->rng_seed = seed;
  return AOG_OK;
}

int aog_get_screens_f64(aog_env* e, double* psi_dev, int first, int count, void* stream) {
  if (!e || !psi_dev) return fail(AOG_ERR_INVALID, "aog_get_screens_f64: null argument");
  if (!e->screens_ready) return fail(AOG_ERR_STATE, "aog_get_screens_f64 before any screen was installed");
  if (int rc = refuse_pre_evolved(e, "aog_get_screens_f64")) return rc;
  if (first < 0 || count < 0 || first + count > e->B) return fail(AOG_ERR_INVALID, "aog_get_screens_f64: env range outside [0,%d)", e->B);
  if (count == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil;
  const size_t n = (size_t)count * N * N;
  if (e->cfg.atm_dynamic) {
    hipLaunchKernelGGL(aog::k_unroll_master, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, e->psi_master, e->origin, psi_dev, first, count, N);
  } else {
    HIP_TRY(hipMemsetAsync(psi_dev, 0, sizeof(double) * n, s));
    const bool fast = e->cfg.precision == AOG_PRECISION_FAST;
    hipLaunchKernelGGL(aog::k_screens_from_store, dim3((e->n_ap + 255) / 256, count), dim3(256), 0, s, fast ? e->psi_tile : nullptr,
                       fast ? nullptr : e->psi64, e->ap_index, psi_dev, first, e->n_ap, e->n_ptiles, N * N, 2.0 * M_PI * e->cfg.wavelength_wfs);
  }
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_set_screen_method(aog_env* e, int method) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_screen_method: null handle");
  if (method != AOG_SCREENS_TWOBAND && method != AOG_SCREENS_HCIPY) return fail(AOG_ERR_INVALID, "aog_set_screen_method: unknown method %d", method);
  e->screen_method = method;
  return AOG_OK;
}

// (m x m) complex64 work buffer + batched 2-D plan of the hipFFT route
static int ensure_fft_plan(aog_env* e, int m, int N) {
  if (e->fft_m == m) return AOG_OK;
  if (e->fft_plan) {
    HIP_TRY(hipDeviceSynchronize());   // (the old plan's work buffers may still be in use on the caller's stream)
    hipfftDestroy((hipfftHandle)(uintptr_t)e->fft_plan);
    e->fft_plan = nullptr;
    e->fft_m = 0;
  }
  dev_release(e, &e->fft_work);
  dev_release(e, &e->fft_crop);
  // batch so that the complex64 work buffer stays under ~2 GiB
  const size_t per = (size_t)m * m * 8;
  int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)e->B, ((size_t)2 << 30) / per));
  int rc;
  if ((rc = dev_alloc(e, &e->fft_work, (size_t)batch * m * m * 2, false)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->fft_crop, (size_t)batch * N * N, false)) != AOG_OK) return rc;
  hipfftHandle plan;
  int dims[2] = {m, m};
  if (hipfftPlanMany(&plan, 2, dims, nullptr, 1, m * m, nullptr, 1, m * m, HIPFFT_C2C, batch) != HIPFFT_SUCCESS)
    return fail(AOG_ERR_HIP, "hipfftPlanMany(%d x %d, batch %d) failed", m, m, batch);
  e->fft_plan = (void*)(uintptr_t)plan;
  e->fft_m = m;
  e->fft_batch = batch;
  return AOG_OK;
}

// Two-band synthesis (aogym_kernels.h, "K8 (two-band form)"): low band on hcipy's (q N)^2 grid below 2 cycles per pupil diameter, high band
// on the (2 N)^2 grid, variance split by w_high = smootherstep((f^2 - f1^2) / (f2^2 - f1^2)), f1 = q / 2, f2 = 2 q in units of du_fine.
static int generate_twoband(aog_env* e, int first, int count, int qf, double cn_squared, double outer_scale, double pixel_pitch, hipStream_t s) {
  const int N = e->cfg.n_pupil, KL = 2 * qf, mH = 2 * N, Mf = qf * N;
  const double duH = 2.0 * M_PI / ((double)mH * pixel_pitch), duL = 2.0 * M_PI / ((double)Mf * pixel_pitch);
  const double u0 = 2.0 * M_PI / outer_scale;
  const double r0 = std::pow(0.423 * 4.0 * M_PI * M_PI, -3.0 / 5.0);
  // sample amplitude in the screen's final unit: sqrt(PSD) du / (2 pi) sqrt(Cn^2) = A0 (f^2 + u0^2)^(-11/12) sqrt(Cn^2) / (m delta)
  const double A0 = std::sqrt(0.0229 * std::pow(r0, -5.0 / 3.0)) * std::pow(2.0 * M_PI, 11.0 / 6.0) * std::sqrt(cn_squared);
  aog::Screen2Args a{};
  a.N = N;
  a.qf = qf;
  a.KL = KL;
  a.seed = e->rng_seed;
  a.gen = e->screen_gen;
  a.env_base = e->cfg.env_id_base;
  a.duH = (float)duH;
  a.duL = (float)duL;
  a.u0sq = (float)(u0 * u0);
  a.ampH = (float)(A0 / ((double)mH * pixel_pitch));
  a.ampL = (float)(A0 / ((double)Mf * pixel_pitch));
  const double f1 = 0.5 * qf, f2 = 2.0 * qf;
  a.win.inv_du2 = (float)(1.0 / (duL * duL));
  a.win.f1sq = (float)(f1 * f1);
  a.win.inv_band = (float)(1.0 / (f2 * f2 - f1 * f1));
  const int LW = N % 64 == 0 ? 64 : (N % 60 == 0 ? 60 : 0);
  const int R = LW ? N / LW : 0;
  if ((R == 1 || R == 2 || R == 4 || R == 8) && !getenv("AOG_SCREENS_FULLFFT")) {
    const size_t per_env = aog::screen2_T_elems(N, KL, 32 / R) * 2;   // floats of T (column tiles of 32 / R outputs)
    if (e->syn_m != -Mf) {   // (workspace key: negative = two-band layout)
      int rc;
      const int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)e->B, ((size_t)4 << 30) / (per_env * sizeof(float))));
      if (e->syn_T) HIP_TRY(hipDeviceSynchronize());   // (a workspace of another method / oversampling may still be in use)
      dev_release(e, &e->syn_T);
      dev_release(e, &e->syn_out);
      if ((rc = dev_alloc(e, &e->syn_T, per_env * batch, false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->syn_out, (size_t)batch * N * N, false)) != AOG_OK) return rc;
      e->syn_batch = batch;
      e->syn_m = -Mf;
    }
    a.T = reinterpret_cast<float2*>(e->syn_T);
    a.out = e->syn_out;
    const size_t lds = (size_t)4 * 64 * 65 * sizeof(float), lds_cols = (size_t)aog::kColsWaves * 64 * 65 * sizeof(float);
    auto rows = LW == 64 ? (R == 1 ? aog::k_screen2_rows<1, 64> : R == 2 ? aog::k_screen2_rows<2, 64> : R == 4 ? aog::k_screen2_rows<4, 64> : aog::k_screen2_rows<8, 64>)
                         : (R == 1 ? aog::k_screen2_rows<1, 60> : R == 2 ? aog::k_screen2_rows<2, 60> : R == 4 ? aog::k_screen2_rows<4, 60> : aog::k_screen2_rows<8, 60>);
    auto cols = LW == 64 ? (R == 1 ? aog::k_screen2_cols<1, 64> : R == 2 ? aog::k_screen2_cols<2, 64> : R == 4 ? aog::k_screen2_cols<4, 64> : aog::k_screen2_cols<8, 64>)
                         : (R == 1 ? aog::k_screen2_cols<1, 60> : R == 2 ? aog::k_screen2_cols<2, 60> : R == 4 ? aog::k_screen2_cols<4, 60> : aog::k_screen2_cols<8, 60>);
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(rows), lds, e->device)) return rc;
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(cols), lds_cols, e->device)) return rc;
    const int NL = 32 / R;
    const int nHgroups = (N + 1 + NL - 1) / NL, nHblocks = (nHgroups + 3) / 4, nLblocks = (KL + 3) / 4;
    const int n_launch = (count + e->syn_batch - 1) / e->syn_batch;
    const int per_launch = (count + n_launch - 1) / n_launch;   // even shares (no short tail launch)
    for (int done = 0; done < count; done += per_launch) {
      const int nb = std::min(per_launch, count - done);
      a.first_local = first + done;
      {
        TimedRegion tr(e, s, AOG_PROF_SCREEN_ROWS);
        hipLaunchKernelGGL(rows, dim3(nHblocks + nLblocks, nb), dim3(256), lds, s, a);
      }
      {
        TimedRegion tr(e, s, AOG_PROF_SCREEN_COLS);
        hipLaunchKernelGGL(cols, dim3((N + aog::kColsWaves * NL - 1) / (aog::kColsWaves * NL), nb), dim3(64 * aog::kColsWaves), lds_cols, s, a);
      }
      HIP_TRY(hipGetLastError());
      int rc = set_screens<float>(e, e->syn_out, first + done, nb, s);
      if (rc != AOG_OK) return rc;
    }
  } else {
    // general route: high band by spectrum fill + hipFFT + crop, low band by direct sums
    if (int rc = ensure_fft_plan(e, mH, N)) return rc;
    hipfftHandle plan = (hipfftHandle)(uintptr_t)e->fft_plan;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return fail(AOG_ERR_HIP, "hipfftSetStream failed");
    if (!e->low_c || e->low_key != KL * 65536 + e->fft_batch) {
      int rc;
      if (e->low_c) HIP_TRY(hipDeviceSynchronize());
      dev_release(e, &e->low_c);
      dev_release(e, &e->low_T);
      if ((rc = dev_alloc(e, &e->low_c, (size_t)e->fft_batch * KL * 2 * KL * 2, false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->low_T, (size_t)e->fft_batch * KL * N * 2, false)) != AOG_OK) return rc;
      e->low_key = KL * 65536 + e->fft_batch;
    }
    for (int done = 0; done < count; done += e->fft_batch) {
      const int nb = std::min(e->fft_batch, count - done);
      const int lw = aog::spectrum_lane_width(N), n_r = (N + lw - 1) / lw;
      const size_t calls = (size_t)2 * lw * ((n_r + 3) / 4) * mH;
      a.first_local = first + done;
      hipLaunchKernelGGL(aog::k_spectrum_fill, dim3((unsigned)((calls + 255) / 256), nb), dim3(256), 0, s, reinterpret_cast<float2*>(e->fft_work), mH, 2,
                         first + done, e->cfg.env_id_base, e->rng_seed, e->screen_gen, a.duH, a.u0sq, a.ampH, 1, a.win);
      HIP_TRY(hipGetLastError());
      if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(e->fft_work), reinterpret_cast<hipfftComplex*>(e->fft_work), HIPFFT_BACKWARD) !=
          HIPFFT_SUCCESS)
        return fail(AOG_ERR_HIP, "hipfftExecC2C failed");
      hipLaunchKernelGGL(aog::k_screen_crop, dim3((N * N + 255) / 256, nb), dim3(256), 0, s, reinterpret_cast<const float2*>(e->fft_work), e->fft_crop, mH,
                         N, 1.0f);
      hipLaunchKernelGGL(aog::k_lowband_spectrum, dim3((KL * 2 * KL + 255) / 256, nb), dim3(256), 0, s, reinterpret_cast<float2*>(e->low_c), a);
      hipLaunchKernelGGL(aog::k_lowband_lines, dim3((KL * N + 255) / 256, nb), dim3(256), 0, s, reinterpret_cast<const float2*>(e->low_c),
                         reinterpret_cast<float2*>(e->low_T), N, KL, Mf);
      hipLaunchKernelGGL(aog::k_lowband_add, dim3((N * N + 255) / 256, nb), dim3(256), 0, s, reinterpret_cast<const float2*>(e->low_T), e->fft_crop, N, KL, Mf);
      HIP_TRY(hipGetLastError());
      int rc = set_screens<float>(e, e->fft_crop, first + done, nb, s);
      if (rc != AOG_OK) return rc;
    }
  }
  hipLaunchKernelGGL(aog::k_bump_generation, dim3((count + 255) / 256), dim3(256), 0, s, e->screen_gen + first, count);
  HIP_TRY(hipGetLastError());
  return clear_poison_if_whole(e, first, count, s);
}

int aog_generate_screens(aog_env* e, int first, int count, int oversampling, double cn_squared, double outer_scale, double pixel_pitch,
                         void* stream) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_generate_screens: null handle");
  if (!e->tables_ready) return fail(AOG_ERR_STATE, "aog_generate_screens before aog_upload_tables");
  if (first < 0 || count < 0 || first + count > e->B) return fail(AOG_ERR_INVALID, "aog_generate_screens: env range outside [0,%d)", e->B);
  if (oversampling < 1 || oversampling > 32 || !(cn_squared > 0) || !(outer_scale > 0) || !(pixel_pitch > 0))
    return fail(AOG_ERR_INVALID, "aog_generate_screens: bad parameter");
  if (count == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil, m = N * oversampling;
  if ((m & 1) != 0) return fail(AOG_ERR_UNSUPPORTED, "aog_generate_screens: odd FFT size");
  // two-band form: needs a fine grid at least 4x oversampled (the low band ends at 2 cycles per pupil diameter, the coarse grid samples
  // every half cycle) and an even pupil; anything else is drawn literally
  if (e->screen_method == AOG_SCREENS_TWOBAND && oversampling >= 4 && oversampling % 2 == 0 && N % 4 == 0)
    return generate_twoband(e, first, count, oversampling, cn_squared, outer_scale, pixel_pitch, s);
  // pruned synthesis (no (qN)^2 array): N = 64 R or 60 R with R in {1, 2, 4, 8} (64 .. 512; 60, 120, 240, 480) and power-of-two
  // oversampling
  const int LW = N % 64 == 0 ? 64 : (N % 60 == 0 ? 60 : 0);
  const int Rr = LW ? N / LW : 0;
  const bool pow2 = (oversampling & (oversampling - 1)) == 0 && (Rr == 1 || Rr == 2 || Rr == 4 || Rr == 8);
  if (pow2 && !getenv("AOG_SCREENS_FULLFFT")) {
    const int lines = m / 2 + 1;                 // half-plane synthesis: spectrum lines 0 .. m/2 (k_screen_rows)
    const size_t per_env = (size_t)lines * N * 2;   // floats of T
    int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)e->B, ((size_t)2 << 30) / (per_env * sizeof(float))));
    if (e->syn_m != m) {   // (a different oversampling or method: the old workspace is given back first)
      int rc;
      if (e->syn_T) HIP_TRY(hipDeviceSynchronize());
      dev_release(e, &e->syn_T);
      dev_release(e, &e->syn_out);
      if ((rc = dev_alloc(e, &e->syn_T, per_env * batch, false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &e->syn_out, (size_t)batch * N * N, false)) != AOG_OK) return rc;
      e->syn_batch = batch;
      e->syn_m = m;
    }
    const double du = 2.0 * M_PI / ((double)m * pixel_pitch);
    const double u0 = 2.0 * M_PI / outer_scale;
    const double r0 = std::pow(0.423 * 4.0 * M_PI * M_PI, -3.0 / 5.0);
    const double amp_scale = std::sqrt(0.0229 * std::pow(r0, -5.0 / 3.0)) * std::pow(2.0 * M_PI, 11.0 / 6.0) * (2.0 * M_PI) / du;
    aog::ScreenSynthArgs a{};
    a.T = reinterpret_cast<float2*>(e->syn_T);
    a.out = e->syn_out;
    a.N = N;
    a.q = oversampling;
    a.seed = e->rng_seed;
    a.gen = e->screen_gen;
    a.env_base = e->cfg.env_id_base;
    a.du = (float)du;
    a.u0sq = (float)(u0 * u0);
    a.amp_scale = (float)amp_scale;
    a.crop_scale = (float)(std::sqrt(cn_squared) / ((double)m * m * pixel_pitch * pixel_pitch));
    const size_t lds = (size_t)4 * 64 * 65 * sizeof(float), lds_cols = (size_t)aog::kColsWaves * 64 * 65 * sizeof(float);
    const int R = Rr;
    auto rows = LW == 64 ? (R == 1 ? aog::k_screen_rows<1, 64> : R == 2 ? aog::k_screen_rows<2, 64> : R == 4 ? aog::k_screen_rows<4, 64> : aog::k_screen_rows<8, 64>)
                         : (R == 1 ? aog::k_screen_rows<1, 60> : R == 2 ? aog::k_screen_rows<2, 60> : R == 4 ? aog::k_screen_rows<4, 60> : aog::k_screen_rows<8, 60>);
    auto cols = LW == 64 ? (R == 1 ? aog::k_screen_cols<1, 64> : R == 2 ? aog::k_screen_cols<2, 64> : R == 4 ? aog::k_screen_cols<4, 64> : aog::k_screen_cols<8, 64>)
                         : (R == 1 ? aog::k_screen_cols<1, 60> : R == 2 ? aog::k_screen_cols<2, 60> : R == 4 ? aog::k_screen_cols<4, 60> : aog::k_screen_cols<8, 60>);
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(rows), lds, e->device)) return rc;
    if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(cols), lds_cols, e->device)) return rc;
    const int n_launch = (count + e->syn_batch - 1) / e->syn_batch;
    const int per_launch = (count + n_launch - 1) / n_launch;   // even shares (no short tail launch)
    for (int done = 0; done < count; done += per_launch) {
      const int nb = std::min(per_launch, count - done);
      a.first_local = first + done;
      {
        TimedRegion tr(e, s, AOG_PROF_SCREEN_ROWS);
        hipLaunchKernelGGL(rows, dim3((lines + 3) / 4, nb), dim3(256), lds, s, a);
      }
      {
        TimedRegion tr(e, s, AOG_PROF_SCREEN_COLS);
        hipLaunchKernelGGL(cols, dim3((N + aog::kColsWaves - 1) / aog::kColsWaves, nb), dim3(64 * aog::kColsWaves), lds_cols, s, a);
      }
      HIP_TRY(hipGetLastError());
      int rc = set_screens<float>(e, e->syn_out, first + done, nb, s);
      if (rc != AOG_OK) return rc;
    }
    hipLaunchKernelGGL(aog::k_bump_generation, dim3((count + 255) / 256), dim3(256), 0, s, e->screen_gen + first, count);
    HIP_TRY(hipGetLastError());
    return clear_poison_if_whole(e, first, count, s);
  }
  if (int rc = ensure_fft_plan(e, m, N)) return rc;
  hipfftHandle plan = (hipfftHandle)(uintptr_t)e->fft_plan;
  if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return fail(AOG_ERR_HIP, "hipfftSetStream failed");
  const double du = 2.0 * M_PI / ((double)m * pixel_pitch);
  const double u0 = 2.0 * M_PI / outer_scale;
  const double r0 = std::pow(0.423 * 4.0 * M_PI * M_PI, -3.0 / 5.0);  // Fried parameter for Cn^2 = 1 at 1 m
  // a = sqrt(0.0229 r0^(-5/3)) (2 pi)^(11/6) (f^2 + u0^2)^(-11/12) (2 pi) / du
  const double amp_scale = std::sqrt(0.0229 * std::pow(r0, -5.0 / 3.0)) * std::pow(2.0 * M_PI, 11.0 / 6.0) * (2.0 * M_PI) / du;
  const float crop_scale = (float)(std::sqrt(cn_squared) / ((double)m * m * pixel_pitch * pixel_pitch));
  for (int done = 0; done < count; done += e->fft_batch) {
    const int nb = std::min(e->fft_batch, count - done);
    const int q = m / N, lw = aog::spectrum_lane_width(N), n_r = (N + lw - 1) / lw;
    const size_t calls = (size_t)q * lw * ((n_r + 3) / 4) * m;
    hipLaunchKernelGGL(aog::k_spectrum_fill, dim3((unsigned)((calls + 255) / 256), nb), dim3(256), 0, s, reinterpret_cast<float2*>(e->fft_work), m, q,
                       first + done, e->cfg.env_id_base, e->rng_seed, e->screen_gen, (float)du, (float)(u0 * u0), (float)amp_scale, 0, aog::BandWindow{});
    HIP_TRY(hipGetLastError());
    // the plan is batched for fft_batch transforms; surplus slots of a short last chunk hold stale (finite) data and are ignored
    if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(e->fft_work), reinterpret_cast<hipfftComplex*>(e->fft_work), HIPFFT_BACKWARD) !=
        HIPFFT_SUCCESS)
      return fail(AOG_ERR_HIP, "hipfftExecC2C failed");
    hipLaunchKernelGGL(aog::k_screen_crop, dim3((N * N + 255) / 256, nb), dim3(256), 0, s, reinterpret_cast<const float2*>(e->fft_work), e->fft_crop, m,
                       N, crop_scale);
    HIP_TRY(hipGetLastError());
    int rc = set_screens<float>(e, e->fft_crop, first + done, nb, s);
    if (rc != AOG_OK) return rc;
  }
  hipLaunchKernelGGL(aog::k_bump_generation, dim3((count + 255) / 256), dim3(256), 0, s, e->screen_gen + first, count);
  HIP_TRY(hipGetLastError());
  return clear_poison_if_whole(e, first, count, s);
}

int aog_upload_sh(aog_env* e, const aog_sh_tables* t) {
  if (!e || !t) return fail(AOG_ERR_INVALID, "aog_upload_sh: null argument");
  if (!e->tables_ready) return fail(AOG_ERR_STATE, "aog_upload_sh before aog_upload_tables");
  if (e->cfg.precision != AOG_PRECISION_FAST) return fail(AOG_ERR_UNSUPPORTED, "aog_upload_sh: the Shack-Hartmann chain is built for the fast precision only");
  if (e->sh_ready) return fail(AOG_ERR_STATE, "aog_upload_sh: already uploaded");
  if (t->n_sub < 1 || !t->sub_slot || !t->centres || !t->slopes_ref || !t->reconstruction || !t->mla_phase || !t->transfer || !t->x_det)
    return fail(AOG_ERR_INVALID, "aog_upload_sh: bad table");
  const int N = e->cfg.n_pupil;
  const size_t N2 = (size_t)N * N;
  for (size_t i = 0; i < N2; ++i)
    if (t->sub_slot[i] < -1 || t->sub_slot[i] >= t->n_sub) return fail(AOG_ERR_INVALID, "aog_upload_sh: sub_slot out of range");
  HIP_TRY(hipSetDevice(e->device));
  int rc;
  auto up = [&](auto** dst, const auto* src, size_t count) -> int {
    if ((rc = dev_alloc(e, dst, count, false)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(*dst, src, sizeof(**dst) * count, hipMemcpyHostToDevice));
    return AOG_OK;
  };
  e->sh_n_sub = t->n_sub;
  if ((rc = up(&e->sh_slot, t->sub_slot, N2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_centres, t->centres, (size_t)t->n_sub * 2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_ref, t->slopes_ref, (size_t)t->n_sub * 2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_recon, t->reconstruction, (size_t)e->A * t->n_sub * 2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_mla, t->mla_phase, N2 * 2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_tf, t->transfer, N2 * 4 * 2)) != AOG_OK) return rc;
  if ((rc = up(&e->sh_xdet, t->x_det, (size_t)N)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->sh_act, (size_t)e->B * e->A)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->sh_act16, (size_t)e->n_etiles * e->A_pad * 32 * 2)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->sh_phase, (size_t)e->n_etiles * e->n_ptiles * 1024)) != AOG_OK) return rc;
  e->sh_double = t->fft_double != 0;
  // pruned passes: lines of 2N = LW RL with LW = 64 (N = 128, 256, 512) or 60 (N = 240: the reference's pupil, and 480)
  const int sh_lw = aog::spectrum_lane_width(N);
  e->sh_pruned = (!e->sh_double && (N == 128 || N == 256 || N == 512 || N == 240 || N == 480)) ? 2 * N / sh_lw : 0;
  if (e->sh_pruned) {
    // pruned three-pass propagation (k_sh_rows_fwd / k_sh_cols / k_sh_rows_inv): F1T [B][2N][N] in sh_pad; compact field [B][N][N] and
    // GT [B][2N][N] in sh_in (zeroed once: pixels outside the aperture are never written)
    const int L = 2 * N, RL = e->sh_pruned, BC = 64 / RL;
    std::vector<float> tw((size_t)L * 2), tfq((size_t)(L / BC) * 64 * 64 * 2);
    for (int j = 0; j < L; ++j) {
      tw[2 * j] = (float)cos(2.0 * M_PI * j / L);
      tw[2 * j + 1] = (float)sin(2.0 * M_PI * j / L);
    }
    for (int cg = 0; cg < L / BC; ++cg)
      for (int i = 0; i < 64; ++i)             // register i = bb RL + r of layout A: ky = lane + 64 r, kx = cg BC + bb (k_sh_cols)
        for (int lane = 0; lane < 64; ++lane) {
          const int bb = i / RL, r = i % RL;
          const size_t dst = (((size_t)cg * 64 + i) * 64 + lane) * 2;
          if (lane >= sh_lw) { tfq[dst] = 0.f; tfq[dst + 1] = 0.f; continue; }   // (lanes LW .. 63 hold nothing)
          const size_t src = ((size_t)(lane + sh_lw * r) * L + (size_t)cg * BC + bb) * 2;
          tfq[dst] = (float)t->transfer[src];
          tfq[dst + 1] = (float)t->transfer[src + 1];
        }
    if ((rc = up(&e->sh_tw, tw.data(), tw.size())) != AOG_OK) return rc;
    if ((rc = up(&e->sh_tfq, tfq.data(), tfq.size())) != AOG_OK) return rc;
    {
      // Does the transfer function factorise, H[ky][kx] = hx[kx] hy[ky] (the paraxial Fresnel one does)?  hx = H[0][.], hy = H[.][0] / H[0][0];
      // checked on every element in float64.  If so the propagation runs as the separable two-pass form (k_sh_rows_sep / k_sh_cols_sep).
      auto H = [&](int ky, int kx, int c) { return t->transfer[((size_t)ky * L + kx) * 2 + c]; };
      const double d0 = H(0, 0, 0) * H(0, 0, 0) + H(0, 0, 1) * H(0, 0, 1);
      std::vector<double> hx((size_t)L * 2), hy((size_t)L * 2);
      double worst = d0 > 0 ? 0.0 : 1.0;
      if (d0 > 0) {
        for (int k = 0; k < L; ++k) {
          hx[2 * k] = H(0, k, 0);
          hx[2 * k + 1] = H(0, k, 1);
          hy[2 * k] = (H(k, 0, 0) * H(0, 0, 0) + H(k, 0, 1) * H(0, 0, 1)) / d0;      // H[k][0] conj(H[0][0]) / |H[0][0]|^2
          hy[2 * k + 1] = (H(k, 0, 1) * H(0, 0, 0) - H(k, 0, 0) * H(0, 0, 1)) / d0;
        }
        for (int ky = 0; ky < L; ++ky)
          for (int kx = 0; kx < L; ++kx) {
            const double re = hx[2 * kx] * hy[2 * ky] - hx[2 * kx + 1] * hy[2 * ky + 1], im = hx[2 * kx] * hy[2 * ky + 1] + hx[2 * kx + 1] * hy[2 * ky];
            worst = std::max(worst, std::max(std::fabs(re - H(ky, kx, 0)), std::fabs(im - H(ky, kx, 1))));
          }
      }
      e->sh_sep_rl = (worst <= 1e-9 && !getenv("AOG_SH_THREE_PASS")) ? RL : 0;
      // work buffers.  Separable form: phase grid [B][N][N] fp32 (sh_in) + the one intermediate G1 [B][N][N] complex64 (sh_pad).  Three-pass
      // form: phase grid + GT [B][2N][N] complex64 (sh_in, zeroed once) + F1T [B][2N][N] complex64 (sh_pad).
      char* p1 = nullptr;
      char* p2 = nullptr;
      const size_t pad_bytes = e->sh_sep_rl ? (size_t)e->B * N2 * sizeof(float) * 2 : (size_t)e->B * N2 * 2 * sizeof(float) * 2;
      const size_t in_bytes = e->sh_sep_rl ? (size_t)e->B * N2 * sizeof(float) : (size_t)e->B * N2 * 3 * sizeof(float) * 2;
      if ((rc = dev_alloc(e, &p1, pad_bytes, false)) != AOG_OK) return rc;
      if ((rc = dev_alloc(e, &p2, in_bytes, true)) != AOG_OK) return rc;
      e->sh_pad = p1;
      e->sh_in = p2;
      if (e->sh_sep_rl) {
        std::vector<float> hxq((size_t)sh_lw * 64 * 2), hyq((size_t)RL * 64 * 2, 0.f);
        for (int k2 = 0; k2 < sh_lw; ++k2)
          for (int lane = 0; lane < 64; ++lane) {
            const int kx = lane / BC + RL * k2;
            hxq[((size_t)k2 * 64 + lane) * 2] = (float)hx[2 * kx];
            hxq[((size_t)k2 * 64 + lane) * 2 + 1] = (float)hx[2 * kx + 1];
          }
        for (int r = 0; r < RL; ++r)
          for (int lane = 0; lane < sh_lw; ++lane) {
            const int ky = lane + sh_lw * r;
            hyq[((size_t)r * 64 + lane) * 2] = (float)hy[2 * ky];
            hyq[((size_t)r * 64 + lane) * 2 + 1] = (float)hy[2 * ky + 1];
          }
        if ((rc = up(&e->sh_hxq, hxq.data(), hxq.size())) != AOG_OK) return rc;
        if ((rc = up(&e->sh_hyq, hyq.data(), hyq.size())) != AOG_OK) return rc;
      }
    }
    if ((rc = dev_alloc(e, &e->sh_sums, (size_t)e->B * t->n_sub * 3)) != AOG_OK) return rc;
    std::vector<int32_t> apidx((size_t)e->n_ap), yx((size_t)e->n_ap);
    HIP_TRY(hipMemcpy(apidx.data(), e->ap_index, sizeof(int32_t) * e->n_ap, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->n_ap; ++i) yx[i] = ((apidx[i] / N) << 16) | (apidx[i] % N);
    std::vector<float> mla32(N2 * 2);
    for (size_t i = 0; i < N2 * 2; ++i) mla32[i] = (float)t->mla_phase[i];
    if ((rc = up(&e->sh_ap_yx, yx.data(), yx.size())) != AOG_OK) return rc;
    if ((rc = up(&e->sh_mla32, mla32.data(), mla32.size())) != AOG_OK) return rc;
    // the micro-lens factor's argument in revolutions per packed aperture pixel: added to the phase by k_phase_mfma<.., GRID>; the phase
    // grid the first pass reads starts out as "outside the aperture" everywhere (only aperture pixels are ever written)
    std::vector<float> mrev((size_t)e->n_ap);
    for (int i = 0; i < e->n_ap; ++i)
      mrev[i] = (float)(atan2(t->mla_phase[(size_t)apidx[i] * 2 + 1], t->mla_phase[(size_t)apidx[i] * 2]) / (2.0 * M_PI));
    if ((rc = up(&e->sh_ftab, mrev.data(), mrev.size())) != AOG_OK) return rc;
    {
      std::vector<float> fill((size_t)N2, aog::kShOutside);
      for (int b = 0; b < e->B; ++b)
        HIP_TRY(hipMemcpy(static_cast<float*>(e->sh_in) + (size_t)b * N2, fill.data(), sizeof(float) * N2, hipMemcpyHostToDevice));
    }
  } else {
    const size_t cbytes = e->sh_double ? sizeof(double) * 2 : sizeof(float) * 2;
    char* p1 = nullptr;
    char* p2 = nullptr;
    if ((rc = dev_alloc(e, &p1, (size_t)e->B * N2 * 4 * cbytes, false)) != AOG_OK) return rc;
    // zero-padded INPUT of the forward transform: only aperture pixels are ever written (k_sh_field), the padding stays zero because
    // the forward FFT runs out of place into sh_pad — no memset per call
    if ((rc = dev_alloc(e, &p2, (size_t)e->B * N2 * 4 * cbytes, true)) != AOG_OK) return rc;
    e->sh_pad = p1;
    e->sh_in = p2;
  }
  if (!e->sh_double && !e->sh_pruned) {
    std::vector<float> tf32(N2 * 4 * 2);
    for (size_t i = 0; i < tf32.size(); ++i) tf32[i] = (float)t->transfer[i];
    if ((rc = up(&e->sh_tf32, tf32.data(), tf32.size())) != AOG_OK) return rc;
  }
  if ((rc = dev_alloc(e, &e->sh_image, (size_t)e->B * N2, false)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->sh_noisy, (size_t)e->B * N2, false)) != AOG_OK) return rc;
  if (!e->sh_pruned) {
    hipfftHandle plan;
    int dims[2] = {2 * N, 2 * N};
    if (hipfftPlanMany(&plan, 2, dims, nullptr, 1, 4 * N * N, nullptr, 1, 4 * N * N, e->sh_double ? HIPFFT_Z2Z : HIPFFT_C2C, e->B) != HIPFFT_SUCCESS)
      return fail(AOG_ERR_HIP, "hipfftPlanMany(%s %d x %d, batch %d) failed", e->sh_double ? "Z2Z" : "C2C", 2 * N, 2 * N, e->B);
    e->sh_plan = (void*)(uintptr_t)plan;
  }
  e->sh_amp = t->field_amplitude;
  e->sh_scale = t->image_scale;
  e->sh_gain = t->gain;
  e->sh_leak = t->leakage;
  e->sh_ready = true;
  return AOG_OK;
}

int aog_sh_image(aog_env* e, double* image_dev, void* stream) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_sh_image: null handle");
  if (!e->sh_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_sh_image before aog_upload_sh / aog_set_screens");
  if (int rcp = refuse_pre_evolved(e, "aog_sh_image")) return rcp;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil;
  const size_t per = (size_t)4 * N * N;
  hipfftHandle plan = (hipfftHandle)(uintptr_t)e->sh_plan;
  if (!e->sh_pruned && hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return fail(AOG_ERR_HIP, "hipfftSetStream failed");
  if (int rct = ensure_tiles(e, s)) return rct;
  {
    const int n = e->B * e->A_pad;
    hipLaunchKernelGGL(aog::k_sh_act16, dim3((n + 255) / 256), dim3(256), 0, s, e->sh_act, e->sh_act16, e->B, e->A, e->A_pad,
                       2.0 / e->cfg.wavelength_wfs);
    if (e->sh_pruned) {
      TimedRegion tr(e, s, AOG_PROF_SH_FIELD);
      aog_host::launch_phase_field(e, s, e->sh_act16, static_cast<float*>(e->sh_in), (size_t)N * N, N, true);   // reduced phases on the pupil grid: the first pass forms the field
    } else {
      aog_host::launch_phase(e, s, e->sh_act16, e->sh_phase);
    }
  }
  const double norm = 1.0 / (double)per;  // hipFFT's inverse is un-normalised
  const dim3 g_ap((e->n_ap + 255) / 256, e->B), g_per((unsigned)((per + 255) / 256), e->B), g_img((N * N + 255) / 256, e->B);
  if (e->sh_pruned) {
    float2* field = static_cast<float2*>(e->sh_in);
    float2* GT = field + (size_t)e->B * N * N;
    float2* F1T = static_cast<float2*>(e->sh_pad);
    const float2* tw = reinterpret_cast<const float2*>(e->sh_tw);
    const size_t lds = sizeof(float) * 64 * 65 * aog::kShFftWaves, lds_fused = lds + sizeof(double) * 3 * e->sh_n_sub * aog::kShFftWaves;
    const double scale = e->sh_scale * norm * norm;
    // nobody asked for the image (SH_step): photon noise and the estimator's per-lenslet sums are taken inside the last pass
    // (when the per-wave lenslet tables do not fit the LDS beside the transform planes, the unfused pass + k_sh_noise + estimator run instead)
    const bool fused = image_dev == nullptr && lds_fused <= kLdsBytes;
    aog::ShFuseArgs fa{};
    if (fused) {
      e->sh_calls += 1;   // (the noise call the following aog_sh_update(null) would have made)
      fa.sub_slot = e->sh_slot;
      fa.x_det = e->sh_xdet;
      fa.sums = e->sh_sums;
      fa.n_sub = e->sh_n_sub;
      fa.env_base = (size_t)e->cfg.env_id_base;
      fa.seed = e->rng_seed;
      fa.call = e->sh_calls;
      zero_words(e->sh_sums, (size_t)e->B * e->sh_n_sub * 3 * 2, s);
    }
    e->sh_sums_ready = fused;
    auto run = [&](auto rlc, auto lwc) -> int {
      constexpr int RL = decltype(rlc)::v, LW = decltype(lwc)::v, BC = 64 / RL;
      const int L = LW * RL;
      if (e->sh_sep_rl) {
        // separable transfer function: rows (forward, x hx, inverse, keep x < N) then columns (forward, x hy, inverse, keep y < N) over an
        // N x N intermediate (in sh_pad): two passes, 20 N^2 bytes per env instead of three passes and 68 N^2
        float2* G1 = F1T;
        if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_rows_sep<RL, LW>), lds, e->device)) return rc;
        if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_cols_sep<RL, LW, false>), lds, e->device)) return rc;
        if (fused)
          if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_cols_sep<RL, LW, true>), lds_fused, e->device)) return rc;
        const dim3 g1((N / BC + aog::kShFftWaves - 1) / aog::kShFftWaves, e->B);   // pass 1: groups of BC rows; pass 2: groups of BC columns
        {
          TimedRegion tr(e, s, AOG_PROF_SH_ROWS_FWD);
          hipLaunchKernelGGL((aog::k_sh_rows_sep<RL, LW>), g1, dim3(64 * aog::kShFftWaves), lds, s, reinterpret_cast<const float*>(field), G1, tw,
                             reinterpret_cast<const float2*>(e->sh_hxq), (float)e->sh_amp);
        }
        TimedRegion tr(e, s, AOG_PROF_SH_COLS);
        if (fused) hipLaunchKernelGGL((aog::k_sh_cols_sep<RL, LW, true>), g1, dim3(64 * aog::kShFftWaves), lds_fused, s, G1, e->sh_image, tw,
                                      reinterpret_cast<const float2*>(e->sh_hyq), scale, fa);
        else hipLaunchKernelGGL((aog::k_sh_cols_sep<RL, LW, false>), g1, dim3(64 * aog::kShFftWaves), lds, s, G1, e->sh_image, tw,
                                reinterpret_cast<const float2*>(e->sh_hyq), scale, fa);
        return AOG_OK;
      }
      if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_rows_fwd<RL, LW, true>), lds, e->device)) return rc;
      if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_cols<RL, LW>), lds, e->device)) return rc;
      if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_rows_inv<RL, LW, false>), lds, e->device)) return rc;
      if (fused)
        if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_rows_inv<RL, LW, true>), lds_fused, e->device)) return rc;
      const dim3 g_rows((N / BC + aog::kShFftWaves - 1) / aog::kShFftWaves, e->B), g_cols((L / BC + aog::kShFftWaves - 1) / aog::kShFftWaves, e->B);
      {
        TimedRegion tr(e, s, AOG_PROF_SH_ROWS_FWD);
        hipLaunchKernelGGL((aog::k_sh_rows_fwd<RL, LW, true>), g_rows, dim3(64 * aog::kShFftWaves), lds, s, field, F1T, tw, (float)e->sh_amp);
      }
      {
        TimedRegion tr(e, s, AOG_PROF_SH_COLS);
        hipLaunchKernelGGL((aog::k_sh_cols<RL, LW>), g_cols, dim3(64 * aog::kShFftWaves), lds, s, F1T, GT, reinterpret_cast<const float2*>(e->sh_tfq), tw);
      }
      TimedRegion tr(e, s, AOG_PROF_SH_ROWS_INV);
      if (fused) hipLaunchKernelGGL((aog::k_sh_rows_inv<RL, LW, true>), g_rows, dim3(64 * aog::kShFftWaves), lds_fused, s, GT, e->sh_image, tw, scale, fa);
      else hipLaunchKernelGGL((aog::k_sh_rows_inv<RL, LW, false>), g_rows, dim3(64 * aog::kShFftWaves), lds, s, GT, e->sh_image, tw, scale, fa);
      return AOG_OK;
    };
    int rcp;
    if (N % 64 == 0) rcp = e->sh_pruned == 4 ? run(aog::IC<4>{}, aog::IC<64>{}) : e->sh_pruned == 8 ? run(aog::IC<8>{}, aog::IC<64>{}) : run(aog::IC<16>{}, aog::IC<64>{});
    else rcp = e->sh_pruned == 8 ? run(aog::IC<8>{}, aog::IC<60>{}) : run(aog::IC<16>{}, aog::IC<60>{});
    if (rcp) return rcp;
  } else if (e->sh_double) {
    e->sh_sums_ready = false;
    double2* in = static_cast<double2*>(e->sh_in);
    double2* pad = static_cast<double2*>(e->sh_pad);
    hipLaunchKernelGGL(aog::k_sh_field<double2>, g_ap, dim3(256), 0, s, e->sh_phase, e->ap_index, reinterpret_cast<const double2*>(e->sh_mla), in, e->n_ap,
                       e->n_ptiles, N, e->sh_amp, per, 2 * N);
    HIP_TRY(hipGetLastError());
    hipfftDoubleComplex* buf = reinterpret_cast<hipfftDoubleComplex*>(pad);
    if (hipfftExecZ2Z(plan, reinterpret_cast<hipfftDoubleComplex*>(in), buf, HIPFFT_FORWARD) != HIPFFT_SUCCESS)
      return fail(AOG_ERR_HIP, "hipfftExecZ2Z forward failed");
    hipLaunchKernelGGL(aog::k_sh_transfer<double2>, g_per, dim3(256), 0, s, pad, reinterpret_cast<const double2*>(e->sh_tf), per);
    if (hipfftExecZ2Z(plan, buf, buf, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return fail(AOG_ERR_HIP, "hipfftExecZ2Z backward failed");
    hipLaunchKernelGGL(aog::k_sh_intensity<double2>, g_img, dim3(256), 0, s, pad, e->sh_image, N, e->sh_scale * norm * norm);
  } else {
    e->sh_sums_ready = false;
    float2* in = static_cast<float2*>(e->sh_in);
    float2* pad = static_cast<float2*>(e->sh_pad);
    hipLaunchKernelGGL(aog::k_sh_field<float2>, g_ap, dim3(256), 0, s, e->sh_phase, e->ap_index, reinterpret_cast<const double2*>(e->sh_mla), in, e->n_ap,
                       e->n_ptiles, N, e->sh_amp, per, 2 * N);
    HIP_TRY(hipGetLastError());
    hipfftComplex* buf = reinterpret_cast<hipfftComplex*>(pad);
    if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(in), buf, HIPFFT_FORWARD) != HIPFFT_SUCCESS)
      return fail(AOG_ERR_HIP, "hipfftExecC2C forward failed");
    hipLaunchKernelGGL(aog::k_sh_transfer<float2>, g_per, dim3(256), 0, s, pad, reinterpret_cast<const float2*>(e->sh_tf32), per);
    if (hipfftExecC2C(plan, buf, buf, HIPFFT_BACKWARD) != HIPFFT_SUCCESS) return fail(AOG_ERR_HIP, "hipfftExecC2C backward failed");
    hipLaunchKernelGGL(aog::k_sh_intensity<float2>, g_img, dim3(256), 0, s, pad, e->sh_image, N, e->sh_scale * norm * norm);
  }
  HIP_TRY(hipGetLastError());
  if (image_dev) HIP_TRY(hipMemcpyAsync(image_dev, e->sh_image, sizeof(double) * (size_t)e->B * N * N, hipMemcpyDeviceToDevice, s));
  return AOG_OK;
}

int aog_sh_update(aog_env* e, const double* noisy_image_dev, double* action_dev, void* stream) {
  if (!e || !action_dev) return fail(AOG_ERR_INVALID, "aog_sh_update: null argument");
  if (!e->sh_ready) return fail(AOG_ERR_STATE, "aog_sh_update before aog_upload_sh");
  if (int rc = refuse_pre_evolved(e, "aog_sh_update")) return rc;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil;
  const double* img = noisy_image_dev;
  const double* sums_in = nullptr;
  if (!img && e->sh_sums_ready) {
    sums_in = e->sh_sums;   // the preceding aog_sh_image(null) already drew the noise and summed the lenslets
  } else if (!img) {
    e->sh_calls += 1;
    hipLaunchKernelGGL(aog::k_sh_noise, dim3((unsigned)((N * N + 255) / 256), e->B), dim3(256), 0, s, e->sh_image, e->sh_noisy, N,
                       (size_t)e->cfg.env_id_base, e->rng_seed, e->sh_calls, e->sh_pruned ? e->sh_sep_rl : 0);
    img = e->sh_noisy;
  }
  aog::ShEstimateArgs p{};
  p.image = img;
  p.sums_in = sums_in;
  e->sh_sums_ready = false;
  p.sub_slot = e->sh_slot;
  p.x_det = e->sh_xdet;
  p.centres = e->sh_centres;
  p.slopes_ref = e->sh_ref;
  p.recon = e->sh_recon;
  p.sh_act = e->sh_act;
  p.action_out = action_dev;
  p.N = N;
  p.n_sub = e->sh_n_sub;
  p.A = e->A;
  p.gain = e->sh_gain;
  p.leakage = e->sh_leak;
  hipLaunchKernelGGL(aog::k_sh_estimate, dim3(e->B), dim3(256), sizeof(double) * 5 * e->sh_n_sub, s, p);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

extern "C++" {
namespace {
struct StatePart {
  void* ptr;
  size_t bytes;
};
std::vector<StatePart> state_parts(const aog_env* e) {
  std::vector<StatePart> v;
  const size_t N2 = (size_t)e->cfg.n_pupil * e->cfg.n_pupil;
  auto add = [&](void* p, size_t b) { if (p && b) v.push_back({p, b}); };
  add(e->act_dm, sizeof(double) * e->B * e->A);
  add(e->t_render, sizeof(int32_t) * e->B);
  add(e->screen_gen, sizeof(uint32_t) * e->B);
  if (e->ring_direct) {
    // (the fp32 layouts of a ring-direct handle are functions of the master screens, origins and reference pistons saved below)
  } else if (e->cfg.precision == AOG_PRECISION_FAST) {
    add(e->psi_tile, sizeof(float) * (size_t)e->n_etiles * e->n_ptiles * 1024);
    add(e->psi_rev, sizeof(float) * (size_t)e->n_quads * e->Bp * 4);
  } else {
    add(e->psi64, sizeof(double) * (size_t)e->B * e->n_ap);
  }
  if (e->cfg.atm_dynamic) {
    add(e->psi_master, sizeof(double) * e->B * N2);
    add(e->origin, sizeof(int32_t) * 2 * e->B);
    add(e->ext_counter, sizeof(uint32_t) * e->B);
    add(e->psi_offset, sizeof(double) * e->B);
    add(e->psi_sum, sizeof(double) * e->B);
  }
  if (e->sh_ready) add(e->sh_act, sizeof(double) * e->B * e->A);
  return v;
}
}  // namespace
}  // extern "C++"

namespace {
struct StateTail {  // host-side counters that steer the device RNG streams; stored in the last 256 bytes of the blob
  int64_t timestep;
  uint64_t rng_seed;
  uint32_t sh_calls, steps_since_reset;
};
}  // namespace

int64_t aog_state_bytes(const aog_env* e) {
  if (!e) return -1;
  int64_t n = 256;
  for (const auto& p : state_parts(e)) n += (int64_t)((p.bytes + 255) / 256 * 256);
  return n;
}

int aog_get_state(aog_env* e, void* blob_dev, int64_t* timestep_out, void* stream) {
  if (!e || !blob_dev) return fail(AOG_ERR_INVALID, "aog_get_state: null argument");
  if (int rc = refuse_pre_evolved(e, "aog_get_state")) return rc;
  HIP_TRY(hipSetDevice(e->device));
  size_t off = 0;
  for (const auto& p : state_parts(e)) {
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(blob_dev) + off, p.ptr, p.bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    off += (p.bytes + 255) / 256 * 256;
  }
  StateTail tail{e->timestep, e->rng_seed, e->sh_calls, (uint32_t)e->steps_since_reset};
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  HIP_TRY(hipMemcpy(static_cast<char*>(blob_dev) + off, &tail, sizeof tail, hipMemcpyHostToDevice));
  if (timestep_out) *timestep_out = e->timestep;
  return AOG_OK;
}

int aog_set_state(aog_env* e, const void* blob_dev, int64_t timestep, void* stream) {
  if (!e || !blob_dev) return fail(AOG_ERR_INVALID, "aog_set_state: null argument");
  if (!e->tables_ready) return fail(AOG_ERR_STATE, "aog_set_state before aog_upload_tables");
  e->pro_pending = false;   // (a restored state replaces the mirror: a pending pipelined action is forgotten)
  HIP_TRY(hipSetDevice(e->device));
  if (e->pre_evolved) {   // a restored state replaces everything the pending extrusion touches: let it finish, then forget it
    HIP_TRY(hipStreamSynchronize(e->ext_stream));
    e->pre_evolved = false;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t off = 0;
  for (const auto& p : state_parts(e)) {
    HIP_TRY(hipMemcpyAsync(p.ptr, static_cast<const char*>(blob_dev) + off, p.bytes, hipMemcpyDeviceToDevice, s));
    off += (p.bytes + 255) / 256 * 256;
  }
  StateTail tail{};
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipMemcpy(&tail, static_cast<const char*>(blob_dev) + off, sizeof tail, hipMemcpyDeviceToHost));
  e->timestep = timestep;
  e->rng_seed = tail.rng_seed;
  e->sh_calls = tail.sh_calls;
  e->steps_since_reset = tail.steps_since_reset;
  e->sh_sums_ready = false;
  if (e->host_flag && *static_cast<volatile int*>(e->host_flag)) {   // a restored state replaces every screen: the handle is usable again
    HIP_TRY(hipMemset(e->dev_status, 0, sizeof(int)));
    *static_cast<volatile int*>(e->host_flag) = 0;
  }
  if (e->ring_direct) {
    int rc = ring_from_master(e, 0, e->B, 1, s);
    if (rc != AOG_OK) return rc;
  }
  // derived operand layouts follow the restored actuators
  const int n = e->B * e->A_pad;
  hipLaunchKernelGGL(aog::k_load_actuators, dim3((n + 255) / 256), dim3(256), 0, s, e->act_dm, e->act_rev, e->act16, e->B, e->A, e->A_pad, e->Bp,
                     2.0 / e->cfg.wavelength_wfs);
  HIP_TRY(hipGetLastError());
  e->screens_ready = true;
  return AOG_OK;
}

int aog_get_phase_screen(aog_env* e, int env_index, float* phase_dev, void* stream) {
  if (!e || !phase_dev) return fail(AOG_ERR_INVALID, "aog_get_phase_screen: null argument");
  if (!e->screens_ready) return fail(AOG_ERR_STATE, "aog_get_phase_screen before aog_set_screens");
  if (e->cfg.precision != AOG_PRECISION_FAST) return fail(AOG_ERR_UNSUPPORTED, "aog_get_phase_screen: fast precision handles only");
  if (env_index < 0 || env_index >= e->B) return fail(AOG_ERR_INVALID, "aog_get_phase_screen: env %d outside [0,%d)", env_index, e->B);
  if (int rcp = refuse_pre_evolved(e, "aog_get_phase_screen")) return rcp;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t N2 = (size_t)e->cfg.n_pupil * e->cfg.n_pupil;
  if (int rct = ensure_tiles(e, s)) return rct;
  HIP_TRY(hipMemsetAsync(phase_dev, 0, sizeof(float) * N2, s));
  hipLaunchKernelGGL(aog::k_phase_screen, dim3((e->n_ap + 255) / 256), dim3(256), 0, s, e->psi_tile, e->ap_index, phase_dev, env_index, e->n_ap,
                     e->n_ptiles);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_set_return_accumulator(aog_env* e, float* returns_dev) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_return_accumulator: null handle");
  e->ret_acc = returns_dev;
  return AOG_OK;
}

int aog_device_status(aog_env* e, int32_t* status_out) {
  if (!e || !status_out) return fail(AOG_ERR_INVALID, "aog_device_status: null argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  int v[16];
  HIP_TRY(hipMemcpy(v, e->dev_status, sizeof v, hipMemcpyDeviceToHost));
  *status_out = v[0] | *static_cast<volatile int*>(e->host_flag);
  if (getenv("AOG_EXTRUDE_TIMING")) {   // developer aid: phase clocks (10 ns ticks) of workgroup 0 of k_extrude16_split
    if (v[1]) fprintf(stderr, "[aogym] extrude16_split WG0 ticks: gather %d noise %d compute %d (matrix passes %d, exchange %d) barrier %d rounds %d matrix passes run %d, shader clocks in them / 16: %d\n", v[4], v[5], v[6], v[9], v[10], v[7], v[8], v[11], v[12]);
    const int one = 1;
    HIP_TRY(hipMemcpy(e->dev_status + 1, &one, sizeof one, hipMemcpyHostToDevice));
  }
  return AOG_OK;
}

int aog_get_actuators(aog_env* e, double* act_dev, void* stream) {
  if (!e || !act_dev) return fail(AOG_ERR_INVALID, "aog_get_actuators: null argument");
  if (int rc = refuse_pre_evolved(e, "aog_get_actuators")) return rc;   // (pipelined stepping: the mirror already holds the next action)
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(act_dev, e->act_dm, sizeof(double) * e->B * e->A, hipMemcpyDeviceToDevice,
                         static_cast<hipStream_t>(stream)));
  return AOG_OK;
}

int aog_set_actuators(aog_env* e, const double* act_dev, void* stream) {
  if (!e || !act_dev) return fail(AOG_ERR_INVALID, "aog_set_actuators: null argument");
  e->pro_pending = false;   // (whatever a pipelined step had loaded is replaced)
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(e->act_dm, act_dev, sizeof(double) * e->B * e->A, hipMemcpyDeviceToDevice, s));
  const int n = e->B * e->A_pad;
  hipLaunchKernelGGL(aog::k_load_actuators, dim3((n + 255) / 256), dim3(256), 0, s, e->act_dm, e->act_rev, e->act16,
                     e->B, e->A, e->A_pad, e->Bp, 2.0 / e->cfg.wavelength_wfs);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_reset(aog_env* e, const uint8_t* mask, float* obs_raw, uint16_t* obs, void* stream) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_reset: null handle");
  if (!e->tables_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_reset before aog_upload_tables/aog_set_screens");
  if (int rc = check_poisoned(e, "aog_reset")) return rc;
  if (int rc = refuse_pre_evolved(e, "aog_reset")) return rc;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!mask) e->steps_since_reset = 0;
  {
    const int n = e->B * e->A;
    hipLaunchKernelGGL(aog::k_reset_state, dim3((n + 255) / 256), dim3(256), 0, s, mask, e->act_dm, e->t_render, e->B, e->A,
                       e->cfg.flat_mirror_start);
    const int n2 = e->B * e->A_pad;
    hipLaunchKernelGGL(aog::k_load_actuators, dim3((n2 + 255) / 256), dim3(256), 0, s, e->act_dm, e->act_rev, e->act16,
                       e->B, e->A, e->A_pad, e->Bp, 2.0 / e->cfg.wavelength_wfs);
    HIP_TRY(hipGetLastError());
  }
  int rc = launch_fused(e, s);
  if (rc != AOG_OK) return rc;
  return launch_epilogue(e, false, obs_raw, obs, nullptr, nullptr, nullptr, nullptr, s);
}

static int step_impl(aog_env* e, const float* action, const float* action_next, bool pipelined, float* obs_raw, uint16_t* obs, float* reward,
                     uint8_t* done, float* power, float* strehl, void* stream);
int aog_step(aog_env* e, const float* action, float* obs_raw, uint16_t* obs, float* reward, uint8_t* done, float* power,
             float* strehl, void* stream) {
  return step_impl(e, action, nullptr, false, obs_raw, obs, reward, done, power, strehl, stream);
}
int aog_step_pipelined(aog_env* e, const float* action, const float* action_next, float* obs_raw, uint16_t* obs, float* reward, uint8_t* done,
                       float* power, float* strehl, void* stream) {
  return step_impl(e, action, action_next, true, obs_raw, obs, reward, done, power, strehl, stream);
}
static int step_impl(aog_env* e, const float* action, const float* action_next, bool pipelined, float* obs_raw, uint16_t* obs, float* reward,
                     uint8_t* done, float* power, float* strehl, void* stream) {
  if (!e || !action) return fail(AOG_ERR_INVALID, "aog_step: null argument");
  if (!pipelined && e->pro_pending)
    return fail(AOG_ERR_STATE, "aog_step: a pipelined step has already loaded the next action (continue with aog_step_pipelined)");
  if (pipelined && e->lookahead) return fail(AOG_ERR_UNSUPPORTED, "aog_step_pipelined: not together with aog_set_lookahead");
  if (!e->tables_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_step before aog_upload_tables/aog_set_screens");
  if (int rc = check_poisoned(e, "aog_step")) return rc;
  if (e->cfg.reward_type == AOG_REWARD_SMF_SSIM && e->n_obs < 7)
    return fail(AOG_ERR_INVALID, "win_size exceeds image extent (smf_ssim needs obs_dim**2 >= 7; AO_env.py:495)");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  e->timestep += 1;  // AO_env.py:123
  e->steps_since_reset += 1;
  bool join_ext = false;
  e->sh_sums_ready = false;   // (an aog_sh_image(NULL) not followed by its aog_sh_update is void once the env has stepped)
  if (e->cfg.atm_dynamic) {
    if (e->pre_evolved) {   // the previous step launched this step's extrusion on the library's stream: join it
      if (e->next_noise) return fail(AOG_ERR_STATE, "aog_step: extrusion normals were supplied for a step whose extrusion already ran (lookahead "
                                     "draws from the device stream; switch it off for host-supplied normals)");
      join_ext = true;   // (joined just ahead of the fused kernel: the prologue does not read the screens)
      e->pre_evolved = false;
    } else {
      int rce = evolve_layer(e, s, e->timestep);
      if (rce != AOG_OK) return rce;
    }
  }
  // each fused kernel reads one operand layout: write only that one (the float64 device kernel and the VALU kernel read act_rev)
  const bool mfma_fast = e->kernel == AOG_KERNEL_MFMA && e->cfg.precision == AOG_PRECISION_FAST && !e->sh_ready;
  if (!e->pro_pending) {   // (pipelined: the previous call's last launch already turned this step's action into actuators)
    hipLaunchKernelGGL(join_ext ? aog::k_prologue<false> : aog::k_prologue<true>, dim3((e->B + aog::kProEnvs - 1) / aog::kProEnvs), dim3(64 * aog::kProEnvs), 0, s, action, e->gram, e->act_dm,
                       mfma_fast ? nullptr : e->act_rev, e->act16, e->B, e->A,
                       e->A_pad, e->Bp, e->cfg.sh_operation, e->cfg.surface_rms_target, 2.0 / e->cfg.wavelength_wfs);
    HIP_TRY(hipGetLastError());
  }
  e->pro_pending = false;
  if (join_ext) HIP_TRY(hipStreamWaitEvent(s, e->ev_ext_done, 0));
  int rc = launch_fused(e, s);
  if (rc != AOG_OK) return rc;
  // lookahead: step t + 1's wind shift needs nothing from this step's outputs (AO_env.py:125 vs :132-142), only that the fused kernel
  // has finished reading the ring.  Not on an episode's last step: reset() observes the atmosphere as this step left it (AO_env.py:84).
  if (e->cfg.atm_dynamic && e->lookahead && !e->next_noise && e->steps_since_reset < e->cfg.max_steps) {
    HIP_TRY(hipEventRecord(e->ev_fused_done, s));
    HIP_TRY(hipStreamWaitEvent(e->ext_stream, e->ev_fused_done, 0));
    if (int rce = evolve_layer(e, e->ext_stream, e->timestep + 1)) return rce;
    HIP_TRY(hipEventRecord(e->ev_ext_done, e->ext_stream));
    e->pre_evolved = true;
  }
  const int rce = launch_epilogue(e, true, obs_raw, obs, reward, done, power, strehl, s, pipelined ? action_next : nullptr);
  if (rce == AOG_OK && pipelined && action_next) e->pro_pending = true;
  return rce;
}

int aog_focal_image(aog_env* e, int env_index, float* field_dev, void* stream) {
  if (!e || !field_dev) return fail(AOG_ERR_INVALID, "aog_focal_image: null argument");
  if (!e->tables_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_focal_image before aog_upload_tables/aog_set_screens");
  if (!e->n_focal) return fail(AOG_ERR_STATE, "aog_focal_image: focal_m1/focal_m2 were not uploaded");
  if (env_index < 0 || env_index >= e->B) return fail(AOG_ERR_INVALID, "aog_focal_image: env %d outside [0,%d)", env_index, e->B);
  if (int rcp = refuse_pre_evolved(e, "aog_focal_image")) return rcp;
  const bool fast = e->cfg.precision == AOG_PRECISION_FAST;
  if (fast && e->focal_m1s) return aog_focal_images(e, env_index, 1, field_dev, stream);   // the batched matrix-core path
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil, nf = e->n_focal;
  HIP_TRY(hipMemsetAsync(e->focal_E, 0, sizeof(double) * 2 * N * N, s));
  hipLaunchKernelGGL(aog::k_focal_field, dim3((e->n_ap + 255) / 256), dim3(256), 0, s, fast ? e->psi_tile : nullptr,
                     fast ? nullptr : e->psi64, e->modes_f32, e->modes64, e->act_rev, e->act_dm, e->ap_index,
                     reinterpret_cast<double2*>(e->focal_E), env_index, e->n_ap, e->n_ptiles, e->A, e->A_pad, e->Bp, e->cfg.wavelength_wfs);
  hipLaunchKernelGGL(aog::k_cgemm_small, dim3((nf * N + 255) / 256), dim3(256), 0, s, reinterpret_cast<const double2*>(e->focal_m1),
                     reinterpret_cast<const double2*>(e->focal_E), reinterpret_cast<double2*>(e->focal_T), (float2*)nullptr, nf, N, N);
  hipLaunchKernelGGL(aog::k_cgemm_small, dim3((nf * nf + 255) / 256), dim3(256), 0, s, reinterpret_cast<const double2*>(e->focal_T),
                     reinterpret_cast<const double2*>(e->focal_m2), (double2*)nullptr, reinterpret_cast<float2*>(field_dev), nf, N, nf);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_focal_images(aog_env* e, int first, int count, float* field_dev, void* stream) {
  if (!e || !field_dev) return fail(AOG_ERR_INVALID, "aog_focal_images: null argument");
  if (!e->tables_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_focal_images before aog_upload_tables/aog_set_screens");
  if (!e->n_focal) return fail(AOG_ERR_STATE, "aog_focal_images: focal_m1/focal_m2 were not uploaded");
  if (first < 0 || count < 0 || first + count > e->B) return fail(AOG_ERR_INVALID, "aog_focal_images: env range outside [0,%d)", e->B);
  if (e->cfg.precision != AOG_PRECISION_FAST || !e->focal_m1s)
    return fail(AOG_ERR_UNSUPPORTED, "aog_focal_images: fast-precision handles only (use aog_focal_image on a float64 validation handle)");
  if (int rcp = refuse_pre_evolved(e, "aog_focal_images")) return rcp;
  if (count == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = e->cfg.n_pupil, nf = e->n_focal;
  int rc;
  const int Nxp = round_up(N, 128), Nyp = round_up(N, 16), nfp = round_up(nf, 128);
  const size_t grid_env = (size_t)Nyp * Nxp, t16_env = (size_t)(Nxp / 32) * (nfp / 32) * 2 * 4 * 64 * 8;
  if (!e->focal_grid) {
    // work buffers on first use, for a chunk of whole env tiles: the phase grid (every pixel starts out as "outside the aperture": only
    // aperture pixels are ever written) and T' (split f16, pass 2's operand order)
    const size_t cap = std::max<size_t>(32, (((size_t)256 << 20) / std::max(grid_env * 4, t16_env * 2)) / 32 * 32);
    e->focal_chunk = (int)std::min<size_t>((size_t)e->n_etiles * 32, cap);
    if (const char* v = getenv("AOG_FOCAL_CHUNK")) e->focal_chunk = std::max(32, std::min(e->focal_chunk, atoi(v) / 32 * 32));   // (tests: several chunks at small sizes)
    if ((rc = dev_alloc(e, &e->focal_grid, (size_t)e->focal_chunk * grid_env, false)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->focal_T16, (size_t)e->focal_chunk * t16_env, false)) != AOG_OK) return rc;
    if ((rc = dev_alloc(e, &e->focal_act_ll, (size_t)e->n_etiles * 32 * e->A_pad, true)) != AOG_OK) return rc;
    std::vector<float> fill(grid_env, aog::kShOutside);
    for (int i = 0; i < e->focal_chunk; ++i)
      HIP_TRY(hipMemcpy(e->focal_grid + (size_t)i * grid_env, fill.data(), sizeof(float) * grid_env, hipMemcpyHostToDevice));
  }
  // psi_tile is always current for quasi_static / semi_dynamic handles; dynamic ones refresh it here when the step kernel does not use it
  if ((rc = ensure_tiles(e, s)) != AOG_OK) return rc;
  if (e->cfg.atm_dynamic && !e->ring_direct && e->kernel != AOG_KERNEL_MFMA && (rc = pack_from_master(e, 0, e->B, s)) != AOG_OK) return rc;
  // u = psi + Mt a with the CURRENT mirror state of every env (act16 is rewritten from act_dm: the VALU step kernel does not keep it)
  {
    const int n = e->B * e->A_pad;
    hipLaunchKernelGGL(aog::k_load_actuators, dim3((n + 255) / 256), dim3(256), 0, s, e->act_dm, e->act_rev, e->act16, e->B, e->A, e->A_pad, e->Bp,
                       2.0 / e->cfg.wavelength_wfs, e->focal_act_ll);
  }
  for (int env0 = first / 32 * 32; env0 < first + count; env0 += e->focal_chunk) {
    const int env1 = std::min(first + count, env0 + e->focal_chunk);          // envs [lo, env1) of this chunk are asked for
    const int lo = std::max(first, env0), n_et = (env1 - env0 + 31) / 32;
    aog_host::launch_phase_grid(e, s, e->act16, e->focal_grid, grid_env, Nxp, env0 / 32, n_et);
    const size_t skip = (size_t)(lo - env0);
    hipLaunchKernelGGL(aog::k_focal_pass1, dim3(Nxp / 128, nfp / 128, env1 - lo), dim3(256), 0, s, e->focal_grid + skip * grid_env,
                       reinterpret_cast<const aog::f16x8*>(e->focal_m1s), reinterpret_cast<aog::f16x8*>(e->focal_T16), Nxp, Nyp, nfp);
    hipLaunchKernelGGL(aog::k_focal_pass2, dim3(nfp / 128, nfp / 128, env1 - lo), dim3(256), 0, s, reinterpret_cast<const aog::f16x8*>(e->focal_T16),
                       reinterpret_cast<const aog::f16x8*>(e->focal_m2s), reinterpret_cast<float2*>(field_dev) + (size_t)(lo - first) * nf * nf, Nxp, nfp,
                       nf, e->focal_unscale);
    HIP_TRY(hipGetLastError());
  }
  return AOG_OK;
}

int aog_actor_act(const aog_actor* n, int device, const void* obs_dev, int obs_is_f16, float* mean_dev, float* action_dev, float* log_prob_dev,
                  void* stream) {
  if (!n || !obs_dev) return fail(AOG_ERR_INVALID, "aog_actor_act: null argument");
  if (n->batch < 0 || n->state_dim < 1 || n->hidden_dim < 1 || n->act_dim < 1 || n->hidden_dim > 1024 || n->state_dim > 1024 || n->act_dim > 4096)
    return fail(AOG_ERR_INVALID, "aog_actor_act: bad dimensions (batch %d, state %d, hidden %d, act %d)", n->batch, n->state_dim, n->hidden_dim, n->act_dim);
  if (!n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->w3 || !n->b3 || !n->wo || !n->bo) return fail(AOG_ERR_INVALID, "aog_actor_act: null weight pointer");
  if (((uintptr_t)n->w1 | (uintptr_t)n->w2 | (uintptr_t)n->w3 | (uintptr_t)n->wo) & 15) return fail(AOG_ERR_INVALID, "aog_actor_act: weight matrices must be 16-byte aligned");
  if (!(n->dropout_p >= 0.f && n->dropout_p < 1.f) || !(n->cov_var > 0.f)) return fail(AOG_ERR_INVALID, "aog_actor_act: dropout_p must be in [0,1), cov_var > 0");
  if (n->batch == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(device));
  aog::ActorArgs a{};
  a.obs = obs_dev;
  a.obs_f16 = obs_is_f16 ? 1 : 0;
  a.w1 = n->w1; a.b1 = n->b1; a.w2 = n->w2; a.b2 = n->b2; a.w3 = n->w3; a.b3 = n->b3; a.wo = n->wo; a.bo = n->bo;
  a.mean = mean_dev; a.action = action_dev; a.log_prob = log_prob_dev;
  a.B = n->batch; a.S = n->state_dim; a.H = n->hidden_dim; a.A = n->act_dim;
  a.kpad = round_up(std::max(n->state_dim, n->hidden_dim), 16);
  a.p_drop = n->dropout_p;
  a.keep_scale = 1.0f / (1.0f - n->dropout_p);
  a.std = std::sqrt(n->cov_var);
  a.logp_const = 0.5f * (float)n->act_dim * std::log(2.0f * (float)M_PI * n->cov_var);
  a.seed = n->seed;
  a.call_lo = (uint32_t)n->call_index;
  a.call_hi = (uint32_t)(n->call_index >> 32);
  a.env_base = n->env_id_base;
  const size_t lds = ((size_t)2 * a.kpad * 16 + 16 + (size_t)aog::kActorWFloats) * sizeof(float);
  if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_actor_act), lds, device)) return rc;
  hipLaunchKernelGGL(aog::k_actor_act, dim3((n->batch + 15) / 16), dim3(aog::kActorThreads), lds, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_set_lookahead(aog_env* e, int enable) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_lookahead: null handle");
  if (!e->cfg.atm_dynamic) return fail(AOG_ERR_STATE, "aog_set_lookahead: only dynamic-atmosphere handles evolve their screens inside aog_step");
  HIP_TRY(hipSetDevice(e->device));
  if (enable && !e->ext_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&e->ext_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fused_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_ext_done, hipEventDisableTiming));
  }
  e->lookahead = enable != 0;   // (an extrusion already launched ahead stays valid: the next aog_step joins it)
  return AOG_OK;
}

int aog_selftest_barrier_timeout(aog_env* e, void* stream) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_selftest_barrier_timeout: null handle");
  if (!e->cfg.atm_dynamic || !e->layer_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_selftest_barrier_timeout: needs a dynamic handle with layer and screens");
  if (!e->ext_bar || getenv("AOG_EXTRUDE_SIMPLE") || getenv("AOG_EXTRUDE_NOSPLIT") || ext_split_lds(e) > kLdsBytes)
    return fail(AOG_ERR_UNSUPPORTED, "aog_selftest_barrier_timeout: this handle does not use the split extrusion kernel");
  if (int rcp = refuse_pre_evolved(e, "aog_selftest_barrier_timeout")) return rcp;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  e->ext_spin_limit = 1u << 10;
  e->ext_absent_part = 1;
  e->timestep += 1;
  const int rc = evolve_layer(e, s, e->timestep);
  e->ext_spin_limit = 1u << 24;
  e->ext_absent_part = -1;
  if (rc != AOG_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return AOG_OK;
}

int aog_selftest_poisson(const double* lam_dev, double* out_dev, int n_env, int n, uint64_t seed, uint32_t call, void* stream) {
  if (!lam_dev || !out_dev || n_env < 1 || n < 1) return fail(AOG_ERR_INVALID, "aog_selftest_poisson: bad argument");
  hipLaunchKernelGGL(aog::k_sh_noise, dim3((unsigned)((n * n + 255) / 256), n_env), dim3(256), 0, static_cast<hipStream_t>(stream), lam_dev, out_dev, n,
                     (size_t)0, (unsigned long long)seed, call, 0);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_selftest_sincos(const float* u_dev, float* sin_dev, float* cos_dev, int n, int flavour, void* stream) {
  if (!u_dev || !sin_dev || !cos_dev || n < 0 || flavour < 0 || flavour > 2) return fail(AOG_ERR_INVALID, "aog_selftest_sincos: bad argument");
  if (n == 0) return AOG_OK;
  hipLaunchKernelGGL(aog::k_selftest_sincos, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), u_dev, sin_dev,
                     cos_dev, n, flavour);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

int aog_profile_block(aog_env* e, int launches) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_profile_block: null handle");
  if (launches < 1 || launches > 64) return fail(AOG_ERR_INVALID, "aog_profile_block: %d launches per block (1 .. 64)", launches);
  e->profile_block = launches;
  return AOG_OK;
}

int aog_profile_enable(aog_env* e, int enable) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_profile_enable: null handle");
  e->profile = enable != 0;
  e->profile_every = enable > 1 ? enable : 1;
  e->profile_phase = 0;
  e->events_used = 0;
  if (e->profile && e->events.size() < 1024) {
    // event pairs are created here, not inside the caller's timed region (a few microseconds each; the pool still grows on demand)
    HIP_TRY(hipSetDevice(e->device));
    while (e->events.size() < 1024) {
      hipEvent_t a = nullptr, b = nullptr;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      e->events.emplace_back(a, b);
    }
    // first use of timed events sets up runtime state (milliseconds): do it here
    float ms = 0;
    HIP_TRY(hipEventRecord(e->events[0].first, nullptr));
    HIP_TRY(hipEventRecord(e->events[0].second, nullptr));
    HIP_TRY(hipEventSynchronize(e->events[0].second));
    HIP_TRY(hipEventElapsedTime(&ms, e->events[0].first, e->events[0].second));
  }
  return AOG_OK;
}

int aog_profile_read(aog_env* e, double* mean_ms, int* launches) {
  if (!e || !mean_ms || !launches) return fail(AOG_ERR_INVALID, "aog_profile_read: null argument");
  HIP_TRY(hipSetDevice(e->device));
  for (int k = 0; k < AOG_PROF_COUNT; ++k) {
    e->prof_ms[k] = 0;
    e->prof_n[k] = 0;
  }
  for (size_t i = 0; i < e->events_used; ++i) {
    HIP_TRY(hipEventSynchronize(e->events[i].second));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e->events[i].first, e->events[i].second));
    const int k = i < e->event_kernel.size() ? e->event_kernel[i] : AOG_PROF_FUSED;
    e->prof_ms[k] += ms;
    e->prof_n[k] += 1;
  }
  *launches = e->prof_n[AOG_PROF_FUSED];
  *mean_ms = e->prof_n[AOG_PROF_FUSED] ? e->prof_ms[AOG_PROF_FUSED] / (double)e->prof_n[AOG_PROF_FUSED] : 0.0;
  e->events_used = 0;
  e->profile_phase = 0;   // the next launch opens a timed block: a short measurement after a read still gets its samples
  return AOG_OK;
}

int aog_profile_read_kernel(aog_env* e, int which, double* mean_ms, int* launches) {
  if (!e || !mean_ms || !launches) return fail(AOG_ERR_INVALID, "aog_profile_read_kernel: null argument");
  if (which < 0 || which >= AOG_PROF_COUNT) return fail(AOG_ERR_INVALID, "aog_profile_read_kernel: unknown kernel id %d", which);
  *launches = e->prof_n[which];
  *mean_ms = e->prof_n[which] ? e->prof_ms[which] / (double)e->prof_n[which] : 0.0;
  return AOG_OK;
}

}  // extern "C"
