// libaogym.so — C-ABI (include/aogym.h) over the gfx950 kernels in aogym_kernels.h.
// Host side only: argument checking, table conversion/upload, launch geometry, stream-ordered launches.
#include "host_common.h"
#include "k_pack.h"
#include "k_step.h"

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
#include <type_traits>
#include <vector>

namespace aog_host {

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
}  // namespace aog_host
using namespace aog_host;


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
namespace aog_host {

int dev_alloc_bytes(aog_env* e, void** out, size_t bytes, bool zero) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  if (zero) HIP_TRY(hipMemset(p, 0, bytes));
  e->allocs.push_back(p);
  e->alloc_bytes.push_back(bytes);
  e->dev_bytes += (int64_t)bytes;
  *out = p;
  return AOG_OK;
}

void dev_release_ptr(aog_env* e, void** ptr) {
  if (!*ptr) return;
  for (size_t i = 0; i < e->allocs.size(); ++i)
    if (e->allocs[i] == *ptr) {
      e->dev_bytes -= (int64_t)e->alloc_bytes[i];   // (aog_info.device_bytes stays what the handle owns)
      e->allocs.erase(e->allocs.begin() + (long)i);
      e->alloc_bytes.erase(e->alloc_bytes.begin() + (long)i);
      break;
    }
  (void)hipFree(*ptr);
  *ptr = nullptr;
}

// zero `n_words` 32-bit words at p on stream s with a kernel of the library (see k_zero_words for why not hipMemsetAsync)
void zero_words(void* p, size_t n_words, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<size_t>((n_words + 255) / 256, 4096);
  if (n_words) hipLaunchKernelGGL(aog::k_zero_words, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(p), n_words);
}

// A bounded inter-workgroup wait of an earlier launch timed out (k_extrude16_split): every screen that launch touched is suspect.
// The flag lives in pinned host memory, so this costs one load and no synchronisation; it is seen at the latest by the call after
// the one whose launch tripped it.  Installing fresh screens for the whole batch (aog_set_screens / aog_set_state) clears it.
int check_poisoned(const aog_env* e, const char* who) {
  if (e->host_flag && *static_cast<volatile const int*>(e->host_flag) != 0)
    return fail(AOG_ERR_STATE, "%s: %s; the screens of this handle are invalid (install new screens for the whole batch or restore a saved state)", who,
                (*static_cast<volatile const int*>(e->host_flag) & 2) ? "an earlier aog_step failed after its counters had moved"
                                                                     : "an inter-workgroup wait of the dynamic-atmosphere kernel timed out in an earlier step");
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

int load_actuators(aog_env* e, hipStream_t s, _Float16* act_ll) {
  const int n = e->B * e->A_pad;
  hipLaunchKernelGGL(aog::k_load_actuators, dim3((n + 255) / 256), dim3(256), 0, s, e->act_dm, e->act_rev, e->act16, e->B, e->A, e->A_pad, e->Bp,
                     2.0 / e->cfg.wavelength_wfs, act_ll);
  HIP_TRY(hipGetLastError());
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
}  // namespace aog_host


namespace aog_host {
#define AOG_PHASE_DECL(A)                                                                                                          \
  int launch_phase_apad##A(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile);                                     \
  int launch_phase_field_apad##A(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, bool grid);      \
  int launch_phase_grid_apad##A(aog_env* e, hipStream_t s, const _Float16* act16, const aog::PhaseFieldArgs& fa, int etile0, int n_et);
AOG_PHASE_DECL(16) AOG_PHASE_DECL(32) AOG_PHASE_DECL(64) AOG_PHASE_DECL(128)
#undef AOG_PHASE_DECL
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
    case 16: launch_phase_field_apad16(e, s, act16, fa, grid); break;
    case 32: launch_phase_field_apad32(e, s, act16, fa, grid); break;
    case 64: launch_phase_field_apad64(e, s, act16, fa, grid); break;
    default: launch_phase_field_apad128(e, s, act16, fa, grid); break;
  }
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
    case 16: launch_phase_grid_apad16(e, s, act16, fa, etile0, n_et); break;
    case 32: launch_phase_grid_apad32(e, s, act16, fa, etile0, n_et); break;
    case 64: launch_phase_grid_apad64(e, s, act16, fa, etile0, n_et); break;
    default: launch_phase_grid_apad128(e, s, act16, fa, etile0, n_et); break;
  }
}
void launch_phase(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile) {
  switch (e->A_pad) {
    case 16: launch_phase_apad16(e, s, act16, out_tile); break;
    case 32: launch_phase_apad32(e, s, act16, out_tile); break;
    case 64: launch_phase_apad64(e, s, act16, out_tile); break;
    default: launch_phase_apad128(e, s, act16, out_tile); break;
  }
}
}  // namespace aog_host

namespace {

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

template <typename T>
int set_screens(aog_env* e, const T* psi, int first, int count, hipStream_t s, bool means_ready = false) {
  if (!e || !psi) return fail(AOG_ERR_INVALID, "aog_set_screens: null argument");
  if (!e->tables_ready) return fail(AOG_ERR_STATE, "aog_set_screens before aog_upload_tables");
  if (first < 0 || count < 0 || first + count > e->B)
    return fail(AOG_ERR_INVALID, "aog_set_screens: env range [%d,%d) outside [0,%d)", first, first + count, e->B);
  if (int rc = refuse_pre_evolved(e, "aog_set_screens")) return rc;
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
  if (count == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(e->device));
  const double inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  const int N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  if (e->cfg.atm_dynamic) {
    if (int rcs = std::is_same<T, double>::value ? store_master_f64(e, reinterpret_cast<const double*>(psi), first, count, s)
                                                 : store_master_f32(e, reinterpret_cast<const float*>(psi), first, count, s))
      return rcs;
    int rc = e->ring_direct ? ring_from_master(e, first, count, 0, s) : pack_from_master(e, first, count, s);
    if (rc != AOG_OK) return rc;
  } else if (e->cfg.precision == AOG_PRECISION_FAST && count >= 8) {
    // batches: aperture means, then tiled conversion with whole-line stores (k_pack_tiles)
    if (!e->pack_mean) {
      int rc = dev_alloc(e, &e->pack_mean, (size_t)e->B, false);
      if (rc != AOG_OK) return rc;
    }
    TimedRegion tr(e, s, AOG_PROF_PACK);
    // (means_ready: the synthesis' last pass summed the aperture while it had the screens in registers: pack_mean[0 .. count) is there)
    if (!means_ready) hipLaunchKernelGGL((aog::k_screen_means<T>), dim3(count), dim3(256), 0, s, psi, e->ap_index, e->pack_mean, N2, e->n_ap);
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

}  // namespace

namespace aog_host {
int set_screens_f32(aog_env* e, const float* psi, int first, int count, hipStream_t s, bool means_ready) { return set_screens<float>(e, psi, first, count, s, means_ready); }
}  // namespace aog_host

extern "C" {

int aog_abi_version(void) { return AOG_ABI_VERSION; }

#ifndef AOG_BUILD_ID
#define AOG_BUILD_ID "unidentified"
#endif
// (the marker lets build.py read the id out of the file without loading it)
static const char kBuildIdMarker[] = "AOG_BUILD_ID=" AOG_BUILD_ID;
const char* aog_build_id(void) { return kBuildIdMarker + 13; }

const char* aog_last_error(void) { return g_last_error.c_str(); }

int64_t aog_struct_size(int which) {
  switch (which) {
    case 0: return (int64_t)sizeof(aog_config);
    case 1: return (int64_t)sizeof(aog_tables);
    case 2: return (int64_t)sizeof(aog_layer_tables);
    case 3: return (int64_t)sizeof(aog_sh_tables);
    case 4: return (int64_t)sizeof(aog_actor);
    case 5: return (int64_t)sizeof(aog_info);
    case 6: return (int64_t)sizeof(aog_layer_composite);
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
    // (every table count since round 4: the many-table variants keep their float64 sums in LDS and run chunks as long as o = 2's; eight waves
    // of them need 8 x 2 LIVE x 512 B of LDS — 128 KB at o = 5 — beside the chunk's science rows: one workgroup per CU, which is what this form runs)
    const bool asym = e->kernel == AOG_KERNEL_MFMA && e->n_etiles >= 4 && !getenv("AOG_FUSED_4WAVE");
    e->mfma_waves = asym ? 8 : 4;
    e->mfma_heavy = asym ? 672 : 0;
    const int wp = e->mfma_waves / e->mfma_we;
    const int wg_y = (e->n_etiles + e->mfma_we - 1) / e->mfma_we;
    // P pixel chunks (proportional split of the tiles), 8 waves per CU when the batch allows
    int Pm = cfg->pixel_chunks > 0 ? cfg->pixel_chunks : std::max(1, (asym ? 256 : 256 * 2) / wg_y);
    // (every variant keeps float64 sums now — registers for o = 2, an LDS plane per wave beyond — so a chunk may be long: as long as its
    // science rows (128 B a tile) fit in the LDS beside that plane and the actuator operands; fused_inst.hip lays the same areas out)
    const int live = e->MRW <= 8 ? 0 : (e->MRW <= 16 ? 8 : (e->MRW <= 24 ? 12 : 16));
    const size_t lds_fixed = (size_t)e->mfma_waves * 2 * live * 64 * sizeof(double) +
                             ((e->A_pad > 64 || cfg->atm_dynamic) ? (size_t)e->mfma_waves * (e->A_pad / 16) * 2 * 64 * 16 + (size_t)e->mfma_waves * 32 * 36 * 4 : 0) + 64;
    const int max_tpc = std::min(4096, (int)((aog_host::kLdsBytes - lds_fixed) / 128));
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
  TRY_ALLOC(dev_alloc(e, &e->dev_status, 16 + 4 * 2048));   // (16 status words; the rest: developer read-outs)
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
  if (e->x8_plan_stream) {
    (void)hipStreamSynchronize(e->x8_plan_stream);
    (void)hipStreamDestroy(e->x8_plan_stream);
    (void)hipEventDestroy(e->x8_ev_evolved);
    (void)hipEventDestroy(e->x8_ev_planned);
  }
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
  {
    const int Nw = (e->cfg.n_pupil + 31) / 32;
    std::vector<uint32_t> bits((size_t)e->cfg.n_pupil * Nw, 0u);
    for (int p = 0; p < n_ap; ++p) {
      const int f = t->ap_index[p], iy = f / e->cfg.n_pupil, ix = f % e->cfg.n_pupil;
      bits[(size_t)iy * Nw + (ix >> 5)] |= 1u << (ix & 31);
    }
    if (!e->ap_bits) {
      const int rcb = dev_alloc(e, &e->ap_bits, bits.size(), false);
      if (rcb != AOG_OK) return rcb;
    }
    HIP_TRY(hipMemcpy(e->ap_bits, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice));
  }
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

int aog_set_rng_seed(aog_env* e, uint64_t seed) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_rng_seed: null handle");
  if (int rcp = refuse_pre_evolved(e, "aog_set_rng_seed")) return rcp;   // (the extrusion launched ahead already drew from the old seed)
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
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
    if (int rcu = unroll_master(e, psi_dev, first, count, s)) return rcu;
  } else {
    HIP_TRY(hipMemsetAsync(psi_dev, 0, sizeof(double) * n, s));
    const bool fast = e->cfg.precision == AOG_PRECISION_FAST;
    hipLaunchKernelGGL(aog::k_screens_from_store, dim3((e->n_ap + 255) / 256, count), dim3(256), 0, s, fast ? e->psi_tile : nullptr,
                       fast ? nullptr : e->psi64, e->ap_index, psi_dev, first, e->n_ap, e->n_ptiles, N * N, 2.0 * M_PI * e->cfg.wavelength_wfs);
  }
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
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (likewise what the int8 extrusion prepared ahead)
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
#ifdef AOG_DEV
  if (getenv("AOG_X8_DEV") && (atoi(getenv("AOG_X8_DEV")) & 1024)) {
    fprintf(stderr, "[aogym] k_x8_product: cycles per step max %d min %d, most steps %d, workgroups %d; 10 ns ticks from the first workgroup's start: last start %d, last loop end %d, last end %d; mean start-to-loop-end %d\n", v[8], v[9], v[10], v[11], v[3] - v[2], v[4] - v[2], v[5] - v[2], v[11] ? v[6] / v[11] : 0);
    if (const char* path = getenv("AOG_X8_DEV_DUMP")) {   // per-workgroup records: steps | k << 8 | phase << 12 | xcc << 16 | hw cu/sh/se << 20, cycles per step, start, loop end (10 ns ticks)
      static int rec[4 * 2048];
      HIP_TRY(hipMemcpy(rec, e->dev_status + 16, sizeof rec, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "w")) {
        for (int w = 0; w < v[12] && w < 2048; ++w)
          fprintf(f, "%d %d %d %d %d %d %d %d\n", rec[4 * w] & 255, (rec[4 * w] >> 8) & 15, (rec[4 * w] >> 12) & 1, (rec[4 * w] >> 16) & 15, (rec[4 * w] >> 20) & 255, rec[4 * w + 1], rec[4 * w + 2] - v[2], rec[4 * w + 3] - v[2]);
        fclose(f);
      }
    }
    const int z6[6] = {1 << 30, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(e->dev_status + 2, z6, sizeof z6, hipMemcpyHostToDevice));
    const int init[8] = {0, 1 << 30, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpy(e->dev_status + 8, init, sizeof init, hipMemcpyHostToDevice));
  }
#endif
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
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
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
static int step_body(aog_env* e, const float* action, const float* action_next, bool pipelined, float* obs_raw, uint16_t* obs, float* reward,
                     uint8_t* done, float* power, float* strehl, void* stream, bool* mutated) {
  if (!e || !action) return fail(AOG_ERR_INVALID, "aog_step: null argument");
  if (!pipelined && e->pro_pending)
    return fail(AOG_ERR_STATE, "aog_step: a pipelined step has already loaded the next action (continue with aog_step_pipelined)");
  if (pipelined && e->lookahead) return fail(AOG_ERR_UNSUPPORTED, "aog_step_pipelined: not together with aog_set_lookahead");
  if (!e->tables_ready || !e->screens_ready) return fail(AOG_ERR_STATE, "aog_step before aog_upload_tables/aog_set_screens");
  if (int rc = check_poisoned(e, "aog_step")) return rc;
  if (e->cfg.reward_type == AOG_REWARD_SMF_SSIM && e->n_obs < 7)
    return fail(AOG_ERR_INVALID, "win_size exceeds image extent (smf_ssim needs obs_dim**2 >= 7; AO_env.py:495)");
  if (e->cfg.atm_dynamic && e->pre_evolved && e->next_noise)
    return fail(AOG_ERR_STATE, "aog_step: extrusion normals were supplied for a step whose extrusion already ran (lookahead draws from the device "
                "stream; switch it off for host-supplied normals)");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  *mutated = true;   // from here on a failure leaves counters, ring and mirror out of step with each other: step_impl poisons the handle
  e->timestep += 1;  // AO_env.py:123
  e->steps_since_reset += 1;
  bool join_ext = false;
  e->sh_sums_ready = false;   // (an aog_sh_image(NULL) not followed by its aog_sh_update is void once the env has stepped)
  if (e->cfg.atm_dynamic) {
    if (e->pre_evolved) {   // the previous step launched this step's extrusion on the library's stream: join it
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

static int step_impl(aog_env* e, const float* action, const float* action_next, bool pipelined, float* obs_raw, uint16_t* obs, float* reward,
                     uint8_t* done, float* power, float* strehl, void* stream) {
  bool mutated = false;
  const int rc = step_body(e, action, action_next, pipelined, obs_raw, obs, reward, done, power, strehl, stream, &mutated);
  // A launch or a dynamic-LDS request that fails AFTER the step counters moved (and perhaps after the next extrusion was queued) leaves the
  // handle's counters, screens and mirror inconsistent: mark it unusable (bit 1 of the status word; cleared like a barrier timeout, by
  // installing screens for the whole batch or restoring a saved state) instead of letting later steps run on it.
  if (rc != AOG_OK && mutated && e && e->host_flag) *static_cast<volatile int*>(e->host_flag) |= 2;
  return rc;
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
