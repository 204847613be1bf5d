// Dynamic atmosphere (AO_env.py:125): layer tables, wind, the per-step extrusion launch, master / ring maintenance.
#include "host_common.h"
#include "k_pack.h"
#include "k_extrude.h"
#include "k_extrude_i8.h"

using namespace aog_host;

// ---- int8 composite extrusion: host side -----------------------------------------------------------------------------------------------
namespace {
struct X8Host {
  aog::X8Table tab[2][aog::kX8MaxK + 1] = {};
};

void x8_digits_host(long long x, int nd, int8_t* d) {   // balanced base-128 digits, most significant first (as the device's x8_digits)
  for (int t = nd - 1; t > 0; --t) {
    const int dg = (int)((x + 64) & 127) - 64;
    d[t] = (int8_t)dg;
    x = (x - dg) >> 7;
  }
  d[0] = (int8_t)x;
}

// The step's whole-pixel shifts never exceed floor(max wind component x delta_t / pitch) + 1 per axis (difference of two roundings)
int x8_needed_k(const aog_env* e) { return (int)std::floor(e->max_wind * e->delta_t / e->pitch) + 1; }

bool x8_usable(const aog_env* e) {
  if (e->ext_mode == AOG_EXTRUDE_F64 || getenv("AOG_EXTRUDE_F64") || !e->x8_host) return false;
  const int kcap = std::min(e->x8_kmax[0], e->x8_kmax[1]);
  return kcap >= 1 && x8_needed_k(e) <= kcap;
}

int x8_ensure_buffers(aog_env* e) {
  if (e->x8_Z8) return AOG_OK;
  const X8Host* h = static_cast<const X8Host*>(e->x8_host);
  const int kcap = std::min(e->x8_kmax[0], e->x8_kmax[1]);
  int ks_max = 0, rt_max = 0;
  for (int a = 0; a < 2; ++a)
    for (int k = 1; k <= kcap; ++k) {
      ks_max = std::max(ks_max, h->tab[a][k].KsAmax + h->tab[a][k].KsB);   // (the normals' steps sit behind the K-shift stencil's)
      rt_max = std::max(rt_max, h->tab[a][k].RT);
    }
  e->x8_tiles64_max = (e->B + 63) / 64;   // envs are packed densely, most shifts first
  e->x8_slots_max = e->x8_tiles64_max * 64;
  e->x8_KsTot_max = ks_max;   // (one 16-sample chunk per thread of k_x8_prepare: 2 KsTot chunks per env)
  e->x8_rt_max = rt_max;
  int rc;
  if ((rc = dev_alloc(e, &e->x8_dxy, (size_t)2 * e->B)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_slot, (size_t)2 * e->B)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_list, (size_t)2 * e->x8_slots_max)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_tile_k, (size_t)2 * e->x8_tiles64_max)) != AOG_OK) return rc;
  // k_x8_plan deals groups of at most 4 row pairs x chunk tiles to the least loaded XCD: no queue is longer than the mean + one group
  e->x8_items_max = 8 * ((e->x8_tiles64_max * ((rt_max + 1) / 2) + 7) / 8 + 4 * aog::x8_chunk_tiles(e->x8_tiles64_max));
  if ((rc = dev_alloc(e, &e->x8_items, (size_t)2 * e->x8_items_max)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_rec, (size_t)4 * e->x8_slots_max)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_colbuf, (size_t)e->x8_slots_max * kcap * h->tab[1][1].Np)) != AOG_OK) return rc;
  if ((rc = dev_alloc(e, &e->x8_Z8, ((size_t)2 * e->x8_tiles64_max * ks_max + aog::kX8PadSteps) * 5 * 1024)) != AOG_OK) return rc;   // (zeroed: unused columns hold zeros, not junk)
  return AOG_OK;
}

// one step of every env's wind shifts as plan -> (prepare, product) x 2
int x8_evolve(aog_env* e, hipStream_t s, long long step_index) {
  if (int rc = x8_ensure_buffers(e)) return rc;
  aog::X8Args p{};
  p.tables = static_cast<const aog::X8Table*>(e->x8_tables_dev);
  p.master = e->psi_master;
  p.ring = e->ring_direct ? e->psi_ring : nullptr;
  p.ring_ref = e->psi_offset;
  p.ring_inv = 1.0 / (2.0 * M_PI * e->cfg.wavelength_wfs);
  p.origin = e->origin;
  p.ext_counter = e->ext_counter;
  p.velocity = e->velocity;
  p.noise = e->next_noise;
  p.max_ext = e->next_noise_max_ext;
  p.N = e->cfg.n_pupil;
  p.B = e->B;
  p.kcap = std::min(e->x8_kmax[0], e->x8_kmax[1]);
  p.t_prev = (double)(step_index - 1) * e->delta_t;
  p.t_new = (double)step_index * e->delta_t;
  p.pitch = e->pitch;
  p.seed = e->rng_seed;
  p.env_base = e->cfg.env_id_base;
  p.dxy = e->x8_dxy;
  p.slot = e->x8_slot;
  p.list = e->x8_list;
  p.tile_k = e->x8_tile_k;
  p.items = e->x8_items;
  p.items_max = e->x8_items_max;
  p.tiles64_max = e->x8_tiles64_max;
  p.slots_max = e->x8_slots_max;
  p.Z8 = e->x8_Z8;
  p.KsTot_max = e->x8_KsTot_max;
  p.rec = e->x8_rec;
  p.colbuf = e->x8_colbuf;
  p.status = e->dev_status;
#ifdef AOG_DEV
  if (const char* v = getenv("AOG_X8_DEV")) p.dev = atoi(v);
#endif
  // What step t + 1 needs that step t's fused kernel, epilogue and the caller's policy query do not touch runs beside them on a stream of
  // its own (lowest priority): the PLAN (clock and winds only) and the whole x PHASE (prepare + product of phase 0: it reads the screens as
  // step t's extrusion left them and writes operands and the staged columns — the screens themselves first change in phase 1).  Used only
  // if it was made for this very step and state; anything that changes the state it read drops it first (x8_drop_ahead), host-supplied
  // normals arriving for a step whose x phase drew from the device stream redo that phase (nothing of it was committed).
  const bool ahead = !getenv("AOG_X8_NO_PLAN_AHEAD");
  int have = 0;   // 1: the plan is there, 2: and phase 0
  if (e->x8_plan_step >= 0) {
    HIP_TRY(hipStreamWaitEvent(s, e->x8_ev_planned, 0));   // (-1: none made; -2: dropped after a stream synchronise)
    if (e->x8_plan_step == step_index) have = e->x8_ahead_level;
  }
  if (have == 2 && e->next_noise) have = 1;
  e->x8_plan_step = -1;
  const dim3 gprep(e->B), bprep(round_up(2 * e->x8_KsTot_max, 64)), gprod(e->x8_items_max);
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_x8_product), aog::kX8ProductLds, e->device)) return rc;
  if (have < 1) hipLaunchKernelGGL(aog::k_x8_plan, dim3(1), dim3(aog::kX8PlanThreads), 0, s, p);
  for (int phase = have < 2 ? 0 : 1; phase < 2; ++phase) {
    hipLaunchKernelGGL(aog::k_x8_prepare, gprep, bprep, 0, s, p, phase);
    hipLaunchKernelGGL(aog::k_x8_product, gprod, dim3(512), aog::kX8ProductLds, s, p, phase);
  }
  HIP_TRY(hipGetLastError());
  if (ahead) {
    if (!e->x8_plan_stream) {
      int least = 0, greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_TRY(hipStreamCreateWithPriority(&e->x8_plan_stream, hipStreamNonBlocking, least));
      HIP_TRY(hipEventCreateWithFlags(&e->x8_ev_evolved, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->x8_ev_planned, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(e->x8_ev_evolved, s));   // (this step's kernels read the plan's arrays and write the screens)
    HIP_TRY(hipStreamWaitEvent(e->x8_plan_stream, e->x8_ev_evolved, 0));
    p.t_prev = (double)step_index * e->delta_t;
    p.t_new = (double)(step_index + 1) * e->delta_t;
    hipLaunchKernelGGL(aog::k_x8_plan, dim3(1), dim3(aog::kX8PlanThreads), 0, e->x8_plan_stream, p);
    e->x8_ahead_level = 1;
    // (not the x phase after an episode's last step — a reset follows — nor while the caller supplies the normals)
    if (!getenv("AOG_X8_NO_PHASE_AHEAD") && !e->next_noise && (e->cfg.max_steps <= 0 || e->steps_since_reset < e->cfg.max_steps)) {
      hipLaunchKernelGGL(aog::k_x8_prepare, gprep, bprep, 0, e->x8_plan_stream, p, 0);
      hipLaunchKernelGGL(aog::k_x8_product, gprod, dim3(512), aog::kX8ProductLds, e->x8_plan_stream, p, 0);
      e->x8_ahead_level = 2;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->x8_ev_planned, e->x8_plan_stream));
    e->x8_plan_step = step_index + 1;
  }
  return AOG_OK;
}
}  // namespace

// a plan made ahead read the winds and the clock as they were: whoever changes either (aog_set_wind, ...) calls this first
int aog_host::x8_drop_ahead(aog_env* e) {
  if (e->x8_plan_stream) HIP_TRY(hipStreamSynchronize(e->x8_plan_stream));
  if (e->x8_plan_step >= 0) e->x8_plan_step = -2;   // (made, finished, not to be used: no wait needed either — but the event exists; -2 keeps the branch above simple)
  return AOG_OK;
}

namespace aog_host {
// float64 ring-buffer master screens of envs [first, first+count) -> the fused kernels' fp32 layouts
int pack_from_master(aog_env* e, int first, int count, hipStream_t s, bool per_step) {
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
  if (x8_usable(e)) {
    // int8 matrix-core form: the step's x shifts, then its y shifts, each as one exact fixed-point product (k_extrude_i8.h)
    if (int rc = x8_evolve(e, s, step_index)) return rc;
  } else if (e->ext_bar && !getenv("AOG_EXTRUDE_SIMPLE") && !getenv("AOG_EXTRUDE_NOSPLIT") && ext_split_lds(e) <= kLdsBytes) {
    // float64 matrix-core form with each 16-env group's rows split over four workgroups + group barrier.  Since round 4 this is the
    // VALIDATION form (the int8 composite form above is the fast one), so its barrier runs the full agent-scope release / acquire protocol
    // of MI355X_MICROARCH.md in every round; the same-XCD short form of round 3 (writer without the L2 write-back when the four partners
    // measured that they share an XCD: 30 us per step faster, one intermittent mismatch in its history) is opt-in: AOG_EXTRUDE_SAME_XCD=1
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
                         e->host_flag_dev, g0, e->ext_spin_limit, e->ext_absent_part, bar_next, getenv("AOG_EXTRUDE_SAME_XCD") ? 0 : 1);
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

template <typename T>
static int store_master_t(aog_env* e, const T* psi, int first, int count, hipStream_t s) {
  const int N2 = e->cfg.n_pupil * e->cfg.n_pupil;
  const size_t n = (size_t)count * N2;
  hipLaunchKernelGGL((aog::k_store_master<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, psi, e->psi_master, e->origin,
                     e->ext_counter, first, count, N2);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}
int store_master_f64(aog_env* e, const double* psi, int first, int count, hipStream_t s) { return store_master_t(e, psi, first, count, s); }
int store_master_f32(aog_env* e, const float* psi, int first, int count, hipStream_t s) { return store_master_t(e, psi, first, count, s); }

int unroll_master(aog_env* e, double* psi_dev, int first, int count, hipStream_t s) {
  const int N = e->cfg.n_pupil;
  const size_t n = (size_t)count * N * N;
  hipLaunchKernelGGL(aog::k_unroll_master, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, e->psi_master, e->origin, psi_dev, first, count, N);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}
}  // namespace aog_host

extern "C" {

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
  if (int rcp = refuse_pre_evolved(e, "aog_upload_layer")) return rcp;
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
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

int aog_upload_layer_composite(aog_env* e, const aog_layer_composite* t) {
  if (!e || !t) return fail(AOG_ERR_INVALID, "aog_upload_layer_composite: null argument");
  if (!e->cfg.atm_dynamic || !e->layer_ready) return fail(AOG_ERR_STATE, "aog_upload_layer_composite: needs a dynamic handle after aog_upload_layer");
  if (t->axis < 0 || t->axis > 1 || t->k_max < 1 || t->k_max > aog::kX8MaxK || t->n_old < 1 || !t->old_yx || !t->A || !t->B)
    return fail(AOG_ERR_INVALID, "aog_upload_layer_composite: bad argument (axis %d, k_max %d of at most %d, n_old %d)", t->axis, t->k_max, aog::kX8MaxK, t->n_old);
  if (e->x8_kmax[t->axis]) return fail(AOG_ERR_STATE, "aog_upload_layer_composite: axis %d already uploaded", t->axis);
  if (int rcp = refuse_pre_evolved(e, "aog_upload_layer_composite")) return rcp;
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
  const int N = e->cfg.n_pupil, K = t->k_max, U = t->n_old, Np = round_up(N, 64);
  if (round_up(U, 32) + K * Np > aog::kX8PrepMaxThreads * 16)
    return fail(AOG_ERR_UNSUPPORTED, "aog_upload_layer_composite: %d stencil samples + %d normals per step (at most %d together)", U, K * Np, aog::kX8PrepMaxThreads * 16);
  for (int c = 0; c < U; ++c)
    if ((t->old_yx[c] >> 16) < 0 || (t->old_yx[c] >> 16) >= N || (t->old_yx[c] & 0xffff) >= N) return fail(AOG_ERR_INVALID, "aog_upload_layer_composite: old_yx[%d] outside the screen", c);
  HIP_TRY(hipSetDevice(e->device));
  if (!e->x8_host) e->x8_host = new X8Host();
  X8Host* h = static_cast<X8Host*>(e->x8_host);
  const bool vertical = t->axis == 0;
  const double mid = 0.5 * (double)(N - 1);
  // the first shift whose rows touch each union column (exact zeros elsewhere: compose_extrusions never writes them); columns in that order:
  // the rows of shift j — and an env that shifts k >= j times — need only the first U_j columns
  std::vector<int> first_use((size_t)U, K + 1);
  for (int j = 1; j <= K; ++j)
    for (int i = 0; i < N; ++i) {
      const double* row = t->A + (size_t)((j - 1) * N + i) * U;
      for (int c = 0; c < U; ++c)
        if (row[c] != 0.0 && first_use[c] > j) first_use[c] = j;
    }
  std::vector<int> cols;
  std::vector<int> Uk((size_t)K + 1, 0);
  for (int j = 1; j <= K; ++j) {
    for (int c = 0; c < U; ++c)
      if (first_use[c] == j) cols.push_back(c);
    Uk[j] = (int)cols.size();
  }
  const int UK = Uk[K];
  if (Uk[1] < 2) return fail(AOG_ERR_INVALID, "aog_upload_layer_composite: the first shift reads %d columns", Uk[1]);
  int rc;
  auto up = [&](auto** dst, const auto* src, size_t count) -> int {
    if ((rc = dev_alloc(e, dst, count, false)) != AOG_OK) return rc;
    HIP_TRY(hipMemcpy(*dst, src, sizeof(**dst) * count, hipMemcpyHostToDevice));
    return AOG_OK;
  };
  // ONE table per axis: the rows of shift j of the K-shift operator are the rows of shift j of every k-shift operator, k >= j (slice j depends
  // on the old screen and on the normals of shifts 1 .. j only), so envs of every shift count share it
  const int KsAK = (UK + 31) / 32, KsBK = K * Np / 32, RT = K * Np / 32, KsT = KsAK + KsBK + aog::kX8PadSteps;
  double amax = 0.0, bmax = 0.0;
  for (int r = 0; r < K * N; ++r) {
    for (int c : cols) amax = std::max(amax, std::fabs(t->A[(size_t)r * U + c]));
    for (int c = 0; c < K * N; ++c) bmax = std::max(bmax, std::fabs(t->B[(size_t)r * K * N + c]) * e->sqrt_cn2);
  }
  if (!(amax > 0.0) || !(bmax > 0.0) || !std::isfinite(amax) || !std::isfinite(bmax)) return fail(AOG_ERR_INVALID, "aog_upload_layer_composite: empty or non-finite operator");
  const int log2_qa = std::ilogb(amax) + 1 - 34, log2_qb = std::ilogb(bmax) + 1 - 34;   // |A| / qa, |B| / qb < 2^34 (5 digits each)
  std::vector<int32_t> yx((size_t)KsAK * 32);
  std::vector<int8_t> T8((size_t)RT * KsT * 5 * 1024, 0);
  std::vector<double> r1((size_t)RT * 32, 0.0), r2((size_t)RT * 32, 0.0);
  std::vector<double> xa((size_t)UK);
  for (int cc = 0; cc < UK; ++cc) {
    const int32_t q = t->old_yx[cols[cc]];
    xa[cc] = (double)(vertical ? (q & 0xffff) : (q >> 16)) - mid;
  }
  for (int cc = 0; cc < KsAK * 32; ++cc) yx[cc] = t->old_yx[cols[cc < UK ? cc : 0]];
  for (int j = 1; j <= K; ++j)
    for (int i = 0; i < N; ++i) {
      const int row = (j - 1) * Np + i, rt = row >> 5;
      const double* arow = t->A + (size_t)((j - 1) * N + i) * U;
      long double s1 = 0.0L, s2 = 0.0L;
      for (int cc = 0; cc < Uk[j]; ++cc) {
        const double a = arow[cols[cc]];
        s1 += a;
        s2 += (long double)a * xa[cc];
        int8_t d[5];
        x8_digits_host(std::llrint(std::ldexp(a, -log2_qa)), 5, d);
        const size_t base = (((size_t)rt * KsT + (cc >> 5)) * 5) * 1024 + (size_t)((row & 31) + 32 * ((cc & 31) >> 4)) * 16 + (cc & 15);
        for (int dg = 0; dg < 5; ++dg) T8[base + (size_t)dg * 1024] = d[dg];
      }
      r1[row] = (double)s1;
      r2[row] = (double)s2;
      const double* brow = t->B + (size_t)((j - 1) * N + i) * K * N;
      for (int jj = 1; jj <= j; ++jj)
        for (int ii = 0; ii < N; ++ii) {
          const double b = brow[(size_t)(jj - 1) * N + ii] * e->sqrt_cn2;
          if (b == 0.0) continue;
          const int cc = (jj - 1) * Np + ii;
          int8_t d[5];
          x8_digits_host(std::llrint(std::ldexp(b, -log2_qb)), 5, d);
          const size_t base = (((size_t)rt * KsT + KsAK + (cc >> 5)) * 5) * 1024 + (size_t)((row & 31) + 32 * ((cc & 31) >> 4)) * 16 + (cc & 15);
          for (int dg = 0; dg < 5; ++dg) T8[base + (size_t)dg * 1024] = d[dg];
        }
    }
  int32_t* d_yx = nullptr;
  int8_t* d_T8 = nullptr;
  double *d_r1 = nullptr, *d_r2 = nullptr;
  if ((rc = up(&d_yx, yx.data(), yx.size())) != AOG_OK) return rc;
  if ((rc = up(&d_T8, T8.data(), T8.size())) != AOG_OK) return rc;
  if ((rc = up(&d_r1, r1.data(), r1.size())) != AOG_OK) return rc;
  if ((rc = up(&d_r2, r2.data(), r2.size())) != AOG_OK) return rc;
  double sx = 0.0, sxx = 0.0;
  for (int k = 1, cc = 0; k <= K; ++k) {   // per shift count: how much of the table an env of that count (prepare) or a row of that shift (product) uses
    for (; cc < Uk[k]; ++cc) {
      sx += xa[cc];
      sxx += xa[cc] * xa[cc];
    }
    aog::X8Table& tb = h->tab[t->axis][k];
    tb = aog::X8Table{};
    tb.yx = d_yx; tb.T8 = d_T8; tb.r1 = d_r1; tb.r2 = d_r2;
    tb.k = k; tb.U = Uk[k]; tb.KsA = (Uk[k] + 31) / 32; tb.KsB = k * Np / 32; tb.KsAmax = KsAK; tb.KsT = KsT; tb.RT = k * Np / 32; tb.Np = Np;
    tb.log2_qa = log2_qa;
    tb.log2_cn = log2_qa - log2_qb;
    tb.ez_floor = 3 - tb.log2_cn;
    tb.sx = sx;
    tb.sxx = sxx;
  }
  e->x8_kmax[t->axis] = K;
  if (!e->x8_tables_dev) {
    void* dp = nullptr;
    if ((rc = dev_alloc_bytes(e, &dp, sizeof h->tab, true)) != AOG_OK) return rc;
    e->x8_tables_dev = dp;
  }
  HIP_TRY(hipMemcpy(e->x8_tables_dev, h->tab, sizeof h->tab, hipMemcpyHostToDevice));
  return AOG_OK;
}

int aog_set_extrusion_mode(aog_env* e, int mode) {
  if (!e) return fail(AOG_ERR_INVALID, "aog_set_extrusion_mode: null handle");
  if (mode != AOG_EXTRUDE_AUTO && mode != AOG_EXTRUDE_F64) return fail(AOG_ERR_INVALID, "aog_set_extrusion_mode: unknown mode %d", mode);
  if (int rcp = refuse_pre_evolved(e, "aog_set_extrusion_mode")) return rcp;
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
  e->ext_mode = mode;
  return AOG_OK;
}

int aog_set_wind(aog_env* e, const double* velocity_dev, double max_abs_component, void* stream) {
  if (!e || !velocity_dev || !(max_abs_component >= 0)) return fail(AOG_ERR_INVALID, "aog_set_wind: bad argument");
  e->max_wind = max_abs_component;
  if (!e->cfg.atm_dynamic) return fail(AOG_ERR_STATE, "aog_set_wind: handle was not created with atm_dynamic = 1");
  if (int rcp = refuse_pre_evolved(e, "aog_set_wind")) return rcp;   // (an extrusion launched ahead may still be reading the old wind)
  HIP_TRY(hipSetDevice(e->device));
  if (int rcd = x8_drop_ahead(e)) return rcd;
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
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step read the state this call changes)
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  e->ext_spin_limit = 1u << 10;
  e->ext_absent_part = 1;
  e->timestep += 1;
  const int mode = e->ext_mode;
  e->ext_mode = AOG_EXTRUDE_F64;   // (the barrier lives in the float64 round kernel: the int8 form has none)
  const int rc = evolve_layer(e, s, e->timestep);
  e->ext_mode = mode;
  e->ext_spin_limit = 1u << 24;
  e->ext_absent_part = -1;
  if (rc != AOG_OK) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return AOG_OK;
}

}  // extern "C"
