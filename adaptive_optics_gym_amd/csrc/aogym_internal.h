// Internal (non-ABI) declarations shared by the translation units of libaogym.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

#include "../../include/aogym.h"

struct aog_env {
  aog_config cfg{};
  int device = 0;
  bool tables_ready = false;
  bool screens_ready = false;
  int B = 0, Bp = 0, A = 0, A_pad = 0, n_ap = 0, n_ap_pad = 0, n_quads = 0, n_ptiles = 0, n_etiles = 0;
  int MRW = 0, MRS = 0;          // padded table counts of the fast kernels
  int MRW_used = 0, MRS_used = 0;
  int n_obs = 0, n_out = 0;
  int kernel = AOG_KERNEL_VALU;  // resolved
  int sincos_hw = 0;
  // launch geometry
  int valu_chunks = 0, valu_qpc = 0;
  int mfma_we = 1, mfma_chunks_x = 0, mfma_tpc = 0;
  int mfma_waves = 4;            // waves per workgroup of k_fused_tab (8 with asymmetric pairs)
  int mfma_heavy = 0;            // asymmetric wave pairs: share (x / 1024) of a chunk's tiles that the prioritised sub-chunk takes; 0 = off
  int n_chunks = 0;              // partial slabs the epilogue sums
  int64_t dev_bytes = 0;
  // constant tables
  int32_t* ap_index = nullptr;
  uint32_t* ap_bits = nullptr;   // [N][ceil(N / 32)] the aperture as a bit mask (bit x & 31 of word x >> 5 of row y): k_screen2_cols' aperture sums
  double* syn_part = nullptr;    // [synthesis batch][column tiles] aperture sums of the screens just drawn, per column tile (k_screen2_cols -> k_mean_from_parts)
  size_t syn_part_elems = 0;
  float* modes_f32 = nullptr;    // [n_ap_pad][A_pad]
  _Float16* modes16 = nullptr;   // [n_ptiles][A_pad/16][hi|lo][64][8]
  float* tabs_f32 = nullptr;     // [n_ap_pad][TROW]
  _Float16* tab16 = nullptr;     // table-MFMA form: [n_ptiles][step 2][hi|lo][lane 64][8] A operands of the wfs tables
  float* sci_tile = nullptr;     // [n_ptiles][h 2][16] science table in accumulator order
  double* gram = nullptr;        // [A][A]
  double* wfs_coef = nullptr;    // [n_out][MRW_used][2]
  double* sci_coef = nullptr;    // [MRS_used][2]
  double* modes64 = nullptr;     // validation: [n_ap][A]
  double* tabs64 = nullptr;      // validation: [n_ap][MRW_used+MRS_used]
  // Shack-Hartmann chain (K10)
  bool sh_ready = false;
  int sh_n_sub = 0;
  int32_t* sh_slot = nullptr;
  double* sh_centres = nullptr;
  double* sh_ref = nullptr;
  double* sh_recon = nullptr;
  double* sh_mla = nullptr;       // [N*N] complex
  double* sh_tf = nullptr;        // [2N][2N] complex
  double* sh_xdet = nullptr;
  double* sh_act = nullptr;       // [B][A] deformable_mirror_shack.actuators
  _Float16* sh_act16 = nullptr;   // same, B-operand layout
  float* sh_phase = nullptr;      // psi_tile layout: wfs phase (rev) through the shack mirror
  void* sh_pad = nullptr;         // [B][2N][2N] complex work buffer (complex64, or complex128 when sh_double)
  void* sh_in = nullptr;          // [B][2N][2N] zero-padded forward input (padding never written)
  float* sh_tf32 = nullptr;       // [2N][2N] complex64 copy of the transfer function
  bool sh_double = false;         // complex128 transforms (aog_sh_tables.fft_double)
  double* sh_image = nullptr;     // [B][N*N]
  double* sh_noisy = nullptr;     // [B][N*N]
  void* sh_plan = nullptr;        // hipfftHandle (Z2Z, batch B)
  int sh_pruned = 0;              // L / 64 (4, 8, 16) when the pruned three-pass propagation is used (complex64, N = 128 / 256 / 512); 0 = hipFFT 2-D
  float* sh_tw = nullptr;         // [L] complex64 e^{+2 pi i j / L}
  int32_t* sh_ap_yx = nullptr;    // [n_ap] iy << 16 | ix of aperture pixel p
  float* sh_mla32 = nullptr;      // [N*N] complex64 micro-lens phase factor
  float* sh_ftab = nullptr;       // [n_ap] argument of the micro-lens factor in revolutions per packed aperture pixel (pruned route)
  double* sh_sums = nullptr;      // [B][n_sub][3] noisy per-lenslet sums of the fused row pass (aog_sh_image without an image pointer)
  bool sh_sums_ready = false;     // set by that call, consumed by the next aog_sh_update(null)
  float* sh_tfq = nullptr;        // [L / BC][64][64] complex64 transfer function in the column pass's lane / register order
  int sh_sep_rl = 0;              // RL of the separable two-pass propagation (transfer function = hx(kx) hy(ky)); 0 = three-pass form
  float* sh_hxq = nullptr;        // [LW][64] complex64 hx[lane / BC + RL k2]   (pass 1, layout B)
  float* sh_hyq = nullptr;        // [RL][64] complex64 hy[lane + LW r]         (pass 2, layout A)
  double sh_amp = 0, sh_scale = 0, sh_gain = 0, sh_leak = 0;
  uint32_t sh_calls = 0;
  // device screen synthesis (K8)
  void* fft_plan = nullptr;      // hipfftHandle
  int fft_m = 0, fft_batch = 0;
  float* fft_work = nullptr;     // [fft_batch][m][m] complex64
  float* fft_crop = nullptr;     // [fft_batch][N][N]
  float* syn_T = nullptr;        // pruned synthesis: [syn_batch][m][N] complex64 (lines after pass A)
  float* syn_out = nullptr;      // [syn_batch][N][N]
  int syn_batch = 0, syn_m = 0;     // syn_m: q N of the literal layout, -(q N) of the two-band layout
  int screen_method = AOG_SCREENS_TWOBAND;
  float* low_c = nullptr;        // general route of the two-band form: low-band spectrum [fft_batch][KL][2 KL] complex64
  float* low_T = nullptr;        // and its lines [fft_batch][KL][N] complex64
  int low_key = 0;
  uint32_t* screen_gen = nullptr;   // [B] screens synthesised so far per env (Philox stream position of k_screen_rows / k_spectrum_fill)
  // focal-image export (optional)
  int n_focal = 0;
  double* focal_m1 = nullptr;    // [n_focal][N] complex
  double* focal_m2 = nullptr;    // [N][n_focal] complex
  double* focal_E = nullptr;     // [N][N] complex scratch
  double* focal_T = nullptr;     // [n_focal][N] complex scratch
  _Float16* focal_m1s = nullptr; // split-f16 operand tiles of the batched matrix-core path (aog_focal_images): m1 2^e1, [v block][k-step][4][64][8]
  _Float16* focal_m2s = nullptr; // m2 2^e2, [u block][x tile][2][4][64][8] (x in the order pass 1's accumulators hold it)
  float focal_unscale = 1.f;     // 2^-(e1 + e2)
  int32_t* focal_ap_yx = nullptr;  // [n_ap] iy << 16 | ix of aperture pixel p
  float* focal_grid = nullptr;   // [focal_chunk][Nyp][Nxp] reduced phases (revolutions), kShOutside outside the aperture
  _Float16* focal_act_ll = nullptr;  // [n_etiles][A_pad / 16][64][8] third f16 term of the actuators (K4 phases)
  _Float16* focal_T16 = nullptr; // [focal_chunk][Nxp / 32][nfp / 32][2][4][64][8]: T' = m1' E, split, pass 2's operand order
  int focal_chunk = 0;
  // state
  float* psi_rev = nullptr;      // [n_quads][Bp][4]  (handles that run the VALU kernel only)
  double* pack_mean = nullptr;   // [B] aperture means of the screens being installed (k_screen_means -> k_pack_tiles)
  float* psi_tile = nullptr;     // [Bp/32][n_ptiles][4][64][4]
  double* psi64 = nullptr;       // validation: [B][n_ap]
  double* act_dm = nullptr;      // [B][A]
  float* act_rev = nullptr;      // [A_pad][Bp]
  _Float16* act16 = nullptr;     // [Bp/32][A_pad/16][hi|lo][64][8]
  int32_t* t_render = nullptr;   // [B]
  // dynamic atmosphere (cfg.atm_dynamic)
  bool layer_ready = false;
  // ring-direct form (fast MFMA handles): the fused kernel reads the fp32 ring copy of the master screens itself, no per-step repack
  bool ring_direct = false;
  bool tiles_stale = false;      // psi_tile (used by the focal-field / Shack-Hartmann / phase-screen paths) is older than the master screens
  float* psi_ring = nullptr;     // [B][N][N + 4] fp32, see aog::DynPsi
  uint32_t* quad_desc = nullptr; // [n_ptiles * 2][4]
  uint32_t* quad_cont = nullptr; // [n_ptiles * 2][4]
  unsigned* ext_bar = nullptr;   // group-barrier tickets of k_extrude16_split: two sets that alternate between steps (each launch zeroes the other set)
  int ext_bar_phase = 0;
  int* dev_status = nullptr;     // sticky device-side error word (1 = a bounded spin timed out)
  int* host_flag = nullptr;      // the same flag in pinned, device-mapped host memory: read by the host without a synchronisation
  int* host_flag_dev = nullptr;  // its device address
  int n_ext_groups = 0;
  int ext_resident = 0;          // workgroups of k_extrude16_split one launch may hold (occupancy query x CUs; 0 = not asked yet)
  unsigned ext_spin_limit = 1u << 24;   // polls before a barrier wait gives up (seconds)
  int ext_absent_part = -1;      // aog_selftest_barrier_timeout: the part that never arrives
  int32_t* ext_perm = nullptr;   // [n_ext_groups * 16] group slot -> env id, -1 = padding (envs sorted by wind, see aog_set_wind)
  double max_wind = 0;           // max |component| of any env's velocity (bounds the rounds per step)
  long long timestep = 0;        // AOEnv.timestep: monotone over episodes (AO_env.py:123)
  // lookahead (aog_set_lookahead): the wind extrusion of step t + 1 is launched by aog_step(t) on a stream of the library's own, behind
  // the fused kernel of step t, and runs beside the step's epilogue and whatever the caller does before aog_step(t + 1) (its policy query)
  bool lookahead = false;
  bool pre_evolved = false;      // the master screens / ring already stand at timestep + 1
  long long steps_since_reset = 0;   // steps since the last whole-batch aog_reset (lock-step episodes end at cfg.max_steps)
  hipStream_t ext_stream = nullptr;
  hipEvent_t ev_fused_done = nullptr, ev_ext_done = nullptr;
  double* psi_master = nullptr;  // [B][N*N] float64 toroidal screens
  int32_t* origin = nullptr;     // [B][2]
  uint32_t* ext_counter = nullptr;  // [B]
  double* velocity = nullptr;    // [B][2]
  double* psi_offset = nullptr;  // [B] piston offset used by the per-step repack
  double* psi_sum = nullptr;     // [B] aperture sums accumulated by the last repack
  int32_t* stencil_v = nullptr;
  int32_t* stencil_h = nullptr;
  int32_t* stencil_v_yx = nullptr;  // (sy << 16 | sx)
  int32_t* stencil_h_yx = nullptr;
  double* At_v = nullptr;        // [nz_v][N]
  double* Bt_v = nullptr;        // [N][N]
  double* At_h = nullptr;
  double* Bt_h = nullptr;
  double* Wa_v = nullptr;        // MFMA-blocked copies [row block][k/8][lane][2] (k_extrude16_split)
  double* Wb_v = nullptr;
  double* Wa_h = nullptr;
  double* Wb_h = nullptr;
  int nz_v = 0, nz_h = 0;
  // int8 composite extrusion (aog_upload_layer_composite; kernels in k_extrude_i8.h).  x8_host: host copies of the operator tables (opaque here)
  void* x8_host = nullptr;
  void* x8_tables_dev = nullptr;   // aog::X8Table [2][kX8MaxK + 1]
  int x8_kmax[2] = {0, 0};         // k_max uploaded per axis (0 = none)
  int ext_mode = 0;                // AOG_EXTRUDE_*
  int32_t* x8_dxy = nullptr;
  int32_t* x8_slot = nullptr;
  int32_t* x8_list = nullptr;
  int32_t* x8_tile_k = nullptr;
  int32_t* x8_items = nullptr;
  int8_t* x8_Z8 = nullptr;
  double* x8_rec = nullptr;
  double* x8_colbuf = nullptr;
  hipStream_t x8_plan_stream = nullptr;   // the plan of step t + 1 runs here beside step t's fused kernel (x8_evolve)
  hipEvent_t x8_ev_evolved = nullptr, x8_ev_planned = nullptr;
  int x8_ahead_level = 0;                 // what was made ahead for x8_plan_step: 1 the plan, 2 the plan and the x phase
  long long x8_plan_step = -1;            // step the arrays of k_x8_plan were last made for ahead of time (-1 none, -2 dropped)
  int x8_tiles64_max = 0, x8_slots_max = 0, x8_KsTot_max = 0, x8_rt_max = 0, x8_items_max = 0;
  int near_v = 0, near_h = 0;    // stencil samples in the two newest slices come first in the uploaded order (aog_upload_layer)
  double sqrt_cn2 = 0, pitch = 0, delta_t = 0;
  const double* next_noise = nullptr;
  int next_noise_max_ext = 0;
  unsigned long long rng_seed = 1234;
  double* partials = nullptr;
  size_t partial_elems = 0;
  // profiling of the fused kernel
  float* ret_acc = nullptr;      // caller-owned episode-return accumulator (aog_set_return_accumulator)
  bool profile = false;
  bool pro_pending = false;      // aog_step_pipelined: the actuators already hold the NEXT step's action (its prologue rode with the last epilogue)
  int profile_block = 8;         // launches per timed block (aog_profile_block)
  int profile_every = 1;         // time every n-th launch of the fused kernel (aog_profile_enable(env, n))
  unsigned profile_phase = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<int> event_kernel;    // AOG_PROF_* id of each used event pair
  size_t events_used = 0;
  double prof_ms[AOG_PROF_COUNT] = {};   // totals of the last aog_profile_read, per kernel id
  int prof_n[AOG_PROF_COUNT] = {};
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;   // size of allocs[i] (dev_release gives work buffers back before the handle is destroyed)
};

namespace aog_host {
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the FUNCTION (per device), not to a handle: remember the largest request
// made for each (function, device) in this process and only ever raise it, so that handles of different shapes coexist.
int ensure_dynamic_lds(const void* fn, size_t bytes, int device);
#ifdef AOG_DEV
extern long long* dev_timeline;   // per-wave time stamps of the last fused launch (AOG_DEV_TIMELINE=1)
#endif
// Fused-kernel launchers, one translation unit per padded mode count so the build parallelises
// (fused_inst.hip compiled with -DAOG_INST_APAD=16|32|64|128).  Return 0 or the aog_status of a failed dynamic-LDS request.
int launch_fused_apad16(aog_env* e, hipStream_t s);
int launch_fused_apad32(aog_env* e, hipStream_t s);
int launch_fused_apad64(aog_env* e, hipStream_t s);
int launch_fused_apad128(aog_env* e, hipStream_t s);
// phase-only contraction u = psi + Mt a for every (pixel, env) with the actuator operands `act16`, written in the psi_tile layout
void launch_phase(aog_env* e, hipStream_t s, const _Float16* act16, float* out_tile);
void launch_phase_grid(aog_env* e, hipStream_t s, const _Float16* act16, float* grid, size_t env_stride, int row_stride, int etile0, int n_et);
void launch_phase_field(aog_env* e, hipStream_t s, const _Float16* act16, float* field, size_t env_stride, int row_stride, bool grid);   // complex64 field, or (grid) one float of reduced phase per pixel
}  // namespace aog_host
