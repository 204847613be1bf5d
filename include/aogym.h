/*
 * aogym.h — C-ABI of libaogym.so: the MI355X (gfx950) implementation of the AOEnv.step() hot path.
 *
 * The reference (payamparvizi/adaptive_optics_gym) is pure Python on top of hcipy; it has no FFI.
 * The interface each entry point replaces is therefore a *Python* call site in
 * gym_AO/envs/AO_env.py (cited per function).  The reference-side binding a maintainer would add is
 * a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative aog_status; it never throws, never exits;
 *     aog_last_error() returns a thread-local message for the last failure on this thread.
 *   - pointers named *_dev are device pointers owned by the CALLER (e.g. torch tensors); the library
 *     owns only the handle, its constant tables and its per-environment state.
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered and asynchronous.
 *   - one handle per device; a handle is not thread-safe; distinct handles are independent.
 *   - there is no CPU fallback: creating a handle without a HIP device fails with AOG_ERR_HIP.
 */
#ifndef AOGYM_H
#define AOGYM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOG_ABI_VERSION 19

typedef struct aog_env aog_env;

typedef enum {
  AOG_OK = 0,
  AOG_ERR_INVALID = -1,     /* bad argument / unsupported configuration              */
  AOG_ERR_HIP = -2,         /* a HIP runtime call failed (message has the HIP error) */
  AOG_ERR_STATE = -3,       /* call order violated (e.g. step before upload_tables)  */
  AOG_ERR_UNSUPPORTED = -4  /* valid in the reference, not built yet                 */
} aog_status;

enum { AOG_REWARD_STREHL = 0, AOG_REWARD_SMF_SSIM = 1 };           /* AO_env.py:476,487 */
enum { AOG_PRECISION_FAST = 0, AOG_PRECISION_FP64 = 1 };           /* fp32 data / fp64 validation kernel */
enum { AOG_KERNEL_AUTO = 0, AOG_KERNEL_VALU = 1, AOG_KERNEL_MFMA = 2 };

/* Scalar configuration.  Mirrors AOEnv.__init__ kwargs (AO_env.py:17-29) and the constants fixed by
 * parameters_init (AO_env.py:211-247) that the device path consumes. */
typedef struct {
  int32_t abi_version;          /* = AOG_ABI_VERSION                                            */
  int32_t num_envs;             /* B: environments stepped in lock-step                         */
  int32_t n_pupil;              /* N: pupil grid side (reference: 240, AO_env.py:216)            */
  int32_t n_modes;              /* A = act_dim (AO_env.py:223)                                   */
  int32_t obs_dim;              /* o (AO_env.py:236)                                             */
  int32_t n_ap;                 /* aperture pixels (packed list length)                          */
  int32_t n_wfs_tables;         /* real pupil-plane tables at lambda_wfs                         */
  int32_t n_sci_tables;         /* real pupil-plane tables at lambda_sci                         */
  int32_t n_fiber_modes;        /* guided LP modes (3 at V = 2.639)                              */
  int32_t reward_type;          /* AOG_REWARD_*                                                  */
  int32_t sh_operation;         /* 1: action = raw actuators (AO_env.py:115-116)                 */
  int32_t max_steps;            /* timesteps_per_episode (AO_env.py:227)                         */
  int32_t flat_mirror_start;    /* flat_mirror_start_per_episode (AO_env.py:79-80)               */
  int32_t has_rew_threshold;    /* rew_threshold is not None (AO_env.py:500)                     */
  int32_t precision;            /* AOG_PRECISION_*                                               */
  int32_t kernel;               /* AOG_KERNEL_* (fast precision only)                            */
  int32_t pixel_chunks;         /* 0 = auto; number of pixel chunks the fused kernel splits into */
  int32_t atm_dynamic;          /* 1: atm_type == 'dynamic' (float64 master screens + wind extrusion each step) */
  int32_t env_id_base;          /* global id of this handle's env 0 (multi-GPU: rank r of a batch sharded contiguously owns
                                   envs [env_id_base, env_id_base + num_envs)).  Every device random stream — screen synthesis,
                                   extrusion normals, Shack-Hartmann photon noise — is keyed by the GLOBAL env id, so results do
                                   not depend on how the batch is split over handles / GPUs              */
  int32_t reserved0;            /* = 0                                                                                          */
  double wavelength_wfs;        /* 1.5e-6 (AO_env.py:219)                                        */
  double wavelength_sci;        /* 2.2e-6 (AO_env.py:220)                                        */
  double surface_rms_target;    /* 0.1*wavelength_sci (AO_env.py:120)                            */
  double rew_threshold;         /* used iff has_rew_threshold                                    */
  double ssim_ref_peak;         /* 2.8 (AO_env.py:492)                                           */
  double ssim_alpha;            /* 0.8 (AO_env.py:497)                                           */
} aog_config;

/* Host-precomputed constant tables (float64, HOST pointers; copied and converted by the call).
 * They are what AOEnv.__init__ precomputes through hcipy (AO_env.py:42-68, 293-393).
 *
 *   field at lambda_wfs on packed aperture pixel p:  E_p = exp(i*phi_p)  (amplitude folded into tables)
 *   U_m = sum_p cos(phi_p) g_m(p),  V_m = sum_p sin(phi_p) g_m(p)        (g = wfs_tables / sci_tables)
 *   Z_j = sum_m coef[j][m] * (U_m + i V_m)
 *   obs_raw[j] = |Z_j|^2  (j < o^2);  power = sum_k |Z_{o^2+k}|^2  (k < n_fiber_modes);
 *   strehl = |Z_sci|^2.
 */
typedef struct {
  const int32_t* ap_index;   /* [n_ap]  flat pupil index iy*N+ix of packed pixel p (row-major order) */
  const double* modes;       /* [n_ap][n_modes]  DM mode matrix restricted to the aperture (metres of
                                surface per unit actuator; AO_env.py:346-347,352-353)               */
  const double* gram;        /* [n_modes][n_modes]  centred Gram matrix: std_grid(M a)^2 = a' G a     */
  const double* wfs_tables;  /* [n_wfs_tables][n_ap]                                                 */
  const double* sci_tables;  /* [n_sci_tables][n_ap]                                                 */
  const double* wfs_coef;    /* [o^2 + n_fiber_modes][n_wfs_tables][2]  (re, im)                     */
  const double* sci_coef;    /* [1][n_sci_tables][2]                                                 */
  /* optional (NULL = aog_focal_image unsupported): the two matrices of the Fraunhofer matrix Fourier transform onto
   * the n_focal x n_focal fiber focal grid (propagator_fiber, AO_env.py:390), scale factors folded into focal_m1:
   *   F = focal_m1 [n_focal][N] . E [N][N] . focal_m2 [N][n_focal]                                           */
  const double* focal_m1;    /* [n_focal][N][2]                                                        */
  const double* focal_m2;    /* [N][n_focal][2]                                                        */
  int32_t n_focal;           /* 128 (AO_env.py:235)                                                    */
} aog_tables;

typedef struct {
  int32_t abi_version, num_envs, num_envs_padded, n_ap, n_ap_padded, n_modes_padded;
  int32_t pixel_chunks, kernel, n_sums, reserved;
  int64_t device_bytes;      /* bytes of HBM the handle owns */
} aog_info;

int aog_abi_version(void);
/* Identity of the sources this binary was compiled from: the first 32 hex digits of the SHA-256 over every csrc/ header and .hip file and this header
 * (adaptive_optics_gym_amd/build.py::source_id), "+FLAG" appended for developer builds.  The ctypes binding refuses a library whose id
 * differs from the sources it finds beside it. */
const char* aog_build_id(void);
const char* aog_last_error(void);
/* sizeof() of the structs of this header as the library was compiled, so that a binding in another language can verify its own
 * declarations at load time: which = 0 aog_config, 1 aog_tables, 2 aog_layer_tables, 3 aog_sh_tables, 4 aog_actor, 5 aog_info, 6 aog_layer_composite;
 * -1 for any other value. */
int64_t aog_struct_size(int which);

/* AOEnv.__init__ (AO_env.py:17-71): allocate the handle and its state on `device`. */
int aog_create(const aog_config* cfg, int device, aog_env** out);
void aog_destroy(aog_env* env);
int aog_get_info(const aog_env* env, aog_info* out);

/* The part of AOEnv.__init__ that goes through hcipy (pupil_simulation, incoming_wavefront,
 * DM_function, fiber_coupling; AO_env.py:50-64). */
int aog_upload_tables(aog_env* env, const aog_tables* tables);

/* layer._achromatic_screen for envs [first, first+count) (hcipy InfiniteAtmosphericLayer state created at
 * AO_env.py:370 / regenerated at AO_env.py:77).  psi_dev: [count][N][N] achromatic screens (phase * lambda,
 * hcipy's unit), float64 or float32, row-major with x fastest.  The aperture mean of every screen is
 * removed (all outputs are invariant to a global phase) before conversion to the internal fp32 layout. */
int aog_set_screens_f64(aog_env* env, const double* psi_dev, int first, int count, void* stream);
int aog_set_screens_f32(aog_env* env, const float* psi_dev, int first, int count, void* stream);

/* hcipy InfiniteAtmosphericLayer construction products (AO_env.py:370; hcipy _make_stencils / _make_AB_matrices):
 * stencil positions and the auto-regressive extrusion matrices, shared by every env of the handle.  HOST pointers. */
typedef struct {
  int32_t nz_vertical, nz_horizontal;   /* stencil sizes (3N unless samples coincide)                        */
  const int32_t* stencil_vertical;      /* [nz_v] flat logical index iy*N+ix, increasing ('bottom' stencil)  */
  const int32_t* stencil_horizontal;    /* [nz_h] ('left' stencil)                                           */
  const double* A_vertical;             /* [N][nz_v]  new row    = A z + sqrt(Cn^2) B n                      */
  const double* B_vertical;             /* [N][N]                                                            */
  const double* A_horizontal;           /* [N][nz_h]  new column                                             */
  const double* B_horizontal;           /* [N][N]                                                            */
  double sqrt_cn_squared;               /* sqrt of AO_env.py:367                                             */
  double pixel_pitch;                   /* pupil_grid.delta (m)                                              */
  double delta_t;                       /* 1e-3 s (AO_env.py:226)                                            */
} aog_layer_tables;
int aog_upload_layer(aog_env* env, const aog_layer_tables* layer);

/* k_max successive one-pixel extrusions along one axis composed into ONE linear operator (exact: the same samples given the same normals;
 * adaptive_optics_gym_amd/extrusion_host.py::compose_extrusions builds it from the tables above):
 *     [R_1; ...; R_k] = A z_old + sqrt(Cn^2) B [n_1; ...; n_k],   R_j = the slice shift j of a step creates,
 * z_old = the screen BEFORE the first shift at the union of every sample the k stencils reach.  With both axes uploaded (after
 * aog_upload_layer, same k_max) aog_step advances a dynamic atmosphere with two matrix products per step on the int8 matrix cores
 * (csrc/k_extrude_i8.h: operands in base-128 digits, exact int32 accumulation; new samples good to ~1e-9 rad) instead of the chain of
 * one-pixel float64 rounds; operators for k < k_max are cut out of the uploaded one.  k_max must cover the largest whole-pixel shift any
 * env makes per step (floor(max wind component * delta_t / pixel_pitch) + 1) and be <= 8; steps the operators do not cover, float64
 * validation handles and AOG_EXTRUDE_F64 keep the float64 kernels.  HOST pointers.
 * Work ahead: what step t + 1 needs that depends on the clock, the winds and the screens step t's extrusion left — its shift plan and its
 * whole x phase, which writes operands and staged columns only — is launched by aog_step(t) on a low-priority stream of the library's own
 * and runs beside step t's remaining kernels and the caller's work between the two steps.  Invisible to the caller: every call that
 * changes what it read drops it, normals supplied for the next step redo it, results are bit for bit those of the in-line order
 * (AOG_X8_NO_PLAN_AHEAD=1 in the environment runs everything in line; AOG_X8_NO_PHASE_AHEAD=1 only the x phase). */
typedef struct {
  int32_t axis;             /* 0: vertical ('bottom' / 'top': new rows), 1: horizontal ('left' / 'right': new columns)             */
  int32_t k_max;            /* shifts composed                                                                                    */
  int32_t n_old;            /* U: size of the union stencil                                                                       */
  int32_t reserved0;
  const int32_t* old_yx;    /* [U] (sy << 16 | sx): logical position on the screen hcipy's _extrude sees, before the first shift   */
  const double* A;          /* [k_max N][U]: row (j - 1) N + i = sample i of the slice shift j creates                            */
  const double* B;          /* [k_max N][k_max N]: column (j' - 1) N + i' = normal i' of shift j' (without the sqrt(Cn^2) factor)  */
} aog_layer_composite;
int aog_upload_layer_composite(aog_env* env, const aog_layer_composite* op);
enum { AOG_EXTRUDE_AUTO = 0, AOG_EXTRUDE_F64 = 1 };   /* AUTO: the int8 composite form whenever its operators cover the step        */
int aog_set_extrusion_mode(aog_env* env, int mode);

/* layer.velocity of every env: [B][2] float64 (vx, vy) in m/s (hcipy draws the direction at construction).
 * max_abs_component >= max over envs of max(|vx|, |vy|): bounds the whole-pixel shifts per step.
 * Reads the velocities back once to group envs of similar wind for the extrusion kernel: synchronises `stream`. */
int aog_set_wind(aog_env* env, const double* velocity_dev, double max_abs_component, void* stream);

/* Lookahead for rollouts over a dynamic atmosphere.  The wind shift of step t + 1 (layer.t = ..., AO_env.py:125) depends on nothing
 * step t computes — only the product of the field with the evolved screen does (AO_env.py:132).  With lookahead on, aog_step(t) launches
 * the extrusion of step t + 1 on a stream of the library's own as soon as its fused kernel has finished reading the screens; it then runs
 * beside the step's epilogue and beside whatever the caller enqueues before aog_step(t + 1) (its policy query), and aog_step(t + 1) joins it.
 * Results are bit-identical with and without (same kernels, same random streams).  What changes: between the two calls the handle's screens
 * already stand at step t + 1, so aog_reset, aog_get_screens_f64, aog_get_state, aog_set_screens_*, aog_get_phase_screen, aog_focal_image(s)
 * and aog_sh_image return AOG_ERR_STATE there.  The last step of a lock-step episode (cfg.max_steps steps after the last whole-batch aog_reset)
 * never looks ahead, so all of them are available at episode boundaries — where a rollout calls them.  Needs the device random stream
 * (supplying normals with aog_set_extrusion_noise switches lookahead off for that step).  Off by default. */
int aog_set_lookahead(aog_env* env, int enable);

/* Standard normals for the extrusions of the NEXT aog_step: [B][max_ext][N] float64, consumed in hcipy's order (x shifts
 * first, then y).  NULL (default) = on-device Philox4x32-10 stream seeded by aog_set_rng_seed. */
int aog_set_extrusion_noise(aog_env* env, const double* noise_dev, int max_ext, void* stream);
int aog_set_rng_seed(aog_env* env, uint64_t seed);

/* layer._achromatic_screen of envs [first, first + count) as plain [count][N][N] float64 (hcipy's unit: phase * lambda).  Dynamic
 * handles return their float64 master screens; quasi_static / semi_dynamic handles return the stored screen exactly as the step
 * kernels read it: aperture pixels only (0 outside), aperture mean removed, fp32 values widened without rounding. */
int aog_get_screens_f64(aog_env* env, double* psi_dev, int first, int count, void* stream);

/* layer.reset() / layer construction (AO_env.py:77, :370): synthesise new von Karman screens for envs [first, first+count) ON THE
 * DEVICE and install them — hcipy's FiniteAtmosphericLayer + SpectralNoiseFactoryFFT: complex normals on the (oversampling N)^2
 * FFT grid times sqrt(PSD (2 pi)^2 / du^2), inverse FFT (hipFFT/rocFFT), real part of the central N x N crop / delta^2 * sqrt(Cn^2).
 * Normals come from the handle's Philox stream (aog_set_rng_seed); statistically equivalent to hcipy, not draw-for-draw: only the
 * half plane of spectrum lines 0..m/2 is drawn (a conjugate pair of independent complex normals with equal amplitudes contributes
 * to the REAL part exactly like one normal of sqrt(2) x the amplitude), and pupils of 64 R / 60 R pixels never materialise the
 * oversampled array (pruned two-pass transform). */
int aog_generate_screens(aog_env* env, int first, int count, int oversampling, double cn_squared, double outer_scale,
                         double pixel_pitch, void* stream);

/* How aog_generate_screens draws a screen.  Both methods draw the same zero-mean stationary Gaussian field on the N x N pupil up to
 * max |dC(r)| < 1e-4 C(0) over every lag r of the pupil (tests/test_screen_twoband.py evaluates both covariance functions exactly on
 * the host in float64), i.e. they are statistically equivalent to hcipy's layer.reset() (AO_env.py:77), not draw-for-draw.
 *   AOG_SCREENS_TWOBAND (default)  the spectrum samples' variance is split by a smooth radial window into a low band kept on hcipy's
 *       (oversampling N)^2 frequency grid (non-zero below 2 cycles per pupil diameter: 4 oversampling^2 samples) and a high band drawn
 *       on the (2 N)^2 grid (period 2 D: its covariance has decayed before the wrap-around lag).  4 N^2 + 4 q^2 samples per screen
 *       instead of q^2 N^2.  Needs oversampling >= 4 and even, N % 4 == 0; otherwise the literal method is used.
 *   AOG_SCREENS_HCIPY  the literal (oversampling N)^2 draw described above. */
enum { AOG_SCREENS_TWOBAND = 0, AOG_SCREENS_HCIPY = 1 };
int aog_set_screen_method(aog_env* env, int method);

/* Shack-Hartmann baseline controller (AO_env.py:254-290, 396-465).  Tables built on the host by the counterpart of
 * shack_hartmann_init (adaptive_optics_gym_amd/sh_host.py).  HOST pointers, float64. */
typedef struct {
  int32_t n_sub;                /* flux-selected sub-apertures (AO_env.py:418-425)                                 */
  const int32_t* sub_slot;      /* [N*N] slot 0..n_sub-1 of the pixel's lenslet, or -1                             */
  const double* centres;        /* [n_sub][2] lenslet positions (x, y) the estimator subtracts                      */
  const double* slopes_ref;     /* [2 n_sub] reference slopes, all x then all y (AO_env.py:428)                     */
  const double* reconstruction; /* [A][2 n_sub] inverse_tikhonov(response, 1e-3) (AO_env.py:464-465)                */
  const double* mla_phase;      /* [N*N][2] exp(i k opd) of the micro-lens array                                    */
  const double* transfer;       /* [2N][2N][2] Fresnel transfer function on the unshifted 2x-padded FFT grid        */
  const double* x_det;          /* [N] detector coordinate of a column / row (NoiselessDetector(focal_grid))        */
  double field_amplitude;       /* amplitude of wf_wfs on the aperture / magnification                              */
  double image_scale;           /* magnified pixel area x delta_t: image = |E|^2 * image_scale                      */
  double gain, leakage;         /* 0.3, 0.01 (AO_env.py:282-283)                                                    */
  int32_t fft_double;           /* 0 (default): the Fresnel propagation runs complex64 transforms — relative error ~1e-6 of the image
                                   peak, far below the photon noise (>= 1e-3) large_poisson adds before anything reads the image;
                                   1: complex128 transforms (bit-for-bit comparisons of the noise-free image with a float64 oracle) */
  int32_t reserved0;
} aog_sh_tables;
int aog_upload_sh(aog_env* env, const aog_sh_tables* sh);

/* camera.integrate(shwfs(magnifier(deformable_mirror_shack(layer(wf_wfs)))), delta_t); camera.read_out() (AO_env.py:263-274):
 * the noise-free Shack-Hartmann image of every env, [B][N*N] float64.  image_dev may be NULL = "the next call is aog_sh_update(NULL)":
 * the image stays internal, and for pupils of 128 / 240 / 256 / 480 / 512 pixels (complex64) it is not even written — photon noise and the
 * estimator's per-lenslet sums are taken inside the last propagation pass and handed to that aog_sh_update. */
int aog_sh_image(aog_env* env, double* image_dev, void* stream);

/* large_poisson + estimate + slopes_ref + leaky integrator (AO_env.py:275-287) -> deformable_mirror_shack.actuators, which
 * are also returned in action_dev [B][A] float64 (the action SH_step hands to step()).  noisy_image_dev: the image after the
 * caller's own large_poisson (parity with a host RNG stream), or NULL = photon noise from the handle's Philox stream. */
int aog_sh_update(aog_env* env, const double* noisy_image_dev, double* action_dev, void* stream);

/* Checkpointing (the reference has none for the env): every piece of per-env state the handle owns — screens
 * (and ring-buffer origins / RNG stream positions for dynamic handles), mirror and Shack-Hartmann actuators, per-episode step
 * counters — as one opaque device blob of aog_state_bytes() bytes, plus the global step counter.  A blob is only valid for a
 * handle created with the same configuration. */
int64_t aog_state_bytes(const aog_env* env);
int aog_get_state(aog_env* env, void* blob_dev, int64_t* timestep_out, void* stream);
int aog_set_state(aog_env* env, const void* blob_dev, int64_t timestep, void* stream);

/* The sensing-arm pupil phase of one env in radians on the full N x N grid (0 outside the aperture, aperture mean removed):
 * atmosphere only (what render() shows as the phase screen, AO_env.py:87-88,128-129).  float32 [N*N]. */
int aog_get_phase_screen(aog_env* env, int env_index, float* phase_dev, void* stream);

/* Episode-return accumulation of the rollout (algorithm.py:509-510 sums the rewards of an episode on the host): when
 * returns_dev ([B] float32, caller-owned, device) is set, every aog_step also does returns_dev[env] += reward[env] (float32, the
 * same arithmetic as the caller's own `returns += reward`) in its last kernel.  NULL detaches.  The caller zeroes the buffer at
 * episode start and reads it (all-gathers it across ranks) at episode end. */
int aog_set_return_accumulator(aog_env* env, float* returns_dev);

/* Synchronises the device and returns the handle's sticky device-side status word: 0 = fine, 1 = a bounded inter-workgroup wait
 * timed out (results of that step are invalid).  The library also watches the same flag through pinned host memory without
 * synchronising: once it is set, aog_step and aog_reset fail with AOG_ERR_STATE (at the latest from the call after the one whose
 * launch tripped it) until new screens are installed for the whole batch (aog_set_screens_*) or a state is restored. */
int aog_device_status(aog_env* env, int32_t* status_out);

/* deformable_mirror.actuators for all envs (metres; AO_env.py:116).  [B][A] float64 device pointers. */
int aog_get_actuators(aog_env* env, double* act_dev, void* stream);
int aog_set_actuators(aog_env* env, const double* act_dev, void* stream);

/* AOEnv.reset (AO_env.py:74-103) for the envs with mask_dev[b] != 0 (NULL = all): flatten the mirror iff
 * flat_mirror_start, zero the per-episode step counter, and return the observation of EVERY env
 * (obs_raw_dev [B][o^2] float32 before the cast, obs_dev [B][o^2] IEEE half; either may be NULL). */
int aog_reset(aog_env* env, const uint8_t* mask_dev, float* obs_raw_dev, uint16_t* obs_dev, void* stream);

/* AOEnv.step (AO_env.py:106-153) incl. reward_function (AO_env.py:468-503).
 *   action_dev  [B][A] float32
 *   obs_raw_dev [B][o^2] float32 (wf.power before the float16 cast, AO_env.py:142)   nullable
 *   obs_dev     [B][o^2] IEEE half (AO_env.py:153)                                   nullable
 *   reward_dev  [B] float32;  done_dev [B] uint8;  power_dev [B] float32 (info["power"])
 *   strehl_dev  [B] float32 (Strehl ratio in [0,1]; nullable) */
int aog_step(aog_env* env, const float* action_dev, float* obs_raw_dev, uint16_t* obs_dev, float* reward_dev,
             uint8_t* done_dev, float* power_dev, float* strehl_dev, void* stream);

/* aog_step for callers that know the NEXT action when they hand over the current one (open-loop action sequences, replayed trajectories,
 * throughput benchmarks on synthetic actions): identical results, one kernel launch less per step.  The last launch of the call carries
 * the epilogue of this step AND the action -> actuator prologue of the next one (they share nothing); the next call then skips its own
 * prologue and `action` must be the `action_next` of the call before (it is not looked at).  action_next = NULL ends the sequence (a plain
 * epilogue).  Between a call with action_next != NULL and the next call the mirror state already belongs to the next step: aog_reset,
 * aog_step, aog_get_state, aog_get_actuators, aog_focal_image(s), aog_sh_* ... fail with AOG_ERR_STATE until the sequence is ended
 * (aog_set_actuators / aog_set_state replace the mirror and end it).  Not together with aog_set_lookahead.  (ABI 17) */
int aog_step_pipelined(aog_env* env, const float* action_dev, const float* action_next_dev, float* obs_raw_dev, uint16_t* obs_dev, float* reward_dev,
                       uint8_t* done_dev, float* power_dev, float* strehl_dev, void* stream);

/* self.wf_wfs_after_foc.electric_field of one env (AO_env.py:138): the n_focal x n_focal focal-plane field of the sensing
 * arm with the current screen and mirror, as interleaved (re, im) float32, row-major (y, x), up to a global phase (the
 * library stores screens with their aperture mean removed).  Off the step() path; used for render()/fiber cross-checks. */
int aog_focal_image(aog_env* env, int env_index, float* field_dev /* [n_focal][n_focal][2] */, void* stream);

/* The same field for envs [first, first + count) in one call: field_dev [count][n_focal][n_focal][2] float32.  Fast-precision handles
 * only.  E = exp(i phi) on the pupil grid from the split-f16 phase contraction of the step kernels, then the two matrices of the
 * Fraunhofer matrix Fourier transform as two batched complex products on the f16 matrix cores (v_mfma_f32_32x32x16_f16, every operand split
 * hi + lo: 22 significant bits per factor, exact products), the pupil field formed inside the first product from a dense phase grid. */
int aog_focal_images(aog_env* env, int first, int count, float* field_dev, void* stream);

/* ---- policy query of the rollout (Actor.forward + Actor.get_action, network.py:17-69; caller algorithm.py:216-296) ----
 * mean = W_o drop(relu(W_3 drop(relu(W_2 drop(relu(W_1 obs + b_1)) + b_2)) + b_3)) + b_o with nn.Dropout(dropout_p) ACTIVE
 * (the reference never leaves training mode while acting), action = mean + sqrt(cov_var) eps, eps ~ N(0, I),
 * log_prob = MultivariateNormal(mean, cov_var I).log_prob(action).  Weights are torch nn.Linear layouts [out][in], float32,
 * device pointers; they are read on every call (the learner updates them between rollouts).  Dropout masks and eps come from
 * Philox4x32-10 keyed by (seed, call_index, global env id, layer, unit): statistically, not bit-wise, torch's streams.
 * One launch per call; independent of any aog_env handle. */
typedef struct aog_actor {
  int32_t batch, state_dim, hidden_dim, act_dim;
  int32_t env_id_base;                  /* global id of obs row 0: dropout masks and eps are keyed by the GLOBAL env id */
  int32_t reserved0;                    /* = 0 */
  const float* w1; const float* b1;     /* [hidden][state],  [hidden] */
  const float* w2; const float* b2;     /* [hidden][hidden], [hidden] */
  const float* w3; const float* b3;     /* [hidden][hidden], [hidden] */
  const float* wo; const float* bo;     /* [act][hidden],    [act]    */
  float dropout_p;                      /* 0.5 in the reference (network.py:39); 0 = evaluation mode */
  float cov_var;                        /* 0.5 in the reference (algorithm.py:107) */
  uint64_t seed, call_index;
} aog_actor;
/* obs: [batch][state_dim], float16 bits (obs_is_f16 = 1: what aog_step writes) or float32; outputs may be NULL. */
int aog_actor_act(const aog_actor* net, int device, const void* obs_dev, int obs_is_f16, float* mean_dev /* [batch][act] */,
                  float* action_dev /* [batch][act] */, float* log_prob_dev /* [batch] */, void* stream);

/* Self-test hook: sin(2 pi u), cos(2 pi u) for n float32 revolutions u_dev with the fused kernels' device code.
 * flavour 0 = polynomial, 1 = v_sin_f32/v_cos_f32 after the exact reduction, 2 = v_sin_f32/v_cos_f32 on raw input. */
int aog_selftest_sincos(const float* u_dev, float* sin_dev, float* cos_dev, int n, int flavour, void* stream);

/* Self-test hook: the Shack-Hartmann camera's photon noise (large_poisson, AO_env.py:272-275) on caller-supplied expected counts:
 * lam_dev / out_dev [n_env][n][n] float64, drawn by the same device code and Philox keying (env, row, column, call) as aog_sh_update(NULL) and the
 * fused row pass.  For distribution tests of the sampler (exact inversion below 12 counts, skew-corrected rounded normal above). */
int aog_selftest_poisson(const double* lam_dev, double* out_dev, int n_env, int n, uint64_t seed, uint32_t call, void* stream);

/* Self-test hook for the failure path of the dynamic atmosphere's inter-workgroup barrier (k_extrude16_split): runs the wind extrusion of
 * one step with one of every group's four workgroups absent and a short poll limit, so the partners' bounded wait gives up exactly as it
 * would if they were not co-resident.  Synchronises.  Afterwards aog_device_status() reports 1 and aog_step / aog_reset fail with
 * AOG_ERR_STATE until new screens are installed for the whole batch or a state is restored; the handle's screens are invalid (that is the
 * point).  Envs must move by at least one pixel in the step for a barrier to be reached.  AOG_ERR_UNSUPPORTED for handles that do not
 * use the split extrusion kernel. */
int aog_selftest_barrier_timeout(aog_env* env, void* stream);

/* Microseconds-resolution timing of the dominant (fused) kernel of the most recent aog_step/aog_reset calls,
 * measured with HIP events on the stream the kernel was launched on.  Enable, run steps, then read the
 * mean duration (ms) and the number of launches averaged.  enable = n > 1 times one block of consecutive launches (aog_profile_block, default 8) in n
 * only — the middle block of every n: the two event records
 * of a timed launch hold the stream for ~6 us, which a throughput measurement running at the same time should not pay on
 * every step. */
int aog_profile_enable(aog_env* env, int enable);
/* Launches per timed block (default 8; 1 .. 64): a window of a few dozen steps takes a short block so that the records stay ~1 % of it
 * (ABI 16).  Takes effect at the next aog_profile_enable / aog_profile_read. */
int aog_profile_block(aog_env* env, int launches);
int aog_profile_read(aog_env* env, double* mean_ms, int* launches);
/* While profiling is enabled the kernels that dominate the other workloads are timed the same way (HIP events on the launch stream,
 * every launch: they run once per reset or take >= 100 us): which = one of AOG_PROF_*; returns the mean duration and the number of launches
 * collected by the LAST aog_profile_read (which drains the events of every kernel).  AOG_PROF_FUSED repeats that call's own result. */
enum {
  AOG_PROF_FUSED = 0,        /* k_fused_tab / k_fused_valu / k_fused_ref                                   */
  AOG_PROF_SCREEN_ROWS = 1,  /* k_screen2_rows / k_screen_rows: spectrum draw + row transforms (K8 pass A) */
  AOG_PROF_SCREEN_COLS = 2,  /* k_screen2_cols / k_screen_cols (K8 pass B)                                 */
  AOG_PROF_PACK = 3,         /* k_screen_means + k_pack_tiles (or k_pack_screens) of a screen installation */
  AOG_PROF_EXTRUDE = 4,      /* k_extrude16_split / k_extrude16 / k_extrude (K7)                           */
  AOG_PROF_SH_FIELD = 5,     /* k_phase_mfma<FIELD> (K10)                                                  */
  AOG_PROF_SH_ROWS_FWD = 6,  /* k_sh_rows_fwd                                                              */
  AOG_PROF_SH_COLS = 7,      /* k_sh_cols                                                                  */
  AOG_PROF_SH_ROWS_INV = 8,  /* k_sh_rows_inv (+ photon noise + lenslet sums when fused)                   */
  AOG_PROF_COUNT = 9
};
int aog_profile_read_kernel(aog_env* env, int which, double* mean_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* AOGYM_H */
