// K4: focal-plane field export (aog_focal_image / aog_focal_images).
#include "host_common.h"
#include "k_focal.h"

using namespace aog_host;

extern "C" {

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
  if ((rc = load_actuators(e, s, e->focal_act_ll)) != AOG_OK) return rc;
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

}  // extern "C"
