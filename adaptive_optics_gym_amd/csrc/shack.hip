// K10: Shack-Hartmann chain (aog_upload_sh / aog_sh_image / aog_sh_update).
#include "host_common.h"
#include "k_shack.h"
#include <hipfft/hipfft.h>

using namespace aog_host;

extern "C" {

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
        const size_t lds_rows = lds + aog::kShHxLdsBytes<RL, LW>;
        if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_rows_sep<RL, LW>), lds_rows, e->device)) return rc;
        if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_cols_sep<RL, LW, false>), lds, e->device)) return rc;
        if (fused)
          if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_sh_cols_sep<RL, LW, true>), lds_fused, e->device)) return rc;
        const dim3 g1((N / BC + aog::kShFftWaves - 1) / aog::kShFftWaves, e->B);   // pass 1: groups of BC rows; pass 2: groups of BC columns
        {
          TimedRegion tr(e, s, AOG_PROF_SH_ROWS_FWD);
          hipLaunchKernelGGL((aog::k_sh_rows_sep<RL, LW>), g1, dim3(64 * aog::kShFftWaves), lds_rows, s, reinterpret_cast<const float*>(field), G1, tw,
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

int aog_selftest_poisson(const double* lam_dev, double* out_dev, int n_env, int n, uint64_t seed, uint32_t call, void* stream) {
  if (!lam_dev || !out_dev || n_env < 1 || n < 1) return fail(AOG_ERR_INVALID, "aog_selftest_poisson: bad argument");
  hipLaunchKernelGGL(aog::k_sh_noise, dim3((unsigned)((n * n + 255) / 256), n_env), dim3(256), 0, static_cast<hipStream_t>(stream), lam_dev, out_dev, n,
                     (size_t)0, (unsigned long long)seed, call, 0);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

}  // extern "C"
