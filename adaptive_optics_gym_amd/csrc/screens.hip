// K8: von Karman screen synthesis inside the library (aog_generate_screens).
#include "host_common.h"
#include "k_screens.h"
#include <hipfft/hipfft.h>

using namespace aog_host;

extern "C" {

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
    // the aperture means ride in pass B when the screens go on to the tiled conversion (set_screens: static atmospheres, >= 8 screens a call)
    a.n_tiles = (N + NL - 1) / NL;
    const bool fused_means = !e->cfg.atm_dynamic && e->cfg.precision == AOG_PRECISION_FAST && e->ap_bits && !getenv("AOG_SCREENS_SEPARATE_MEANS");
    if (fused_means) {
      int rc;
      const size_t need = (size_t)e->syn_batch * a.n_tiles;
      if (e->syn_part_elems < need) {
        if (e->syn_part) HIP_TRY(hipDeviceSynchronize());
        dev_release(e, &e->syn_part);
        if ((rc = dev_alloc(e, &e->syn_part, need, false)) != AOG_OK) return rc;
        e->syn_part_elems = need;
      }
      if (!e->pack_mean && (rc = dev_alloc(e, &e->pack_mean, (size_t)e->B, false)) != AOG_OK) return rc;
      a.ap_bits = e->ap_bits;
      a.part = e->syn_part;
    }
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
      if (fused_means) hipLaunchKernelGGL(aog::k_mean_from_parts, dim3((nb + 255) / 256), dim3(256), 0, s, e->syn_part, e->pack_mean, nb, a.n_tiles, e->n_ap);
      int rc = set_screens_f32(e, e->syn_out, first + done, nb, s, fused_means);
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
      int rc = set_screens_f32(e, e->fft_crop, first + done, nb, s);
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
  if (int rcd = x8_drop_ahead(e)) return rcd;   // (work done ahead for the next step of a dynamic atmosphere read the screens this call replaces)
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
      int rc = set_screens_f32(e, e->syn_out, first + done, nb, s);
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
    int rc = set_screens_f32(e, e->fft_crop, first + done, nb, s);
    if (rc != AOG_OK) return rc;
  }
  hipLaunchKernelGGL(aog::k_bump_generation, dim3((count + 255) / 256), dim3(256), 0, s, e->screen_gen + first, count);
  HIP_TRY(hipGetLastError());
  return clear_poison_if_whole(e, first, count, s);
}

}  // extern "C"
