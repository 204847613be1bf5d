// Host-side helpers shared by the translation units of libaogym.so (not part of the C-ABI).
#pragma once
#include "aogym_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace aog_host {

// sets aog_last_error() of the calling thread and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return ::aog_host::fail(AOG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

constexpr size_t kLdsBytes = 160 * 1024;   // LDS per CU on gfx950

int dev_alloc_bytes(aog_env* e, void** out, size_t bytes, bool zero);
void dev_release_ptr(aog_env* e, void** ptr);
template <typename T>
int dev_alloc(aog_env* e, T** out, size_t count, bool zero = true) {
  void* p = nullptr;
  const int rc = dev_alloc_bytes(e, &p, std::max<size_t>(count, 1) * sizeof(T), zero);
  if (rc == AOG_OK) *out = static_cast<T*>(p);
  return rc;
}
// give a work buffer of the handle back (workspaces that are re-sized when the caller changes the synthesis method or oversampling:
// without this every change would keep the old gigabytes until aog_destroy)
template <typename T>
void dev_release(aog_env* e, T** ptr) {
  void* p = static_cast<void*>(*ptr);
  dev_release_ptr(e, &p);
  *ptr = nullptr;
}

// zero `n_words` 32-bit words at p on stream s with a kernel of the library (see k_zero_words for why not hipMemsetAsync)
void zero_words(void* p, size_t n_words, hipStream_t s);

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

int check_poisoned(const aog_env* e, const char* who);
int refuse_pre_evolved(const aog_env* e, const char* who);
int clear_poison_if_whole(aog_env* e, int first, int count, hipStream_t s);
int set_screens_f32(aog_env* e, const float* psi, int first, int count, hipStream_t s, bool means_ready = false);   // means_ready: pack_mean[0 .. count) holds the aperture means already   // device screens [count][N][N] -> internal layouts
// act_dm -> the operand layouts of the fused kernels (act_ll: optional third f16 term of the actuators, K4)
int load_actuators(aog_env* e, hipStream_t s, _Float16* act_ll = nullptr);
// atmosphere.hip
int pack_from_master(aog_env* e, int first, int count, hipStream_t s, bool per_step = false);
int evolve_layer(aog_env* e, hipStream_t s, long long step_index);
int x8_drop_ahead(aog_env* e);   // before anything the int8 extrusion's work ahead (plan, x phase of the next step) read is changed
int ensure_tiles(aog_env* e, hipStream_t s);
int ring_from_master(aog_env* e, int first, int count, int keep_ref, hipStream_t s);
int store_master_f64(aog_env* e, const double* psi, int first, int count, hipStream_t s);
int store_master_f32(aog_env* e, const float* psi, int first, int count, hipStream_t s);
int unroll_master(aog_env* e, double* psi_dev, int first, int count, hipStream_t s);

}  // namespace aog_host
