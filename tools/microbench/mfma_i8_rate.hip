// (independent accumulators, then ONE accumulator = a dependent chain, then two alternating)
// Cycles per v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 / v_mfma_f32_32x32x16_f16, back to back on one SIMD (one wave per SIMD,
// independent accumulators).    hipcc -O2 --offload-arch=gfx950 mfma_i8_rate.hip -o mfma_i8_rate && ./mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int WHICH>
__global__ void k(long long* out, int* sink, int iters) {
  i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(float)(threadIdx.x + i); bh[i] = (_Float16)(float)i; }
  i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  i32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
  f32x16 e0 = {}, e1 = {}, e2 = {}, e3 = {};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (WHICH == 0) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    } else if (WHICH == 3) {   // ONE accumulator: a chain of dependent instructions
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    } else if (WHICH == 4) {   // two accumulators alternating
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
    } else if (WHICH == 1) {
      d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
    } else {
      e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, e1, 0, 0, 0);
      e2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, e2, 0, 0, 0);
      e3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, e3, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3] + (int)(e0[0] + e1[1] + e2[2] + e3[3]);
}
int main() {
  long long* out; int* sink;
  (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&sink, 4 * 64 * 1024);
  const int iters = 4096;
  const char* names[5] = {"v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_32x32x16_f16", "i8 32x32x32, ONE accumulator", "i8 32x32x32, two accumulators"};
  for (int w = 0; w < 5; ++w)
    for (int blocks : {1, 1024}) {
      if (w == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      if (w == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      if (w == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      if (w == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      if (w == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, sink, iters);
      long long t;
      (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      printf("%-26s %4d waves: %.1f s_memtime ticks per instruction\n", names[w], blocks, (double)t / (4.0 * iters));
    }
  return 0;
}
