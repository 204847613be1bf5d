// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs?  A = 2^-20 (subnormal in f16) in every element, B = 2^10: each output is
// 16 * 2^-10 = 2^-6 if subnormals are kept, 0 if they are flushed.  hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  f16v d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  out[threadIdx.x] = d[0];
}
int main() {
  float* o; hipMalloc(&o, 256);
  for (float av : {9.5367431640625e-07f /* 2^-20 */, 6.103515625e-05f /* 2^-14, smallest normal */, 3.0517578125e-05f /* 2^-15 */}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, av, 1024.f);
    float h[64]; hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
    printf("a = %.3e (f16 %s), b = 1024: out = %.6e, expected with subnormals kept %.6e\n", av, av < 6.1e-5f ? "subnormal" : "normal", h[0], 16.f * av * 1024.f);
  }
  return 0;
}
