// The int8 extrusion product's inner step in isolation: 19 v_mfma_i32_32x32x32_i8 on six accumulators (levels l = s + t <= 5 of 5 x 5 digit
// operands), operands in registers, one wave per SIMD.  Orders: level-major (as the kernel first had it), digit-major (consecutive
// instructions never share an accumulator).    hipcc -O2 --offload-arch=gfx950 mfma_i8_step.hip -o mfma_i8_step && ./mfma_i8_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#ifdef ACC_IN_AGPR
#define MF(acc, x, y) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))
#else
#define MF(acc, x, y) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc, 0, 0, 0)
#endif
template <int ORDER>
__global__ __launch_bounds__(64) void k(long long* out, int* sink, const i32x4* src, int iters, int trivial) {
  i32x4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 64], a2 = src[threadIdx.x + 128], a3 = src[threadIdx.x + 192], a4 = src[threadIdx.x + 256];
  i32x4 z0 = src[threadIdx.x + 320], z1 = src[threadIdx.x + 384], z2 = src[threadIdx.x + 448], z3 = src[threadIdx.x + 512], z4 = src[threadIdx.x + 576];
  if (trivial) { a0 = a1 = a2 = a3 = a4 = i32x4{1, 0, 0, 0}; z0 = z1 = z2 = z3 = z4 = i32x4{0, 1, 0, 0}; }
  i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {   // level-major
      MF(c0, a0, z0);
      MF(c1, a0, z1); MF(c1, a1, z0);
      MF(c2, a0, z2); MF(c2, a1, z1); MF(c2, a2, z0);
      MF(c3, a0, z3); MF(c3, a1, z2); MF(c3, a2, z1); MF(c3, a3, z0);
      MF(c4, a0, z4); MF(c4, a1, z3); MF(c4, a2, z2); MF(c4, a3, z1); MF(c4, a4, z0);
      MF(c5, a1, z4); MF(c5, a2, z3); MF(c5, a3, z2); MF(c5, a4, z1);
    } else {            // digit-major: consecutive instructions never share an accumulator
      MF(c0, a0, z0); MF(c1, a0, z1); MF(c2, a0, z2); MF(c3, a0, z3); MF(c4, a0, z4);
      MF(c1, a1, z0); MF(c2, a1, z1); MF(c3, a1, z2); MF(c4, a1, z3); MF(c5, a1, z4);
      MF(c2, a2, z0); MF(c3, a2, z1); MF(c4, a2, z2); MF(c5, a2, z3);
      MF(c3, a3, z0); MF(c4, a3, z1); MF(c5, a3, z2);
      MF(c4, a4, z0); MF(c5, a4, z1);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[4] + c5[5];
}
int main() {
  long long* out; int* sink; i32x4* src;
  (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&sink, 4 * 64 * 1024); (void)hipMalloc(&src, 16 * 64 * 10);
  int host[64 * 10 * 4];
  srand(3);
  for (int& v : host) v = rand() * 65537;
  (void)hipMemcpy(src, host, sizeof host, hipMemcpyHostToDevice);
  const int iters = 2048;
  for (int trivial = 1; trivial >= 0; --trivial)
    for (int order = 0; order < 2; ++order)
      for (int blocks : {1, 1024}) {
        if (order == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters, trivial);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters, trivial);
        long long t;
        (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        printf("%-8s %-12s %4d waves: %.1f cycles per step of 19 matrix instructions (%.1f each)\n", trivial ? "trivial" : "random", order ? "digit-major" : "level-major",
               blocks, (double)t / iters, (double)t / iters / 19.0);
      }
  return 0;
}
