// What slows v_mfma_i32_32x32x32_i8 below its 32-cycle issue rate: operand registers that change from one instruction to the next, or
// accumulator reuse distance?  One wave per SIMD.   hipcc -O2 --offload-arch=gfx950 mfma_i8_operands.hip -o mfma_i8_operands && ./mfma_i8_operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, x, y) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc, 0, 0, 0)
template <int V>
__global__ __launch_bounds__(64) void k(long long* out, int* sink, const i32x4* src, int iters) {
  i32x4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 64], a2 = src[threadIdx.x + 128], a3 = src[threadIdx.x + 192];
  i32x4 z0 = src[threadIdx.x + 320], z1 = src[threadIdx.x + 384], z2 = src[threadIdx.x + 448], z3 = src[threadIdx.x + 512];
  i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) { MF(c0, a0, z0); MF(c1, a0, z0); MF(c2, a0, z0); MF(c3, a0, z0); MF(c0, a0, z0); MF(c1, a0, z0); MF(c2, a0, z0); MF(c3, a0, z0); }
    if (V == 1) { MF(c0, a0, z0); MF(c1, a0, z1); MF(c2, a0, z2); MF(c3, a0, z3); MF(c0, a0, z0); MF(c1, a0, z1); MF(c2, a0, z2); MF(c3, a0, z3); }
    if (V == 2) { MF(c0, a0, z0); MF(c1, a1, z1); MF(c2, a2, z2); MF(c3, a3, z3); MF(c0, a1, z0); MF(c1, a2, z1); MF(c2, a3, z2); MF(c3, a0, z3); }
    if (V == 3) { MF(c0, a0, z0); MF(c1, a1, z1); MF(c2, a2, z2); MF(c3, a3, z3); MF(c4, a1, z0); MF(c5, a2, z1); MF(c6, a3, z2); MF(c7, a0, z3); }
    if (V == 4) { MF(c0, a0, z0); MF(c0, a1, z1); MF(c1, a2, z2); MF(c1, a3, z3); MF(c2, a1, z0); MF(c2, a2, z1); MF(c3, a3, z2); MF(c3, a0, z3); }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[4] + c5[5] + c6[6] + c7[7];
}
int main() {
  long long* out; int* sink; i32x4* src;
  (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&sink, 4 * 64 * 1024); (void)hipMalloc(&src, 16 * 64 * 10);
  int host[64 * 10 * 4];
  srand(3);
  for (int& v : host) v = rand() * 65537;
  (void)hipMemcpy(src, host, sizeof host, hipMemcpyHostToDevice);
  const int iters = 2048;
  const char* names[5] = {"4 accumulators, same A, same B", "4 accumulators, same A, B changes", "4 accumulators, A and B change", "8 accumulators, A and B change",
                          "4 accumulators used twice in a row, A and B change"};
  for (int v = 0; v < 5; ++v)
    for (int blocks : {1, 1024}) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters);
      if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters);
      if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters);
      if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters);
      if (v == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, sink, src, iters);
      long long t;
      (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      printf("%-52s %4d waves: %.1f cycles per instruction\n", names[v], blocks, (double)t / iters / 8.0);
    }
  return 0;
}
