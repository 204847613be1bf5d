// Microbenchmark: issue cost (cycles per wave-instruction) of v_sin_f32 / v_cos_f32 / v_exp_f32 / v_fma_f32 / v_pk_fma_f32 and of
// sin+fma interleaved, one wave per SIMD and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 1) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    if (MODE == 4) { REP16(asm volatile("v_sin_f32 %0, %0\n v_fma_f32 %2, %2, %2, %2\n v_cos_f32 %1, %1\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 5) { REP16(asm volatile("v_sin_f32 %0, %0\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %1, %1, %1, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 6) { REP16(asm volatile("v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (MODE == 7) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rsq_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 8);
  const int iters = 200;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / (iters * 64.0);
  printf("%-28s %4d thr/WG (%d waves/SIMD): %.2f clock64 ticks per instr per wave -> x waves/SIMD = %.2f\n", name, threads, threads / 256,
         per, per / (threads / 256.0) );
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int thr : {256, 512, 768, 1024}) {
    run<0>("v_sin_f32", thr); run<6>("v_cos_f32", thr); run<1>("v_exp_f32", thr); run<7>("rcp/rsq/sqrt/log", thr); run<2>("v_fma_f32", thr); run<3>("v_pk_fma_f32", thr);
    run<4>("sin,fma,cos,fma", thr); run<5>("sin,fma,fma,fma", thr);
  }
  return 0;
}
