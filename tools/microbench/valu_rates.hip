// Microbenchmark: THROUGHPUT (cycles per wave-instruction per SIMD) of v_sin_f32 / v_exp_f32 / v_fma_f32 / v_pk_fma_f32 and mixes,
// 16 independent register chains per wave, 1..4 waves per SIMD.  Time base: wall_clock64 (100 MHz) x assumed 2.4 GHz shader clock.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP16(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n" \
  op " %8, %8\n" op " %9, %9\n" op " %10, %10\n" op " %11, %11\n" op " %12, %12\n" op " %13, %13\n" op " %14, %14\n" op " %15, %15" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
#define FMA16(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7\n" \
  op " %8, %8, %8, %8\n" op " %9, %9, %9, %9\n" op " %10, %10, %10, %10\n" op " %11, %11, %11, %11\n" op " %12, %12, %12, %12\n" op " %13, %13, %13, %13\n" op " %14, %14, %14, %14\n" op " %15, %15, %15, %15" \
  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
template <int MODE>
__global__ void k(float* out, long long* ticks, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { OP16("v_sin_f32"); OP16("v_sin_f32"); }
    if (MODE == 1) { OP16("v_exp_f32"); OP16("v_exp_f32"); }
    if (MODE == 2) { FMA16("v_fma_f32"); FMA16("v_fma_f32"); }
    if (MODE == 3) {
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0\nv_pk_fma_f32 %1, %1, %1, %1\nv_pk_fma_f32 %2, %2, %2, %2\nv_pk_fma_f32 %3, %3, %3, %3\n"
                   "v_pk_fma_f32 %4, %4, %4, %4\nv_pk_fma_f32 %5, %5, %5, %5\nv_pk_fma_f32 %6, %6, %6, %6\nv_pk_fma_f32 %7, %7, %7, %7\n"
                   "v_pk_fma_f32 %0, %0, %0, %0\nv_pk_fma_f32 %1, %1, %1, %1\nv_pk_fma_f32 %2, %2, %2, %2\nv_pk_fma_f32 %3, %3, %3, %3\n"
                   "v_pk_fma_f32 %4, %4, %4, %4\nv_pk_fma_f32 %5, %5, %5, %5\nv_pk_fma_f32 %6, %6, %6, %6\nv_pk_fma_f32 %7, %7, %7, %7"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0\nv_pk_fma_f32 %1, %1, %1, %1\nv_pk_fma_f32 %2, %2, %2, %2\nv_pk_fma_f32 %3, %3, %3, %3\n"
                   "v_pk_fma_f32 %4, %4, %4, %4\nv_pk_fma_f32 %5, %5, %5, %5\nv_pk_fma_f32 %6, %6, %6, %6\nv_pk_fma_f32 %7, %7, %7, %7\n"
                   "v_pk_fma_f32 %0, %0, %0, %0\nv_pk_fma_f32 %1, %1, %1, %1\nv_pk_fma_f32 %2, %2, %2, %2\nv_pk_fma_f32 %3, %3, %3, %3\n"
                   "v_pk_fma_f32 %4, %4, %4, %4\nv_pk_fma_f32 %5, %5, %5, %5\nv_pk_fma_f32 %6, %6, %6, %6\nv_pk_fma_f32 %7, %7, %7, %7"
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
    }
    if (MODE == 4) { OP16("v_sin_f32"); FMA16("v_fma_f32"); }   // half transcendental, half FMA
  }
  const long long t1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* tk;
  hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&tk, 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, tk, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, tk, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, tk, 8, hipMemcpyDeviceToHost);
  const double ns = (double)c * 10.0, waves = threads / 256.0;
  const double per_simd = ns * 2.4 / (iters * 32.0 * waves);   // shader cycles (at 2.4 GHz) per wave-instruction per SIMD
  printf("%-22s %d wave(s)/SIMD: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, threads / 256, per_simd);
  hipFree(out); hipFree(tk);
}
int main() {
  for (int thr : {256, 512, 1024}) {
    run<0>("v_sin_f32", thr); run<1>("v_exp_f32", thr); run<2>("v_fma_f32", thr); run<3>("v_pk_fma_f32", thr); run<4>("sin + fma (1:1)", thr);
  }
  return 0;
}
