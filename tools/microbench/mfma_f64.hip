// f64 matrix-core rate on gfx950: N dependent-free v_mfma_f64_16x16x4_f64 per wave, W waves per SIMD; reports cycles (s_memtime) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, int iters, int chains) {
  f64x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  long long w0 = wall_clock64();
  if (chains == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    }
  } else if (chains == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
      }
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
      }
    }
  }
  long long t1 = clock64();
  long long w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
int main(int argc, char** argv) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 1024 * 1024);
  (void)hipMalloc(&cyc, 16);
  for (int grid : {1, 256, 1024})
    for (int threads : {64, 256, 512, 1024})
      for (int chains : {1, 2, 4}) {
        const int iters = 256;   // 4096 matrix ops per wave
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, out, cyc, iters, chains);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, out, cyc, iters, chains);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double n = 16.0 * iters, waves_per_simd = threads / 256.0 > 1 ? threads / 256.0 : 1.0;
        printf("grid %4d threads %4d chains %d: %7.1f clk64/op/wave  %6.1f ns/op/wave (wall)  -> per SIMD op every %6.1f ns; kernel %.3f ms, %.1f TFLOP/s\n", grid, threads,
               chains, h[0] / n, h[1] * 10.0 / n, h[1] * 10.0 / n / waves_per_simd, ms, 2048.0 * n * (threads / 64) * grid / (ms * 1e-3) / 1e12);
      }
  return 0;
}
