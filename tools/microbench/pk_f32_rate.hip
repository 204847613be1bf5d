// Issue rate of packed fp32 (v_pk_fma_f32) against scalar fp32 (v_fma_f32) on gfx950: 8 independent chains per lane, WAVES waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/pk_f32_rate.hip -o tools/microbench/pk_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
  const f2 av = {a, a + 1.f}, bv = {b, b - 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (PK) x[i] = x[i] * av + bv;                                            // one v_pk_fma_f32: two FMAs
      else { x[i].x = x[i].x * a + b; asm volatile("" : "+v"(x[i].x)); }                  // one v_fma_f32
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 16 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 400000;   // ~10 ms per launch
  // the device needs a few hundred ms of load to reach its clocks: warm up first (sub-millisecond launches on a cold device read ~2x slow)
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k<true>, dim3(512), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  for (int waves = 1; waves <= 2; ++waves)
    for (int pk = 0; pk < 2; ++pk) {
      const int grid = 256 * waves;   // 256 CUs x (4 waves = one per SIMD) x waves
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double instr = (double)iters * 8 * waves;   // per SIMD
      printf("%s, %d wave(s) per SIMD: %.2f ms, %.2f ns per wave-instruction per SIMD, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", waves, ms,
             ms * 1e6 / instr, instr * 1024 * 64 * (pk ? 4 : 2) / (ms * 1e-3) / 1e12);
    }
  return 0;
}
