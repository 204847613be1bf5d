// How exactly does v_mfma_f32_32x32x16_f16 sum its 16 exact products (+ C)?  Random f16 operands; every output compared with the float64
// sum.  Reports, in units of 2^-24 x sum|terms| (half an fp32 ulp of the largest possible result is 0.5 there): mean signed error (a bias
// means truncation, not rounding), rms error, worst error — for C = 0, for C of the size of the sum, and for a chain of 8 dependent
// instructions (one accumulator).  hipcc --offload-arch=gfx950 -O2 mfma_f16_accuracy.hip -o mfma_f16_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// A [chain][32 rows][16 k], B [chain][16 k][32 cols] as plain arrays; lane l: row / col l & 31, k = 8 (l >> 5) + j
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int chain) {
  const int l = threadIdx.x;
  f16v d;
  for (int r = 0; r < 16; ++r) d[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
  for (int c = 0; c < chain; ++c) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = A[((size_t)c * 32 + (l & 31)) * 16 + 8 * (l >> 5) + j];
      b[j] = B[((size_t)c * 16 + 8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[r];
}
int main() {
  const int CH = 8;
  _Float16 *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, CH * 512 * 2); hipMalloc(&dB, CH * 512 * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
  srand(5);
  for (int mode = 0; mode < 4; ++mode) {   // 0: C = 0, one instruction; 1: C ~ sum; 2: chain of 8, C = 0; 3: one instruction, all products positive
    const int chain = mode == 2 ? CH : 1;
    double se = 0, se2 = 0, worst = 0; long n = 0;
    for (int trial = 0; trial < 200; ++trial) {
      std::vector<_Float16> A(CH * 512), B(CH * 512); std::vector<float> C(1024, 0.f), D(1024);
      for (auto& v : A) v = (_Float16)((rand() / (double)RAND_MAX) * (mode == 3 ? 1.0 : 2.0) - (mode == 3 ? 0.0 : 1.0));
      for (auto& v : B) v = (_Float16)((rand() / (double)RAND_MAX) * (mode == 3 ? 1.0 : 2.0) - (mode == 3 ? 0.0 : 1.0));
      if (mode == 1) for (auto& v : C) v = (float)((rand() / (double)RAND_MAX) * 8.0 - 4.0);
      hipMemcpy(dA, A.data(), CH * 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), CH * 1024, hipMemcpyHostToDevice);
      hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, chain);
      hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ex = C[i * 32 + j], mag = std::fabs(ex);
          for (int c = 0; c < chain; ++c)
            for (int kk = 0; kk < 16; ++kk) {
              const double p = (double)A[(c * 32 + i) * 16 + kk] * (double)B[(c * 16 + kk) * 32 + j];
              ex += p; mag += std::fabs(p);
            }
          const double err = ((double)D[i * 32 + j] - ex) / (mag * std::ldexp(1.0, -24));
          se += err; se2 += err * err; worst = std::max(worst, std::fabs(err)); ++n;
        }
    }
    printf("mode %d (%s): mean %+.4f  rms %.4f  worst %.3f   [x 2^-24 sum|terms|]\n", mode,
           mode == 0 ? "1 instr, C = 0" : mode == 1 ? "1 instr, C ~ sum" : mode == 2 ? "chain of 8" : "1 instr, positive products", se / n, std::sqrt(se2 / n), worst);
  }
  return 0;
}
