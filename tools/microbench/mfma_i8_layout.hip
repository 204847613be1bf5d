// Operand lane map of v_mfma_i32_32x32x32_i8 checked with exact integer data: lane l holds A[row l & 31][k = 16 (l >> 5) + b] and
// B[k = 16 (l >> 5) + b][col l & 31] in byte b = 0..15 of its 4-register fragment IF this prints "natural k order: ok"; the sliced
// extrusion kernel only needs the weaker property that byte b of lane group g of A meets byte b of lane group g of B ("paired: ok").
//   hipcc -O2 --offload-arch=gfx950 mfma_i8_layout.hip -o mfma_i8_layout && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const i32x4* a, const i32x4* b, int* c) {
  i32x16 acc = {};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) c[threadIdx.x * 16 + r] = acc[r];
}
int main() {
  std::vector<signed char> A(32 * 32), B(32 * 32);   // A[i][k], B[k][j]
  srand(7);
  for (auto& v : A) v = (signed char)(rand() % 129 - 64);
  for (auto& v : B) v = (signed char)(rand() % 129 - 64);
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: natural order k = 16 g + b; variant 1: an arbitrary permutation of k applied to BOTH operands (must give the same product)
    int perm[32];
    for (int i = 0; i < 32; ++i) perm[i] = variant ? (i * 7 + 3) % 32 : i;
    std::vector<signed char> fa(64 * 16), fb(64 * 16);
    for (int l = 0; l < 64; ++l)
      for (int bb = 0; bb < 16; ++bb) {
        const int kk = perm[16 * (l >> 5) + bb];
        fa[l * 16 + bb] = A[(l & 31) * 32 + kk];
        fb[l * 16 + bb] = B[kk * 32 + (l & 31)];
      }
    void *da, *db, *dc;
    (void)hipMalloc(&da, 1024); (void)hipMalloc(&db, 1024); (void)hipMalloc(&dc, 64 * 16 * 4);
    (void)hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x4*)da, (const i32x4*)db, (int*)dc);
    std::vector<int> c(64 * 16);
    (void)hipMemcpy(c.data(), dc, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int ref = 0;
        for (int kk = 0; kk < 32; ++kk) ref += (int)A[row * 32 + kk] * (int)B[kk * 32 + col];
        bad += ref != c[l * 16 + r];
      }
    printf("%s: %s (%d wrong of 1024)\n", variant ? "paired (same permutation of k on both operands)" : "natural k order", bad ? "WRONG" : "ok", bad);
  }
  return 0;
}
