// Verifies the operand layout assumed for v_mfma_f32_32x32x16_f16 on gfx950:
//   A: lane l holds A[i = l & 31][k = 8 (l >> 5) + e], e = 0..7   (8 halves = 4 VGPRs)
//   B: lane l holds B[k = 8 (l >> 5) + e][j = l & 31]
//   C/D: reg r of lane l = C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
// and times a dependent chain.  build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* A, const float* B, float* C, long long* cyc) {
  int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = (_Float16)B[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
  long long t0 = clock64();
  f32x16 d0 = c, d1 = c;
  for (int it = 0; it < 1000; ++it) {
    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
  }
  long long t1 = clock64();
  if (l == 0) cyc[0] = t1 - t0;
  C[1024 + l] = d0[0] + d1[1];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hC[1024 + 64], ref[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (rand() % 17) - 8; hB[i] = (rand() % 13) - 6; }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC; long long* dc;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dc, 8);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dc);
  long long cy; hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost); hipMemcpy(&cy, dc, 8, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
  printf("layout mismatches: %d of 1024 ; cycles per f16 32x32x16 MFMA (2 independent chains): %.1f\n", bad, cy / 2000.0);
  return bad != 0;
}
