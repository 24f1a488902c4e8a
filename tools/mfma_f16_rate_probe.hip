// Micro-probe: issue cycles per instruction of the 16 x 16 MFMA shapes the strip kernel chooses
// between (one wave per SIMD and two waves per SIMD, four independent accumulators per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f16_rate_probe.bin tools/mfma_f16_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float x = 0.001f * threadIdx.x;
  f16x4 a4, b4;
  f16x8 a8, b8;
  for (int e = 0; e < 4; ++e) a4[e] = (_Float16)(x + e), b4[e] = (_Float16)(x - e);
  for (int e = 0; e < 8; ++e) a8[e] = (_Float16)(x + e), b8[e] = (_Float16)(x - e);
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x + 1.f, acc[i], 0, 0, 0);
        if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMallocManaged(&cyc, 8);
  const int iters = 2000;
  const char* names[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x32_f16"};
  for (int threads = 256; threads <= 512; threads += 256)
    for (int k = 0; k < 3; ++k) {
      for (int rep = 0; rep < 2; ++rep) {
        if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
      }
      printf("%-26s %d waves/SIMD: %.2f cycles per instruction per wave\n", names[k], threads / 256,
             (double)*cyc / (iters * 32.0));
    }
  return 0;
}
