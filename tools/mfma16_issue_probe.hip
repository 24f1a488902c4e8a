// Micro-probe: whole-chip fp32 matrix rate of the GEMM1 step of a fused forward on
// v_mfma_f32_16x16x4_f32, next to the 32x32x2 step the shipped kernel uses (tools/mfma_issue_probe.hip).
// One 512-thread workgroup per CU (2 waves per SIMD); wave w owns 16 output columns of NT node
// tiles (2 row subtiles each): a 16-k step = one weight float4 per lane (register ring from L2,
// prefetch distance 3 steps), 2 NT A fragments by ds_read_b128 (pitch 136: conflict free for the
// (row = lane & 15, k quad = lane >> 4) layout) and 8 NT MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/mfma16_probe tools/mfma16_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

constexpr int P16 = 136;

// MODE bit0: weight ring from global memory, bit1: A fragments from LDS, bit2: interleaved issue
template <int NT, int MODE>
__global__ __launch_bounds__(512) void probe16(const float4* __restrict__ w, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [NT][32][P16]
  for (int i = threadIdx.x; i < NT * 32 * P16; i += blockDim.x) xs[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, kq = lane >> 4;
  f32x4 Z[NT][2];
  for (int m = 0; m < NT; ++m)
    for (int i = 0; i < 2; ++i) Z[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float4* wb = w + (size_t)(wave >> 1) * 64 * 1024 + 32 * (kq & 1) + 16 * (wave & 1) + j + 64 * (kq >> 1);
  float4 ring[4];
  for (int sl = 0; sl < 3; ++sl) ring[sl] = wb[sl * 128];
  typedef const __attribute__((address_space(3))) f32x4* l4;
  const float* xr = xs + j * P16 + 4 * kq;
  f32x4 acur[NT][2];
  for (int m = 0; m < NT; ++m)
    for (int i = 0; i < 2; ++i) acur[m][i] = *(l4)(xr + (m * 32 + 16 * i) * P16);
#pragma unroll 1
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE & 1) ring[(u + 3) & 3] = wb[((it + u + 3) & 511) * 128];
      f32x4 anext[NT][2];
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i)
          anext[m][i] = (MODE & 2) ? *(l4)(xr + (m * 32 + 16 * i) * P16 + 16 * ((it + u + 1) & 7)) : acur[m][i];
      const float4 b = ring[u];
#pragma unroll
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i) Z[m][i] = MF16(acur[m][i][0], b.x, Z[m][i]);
#pragma unroll
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i) Z[m][i] = MF16(acur[m][i][1], b.y, Z[m][i]);
#pragma unroll
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i) Z[m][i] = MF16(acur[m][i][2], b.z, Z[m][i]);
#pragma unroll
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i) Z[m][i] = MF16(acur[m][i][3], b.w, Z[m][i]);
      for (int m = 0; m < NT; ++m)
        for (int i = 0; i < 2; ++i) acur[m][i] = anext[m][i];
      if (MODE & 4) {
        // 8 NT MFMAs, 2 NT LDS reads, one global load: a load behind every fourth MFMA
#pragma unroll
        for (int g = 0; g < 2 * NT; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (g == 2 * NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    }
  }
  float s = 0;
  for (int m = 0; m < NT; ++m)
    for (int i = 0; i < 2; ++i) s += Z[m][i][0] + Z[m][i][1] + Z[m][i][2] + Z[m][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the shipped tiling: two halves of 4 waves, half 0 carries MT0 tiles, half 1 MT1 (32 columns
// per wave, 8-k steps: one weight float4, MT A fragments, 4 MT MFMAs 32x32x2)
template <int MT>
__device__ __forceinline__ void half32(const float4* __restrict__ w, const float* xs, float* out, int iters, int hw) {
  const int lane = threadIdx.x & 63;
  f32x16 Z[MT];
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 16; ++r) Z[m][r] = 0.f;
  const float4* wb = w + (size_t)hw * 64 * 1024 + lane;
  float4 ring[8];
  for (int sl = 0; sl < 7; ++sl) ring[sl] = wb[sl * 64];
  typedef const __attribute__((address_space(3))) f32x4* l4;
  const float* xr = xs + (lane & 31) * 132 + 4 * (lane >> 5);
  f32x4 acur[MT];
  for (int m = 0; m < MT; ++m) acur[m] = *(l4)(xr + m * 32 * 132);
#pragma unroll 1
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      ring[(u + 7) & 7] = wb[((it + u + 7) & 1023) * 64];
      f32x4 anext[MT];
      for (int m = 0; m < MT; ++m) anext[m] = *(l4)(xr + m * 32 * 132 + 8 * ((it + u + 1) & 15));
      const float4 b = ring[u];
      for (int m = 0; m < MT; ++m) Z[m] = MF32(acur[m][0], b.x, Z[m]);
      for (int m = 0; m < MT; ++m) Z[m] = MF32(acur[m][1], b.y, Z[m]);
      for (int m = 0; m < MT; ++m) Z[m] = MF32(acur[m][2], b.z, Z[m]);
      for (int m = 0; m < MT; ++m) Z[m] = MF32(acur[m][3], b.w, Z[m]);
      for (int m = 0; m < MT; ++m) acur[m] = anext[m];
      if (MT == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
  }
  float s = 0;
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 16; ++r) s += Z[m][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MT0, int MT1>
__global__ __launch_bounds__(512) void probe32(const float4* __restrict__ w, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [2 halves][2][32][132]
  for (int i = threadIdx.x; i < 4 * 32 * 132; i += blockDim.x) xs[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int half = threadIdx.x >> 8, hw = (threadIdx.x >> 6) & 3;
  if (half == 0) half32<MT0>(w, xs, out, iters, hw);
  else if (MT1 > 0) half32<(MT1 > 0 ? MT1 : 1)>(w, xs + 2 * 32 * 132, out, iters, hw);
}

template <typename K>
void run(const char* name, K kern, size_t lds, double tiles, int k_per_iter, const float4* w, float* out) {
  const int iters = 8192;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, w, out, 128);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, w, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // flops: tiles x 32 rows x 128 columns x (iters x k_per_iter) x 2 per workgroup
  const double tf = 256.0 * tiles * 32 * 128 * 2.0 * iters * k_per_iter / (ms * 1e-3) / 1e12;
  printf("%-44s wall=%.3f ms  TF=%.1f  (%.3f of 157.3)\n", name, ms, tf, tf / 157.3);
}

int main() {
  float4* w; float* out;
  hipMalloc(&w, 4 * 64 * 1024 * sizeof(float4)); hipMemset(w, 0, 4 * 64 * 1024 * sizeof(float4));
  hipMalloc(&out, 256 * 512 * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x2 halves 2|1 (shipped)", probe32<2, 1>, 4 * 32 * 132 * 4, 3, 8, w, out);
    run("32x32x2 halves 2|2", probe32<2, 2>, 4 * 32 * 132 * 4, 4, 8, w, out);
    run("32x32x2 halves 1|1", probe32<1, 1>, 4 * 32 * 132 * 4, 2, 8, w, out);
    run("32x32x2 halves 2|0", probe32<2, 0>, 4 * 32 * 132 * 4, 2, 8, w, out);
    run("16x16x4 NT=3 ring+lds", probe16<3, 3>, 3 * 32 * P16 * 4, 3, 16, w, out);
    run("16x16x4 NT=3 ring+lds interleaved", probe16<3, 7>, 3 * 32 * P16 * 4, 3, 16, w, out);
    run("16x16x4 NT=3 lds only interleaved", probe16<3, 6>, 3 * 32 * P16 * 4, 3, 16, w, out);
    run("16x16x4 NT=3 mfma only", probe16<3, 0>, 3 * 32 * P16 * 4, 3, 16, w, out);
    run("16x16x4 NT=2 ring+lds interleaved", probe16<2, 7>, 2 * 32 * P16 * 4, 2, 16, w, out);
    run("16x16x4 NT=4 ring+lds interleaved", probe16<4, 7>, 4 * 32 * P16 * 4, 4, 16, w, out);
    run("16x16x4 NT=1 ring+lds interleaved", probe16<1, 7>, 1 * 32 * P16 * 4, 1, 16, w, out);
  }
  return 0;
}
