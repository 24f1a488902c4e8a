// Micro-probe: cycles per v_mfma_f32_32x32x2_f32 for the issue patterns used by the kernels.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/mfma_probe tools/mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE bit0: global ring loads (prefetch distance 3), bit1: ds_read of next A, bit2: sched_barriers,
// bit3: A operand scaled on the VALU before every MFMA (the eigen-space GEMM1's gain scaling)
template <int MODE>
__global__ __launch_bounds__(128) void probe(const float4* __restrict__ w, float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float xs[32 * 132];
  for (int i = threadIdx.x; i < 32 * 132; i += blockDim.x) xs[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 Z[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) Z[i][r] = 0.f;
  const float4* wb0 = w + lane;
  const float4* wb1 = w + 64 * 1024 + lane;
  float4 ring[4][2];
  for (int sl = 0; sl < 4; ++sl) { ring[sl][0] = wb0[sl * 64]; ring[sl][1] = wb1[sl * 64]; }
  const float* xr = &xs[(lane & 31) * 132 + 4 * (lane >> 5)];
  float4 acur = *reinterpret_cast<const float4*>(xr);
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      float4 anext = acur;
      if (MODE & 1) {
        int gi = (it + u4 + 3) & 1023;
        ring[(u4 + 3) & 3][0] = wb0[gi * 64];
        ring[(u4 + 3) & 3][1] = wb1[gi * 64];
      }
      if (MODE & 2) anext = *reinterpret_cast<const float4*>(xr + 8 * ((it + u4 + 1) & 15));
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
      float4 acur2 = acur;
      if (MODE & 8) {
        const float g0 = out[lane], g1 = out[lane + 64];  // two "gains" (tile 0, tile 1)
        acur.x *= g0; acur.y *= g0; acur.z *= g0; acur.w *= g0;
        acur2.x *= g1; acur2.y *= g1; acur2.z *= g1; acur2.w *= g1;
      }
      if (MODE & 8) {
        Z[0] = MF(acur.x, ring[u4][0].x, Z[0]); Z[1] = MF(acur2.x, ring[u4][0].x, Z[1]);
        Z[0] = MF(acur.y, ring[u4][0].y, Z[0]); Z[1] = MF(acur2.y, ring[u4][0].y, Z[1]);
        Z[0] = MF(acur.z, ring[u4][0].z, Z[0]); Z[1] = MF(acur2.z, ring[u4][0].z, Z[1]);
        Z[0] = MF(acur.w, ring[u4][0].w, Z[0]); Z[1] = MF(acur2.w, ring[u4][0].w, Z[1]);
      } else {
      Z[0] = MF(acur.x, ring[u4][0].x, Z[0]); Z[1] = MF(acur.x, ring[u4][1].x, Z[1]);
      Z[0] = MF(acur.y, ring[u4][0].y, Z[0]); Z[1] = MF(acur.y, ring[u4][1].y, Z[1]);
      Z[0] = MF(acur.z, ring[u4][0].z, Z[0]); Z[1] = MF(acur.z, ring[u4][1].z, Z[1]);
      Z[0] = MF(acur.w, ring[u4][0].w, Z[0]); Z[1] = MF(acur.w, ring[u4][1].w, Z[1]);
      }
      acur = anext;
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += Z[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, const float4* w, float* out, long long* cyc) {
  int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(128), 0, 0, w, out, cyc, 128);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(128), 0, 0, w, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double per = (double)h[0] / (iters * 8.0);
  double tf = (double)blocks * 2 * iters * 8.0 * 4096 / (ms * 1e-3) / 1e12;
  printf("%-34s blocks=%5d  clock64/mfma(wave0)=%6.1f  wall=%.3f ms  TF=%.1f\n", name, blocks, per, ms, tf);
}

int main() {
  float4* w; float* out; long long* cyc;
  hipMalloc(&w, 2 * 64 * 1024 * sizeof(float4)); hipMemset(w, 0, 2 * 64 * 1024 * sizeof(float4));
  hipMalloc(&out, 8192 * 128 * sizeof(float)); hipMalloc(&cyc, 8192 * sizeof(long long));
  for (int blocks : {256, 1024}) {
    run<0>("mfma only", blocks, w, out, cyc);
    run<1>("ring loads", blocks, w, out, cyc);
    run<2>("ds_read", blocks, w, out, cyc);
    run<3>("ring + ds_read", blocks, w, out, cyc);
    run<5>("ring, sched_barrier", blocks, w, out, cyc);
    run<6>("ds_read, sched_barrier", blocks, w, out, cyc);
    run<7>("ring + ds_read, sched_barrier", blocks, w, out, cyc);
    run<4>("mfma only, sched_barrier", blocks, w, out, cyc);
    run<15>("ring + ds_read + scaled A, sched_b", blocks, w, out, cyc);
    run<14>("ds_read + scaled A, sched_barrier", blocks, w, out, cyc);
  }
  return 0;
}
