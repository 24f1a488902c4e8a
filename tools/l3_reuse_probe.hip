// Infinity-Cache (L3, 256 MiB) reuse probe for the config-5 Lanczos: is a
// working set of <= 200 MB that ALL 256 compute units re-stream served faster than the HBM stream
// the one-workgroup-per-graph kernel runs at (6.3 - 6.7 TB/s)?
//
// A working set of W bytes is read `passes` times inside ONE launch by 256 x wgs_per_cu workgroups
// of 512 threads; a workgroup owns W / nblocks contiguous bytes (the row block of "its" graph),
// every wave keeps 16 float4 loads per lane in flight (the ring of lnz_lanczos_ritz_large_sym).
//   fixed   : a workgroup re-reads ITS slice every pass (the cooperative kernel's pattern; the slice
//             per XCD, W / 8, is beyond the 4 MiB L2 for W > 32 MiB, so a hit can only be L3)
//   rotated : pass p reads slice (b + 37 p) mod nblocks (no L2 locality even for small W)
//   nt      : non-temporal loads (what the streamed kernel uses) vs the default policy
// W = 4 GiB is the HBM reference (nothing survives), W <= 16 MiB the L2 one.
// Build: hipcc --offload-arch=gfx950 -O3 -o l3_reuse_probe l3_reuse_probe.hip ; prints one JSON line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline f4v ld(const f4v* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

template <bool NT>
__global__ __launch_bounds__(512) void reuse_kernel(const f4v* __restrict__ A, size_t items_per_block,
                                                     int nblocks, int passes, int rot, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = items_per_block / 8;   // float4 items of a wave, contiguous 1 KiB rows of 64 lanes
  float acc = 0.f;
  for (int p = 0; p < passes; ++p) {
    const int slice = (int)(((size_t)blockIdx.x + (size_t)rot * p) % (size_t)nblocks);
    const f4v* base = A + (size_t)slice * items_per_block + (size_t)wave * per_wave + lane;
    f4v buf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) buf[i] = ld<NT>(base + (size_t)i * 64);
    for (size_t k0 = 0; k0 < per_wave; k0 += 16 * 64) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f4v v = buf[i];
        size_t kn = k0 + (size_t)(16 + i) * 64;
        kn = kn < per_wave ? kn : (size_t)i * 64;
        buf[i] = ld<NT>(base + kn);
        acc += v.x + v.y + v.z + v.w;
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t MiB = 1u << 20;
  const size_t sizes[] = {8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 96 * MiB, 128 * MiB, 160 * MiB, 192 * MiB,
                          224 * MiB, 256 * MiB, 384 * MiB, 512 * MiB, 1024 * MiB, 4096 * MiB};
  const size_t maxW = 4096 * MiB;
  f4v* A;
  float* out;
  CK(hipMalloc(&A, maxW));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(A, 0, maxW));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wgs = 1; wgs <= 2; ++wgs)
    for (int nt = 0; nt < 2; ++nt)
      for (int rot = 0; rot <= 37; rot += 37)
        for (size_t W : sizes) {
          const int nblocks = 256 * wgs;
          // a block's slice: a multiple of 8 waves x 16 x 64 float4 = 128 KiB
          size_t items = W / 16 / nblocks;
          items -= items % (8 * 16 * 64);
          if (items == 0) continue;
          const size_t bytes_pass = items * 16 * nblocks;
          const int passes = (int)(bytes_pass >= 1024 * MiB ? 4 : 16);
          float best = 1e30f;
          for (int rep = 0; rep < 4; ++rep) {   // rep 0 warms the caches (first touch comes from HBM)
            CK(hipEventRecord(e0));
            if (nt)
              hipLaunchKernelGGL(reuse_kernel<true>, dim3(nblocks), dim3(512), 0, 0, A, items, nblocks, passes, rot, out);
            else
              hipLaunchKernelGGL(reuse_kernel<false>, dim3(nblocks), dim3(512), 0, 0, A, items, nblocks, passes, rot, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
          }
          printf("{\"wgs_per_cu\": %d, \"nt\": %d, \"pattern\": \"%s\", \"working_set_MiB\": %.1f, \"passes\": %d, "
                 "\"ms\": %.4f, \"TB_per_s\": %.3f}\n",
                 wgs, nt, rot ? "rotated" : "fixed", bytes_pass / (double)MiB, passes, best,
                 (double)bytes_pass * passes / (best * 1e-3) / 1e12);
          fflush(stdout);
        }
  return 0;
}
