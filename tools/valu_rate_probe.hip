// Issue rate of the fp64 vector instructions the large-graph Lanczos kernels lean on, relative to
// v_fma_f32 (4 cycles per wave64 instruction): one wave per SIMD (1024 waves), 16 independent
// chains, so neither dependencies nor occupancy hide anything.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe.bin valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096, CH = 16;

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, float seed) {
  double d[CH];
  float f[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { d[i] = seed + i + threadIdx.x; f[i] = seed + i; }
  const double m = 1.0000001, a = 1e-9;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"((float)m), "v"((float)a));
      if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(m), "v"(a));
      if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(m));
      if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
      if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if (OP == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(f[i]), "+v"(f[(i + 1) % CH]));
      if (OP == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(f[i]) : "v"(f[(i + 1) % CH]));
      if (OP == 7) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(f[i]), "+v"(f[(i + 1) % CH]));
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += d[i] + f[i];
  if (s == 1.2345) out[0] = s;
}

template <int OP>
void run(const char* name, double* out, float base_ms[1]) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<OP>), dim3(1024), dim3(64), 0, 0, out, 1.f);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int t = 0; t < 3; ++t) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), dim3(1024), dim3(64), 0, 0, out, 1.f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (OP == 0) base_ms[0] = best;
  printf("%-22s %.3f ms  = %.2f x v_fma_f32  (~%.1f cycles per wave64 instruction)\n", name, best,
         best / base_ms[0], 4.0 * best / base_ms[0]);
}

int main() {
  double* out; CK(hipMalloc(&out, 8));
  float base[1] = {1.f};
  run<0>("v_fma_f32", out, base);
  run<1>("v_fma_f64", out, base);
  run<2>("v_mul_f64", out, base);
  run<3>("v_add_f64", out, base);
  run<4>("v_cvt_f64_f32", out, base);
  run<5>("v_permlane32_swap_b32", out, base);
  run<6>("v_mov_b32_dpp", out, base);
  run<7>("v_permlane16_swap_b32", out, base);
  return 0;
}
