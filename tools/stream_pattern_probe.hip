// HBM stream probe: does the bandwidth of a non-temporal float4 stream depend on how many
// contiguous bytes a wave reads per matrix row?  256 workgroups x 8 waves, one 2048 x 2048 fp32
// matrix per workgroup (4.3 GB in all, the shape of lnz_lanczos_ritz_large), every wave keeps 16
// float4 loads per lane (16 KiB) in flight.  W = chunks of 256 columns a wave reads per row:
//   W = 1: 1 KiB segments at an 8 KiB stride (the symmetric kernel's block jobs)
//   W = 8: whole 8 KiB rows (the full-stream kernel)
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern_probe stream_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 2048, NW = 8;
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ inline float4 ld_nt(const float* p) {
  f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int W, bool ROT, int WORK, bool DRAIN = false>
__global__ __launch_bounds__(512) void stream_kernel(const float* __restrict__ A, float* out, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* Ab = A + (size_t)blockIdx.x * N * N;
  // the wave's strip: W chunks wide, N * 8 / (NW * W) ... rows so that every wave reads N*N/NW floats
  constexpr int STRIPS = 8 / W;              // column strips per matrix
  constexpr int ROWS = N / (NW / STRIPS);    // rows per wave
  const int strip = wave % STRIPS, rb = wave / STRIPS;
  const float* base = Ab + (size_t)rb * ROWS * N + strip * (256 * W) + 4 * lane;
  constexpr int ITEMS = ROWS * W;            // float4 items of the wave, row-major in the strip
  // ROT: the waves of a workgroup walk their strips from different rows (no DRAM page shared
  // between the waves at any time — the symmetric kernel's situation)
  const int rot = ROT ? wave * (ROWS / NW) : 0;
  auto addr = [&](int k) { return base + (size_t)((k / W + rot) % ROWS) * N + (k % W) * 256; };
  __shared__ double qs[N];
  for (int i = threadIdx.x; i < N; i += 512) qs[i] = 1.0 + i;
  __syncthreads();
  double q0 = qs[4 * lane], q1 = qs[4 * lane + 1], q2 = qs[4 * lane + 2], q3 = qs[4 * lane + 3];
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, pr = 0;
  float acc = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    float4 buf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      buf[i] = ld_nt(addr(i));
    for (int k0 = 0; k0 < ITEMS; k0 += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = buf[i];
        int kn = k0 + 16 + i;
        kn = kn < ITEMS ? kn : i;  // tail: re-read the head (cached)
        buf[i] = ld_nt(addr(kn));
        // DRAIN: what hipcc emits for the symmetric kernel's predicated loads — the first row of
        // a group waits for every load in flight, the one just issued included
        if (DRAIN && i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WORK == 0) acc += v.x + v.y + v.z + v.w;
        if (WORK >= 1) {  // the symmetric SpMV's arithmetic per float4: row dot + column sums, fp64
          const double ax = v.x, ay = v.y, az = v.z, aw = v.w;
          pr += fma(ax, q0, ay * q1) + fma(az, q2, aw * q3);
          const double qr = WORK >= 2 ? qs[(k0 + i) & (N - 1)] : q0;
          c0 = fma(ax, qr, c0); c1 = fma(ay, qr, c1); c2 = fma(az, qr, c2); c3 = fma(aw, qr, c3);
        }
      }
    }
  }
  acc += (float)(c0 + c1 + c2 + c3 + pr);
  if (acc == 123.456f) out[0] = acc;
}

template <int W, bool ROT, int WORK, bool DRAIN = false>
void run(const float* A, float* out, int B) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 4;
  hipLaunchKernelGGL((stream_kernel<W, ROT, WORK, DRAIN>), dim3(B), dim3(512), 0, 0, A, out, 1);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int t = 0; t < 3; ++t) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<W, ROT, WORK, DRAIN>), dim3(B), dim3(512), 0, 0, A, out, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)B * N * N * 4 * reps;
  printf("W=%d rot=%d work=%d drain=%d (%4d B contiguous per row and wave): %.3f ms  %.0f GB/s\n", W, (int)ROT, WORK, (int)DRAIN, W * 1024, best,
         bytes / best / 1e6);
}

// Interleaved rows: wave w reads rows w, w + 8, w + 16, ... — the eight waves of a workgroup
// cover 64 KiB of consecutive memory at any time (the full-stream kernel's order).  TRI: only
// the columns from the row's own 256-column chunk on (the upper block triangle, 56 % of the bytes).
template <bool TRI>
__global__ __launch_bounds__(512) void rows_kernel(const float* __restrict__ A, float* out, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* Ab = A + (size_t)blockIdx.x * N * N;
  float acc = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    for (int I = 0; I < 8; ++I) {
      const int c0 = TRI ? I : 0;
      float4 a0[8], a1[8];
      auto load_row = [&](int r, float4 (&a)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c >= c0) a[c] = ld_nt(Ab + (size_t)r * N + 256 * c + 4 * lane);
      };
      auto use_row = [&](const float4 (&a)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c >= c0) acc += a[c].x + a[c].y + a[c].z + a[c].w;
      };
      int r = 256 * I + wave;
      load_row(r, a0);
      for (; r < 256 * I + 256; r += 16) {
        load_row(r + 8, a1);
        use_row(a0);
        if (r + 16 < 256 * I + 256) load_row(r + 16, a0);
        use_row(a1);
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <bool TRI>
void run_rows(const float* A, float* out, int B) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 4;
  hipLaunchKernelGGL((rows_kernel<TRI>), dim3(B), dim3(512), 0, 0, A, out, 1);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int t = 0; t < 3; ++t) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rows_kernel<TRI>), dim3(B), dim3(512), 0, 0, A, out, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)B * N * N * 4 * reps * (TRI ? 36.0 / 64.0 : 1.0);
  printf("interleaved rows, %s: %.3f ms  %.0f GB/s\n", TRI ? "upper block triangle" : "full rows", best,
         bytes / best / 1e6);
}

int main() {
  const int B = 256;
  float *A, *out;
  CK(hipMalloc(&A, (size_t)B * N * N * 4));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(A, 0, (size_t)B * N * N * 4));
  run<1, false, 0>(A, out, B);
  run<2, false, 0>(A, out, B);
  run<4, false, 0>(A, out, B);
  run<8, false, 0>(A, out, B);
  run<1, true, 0>(A, out, B);
  run<4, true, 0>(A, out, B);
  run<8, true, 0>(A, out, B);
  run<1, true, 1>(A, out, B);
  run<1, true, 2>(A, out, B);
  run<4, true, 2>(A, out, B);
  run<8, false, 2>(A, out, B);
  run<1, true, 2, true>(A, out, B);
  run<1, true, 0, true>(A, out, B);
  run_rows<false>(A, out, B);
  run_rows<true>(A, out, B);
  return 0;
}
