// Backward of the readout head (model/lanczos_net.py:185-194; runner/qm8_runner.py:247 reaches it
// through loss.backward()):
//   z = W_o x + b_o,  a = w_g x + b_g,  y = z * sigmoid(a),  score_b = mean over the masked nodes of y
// Given dL/dscore [B,P] and the stored last conv state X_L [B,32,128] (post ReLU), ONE launch gives
//   dY_L    [B,32,128]  dL/d(pre-activation of the last conv layer) = (dz W_o + da w_g) * [x > 0]
//                       (+ its copy in the compact row numbering of the message matrix)
//   dW_head [P+1,128], db_head [P+1]   (output rows, then the gate row)
//   db_L    [128]                      column sums of dY_L: the last conv layer's bias gradient
// Until r05 this was torch autograd on the stored state: ~45 launches of 4-5 us each, a 17-column
// library GEMM over 32 k rows (85 us) and a column-sum reduction (71 us) — 0.45 ms of a 3.6 ms step
// for 0.4 GFLOP.  Here a workgroup walks its share of the molecules with the head weights in LDS,
// keeps its parameter-gradient partials in registers and writes them once; a second tiny launch adds
// the partials in workgroup order (deterministic: no atomics).
#include "common.hpp"

namespace {

constexpr int DH = 128;
constexpr int XP = 132;     // row pitch of the state / weight tiles in LDS
constexpr int ZP = 36;      // row pitch of z / dz (>= 32 + 4, rows 16-byte aligned)
constexpr int PMAX = 31;    // head width (score columns), + 1 gate row

struct HeadArgs {
  const float* X;           // [B,32,128]
  const uint8_t* mask;      // [B,N]
  const float* gscore;      // [B,P]
  const float* W;           // [P,128] output rows (+ the gate row behind them when Wgate is NULL)
  const float* bias;        // [P] (+ 1)
  const float* Wgate;       // [1,128] or NULL
  const float* bgate;       // [1] or NULL
  const int64_t* row_off;   // [B] or NULL
  float* dY;                // [B,32,128]
  float* dYc;               // [R,128] or NULL
  float* part;              // [n_wg][(P+1)*128 + 32 + 128]
  int B, N, P, n_wg;
};

// P1T: compile-time bound of the head rows P + 1 (17: the QM8 head of 16 properties + the gate row;
// 32: any); EXACT: P + 1 == P1T, the loops over the head rows carry no run-time predicate
template <int P1T, bool EXACT>
__global__ __launch_bounds__(256, 2) void head_backward_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[32 * XP];
  __shared__ __attribute__((aligned(16))) float Ws[(PMAX + 1) * XP];
  __shared__ __attribute__((aligned(16))) float dzs[32 * ZP];
  __shared__ float zs[32 * ZP], bs[PMAX + 1], gss[PMAX + 1];
  __shared__ int extent_s;
  const int tid = threadIdx.x, P = a.P, P1 = EXACT ? P1T : a.P + 1;
  constexpr int P1Q = (P1T + 3) / 4;               // float4 groups of a dz row
  constexpr int NDW = ((P1T + 1) / 2 + 3) / 4 * 4;  // head rows per thread in the weight-gradient pass:
                                                     // thread half rh owns rows [NDW rh, NDW rh + NDW)
  constexpr int NZ = (P1T + 7) / 8;    // ... in the forward recomputation
  const int c = tid & 127, rh = tid >> 7;
  for (int i = tid; i < P1 * 32; i += 256) {
    const int o = i >> 5, k4 = i & 31;
    const float* src = (a.Wgate && o == P) ? a.Wgate : a.W + o * DH;
    *reinterpret_cast<float4*>(&Ws[o * XP + 4 * k4]) = *reinterpret_cast<const float4*>(src + 4 * k4);
  }
  if (tid < P1) bs[tid] = (a.bgate && tid == P) ? a.bgate[0] : a.bias[tid];
  for (int i = tid; i < 32 * ZP; i += 256) dzs[i] = 0.0f;   // (the columns past P + 1 stay zero)
  // this thread's parameter-gradient partials: dW[o][c] for its half's head rows
  float dw[NDW];
#pragma unroll
  for (int i = 0; i < NDW; ++i) dw[i] = 0.0f;
  float dbias = 0.0f;   // column c of dY over rows [16 rh, 16 rh + 16) of every molecule
  float dbo = 0.0f;     // tid < P1: sum of dz[.][tid]
  __syncthreads();
  // the state tile of a molecule: four float4 per thread, the NEXT molecule's in flight under the
  // arithmetic of the current one
  const float4* X4 = reinterpret_cast<const float4*>(a.X);
  float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, x2 = x0, x3 = x0;
#define LNZ_HEAD_FETCH(b_)                                         \
  {                                                                \
    const float4* Xb_ = X4 + (int64_t)(b_) * (32 * DH / 4) + tid;  \
    x0 = Xb_[0], x1 = Xb_[256], x2 = Xb_[512], x3 = Xb_[768];      \
  }
  if ((int)blockIdx.x < a.B) LNZ_HEAD_FETCH(blockIdx.x)
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // ---- the molecule's state tile, incoming gradient and node mask
    {
      const int r = tid >> 5, k4 = tid & 31;   // float4 index tid + 256 i = row r + 8 i, group k4
      *reinterpret_cast<float4*>(&Xs[r * XP + 4 * k4]) = x0;
      *reinterpret_cast<float4*>(&Xs[(r + 8) * XP + 4 * k4]) = x1;
      *reinterpret_cast<float4*>(&Xs[(r + 16) * XP + 4 * k4]) = x2;
      *reinterpret_cast<float4*>(&Xs[(r + 24) * XP + 4 * k4]) = x3;
    }
    if (tid >= 64 && tid < 64 + P) gss[tid - 64] = a.gscore[(int64_t)b * P + tid - 64];
    bool real = false;
    if (tid < 64) {
      real = tid < a.N && a.mask[(int64_t)b * a.N + tid] != 0;
      const unsigned long long m = __ballot(real);
      if (tid == 0) extent_s = m ? 64 - __builtin_clzll(m) : 0;
    }
    if (b + (int)gridDim.x < a.B) LNZ_HEAD_FETCH(b + gridDim.x)
    __syncthreads();
    // ---- z[r][o] = b_o + <W_o, x_r>: thread (r = tid / 8, o = tid % 8 + 8 i)
    {
      const int r = tid >> 3, og = tid & 7;
      float acc[NZ];
#pragma unroll
      for (int i = 0; i < NZ; ++i) acc[i] = 0.0f;
      const float4* xr = reinterpret_cast<const float4*>(&Xs[r * XP]);
#pragma unroll 4
      for (int k4 = 0; k4 < 32; ++k4) {
        const float4 x = xr[k4];
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
          const int o = og + 8 * i;
          if (o < P1) {
            const float4 w = *reinterpret_cast<const float4*>(&Ws[o * XP + 4 * k4]);
            acc[i] = fmaf(x.x, w.x, fmaf(x.y, w.y, fmaf(x.z, w.z, fmaf(x.w, w.w, acc[i]))));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        const int o = og + 8 * i;
        if (o < P1) zs[r * ZP + o] = acc[i] + bs[o];
      }
    }
    __syncthreads();
    // ---- per node row: dz_o = (g_o / n) sigmoid(a), da = sum_o (g_o / n) z_o * s (1 - s)
    if (tid < 32) {
      const unsigned long long m = __ballot(real);
      const float inv = 1.0f / (float)__popcll(m);
      float dg = 0.0f;
      const float s = 1.0f / (1.0f + expf(-zs[tid * ZP + P]));
      for (int o = 0; o < P; ++o) {
        const float dyo = real ? gss[o] * inv : 0.0f;
        dzs[tid * ZP + o] = dyo * s;
        dg = fmaf(dyo, zs[tid * ZP + o], dg);
      }
      dzs[tid * ZP + P] = real ? dg * s * (1.0f - s) : 0.0f;
    }
    __syncthreads();
    // ---- dX = dz W, through the ReLU of the last conv layer; thread (column c, row half rh)
    {
      float wcol[4 * P1Q];
#pragma unroll
      for (int o = 0; o < 4 * P1Q; ++o) wcol[o] = (o < P1T && (EXACT || o < P1)) ? Ws[o * XP + c] : 0.0f;
      const int ext = extent_s;
      const int64_t crow = a.row_off ? a.row_off[b] : 0;
#pragma unroll 2
      for (int r = 16 * rh; r < 16 * rh + 16; ++r) {
        float dx = 0.0f;
        const float4* dzr = reinterpret_cast<const float4*>(&dzs[r * ZP]);   // broadcast reads, four rows of the head at a time
#pragma unroll
        for (int q = 0; q < P1Q; ++q) {
          const float4 d = dzr[q];
          dx = fmaf(d.x, wcol[4 * q], fmaf(d.y, wcol[4 * q + 1], fmaf(d.z, wcol[4 * q + 2], fmaf(d.w, wcol[4 * q + 3], dx))));
        }
        const float g = Xs[r * XP + c] > 0.0f ? dx : 0.0f;
        a.dY[((int64_t)b * 32 + r) * DH + c] = g;
        if (a.dYc && r < ext) a.dYc[(crow + r) * DH + c] = g;
        dbias += g;
      }
    }
    // ---- parameter gradients: dW[o][c] += sum_r dz[r][o] x[r][c], db[o] += sum_r dz[r][o]
#pragma unroll 2
    for (int r = 0; r < 32; ++r) {
      const float x = Xs[r * XP + c];
      const float4* dzr = reinterpret_cast<const float4*>(&dzs[r * ZP + NDW * rh]);
#pragma unroll
      for (int q = 0; q < NDW / 4; ++q) {
        const float4 d = dzr[q];   // (zero past the head's rows)
        dw[4 * q] = fmaf(d.x, x, dw[4 * q]);
        dw[4 * q + 1] = fmaf(d.y, x, dw[4 * q + 1]);
        dw[4 * q + 2] = fmaf(d.z, x, dw[4 * q + 2]);
        dw[4 * q + 3] = fmaf(d.w, x, dw[4 * q + 3]);
      }
    }
    if (tid < P1) {
      float sacc = 0.0f;
      for (int r = 0; r < 32; ++r) sacc += dzs[r * ZP + tid];
      dbo += sacc;
    }
    __syncthreads();
  }
#undef LNZ_HEAD_FETCH
  // ---- this workgroup's partials: [P1][128] weights | [32] head biases | [128] column sums of dY
  float* out = a.part + (int64_t)blockIdx.x * (P1 * DH + 32 + DH);
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int o = NDW * rh + i;
    if (o < P1) out[o * DH + c] = dw[i];
  }
  if (tid < 32) out[P1 * DH + tid] = tid < P1 ? dbo : 0.0f;
  Xs[tid] = dbias;   // (the loop's last barrier is behind us: Xs is free)
  __syncthreads();
  if (tid < DH) out[P1 * DH + 32 + tid] = Xs[tid] + Xs[tid + 128];
}

// out[i] = sum over the workgroups' partials, in workgroup order
__global__ void head_backward_reduce_kernel(const float* __restrict__ part, int n_wg, int per_wg, int P1,
                                            float* __restrict__ dW, float* __restrict__ db,
                                            float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_wg) return;
  float s = 0.0f;
  int w = 0;
  for (; w + 8 <= n_wg; w += 8) {   // eight loads in flight, added in workgroup order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(w + u) * per_wg + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; w < n_wg; ++w) s += part[(int64_t)w * per_wg + i];
  if (i < P1 * DH) dW[i] = s;
  else if (i < P1 * DH + 32) { if (i - P1 * DH < P1) db[i - P1 * DH] = s; }
  else dbias[i - P1 * DH - 32] = s;
}

// Node extents (last real node + 1) of a batch, their exclusive prefix sums (the compact row
// numbering of the message matrix) and their total: one workgroup, one launch.
__global__ __launch_bounds__(1024) void node_extents_kernel(const uint8_t* __restrict__ mask, int B, int N,
                                                            int64_t* __restrict__ extent,
                                                            int64_t* __restrict__ row_off,
                                                            int64_t* __restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    int e = 0;
    if (b < B)
      for (int j = 0; j < N; ++j)
        if (mask[(int64_t)b * N + j]) e = j + 1;
    int incl = e;   // inclusive scan over the wave, then over the waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (b < B) {
      extent[b] = e;
      row_off[b] = before + incl - e;
    }
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (tid == 0) total[0] = carry_s;
}

}  // namespace

extern "C" int lnz_node_extents(const uint8_t* mask, int B, int N, int64_t* extent, int64_t* row_off,
                                int64_t* total, lnz_stream_t stream) {
  LNZ_REQUIRE(mask && extent && row_off && total && B > 0 && N > 0, LNZ_EINVAL,
              "lnz_node_extents: bad arguments (B=%d N=%d)", B, N);
  hipLaunchKernelGGL(node_extents_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, B, N, extent,
                     row_off, total);
  return lnz::check_launch("lnz_node_extents");
}

extern "C" int64_t lnz_head_backward_workspace_floats(int P, int n_wg) {
  if (P < 1 || P > PMAX || n_wg < 1) return 0;
  return (int64_t)n_wg * ((P + 1) * DH + 32 + DH);
}

extern "C" int lnz_head_backward(const float* X_last, const uint8_t* mask, const float* grad_score,
                                 const float* Whead, const float* bhead, const float* Wgate,
                                 const float* bgate, const int64_t* row_off, int B,
                                 int N, int P, int dhid, int n_wg, float* workspace, float* dY,
                                 float* dY_compact, float* dWhead, float* dbhead, float* dbias_last,
                                 lnz_stream_t stream) {
  LNZ_REQUIRE(X_last && mask && grad_score && Whead && bhead && workspace && dY && dWhead && dbhead &&
                  dbias_last && B > 0,
              LNZ_EINVAL, "lnz_head_backward: null pointer or B=%d", B);
  LNZ_REQUIRE(dhid == DH && N >= 1 && N <= 32 && P >= 1 && P <= PMAX, LNZ_ENOTSUP,
              "lnz_head_backward: built for hidden width 128, N <= 32, head width <= 31 (dhid=%d N=%d P=%d)",
              dhid, N, P);
  LNZ_REQUIRE(n_wg >= 1 && n_wg <= 4096, LNZ_EINVAL, "lnz_head_backward: n_wg=%d", n_wg);
  LNZ_REQUIRE(!dY_compact || row_off, LNZ_EINVAL, "lnz_head_backward: dY_compact without row_off");
  LNZ_REQUIRE(!Wgate == !bgate, LNZ_EINVAL, "lnz_head_backward: Wgate and bgate come together");
  HeadArgs a;
  a.X = X_last, a.mask = mask, a.gscore = grad_score, a.W = Whead, a.bias = bhead, a.row_off = row_off;
  a.Wgate = Wgate, a.bgate = bgate;
  a.dY = dY, a.dYc = dY_compact, a.part = workspace;
  a.B = B, a.N = N, a.P = P, a.n_wg = n_wg < B ? n_wg : B;
  if (P + 1 == 17)
    hipLaunchKernelGGL((head_backward_kernel<17, true>), dim3(a.n_wg), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((head_backward_kernel<32, false>), dim3(a.n_wg), dim3(256), 0, (hipStream_t)stream, a);
  int rc = lnz::check_launch("lnz_head_backward");
  if (rc != LNZ_OK) return rc;
  const int per_wg = (P + 1) * DH + 32 + DH;
  hipLaunchKernelGGL(head_backward_reduce_kernel, dim3((per_wg + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, workspace, a.n_wg, per_wg, P + 1, dWhead, dbhead, dbias_last);
  return lnz::check_launch("lnz_head_backward (reduce)");
}
