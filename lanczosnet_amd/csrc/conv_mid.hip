// R9 + R10 + R11 for graphs of 33..128 nodes — the reference's own graph configuration
// (config/graph_lanczos_net.yaml: n in [20, 100], K = 20, one edge type, eight long scales;
// model/lanczos_net_general.py:127-201, model/lanczos_net.py:125-199): ALL conv layers, the head
// and the gated masked mean in ONE launch, exact fp32 on v_mfma_f32_16x16x4_f32.
//
// Until r04 this size class ran on the kernels built for N = 2048 (conv_large.hip): four launches
// per layer, each latency bound at 64 graphs x 100 nodes (0.58 ms for seven layers).  A graph of
// this size fits one compute unit's LDS, but one compute unit per graph would leave three quarters
// of the chip idle at the reference's batch of 64 — so a graph is spread over FOUR workgroups by
// OUTPUT COLUMNS: workgroup (b, q) computes columns [32 q, 32 q + 32) of every layer's state.  Every
// product of a layer needs the full input state but only its own column slice of the weights, so
// the only exchange is one [n, 32]-column slice per workgroup and layer through global memory (L2)
// behind a counter the four workgroups of a graph spin on.  The four are given block indices that
// are equal modulo 8 — on MI355X in SPX mode the same XCD, the same L2 — which HIP does not promise:
//   * placement is CHECKED, not assumed: every workgroup posts its HW_REG_XCC_ID in the graph's
//     placement word at kernel start; four equal ids = one L2, and only then is the exchange fence
//     free (write-through stores + sc1 loads).  Otherwise (CU masks, other partition modes, another
//     part) the publisher releases and the consumer acquires at agent scope — correct under any
//     placement, a few microseconds slower per layer;
//   * forward progress needs the four workgroups of a graph resident together: the launcher sizes
//     every launch to what the device holds at once (occupancy query x compute units, whole groups
//     of 8 graphs = 32 blocks) and issues larger batches as consecutive launches;
//   * every spin is bounded: a peer that never arrives ends in a trap (a failed launch the host
//     sees), never in a hang.
//
// Per layer, with X [n, d] the state in LDS (row pitch 132), V [n, K] the Ritz vectors, g_s [K] the
// spectral gains of long scale s, L_c the edge-type Laplacians, W_c [128, d] the channel blocks of
// the mix weight (reference column order: long scales, then edge types):
//   1. Y = V^T X                      (eigen space; computed transposed, X^T V, so that X is read
//                                      in its row layout; every workgroup computes all of Y)
//   2. P = sum_s (g_s . Y) W_s^T      wave w = scale w (its weight rows straight from global
//                                      memory: no two waves read the same ones), the eight partial
//                                      [K, 32] blocks summed in wave order through LDS
//   3. per edge type c:  T = X W_c^T  (wave w = node rows [16 w, 16 w + 16); W_c's slice staged in
//                        LDS), stored transposed; out += L_c T with the wave's Laplacian rows held
//                        in REGISTERS for the whole kernel (the operators do not change with the
//                        layer)
//   4. out += V P                      (lift)
//   5. X' = relu(out + b) -> the exchange buffer; counter; spin; reload the full X'.
// Fragment indexing (v_mfma_f32_16x16x4_f32): lane = 16 kq + j supplies A[i = j][k = kq] and
// B[k = kq][j]; C/D register r is D[4 kq + r][j].  The k index is enumerated as 16 t + 4 kq + u
// (instruction u of k-block t), the same in A and B, so that a lane's four values of a k-block are
// one 16-byte read of a row-major operand.
#include "common.hpp"
#include "conv_tiles.hpp"

namespace {

constexpr int XP = 132;   // row pitch of the node state (floats)
constexpr int TP = 132;   // row pitch of the transposed buffers: rows = 32 columns / slots, entries = nodes
constexpr int PP = 33;    // row pitch of a wave's long-scale partial block
constexpr int AP = 36;    // row pitch of the summed block

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

struct MidArgs {
  const float* X0;      // [B, N, din0] layer-0 state, din0 a multiple of 16 (zero-padded columns)
  const float* L;       // [B, N, N, C] by element strides
  int64_t sb, sr, sc, sch;
  const float* V;       // [B, N, K]
  const float* G;       // [num_layer, B, S, K]
  const uint8_t* mask;  // [B, N]
  const float* W;       // per layer [128][S + C][din_l], layers behind each other
  const float* bias;    // [num_layer, 128]
  const float* Whead;   // [dout + 1, 128]: head rows, then the gate row
  const float* bhead;   // [dout + 1]
  float* Xwork;         // [num_layer, B, NR, 128] exchange buffer
  int32_t* sync;        // [B * (num_layer + 1)] zero-initialised: per (graph, layer) arrival counters,
                        // then the graphs' placement words
  float* score;         // [B, dout]
  int B, N, K, C, S, num_layer, din0, dout, R;
  int b0, b1;           // this launch's graphs [b0, b1)
  int force_fenced;     // LNZ_MID_FENCED=1: take the release / acquire exchange whatever the placement (tests)
};

constexpr int kSpinLimit = 1 << 24;   // x s_sleep(1) (64 clocks): about half a second

// tid 0 of a workgroup: wait until the low byte of *word reaches 4; returns the word
__device__ __forceinline__ int spin_until_four(const int32_t* word) {
  int v, spins = 0;
  while (((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xff) < 4) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) __builtin_trap();   // a peer workgroup never arrived
  }
  return v;
}

constexpr int mid_lds_floats(int R) {
  return 16 * R * XP + 32 * TP + 32 * XP + 2 * 32 * TP + 32 * AP + 16 * 32;
}

// Layer l's weights and gains into registers (see the kernel): the gains first — vector-memory
// results return in order, and the first phase of the layer waits for them only.
template <int C>
__device__ __forceinline__ void fetch_layer(const MidArgs& a, const int l, const float* Wp, const int b,
                                            const int q, const int tid, const int wave, const int j,
                                            const int kq, f32x4 (&bw)[2][8], f32x4 (&we)[C][2],
                                            float& gnext) {
  const int S = a.S, K = a.K, nch = a.S + C;
  const int din = l ? 128 : a.din0, nk = din >> 4, d4 = din >> 2;
  gnext = 0.0f;
  if (tid < S * 32) {
    const int sidx = tid >> 5, k = tid & 31;
    gnext = k < K ? a.G[(((int64_t)l * a.B + b) * S + sidx) * K + k] : 0.0f;
  }
  const int s0 = wave < S ? wave : 0;
  const float* w0 = Wp + ((int64_t)(32 * q + j) * nch + s0) * din + 4 * kq;
  const float* w1 = w0 + (int64_t)16 * nch * din;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    bw[0][t] = t < nk ? *reinterpret_cast<const f32x4*>(w0 + 16 * t) : zero4();
    bw[1][t] = t < nk ? *reinterpret_cast<const f32x4*>(w1 + 16 * t) : zero4();
  }
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int idx = tid + 512 * h, n = idx / d4, k4 = idx - n * d4;
      we[c][h] = idx < 32 * d4
                     ? *reinterpret_cast<const f32x4*>(Wp + ((int64_t)(32 * q + n) * nch + S + c) * din + 4 * k4)
                     : zero4();
    }
}

template <int C>
__global__ __launch_bounds__(512) void midgraph_forward_kernel(MidArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, kq = lane >> 4;
  // blocks 32 g .. 32 g + 31 = graphs 8 g .. 8 g + 7, four column quarters each; the quarters of a
  // graph are 8 apart (equal modulo 8: one XCD)
  const int idx32 = blockIdx.x & 31;
  const int b = a.b0 + 8 * (blockIdx.x >> 5) + (idx32 & 7), q = idx32 >> 3;
  if (b >= a.b1) return;
  __shared__ int one_l2;   // the four workgroups of this graph run on one XCD (checked below)
  int32_t* place = a.sync + (int64_t)a.B * a.num_layer + b;
  if (tid == 0) {
    // HW_REG_XCC_ID (hwreg 20), bits [3:0]: the XCD this workgroup runs on
    const int xcc = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15;
    __hip_atomic_fetch_add(place, 1 | ((xcc + 1) << (8 + 4 * q)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int N = a.N, K = a.K, S = a.S, R = a.R, NR = 16 * R, B = a.B;
  float* Xs = lds;                    // [NR][XP]
  float* Vt = Xs + NR * XP;           // [32 slots][TP]   Vt[k][node]
  float* Ys = Vt + 32 * TP;           // [32 slots][XP]   Y[k][column of X]
  float* Tt = Ys + 32 * XP;           // [32 columns][TP] T^T of the current edge type
  float* Ws = Tt + 32 * TP;           // [32 columns][TP] the workgroup's rows of W_c
  float* Ps = Tt;                     // [8 waves][32 slots][PP]: the long phase's partial blocks (Tt + Ws)
  float* Pacc = Ws + 32 * TP;         // [32 slots][AP]
  float* gs = Pacc + 32 * AP;         // [S <= 16][32]

  // ---- prologue: the graph's operands that do not change with the layer
  for (int idx = tid; idx < 32 * NR; idx += 512) {
    const int k = idx / NR, node = idx - k * NR;
    Vt[k * TP + node] = (k < K && node < N) ? finite_or_zero(a.V[((int64_t)b * N + node) * K + k]) : 0.0f;
  }
  float vf[2][4];     // V[16 wave + j][16 t + 4 kq + u]: A fragments of the lift
  float Lf[C][8][4];  // L_c[16 wave + j][16 t + 4 kq + u]: A fragments of out += L_c T
  {
    const int node = 16 * wave + j;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = 16 * t + 4 * kq + u;
        vf[t][u] = (wave < R && node < N && k < K) ? finite_or_zero(a.V[((int64_t)b * N + node) * K + k]) : 0.0f;
      }
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int col = 16 * t + 4 * kq + u;
          Lf[c][t][u] = (wave < R && node < N && col < N)
                            ? a.L[(int64_t)b * a.sb + (int64_t)node * a.sr + (int64_t)col * a.sc + (int64_t)c * a.sch]
                            : 0.0f;
        }
  }
  // Equal operator channels are folded: sum_c L_c X W_c^T = L (X (sum_c W_c)^T).  With one edge
  // type the collated L carries the simple graph's Laplacian twice (dataset/graph_data.py:225-262).
  // Decided here, per graph, by comparing the fragments the waves hold anyway.
  bool fold = false;
  if (C > 1) {
    int same = 1;
#pragma unroll
    for (int c = 1; c < C; ++c)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) same &= Lf[c][t][u] == Lf[0][t][u] ? 1 : 0;
    fold = __syncthreads_and(same) != 0;
  }
  {
    const int d4 = a.din0 >> 2;
    for (int idx = tid; idx < NR * d4; idx += 512) {
      const int row = idx / d4, c4 = idx - row * d4;
      f32x4 v = zero4();
      if (row < N) v = *reinterpret_cast<const f32x4*>(a.X0 + ((int64_t)b * N + row) * a.din0 + 4 * c4);
      *reinterpret_cast<f32x4*>(&Xs[row * XP + 4 * c4]) = v;
    }
  }
  const int nch = S + C;
  const float* Wl = a.W;
#ifdef LNZ_MID_PHASES  // clock64 deltas per phase, wave 0 of workgroup (0, 0) -> sync[B * num_layer ..]
  long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, _t0 = clock64();
#define LNZ_PH(i) { const long long _t1 = clock64(); ph[i] += _t1 - _t0; _t0 = _t1; }
#else
#define LNZ_PH(i)
#endif
  // A layer's weights and gains travel in registers from BEFORE the wait for the previous layer's
  // exchange: wave w's rows of long scale w (B fragments of phase 2: 16 x 16 bytes), every thread's
  // share of the edge types' slices (staged into LDS in phase 3), the gains.  The gains go out
  // first: vector-memory results return in order, and phase 1 waits for them only.
  f32x4 bw[2][8];
  f32x4 we[C][2];
  float gnext;
  fetch_layer<C>(a, 0, Wl, b, q, tid, wave, j, kq, bw, we, gnext);
  LNZ_PH(0)  // prologue
  for (int l = 0; l < a.num_layer; ++l) {
    const int din = l ? 128 : a.din0, nk = din >> 4;
    if (tid < S * 32) gs[tid] = gnext;
    __syncthreads();  // Xs (and the prologue's buffers) complete; the previous layer is through with Tt / Ws
    LNZ_PH(1)  // layer head: weight fetch issue, gains
    // ---- 1. Y = V^T X, transposed tiles: wave w = columns [16 w, 16 w + 16) of X, both slot tiles
    if (16 * wave < din) {
      f32x4 y0 = zero4(), y1 = zero4();
      for (int t = 0; t < R; ++t) {
        float xa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = Xs[(16 * t + 4 * kq + u) * XP + 16 * wave + j];
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Vt[j * TP + 16 * t + 4 * kq]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Vt[(16 + j) * TP + 16 * t + 4 * kq]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          y0 = mfma16(xa[u], b0[u], y0);
          y1 = mfma16(xa[u], b1[u], y1);
        }
      }
      // D[i = column 16 w + 4 kq + r][j = slot] -> Y[slot][column]
      *reinterpret_cast<f32x4*>(&Ys[j * XP + 16 * wave + 4 * kq]) = y0;
      *reinterpret_cast<f32x4*>(&Ys[(16 + j) * XP + 16 * wave + 4 * kq]) = y1;
    }
    __syncthreads();
    LNZ_PH(2)  // Y = V^T X
    // ---- 2. long scales: wave w = scales w, w + 8, ...; P_w [32 slots, 32 columns]
    {
      f32x4 p[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
      for (int s = wave; s < S; s += 8) {
        const float g0 = gs[s * 32 + j], g1 = gs[s * 32 + 16 + j];
        if (s >= 8) {  // (more than eight long scales: the later ones are fetched here)
          const float* w0 = Wl + ((int64_t)(32 * q + j) * nch + s) * din + 4 * kq;
          const float* w1 = w0 + (int64_t)16 * nch * din;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            bw[0][t] = t < nk ? *reinterpret_cast<const f32x4*>(w0 + 16 * t) : zero4();
            bw[1][t] = t < nk ? *reinterpret_cast<const f32x4*>(w1 + 16 * t) : zero4();
          }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t >= nk) break;
          f32x4 a0 = *reinterpret_cast<const f32x4*>(&Ys[j * XP + 16 * t + 4 * kq]);
          f32x4 a1 = *reinterpret_cast<const f32x4*>(&Ys[(16 + j) * XP + 16 * t + 4 * kq]);
          const f32x4 b0 = bw[0][t], b1 = bw[1][t];
          a0 *= g0;
          a1 *= g1;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            p[0][0] = mfma16(a0[u], b0[u], p[0][0]);
            p[0][1] = mfma16(a0[u], b1[u], p[0][1]);
            p[1][0] = mfma16(a1[u], b0[u], p[1][0]);
            p[1][1] = mfma16(a1[u], b1[u], p[1][1]);
          }
        }
      }
      float* pw = Ps + wave * 32 * PP;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) pw[(16 * st + 4 * kq + r) * PP + 16 * ct + j] = p[st][ct][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 1024; idx += 512) {
      const int k = idx >> 5, col = idx & 31;
      float acc = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) acc += Ps[(w * 32 + k) * PP + col];  // (waves beyond S wrote zeros)
      Pacc[k * AP + col] = acc;
    }
    __syncthreads();
    LNZ_PH(3)  // long scales + reduction
    // ---- 3. edge types
    f32x4 o0 = zero4(), o1 = zero4();
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (fold && c > 0) continue;  // (workgroup uniform) equal operators: one pass with the summed weights
      {
        const int d4 = din >> 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = tid + 512 * h, n = idx / d4, k4 = idx - n * d4;
          f32x4 v = we[c][h];
          if (fold) {
#pragma unroll
            for (int c2 = 1; c2 < C; ++c2) v += we[c2][h];
          }
          if (idx < 32 * d4) *reinterpret_cast<f32x4*>(&Ws[n * TP + 4 * k4]) = v;
        }
      }
      __syncthreads();
      if (wave < R) {
        f32x4 t0 = zero4(), t1 = zero4();
        for (int t = 0; t < nk; ++t) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(&Xs[(16 * wave + j) * XP + 16 * t + 4 * kq]);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Ws[j * TP + 16 * t + 4 * kq]);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Ws[(16 + j) * TP + 16 * t + 4 * kq]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            t0 = mfma16(xa[u], b0[u], t0);
            t1 = mfma16(xa[u], b1[u], t1);
          }
        }
        // D[i = node 16 w + 4 kq + r][j = column] -> T^T[column][node]
        *reinterpret_cast<f32x4*>(&Tt[j * TP + 16 * wave + 4 * kq]) = t0;
        *reinterpret_cast<f32x4*>(&Tt[(16 + j) * TP + 16 * wave + 4 * kq]) = t1;
      }
      __syncthreads();
      if (wave < R) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t < R) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Tt[j * TP + 16 * t + 4 * kq]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Tt[(16 + j) * TP + 16 * t + 4 * kq]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              o0 = mfma16(Lf[c][t][u], b0[u], o0);
              o1 = mfma16(Lf[c][t][u], b1[u], o1);
            }
          }
        }
      }
      // (the next staging writes Ws only; the barrier behind it orders this GEMM2's reads of Tt
      // before the next GEMM1's stores)
    }
    // ---- 4. lift out += V P, 5. bias + ReLU -> the exchange buffer
    if (wave < R) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b0[u] = Pacc[(16 * t + 4 * kq + u) * AP + j];
          b1[u] = Pacc[(16 * t + 4 * kq + u) * AP + 16 + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o0 = mfma16(vf[t][u], b0[u], o0);
          o1 = mfma16(vf[t][u], b1[u], o1);
        }
      }
    }
    __syncthreads();  // every wave is through with Tt (the last edge type's GEMM2)
    LNZ_PH(4)  // edge types + lift
    // ---- 5. X' = relu(out + b), staged in LDS as the workgroup's [NR, 32] column slice
    float* Os = Tt;   // [NR][AP] (Tt + Ws)
    if (wave < R) {
      const float bi0 = a.bias[l * 128 + 32 * q + j], bi1 = a.bias[l * 128 + 32 * q + 16 + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + 4 * kq + r;
        const bool real = row < N;
        Os[row * AP + j] = real ? fmaxf(o0[r] + bi0, 0.0f) : 0.0f;
        Os[row * AP + 16 + j] = real ? fmaxf(o1[r] + bi1, 0.0f) : 0.0f;
      }
    }
    __syncthreads();
    // ---- 6. exchange.  The slice goes out by WRITE-THROUGH stores (8-byte agent-scope relaxed
    //      atomics lower to global_store ... sc1), drained, then the graph's counter is raised;
    //      the consumers poll it relaxed and read the full state by sc1 loads, which are served
    //      by L2 / memory and never by a stale L1 line — no release / acquire fence (each costs a
    //      whole-L2 write-back or an L1 flush per workgroup and layer; MI355X_MICROARCH.md,
    //      "publish-large").  Each layer has its own region of the exchange buffer.
    {
      float* xo = a.Xwork + ((int64_t)l * B + b) * NR * 128 + 32 * q;
      for (int idx = tid; idx < NR * 16; idx += 512) {
        const int row = idx >> 4, w2 = idx & 15;
        const unsigned long long v = *reinterpret_cast<const unsigned long long*>(&Os[row * AP + 2 * w2]);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(xo + row * 128 + 2 * w2), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (l == 0 && tid == 0) {
      // placement: all four posted at kernel start, long ago; equal ids = one L2
      const int pw = spin_until_four(place);
      const int mine = (pw >> (8 + 4 * q)) & 15;
      one_l2 = ((pw >> 8) & 15) == mine && ((pw >> 12) & 15) == mine && ((pw >> 16) & 15) == mine &&
               ((pw >> 20) & 15) == mine && !a.force_fenced;
    }
    __syncthreads();
    LNZ_PH(5)  // epilogue + publish
    const bool last = l == a.num_layer - 1;
    Wl += (int64_t)128 * nch * din;
    if (!last) fetch_layer<C>(a, l + 1, Wl, b, q, tid, wave, j, kq, bw, we, gnext);
    const bool fenced = !one_l2;   // (workgroup-uniform)
    if (tid == 0) {
      // different L2s: the slice has to leave this one before the counter says it is there
      if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(&a.sync[b * a.num_layer + l], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!last || q == 0) (void)spin_until_four(&a.sync[b * a.num_layer + l]);
    }
    if (last && q != 0) return;  // the head is workgroup (b, 0)'s
    __syncthreads();
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ... and stale lines of this L2 dropped
    LNZ_PH(6)  // wait for the other three
    {
      // the three other workgroups' slices by sc1 loads, the own one from its staging copy
      const float* xi = a.Xwork + ((int64_t)l * B + b) * NR * 128;
      for (int idx = tid; idx < NR * 48; idx += 512) {
        const int row = idx / 48, w3 = idx - row * 48;
        const int qq = w3 >> 4, qsrc = qq + (qq >= q ? 1 : 0), w2 = 16 * qsrc + (w3 & 15);
        const unsigned long long v = __hip_atomic_load(
            reinterpret_cast<const unsigned long long*>(xi + row * 128 + 2 * w2), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<unsigned long long*>(&Xs[row * XP + 2 * w2]) = v;
      }
      for (int idx = tid; idx < NR * 16; idx += 512) {
        const int row = idx >> 4, w2 = idx & 15;
        *reinterpret_cast<unsigned long long*>(&Xs[row * XP + 32 * q + 2 * w2]) =
            *reinterpret_cast<const unsigned long long*>(&Os[row * AP + 2 * w2]);
      }
    }
    LNZ_PH(7)  // reload
  }
  __syncthreads();
  // ---- head (model/lanczos_net.py:185-194): y = (W_h x + b_h) * sigmoid(w_g x + b_g), masked mean
  {
    const int no = a.dout + 1;
    float* Wh = Tt;    // [no][128] head rows + the gate row (<= 32 x 128 floats: Tt + Ws)
    float* yh = Ys;    // [N][no]
    float* gate = Pacc;  // [N]
    for (int idx = tid; idx < no * 32; idx += 512)
      *reinterpret_cast<f32x4*>(&Wh[4 * idx]) = *reinterpret_cast<const f32x4*>(a.Whead + 4 * idx);
    __syncthreads();
    for (int idx = tid; idx < N * no; idx += 512) {
      const int row = idx / no, o = idx - row * no;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
      for (int k = 0; k < 128; k += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&Xs[row * XP + k]);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(&Wh[o * 128 + k]);
        s0 = fmaf(x[0], ww[0], s0), s1 = fmaf(x[1], ww[1], s1);
        s2 = fmaf(x[2], ww[2], s2), s3 = fmaf(x[3], ww[3], s3);
      }
      yh[idx] = (s0 + s1) + (s2 + s3) + a.bhead[o];
    }
    __syncthreads();
    if (tid < N)
      gate[tid] = a.mask[(int64_t)b * N + tid] ? 1.0f / (1.0f + expf(-yh[tid * no + a.dout])) : -1.0f;
    __syncthreads();
    // output o = wave o, o + 8, ...: the lanes take rows lane, lane + 64; fixed-order tree over the lanes
    for (int o = wave; o < a.dout; o += 8) {
      float num = 0.0f, den = 0.0f;
      for (int row = lane; row < N; row += 64) {
        const float g = gate[row];
        if (g >= 0.0f) num = fmaf(yh[row * no + o], g, num), den += 1.0f;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        num += __shfl_xor(num, off, 64);
        den += __shfl_xor(den, off, 64);
      }
      if (lane == 0) a.score[(int64_t)b * a.dout + o] = num / den;
    }
  }
  LNZ_PH(8)  // head
#ifdef LNZ_MID_PHASES
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 9; ++i) a.sync[a.B * (a.num_layer + 1) + i] = (int32_t)(ph[i] >> 4);
#endif
}

}  // namespace

extern "C" int64_t lnz_midgraph_workspace_floats(int B, int N, int num_layer) {
  if (B <= 0 || N <= 0 || num_layer <= 0) return 0;
  return (int64_t)num_layer * B * (16 * ((N + 15) / 16)) * 128;
}

extern "C" int lnz_midgraph_forward(const float* X0, const float* L, int64_t stride_b, int64_t stride_r,
                                    int64_t stride_c, int64_t stride_ch, const float* V, const float* G,
                                    const uint8_t* mask, const float* W, const float* bias,
                                    const float* Whead, const float* bhead, int B, int N, int K, int C,
                                    int S, int num_layer, int din0, int dout, float* Xwork,
                                    int32_t* sync, float* score, lnz_stream_t stream) {
  LNZ_REQUIRE(X0 && L && V && mask && W && bias && Whead && bhead && Xwork && sync && score && B > 0,
              LNZ_EINVAL, "lnz_midgraph_forward: null pointer or B=%d", B);
  LNZ_REQUIRE(N > 0 && N <= 128 && K > 0 && K <= 32 && S >= 0 && S <= 16 && (S == 0 || G) &&
                  num_layer > 0 && dout >= 1 && dout <= 31,
              LNZ_ENOTSUP, "lnz_midgraph_forward: built for N <= 128, K <= 32, <= 16 long scales, "
              "head width <= 31 (N=%d K=%d S=%d dout=%d)", N, K, S, dout);
  LNZ_REQUIRE(C >= 1 && C <= 2, LNZ_ENOTSUP, "lnz_midgraph_forward: built for 1..2 operator channels (C=%d)", C);
  LNZ_REQUIRE(din0 > 0 && din0 % 16 == 0 && din0 <= 128, LNZ_ENOTSUP,
              "lnz_midgraph_forward: input width %d must be a multiple of 16, <= 128 (zero-pad)", din0);
  MidArgs a;
  a.X0 = X0, a.L = L, a.sb = stride_b, a.sr = stride_r, a.sc = stride_c, a.sch = stride_ch;
  a.V = V, a.G = G, a.mask = mask, a.W = W, a.bias = bias, a.Whead = Whead, a.bhead = bhead;
  a.Xwork = Xwork, a.sync = sync, a.score = score;
  a.B = B, a.N = N, a.K = K, a.C = C, a.S = S, a.num_layer = num_layer, a.din0 = din0, a.dout = dout;
  a.R = (N + 15) / 16;
  const size_t bytes = (size_t)mid_lds_floats(a.R) * sizeof(float);
  const void* fn = C == 1 ? (const void*)midgraph_forward_kernel<1> : (const void*)midgraph_forward_kernel<2>;
  // per launch: the attribute is per device
  LNZ_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess,
              LNZ_ENOTSUP, "lnz_midgraph_forward: %zu bytes of LDS per workgroup are not available on this device",
              bytes);
  // The four workgroups of a graph wait for each other: a launch holds only as many blocks as the
  // device keeps resident at once (whole groups of 32 blocks = 8 graphs); a larger batch goes out
  // as consecutive launches on the stream.
  int dev = 0, n_cu = 0, per_cu = 0;
  LNZ_REQUIRE(hipGetDevice(&dev) == hipSuccess &&
                  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, bytes) == hipSuccess,
              LNZ_ELAUNCH, "lnz_midgraph_forward: occupancy query failed");
  const int groups = (int)(((int64_t)n_cu * per_cu) / 32);
  LNZ_REQUIRE(groups >= 1, LNZ_ENOTSUP,
              "lnz_midgraph_forward: the device holds %d x %d workgroups at once, 32 (eight graphs) are needed",
              n_cu, per_cu);
  const char* ff = getenv("LNZ_MID_FENCED");
  a.force_fenced = ff && ff[0] == '1';
  for (int b0 = 0; b0 < B; b0 += 8 * groups) {
    a.b0 = b0;
    a.b1 = b0 + 8 * groups < B ? b0 + 8 * groups : B;
    const int grid = ((a.b1 - a.b0 + 7) / 8) * 32;
    void* params[] = {&a};
    (void)hipLaunchKernel(fn, dim3(grid), dim3(512), params, bytes, (hipStream_t)stream);
    const int rc = lnz::check_launch("lnz_midgraph_forward");
    if (rc != LNZ_OK) return rc;
  }
  return LNZ_OK;
}
