// R8: the filter MLPs of AdaLanczosNet (model/ada_lanczos_net.py:271-272; per conv layer
// Linear(K K S -> 4096) -> ReLU -> Linear(4096 -> 4096) -> ReLU -> Linear(4096 -> 4096) -> ReLU ->
// Linear(4096 -> K K S), M = batch rows) in the default exact-fp32 mode, hand-written.
//
// One launch = one Linear:   out = [relu]( X W^T + bias ),  X [M, K], W [N, K] (torch's
// nn.Linear layout), fp32 operands, v_mfma_f32_32x32x2_f32, fp32 accumulate.
//
// The fp32 matrix pipe does 256 flop / clk / CU, so a 128 x 128 x 32 slice of the product is 4096
// cycles of MFMAs against 32 KB of operands to stage (8 B / clk / CU from L2) and 48-64 KB of
// fragment reads (LDS: 128 B / clk): unlike the fp16 split-precision kernel (f16x3_linear.hip,
// LDS bound) this one has only the matrix pipe to keep busy.  Same skeleton as that kernel:
// 128 x 128 output tiles (M = 1024: 256 tiles, one per CU), K in slices of 32 floats = 128-byte
// LDS rows filled by global_load_lds (16 B per lane, no staging registers) into one of two 32 KB
// stages, one __syncthreads per slice, chunk c of row r in slot c ^ ((r >> 1) & 7) (conflict-free
// ds_read_b128, see f16x3_linear.hip), XCD-aware tile order, bias + ReLU in the epilogue (the
// library path runs them as a second elementwise launch per Linear), split-K with a fixed-order
// reduction when the output has too few tiles to fill the chip.
//
// Fragments: lane (j, hh) reads the 16-byte chunk 2q + hh of its row = k-values 8q + 4hh + 0..3;
// MFMA step u of the block multiplies element u of the A and B chunks — the two lane halves then
// hold k = 8q + u and 8q + 4 + u, the same pairing on both operands.
#include <type_traits>

#include "common.hpp"

namespace {

#ifndef LNZ_F32LIN_WN
#define LNZ_F32LIN_WN 4
#endif
#ifndef LNZ_F32LIN_MI
#define LNZ_F32LIN_MI 16
#endif
// 1: the two wave groups of a workgroup run half a slice apart (see the kernel's PP branch)
#ifndef LNZ_F32LIN_PP
#define LNZ_F32LIN_PP 0
#endif
#ifndef LNZ_F32LIN_BK
#define LNZ_F32LIN_BK 32
#endif
constexpr int BM = 128, BN = 128, BK = LNZ_F32LIN_BK;   // BK = 32 or 64 floats per slice
constexpr int RB = BK * 4;                  // bytes per LDS row
constexpr int RPP = 1024 / RB;              // rows per 1 KB copy piece (one global_load_lds)
constexpr int PPO = 128 / RPP;              // pieces per operand slice
constexpr int kSlice = 128 * RB;            // one operand slice: 128 rows x BK floats
constexpr int kStage = 2 * kSlice;          // x, w
constexpr size_t kLds = 2 * (size_t)kStage; // two stages
// chunk c of row r sits in slot c ^ swz(r): 128-byte rows (BK = 32) share a 256-byte bank row in
// pairs; a 256-byte row (BK = 64) is a bank row of its own and its 16 chunks are rotated by the
// row number (the 16-lane service groups of ds_read_b128 hold rows {0-3, 12-15} at chunk c and rows
// {4-11} at chunk c + 1: the two slot sets are complementary for every c)
__device__ __forceinline__ int swz(const int r) { return BK == 32 ? (r >> 1) & 7 : r & 15; }

// global -> LDS copies of one K slice: NQ pieces (RPP rows x RB bytes = 1 KB each) of one operand,
// as buffer_load_dwordx4 ... lds: the descriptor (operand tile base, wave uniform) sits in SGPRs,
// the per-lane byte offset of a piece (row * ld + chunk, k-independent) in ONE VGPR computed once
// per launch, the slice's k offset in an SGPR and the LDS destination in M0 by scalar arithmetic —
// nothing of a copy is computed on the VALU inside the main loop.  (First version:
// global_load_lds with 64-bit per-lane addresses: two v_lshl_add_u64, a v_add3 and a
// v_readfirstlane -> s_mov m0 in front of every copy; with the copies removed the 4096 x 4096
// Linear ran 8 % faster — the ablation is in DESIGN.md.)
struct PieceOffsets {
  int v[8];
};
template <int NQ>
__device__ __forceinline__ PieceOffsets piece_offsets(const int ld, const int rows_left, const int lane,
                                                      const int q0) {
  constexpr int CPR = BK / 4;                         // 16-B chunks per row
  const int rsub = lane / CPR, slot = lane % CPR;
  PieceOffsets o;
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int r = RPP * (q0 + qq) + rsub;             // row of the 128-row slice
    const int chunk = slot ^ swz(r);                  // which 16-B chunk of the row lands in `slot`
    const int rr = r < rows_left ? r : rows_left - 1; // rows beyond the operand re-read its last row
    o.v[qq] = (rr * ld + 4 * chunk) * 4;
  }
  return o;
}
template <int NQ>
__device__ __forceinline__ void stage_pieces(const __amdgpu_buffer_rsrc_t rsrc, const PieceOffsets& po,
                                             const int k0, unsigned char* lds_op, const int q0) {
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc, (__attribute__((address_space(3))) void*)(lds_op + (q0 + qq) * 1024), 16, po.v[qq],
        k0 * 4, 0, 0);
}

// (a native vector type, not HIP's float4: that struct's union members give its loads the
// may-alias-anything type tag, and the compiler then drains vmcnt — the global_load_lds copies in
// flight — in front of every fragment read that follows a copy)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 frag(const unsigned char* lds_op, const int row, const int chunk) {
  return *reinterpret_cast<const f32x4*>(lds_op + row * RB + ((chunk ^ swz(row)) << 4));
}

// The two fp32 MFMA shapes behind one interface: MI = 32: v_mfma_f32_32x32x2_f32 (16 accumulator
// registers per tile, lane = (row j, k half)), MI = 16: v_mfma_f32_16x16x4_f32 (4 accumulator
// registers per tile, lane = (row j, k quarter)).  Same flop rate; the 16 x 16 shape moves half the
// accumulator bytes through the register file per flop (k = 4 per accumulator update instead of 2).
template <int MI>
struct Mfma;
template <>
struct Mfma<32> {
  typedef f32x16 acc_t;
  [[maybe_unused]] static constexpr int NR = 16;
  static __device__ __forceinline__ acc_t zero() { return lnz::splat16(0.0f); }
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return lnz::mfma32(a, b, c); }
  // row of accumulator register r within the tile, for lane group kq = lane / MI
  static __device__ __forceinline__ int row(int r, int kq) { return lnz::cd_row(r, kq); }
};
template <>
struct Mfma<16> {
  typedef f32x4 acc_t;
  [[maybe_unused]] static constexpr int NR = 4;
  static __device__ __forceinline__ acc_t zero() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int r, int kq) { return 4 * kq + r; }
};

// workgroup barrier that waits for this wave's LDS reads only (its global_load_lds copies stay in
// flight across it) / for everything.  Inline assembly: __syncthreads() carries a release fence,
// which hipcc lowers to vmcnt(0) as well; "memory" keeps the compiler's loads, stores and copies on
// their side of it (register-only instructions are fenced by the sched_barrier around the calls).
__device__ __forceinline__ void barrier_lgkm() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void barrier_all() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// WN = wavefronts along N: 4 = eight waves of 64 x 32 (two per SIMD), 2 = four waves of 64 x 64
// PP = 1 (WN = 4): PING-PONG — the workgroup's waves 0..3 (group A, rows 0..63) and 4..7 (group B,
//   rows 64..127) sit pairwise on the four SIMDs; B runs HALF A SLICE behind A, with a workgroup
//   barrier every half slice (measured: +2.5 % with the first copy path, nothing once the copies
//   cost no VALU work; kept as a build switch).
//
// Work decomposition.  sk_per = 0: one workgroup per 128 x 128 output tile, all of K (M = 1024,
// N = 4096: 256 tiles = one per CU).  sk_per > 0: STREAM-K for outputs with too few tiles to fill
// the chip (N = 1056: 72 tiles): the (tile, k-slice) units, tile major, are dealt in runs of sk_per
// to the workgroups, so every CU multiplies the same number of slices whatever the tile count; a
// run covers the tail of one tile and / or the head of the next (sk_per <= K / BK).  The workgroup
// holding a tile's LAST slice owns the tile: the others store their raw 128 x 128 partial
// (part[workgroup]) and count themselves in (flags[tile], agent-scope release); the owner waits
// for the count, adds the partials in ascending workgroup order and its own on top — a fixed
// order, bit-reproducible — applies bias / ReLU and resets the flag.  A workgroup with two
// segments runs the head of the later tile FIRST (a partial somebody else waits for) and the
// tail it owns last, so nobody waits for work that has not been issued: the owner's contributors
// sit in lower-numbered workgroups and finish their part in their first segment or, with a
// single segment, together with the owner.  All workgroups are resident (grid <= CUs, one
// workgroup per CU by LDS).  (Before: split-K over gridDim.y + a second launch to add the
// partials: 216 of 256 CUs busy, 0.091 ms at N = 1056 against the library's 0.078.)
template <int WN, int MI, int PP>
__global__ __launch_bounds__(128 * WN) void f32_linear_kernel(
    const float* __restrict__ X, const int ldx, const float* __restrict__ W, const int ldw,
    const float* __restrict__ bias, const int relu, const int M, const int N, const int K,
    const int tiles_n, const int bh, float* __restrict__ out, const int ldo,
    float* __restrict__ part, int* __restrict__ flags, const int sk_per) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NW = 2 * WN;          // wavefronts
  typedef Mfma<MI> MM;
  constexpr int RT = 64 / MI;         // MFMA row tiles per wave (64 rows)
  constexpr int CT = 128 / WN / MI;   // MFMA column tiles per wave
  constexpr int KQ = 64 / MI;         // lane groups along k: a fragment group spans 4 KQ k-values
  constexpr int NG = BK / (4 * KQ);   // fragment groups per slice
  constexpr int PW = 2 * PPO / NW;    // copy pieces per wave and slice
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int Tall = K / BK;
  const int tiles_m = (M + BM - 1) / BM;

  // ---- this workgroup's segments: (tile, first slice, slice count)
  int nseg = 1, seg_tile[2], seg_kt0[2], seg_T[2];
  // (stream-K workgroups take their runs in blockIdx order: a remap that gives an XCD consecutive
  // tiles was measured — no difference, the runs of neighbouring workgroups sit at different k)
  const int wg = blockIdx.x;
  if (sk_per > 0) {
    const int u0 = wg * sk_per;
    const int U = tiles_m * tiles_n * Tall;
    const int u1 = u0 + sk_per < U ? u0 + sk_per : U;
    const int t0 = u0 / Tall, t1 = (u1 - 1) / Tall;
    if (t0 == t1) {
      seg_tile[0] = t0, seg_kt0[0] = u0 - t0 * Tall, seg_T[0] = u1 - u0;
      seg_tile[1] = t0, seg_kt0[1] = 0, seg_T[1] = 0;
    } else {  // head of the later tile first, then the tail this workgroup owns
      nseg = 2;
      seg_tile[0] = t1, seg_kt0[0] = 0, seg_T[0] = u1 - t1 * Tall;
      seg_tile[1] = t0, seg_kt0[1] = u0 - t0 * Tall, seg_T[1] = t1 * Tall - u0;
    }
  } else {
    // XCD-aware tile order (f16x3_linear.hip): an XCD's workgroups take a bh x bw block of tiles
    int tm, tn;
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (bh > 0) {
      const int bw = (nwg >> 3) / bh, bpr = tiles_n / bw;
      tm = (xcd / bpr) * bh + slot % bh;
      tn = (xcd % bpr) * bw + slot / bh;
    } else {
      tm = blockIdx.x / tiles_n;
      tn = blockIdx.x - tm * tiles_n;
    }
    seg_tile[0] = tm * tiles_n + tn, seg_kt0[0] = 0, seg_T[0] = Tall;
    seg_tile[1] = 0, seg_kt0[1] = 0, seg_T[1] = 0;
  }

  // this wave's share of the copies: operand `op` (0 = x, 1 = w), pieces [pq0, pq0 + PW)
  const int op = wave * PW / PPO, pq0 = (wave * PW) % PPO;
  const int arow = wr * 64 + (lane % MI), brow = wc * (MI * CT) + (lane % MI), g = lane / MI;

  for (int seg = 0; seg < nseg; ++seg) {
    const int tile = seg ? seg_tile[1] : seg_tile[0];
    const int kt0 = seg ? seg_kt0[1] : seg_kt0[0];
    const int T = seg ? seg_T[1] : seg_T[0];
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const float* src = op == 0 ? X + (int64_t)m0 * ldx : W + (int64_t)n0 * ldw;
    const int ld = op == 0 ? ldx : ldw;
    const int rows_left = op == 0 ? M - m0 : N - n0;
    // (the host checks that a 128-row operand tile spans < 2^31 bytes: 32-bit offsets)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
    const PieceOffsets po = piece_offsets<PW>(ld, rows_left, lane, pq0);

    typename MM::acc_t acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int b = 0; b < CT; ++b) acc[a][b] = MM::zero();

    // Software pipeline at slice granularity: ALL fragments of slice kt + 1 are read into registers
    // (the other half of af / bf) and slice kt + 2 is copied into the stage slice kt was read from,
    // while the 32 CT MFMAs of slice kt run from registers loaded one slice earlier.  A wave
    // that leaves the barrier therefore has a whole slice of register-resident work in front of it:
    // neither the LDS latency nor the copies' landing is ever waited for inside the MFMA stream
    // (first version: fragments read one 8-k block ahead, barrier -> read -> wait -> MFMA at every
    // slice: 0.282 ms at 1024 x 4096 x 4096).
    f32x4 af[2][NG][RT], bf[2][NG][CT];
    auto load_frags = [&](const unsigned char* stage, auto bufc) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
#pragma unroll
        for (int u = 0; u < RT; ++u) af[buf][q][u] = frag(stage, arow + MI * u, KQ * q + g);
#pragma unroll
        for (int u = 0; u < CT; ++u) bf[buf][q][u] = frag(stage + kSlice, brow + MI * u, KQ * q + g);
      }
    };
    auto clamp_k = [&](const int kt) { return (kt0 + (kt < T ? kt : T - 1)) * BK; };
    if (seg > 0) {  // the previous segment's clamped copies and fragment reads are done with LDS
      lnz::wait_vmcnt0();
      __syncthreads();
    }
    stage_pieces<PW>(rsrc, po, clamp_k(0), smem + op * kSlice, pq0);
    stage_pieces<PW>(rsrc, po, clamp_k(1), smem + kStage + op * kSlice, pq0);
    lnz::wait_vmcnt0();
    __syncthreads();
    load_frags(smem, std::integral_constant<int, 0>{});

    // the MFMAs of fragment groups [q0, q0 + NQ) of register set buf.  TRANSPOSED product: A
    // operand = the W fragment, B operand = the x fragment, D[n][m] = sum_k W[n][k] x[m][k] — the
    // same k-ordered fma chain per output element, so the same bits, and four consecutive
    // accumulator registers are four consecutive n of one output row (16-byte stores below)
    auto mfma_groups = [&](auto bufc, auto q0c, auto nqc) {
      constexpr int buf = decltype(bufc)::value, q0 = decltype(q0c)::value, nq = decltype(nqc)::value;
#pragma unroll
      for (int q = q0; q < q0 + nq; ++q) {
#define LNZ_STEP(E)                                                               \
  _Pragma("unroll") for (int a = 0; a < RT; ++a)                                  \
  _Pragma("unroll") for (int b = 0; b < CT; ++b)                                  \
      acc[a][b] = MM::run(bf[buf][q][b].E, af[buf][q][a].E, acc[a][b]);
        LNZ_STEP(x)
        LNZ_STEP(y)
        LNZ_STEP(z)
        LNZ_STEP(w)
#undef LNZ_STEP
      }
    };
    // issue order: one load per gap between MFMAs
    auto order_copies = [&]() {
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // buffer_load ... lds
      }
    };
    auto order_reads = [&]() {
#pragma unroll
      for (int i = 0; i < NG * (RT + CT); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MI == 16 ? 2 : 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // fragment read
      }
    };

    if constexpr (PP == 0) {
      auto slice = [&](const int kt, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        // slice kt + 1 has landed in stage buf ^ 1 and every wave holds slice kt's fragments in
        // registers: stage buf is free for slice kt + 2.  The copies' completion is waited for
        // EXPLICITLY: the compiler does not see that the fragment reads depend on them (without
        // the explicit wait one of the two unrolled barriers came out with lgkmcnt(0) only —
        // intermittent wrong tiles at K >= 4064)
        lnz::wait_vmcnt0();
        __syncthreads();
        stage_pieces<PW>(rsrc, po, clamp_k(kt + 2), smem + buf * kStage + op * kSlice, pq0);
        load_frags(smem + (buf ^ 1) * kStage, std::integral_constant<int, buf ^ 1>{});
        mfma_groups(bufc, std::integral_constant<int, 0>{}, std::integral_constant<int, NG>{});
        order_copies();
        order_reads();
      };
      for (int kt = 0; kt < T; kt += 2) {
        slice(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < T) slice(kt + 1, std::integral_constant<int, 1>{});
      }
    } else {
      // Barrier G(i), i = 0, 1, 2, ...: one every half slice.  A multiplies slice kt between
      // G(2kt) and G(2kt + 2), B between G(2kt + 1) and G(2kt + 3).  Both read the fragments of
      // slice kt + 1 (into the other register set) in the FIRST half of their slice kt: A in
      // (2kt, 2kt + 1), B in (2kt + 1, 2kt + 2) — so stage (kt + 1) % 2 is read during
      // (2kt, 2kt + 2), has to be complete at G(2kt) and is free again from G(2kt + 2).  Slice s
      // is therefore copied in (2s - 4, 2s - 2), by every wave its usual share: A issues it
      // behind G(2s - 4) = the start of its slice s - 2, B behind the same barrier = the middle of
      // its slice s - 3, and each waits for its copies (vmcnt(0)) in front of G(2s - 2).  A wave's
      // LDS reads are complete (lgkmcnt(0)) in front of every barrier.
      static_assert(PP == 0 || (WN == 4 && NG % 2 == 0), "ping-pong: eight waves, even group count");
      constexpr int H = NG / 2;
      typedef std::integral_constant<int, 0> Z0;
      typedef std::integral_constant<int, H> ZH;
      const int grp = __builtin_amdgcn_readfirstlane(wr);
      if (grp == 0) {
        auto slice_a = [&](const int kt, auto bufc) {
          constexpr int buf = decltype(bufc)::value;
          barrier_all();                                                        // G(2 kt)
          stage_pieces<PW>(rsrc, po, clamp_k(kt + 2), smem + buf * kStage + op * kSlice, pq0);
          load_frags(smem + (buf ^ 1) * kStage, std::integral_constant<int, buf ^ 1>{});
          mfma_groups(bufc, Z0{}, ZH{});
          order_copies();
          order_reads();
          barrier_lgkm();                                                       // G(2 kt + 1)
          mfma_groups(bufc, ZH{}, ZH{});
        };
        for (int kt = 0; kt < T; kt += 2) {
          slice_a(kt, std::integral_constant<int, 0>{});
          if (kt + 1 < T) slice_a(kt + 1, std::integral_constant<int, 1>{});
        }
        barrier_lgkm();                                                         // G(2 T)
      } else {
        barrier_lgkm();                                                         // G(0)
        stage_pieces<PW>(rsrc, po, clamp_k(2), smem + op * kSlice, pq0);
        auto slice_b = [&](const int kt, auto bufc) {
          constexpr int buf = decltype(bufc)::value;
          barrier_lgkm();                                                       // G(2 kt + 1)
          load_frags(smem + (buf ^ 1) * kStage, std::integral_constant<int, buf ^ 1>{});
          mfma_groups(bufc, Z0{}, ZH{});
          order_reads();
          barrier_all();                                                        // G(2 kt + 2)
          stage_pieces<PW>(rsrc, po, clamp_k(kt + 3), smem + (buf ^ 1) * kStage + op * kSlice, pq0);
          mfma_groups(bufc, ZH{}, ZH{});
          order_copies();
        };
        for (int kt = 0; kt < T; kt += 2) {
          slice_b(kt, std::integral_constant<int, 0>{});
          if (kt + 1 < T) slice_b(kt + 1, std::integral_constant<int, 1>{});
        }
      }
    }

    // ---- epilogue: register r of lane (j, kq) holds out[m = j][n = MM::row(r, kq)] of its tile;
    // four consecutive registers = four consecutive n of one output row: one 16-byte store per
    // lane (the store tail of 256 workgroups ending together is issue bound)
    const bool whole = kt0 == 0 && T == Tall;
    const bool owner = kt0 + T == Tall;
    const int j = lane % MI, kq = lane / MI;
    // contributors of this tile: the workgroups holding its first and its last slice
    const int gA = sk_per > 0 ? (tile * Tall) / sk_per : 0;
    const int gB = sk_per > 0 ? ((tile + 1) * Tall - 1) / sk_per : 0;
    float* const ptile = part ? part + (int64_t)wg * (BM * BN) : nullptr;
    // Partial tiles travel between workgroups (other CUs, other XCDs' L2s) as 16-byte `sc1`
    // (agent-scope, write-through) stores and `sc1` loads: no release / acquire fence — a fence
    // writes back / invalidates a whole L2 / L1 (MI355X_MICROARCH.md: publishing 64 KB per
    // workgroup 3.0 us write-through against 8.2 us plain + release) — the producer drains its
    // stores (vmcnt(0)) in front of the count, the owner polls the count relaxed and reads with
    // sc1 loads, eight in flight per lane.
    constexpr int AUX_SC1 = 16;
    const int my_off = ((wr * 64 + j) * BN + wc * (MI * CT) + 4 * kq) * 4;   // bytes; + tile offsets below
    auto pos_off = [&](int a, int b, int rg) {  // byte offset of this lane's float4 (a, b, rg) in a partial tile
      return my_off + (MI * a * BN + MI * b + (MM::row(4 * rg, kq) - 4 * kq)) * 4;
    };
    if (!whole && owner) {
      if (tid == 0) {
        // bounded: the producers are earlier workgroups of this launch (every workgroup is resident:
        // grid <= compute units), so the count arrives within microseconds — a workspace that is
        // not zero on entry (or a lost producer) ends in a trap, a failed launch, not in a hang
        int spins = 0;
        while (__hip_atomic_load(flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gB - gA) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 22)) __builtin_trap();   // 4 M x 512 clocks: about a second
        }
        flags[tile] = 0;   // ready for the next launch (nobody else touches it before then)
      }
      __syncthreads();
      // partials in ascending workgroup order, this workgroup's on top: a fixed order
      f32x4 sum[RT][CT][MM::NR / 4];
      for (int gg = gA; gg < gB; ++gg) {
        const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(
            part + (int64_t)gg * (BM * BN), 0, BM * BN * 4, 0x00020000);
        f32x4 tmp[RT][CT][MM::NR / 4];
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int rg = 0; rg < MM::NR / 4; ++rg)
              tmp[a][b][rg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  pr, pos_off(a, b, rg), 0, AUX_SC1));
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int rg = 0; rg < MM::NR / 4; ++rg)
              sum[a][b][rg] = gg == gA ? tmp[a][b][rg] : sum[a][b][rg] + tmp[a][b][rg];
      }
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
          for (int r = 0; r < MM::NR; ++r) acc[a][b][r] = sum[a][b][r >> 2][r & 3] + acc[a][b][r];
    }
    if (!owner) {
      const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(ptile, 0, BM * BN * 4, 0x00020000);
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
          for (int rg = 0; rg < MM::NR / 4; ++rg) {
            f32x4 v;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = acc[a][b][4 * rg + u];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), pr, pos_off(a, b, rg), 0,
                                                   AUX_SC1);
          }
      // publish: every wave drains its write-through stores, workgroup barrier, one lane counts
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(flags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
#pragma unroll
    for (int b = 0; b < CT; ++b) {
#pragma unroll
      for (int rg = 0; rg < MM::NR / 4; ++rg) {
        const int nb = n0 + wc * (MI * CT) + MI * b + MM::row(4 * rg, kq);   // n of register 4 rg
        float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (bias) {
#pragma unroll
          for (int u = 0; u < 4; ++u) bv[u] = nb + u < N ? bias[nb + u] : 0.0f;
        }
#pragma unroll
        for (int a = 0; a < RT; ++a) {
          const int m = m0 + wr * 64 + MI * a + j;
          if (m >= M || nb >= N) continue;
          f32x4 v;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float x = acc[a][b][4 * rg + u] + bv[u];
            if (relu) x = fmaxf(x, 0.0f);
            v[u] = x;
          }
          float* dst = out + (int64_t)m * ldo + nb;
          if ((ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && nb + 3 < N) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (nb + u < N) dst[u] = v[u];
          }
        }
      }
    }
  }
}

}  // namespace

// Stream-K plan: units (tile, slice) per workgroup, or 0 for one workgroup per tile.  Outputs with
// fewer than half a chip's worth of tiles are dealt in runs of `per` slices over about 256
// workgroups (every run at least 8 slices: a run pays one pipeline fill).
static int f32_linear_sk_per(int M, int N, int K, int* grid) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int Tall = K / BK;
  *grid = tiles;
  if (tiles >= 128 || Tall < 16) return 0;
  const int64_t U = (int64_t)tiles * Tall;
  int per = (int)((U + 255) / 256);
  if (per < 8) per = 8;
  if (per >= Tall) return 0;
  *grid = (int)((U + per - 1) / per);
  return per;
}

extern "C" int lnz_f32_linear_splits(int M, int N, int K) {
  // 1 = one workgroup per output tile; otherwise the number of stream-K workgroups: the caller
  // then provides `partials` = lnz_f32_linear_workspace_floats(M, N, K) floats, whose LAST
  // tiles-many words (the tile counters) are ZERO on entry (they are zero again on return)
  if (M <= 0 || N <= 0 || K < BK) return 1;
  int grid = 1;
  return f32_linear_sk_per(M, N, K, &grid) > 0 ? grid : 1;
}

extern "C" int64_t lnz_f32_linear_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K < BK) return 0;
  int grid = 1;
  if (f32_linear_sk_per(M, N, K, &grid) == 0) return 0;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  return (int64_t)grid * BM * BN + tiles;
}

extern "C" int lnz_f32_linear(const float* x, int ldx, const float* w, int ldw, const float* bias,
                              int relu, int M, int N, int K, float* out, int ldo, float* partials,
                              lnz_stream_t stream) {
  LNZ_REQUIRE(x && w && out && M > 0 && N > 0 && K > 0, LNZ_EINVAL,
              "lnz_f32_linear: bad arguments (M=%d N=%d K=%d)", M, N, K);
  LNZ_REQUIRE(K % BK == 0 && ldx >= K && ldw >= K && ldx % 4 == 0 && ldw % 4 == 0 && ldo >= N &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0,
              LNZ_ENOTSUP,
              "lnz_f32_linear: K=%d must be a multiple of %d and the operand rows 16-byte aligned",
              K, BK);
  LNZ_REQUIRE(ldx < (1 << 22) && ldw < (1 << 22), LNZ_ENOTSUP,
              "lnz_f32_linear: row strides %d / %d: a 128-row operand tile must span < 2^31 bytes",
              ldx, ldw);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int grid = tiles_m * tiles_n;
  // stream-K only with a workspace (16-byte aligned: the partial tiles are stored as float4)
  int sk_per = 0;
  if (partials && (reinterpret_cast<uintptr_t>(partials) & 15) == 0) sk_per = f32_linear_sk_per(M, N, K, &grid);
  if (sk_per == 0) grid = tiles_m * tiles_n;
  // block height of an XCD's share of the tiles (0: plain row-major order)
  int bh = 0;
  if (sk_per == 0 && grid % 8 == 0) {
    const int per = grid / 8;
    for (int h = 1; h <= tiles_m && h * h <= per; ++h)
      if (tiles_m % h == 0 && per % h == 0 && tiles_n % (per / h) == 0 &&
          (tiles_m / h) * (tiles_n / (per / h)) == 8)
        bh = h;
  }
  hipStream_t s = (hipStream_t)stream;
  int* flags = sk_per ? reinterpret_cast<int*>(partials + (int64_t)grid * BM * BN) : nullptr;
  auto kfn = f32_linear_kernel<LNZ_F32LIN_WN, LNZ_F32LIN_MI, (LNZ_F32LIN_WN == 4 ? LNZ_F32LIN_PP : 0)>;
  LNZ_DYNAMIC_LDS(kfn, kLds, "f32_linear.hip");
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(128 * LNZ_F32LIN_WN), kLds, s, x, ldx, w, ldw, bias,
                     relu, M, N, K, tiles_n, bh, out, ldo, sk_per ? partials : nullptr, flags, sk_per);
  return lnz::check_launch("lnz_f32_linear");
}
