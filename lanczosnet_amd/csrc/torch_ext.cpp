// torch extension over the C ABI (include/lanczosnet_hip.h): the ops of the LanczosNet forward step
// registered with the dispatcher as torch.ops.lanczosnet.* (BASELINE.json north_star: "hand-written
// HIP C++ kernels bound as a torch extension").  Host code only — the kernels live in
// liblanczosnet_hip.so; this file adds what ATen brings: tensor / dtype / device checks, the device
// guard, the CURRENT HIP stream of the calling thread, output allocation from the caching
// allocator, and no Python-side marshalling on the step's critical path.
//
// Reference native op surface mirrored here as well: `unsorted_segment_sum_forward/_backward`
// taking tensors (operators/src/segment_reduction_cuda.cpp:8-38).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/lanczosnet_hip.h"

namespace {

using at::Tensor;

void check(int code, const char* what) {
  TORCH_CHECK(code == LNZ_OK, what, ": lanczosnet_hip error ", code, ": ", lnz_last_error());
}

// A ROCm build of torch presents its HIP devices under the "cuda" device type: the current stream
// of the calling thread is the masquerading one, the guard the type-erased c10::DeviceGuard.
lnz_stream_t cur_stream() {
  return (lnz_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
}

void need(const Tensor& t, at::ScalarType dt, const char* name, bool contiguous = true) {
  TORCH_CHECK(t.is_cuda(), "lanczosnet: ", name, " must be a HIP (cuda) tensor; there is no CPU path");
  TORCH_CHECK(t.scalar_type() == dt, "lanczosnet: ", name, " must be ", dt, ", got ", t.scalar_type());
  TORCH_CHECK(!contiguous || t.is_contiguous(), "lanczosnet: ", name, " must be contiguous");
}

const void* optr(const c10::optional<Tensor>& t) { return t.has_value() ? t->data_ptr() : nullptr; }

// ---- helpers of the generated raw ops (torch_ext_abi.inc): device pointer of an optional tensor
// (None -> NULL) after checking device and element type
void* raw_ptr(const c10::optional<Tensor>& t, at::ScalarType dt, const char* name) {
  if (!t.has_value()) return nullptr;
  TORCH_CHECK(t->is_cuda(), "lanczosnet: ", name, " must be a HIP (cuda) tensor; there is no CPU path");
  // (uint8 buffers double as bool masks; int32 as uint32 words)
  TORCH_CHECK(t->scalar_type() == dt || (dt == at::kByte && t->scalar_type() == at::kBool),
              "lanczosnet: ", name, " must be ", dt, ", got ", t->scalar_type());
  return t->data_ptr();
}
void* raw_2byte(const c10::optional<Tensor>& t, const char* name) {   // bf16 / fp16 / int16 planes
  if (!t.has_value()) return nullptr;
  TORCH_CHECK(t->is_cuda(), "lanczosnet: ", name, " must be a HIP (cuda) tensor; there is no CPU path");
  TORCH_CHECK(t->element_size() == 2, "lanczosnet: ", name, " must have 2-byte elements, got ",
              t->scalar_type());
  return t->data_ptr();
}
void* raw_any(const c10::optional<Tensor>& t, const char* name) {     // void* workspaces
  if (!t.has_value()) return nullptr;
  TORCH_CHECK(t->is_cuda(), "lanczosnet: ", name, " must be a HIP (cuda) tensor; there is no CPU path");
  return t->data_ptr();
}
const Tensor* first_defined(std::initializer_list<const Tensor*> ts) {
  for (const Tensor* t : ts)
    if (t && t->defined()) return t;
  return nullptr;
}

#include "torch_ext_abi.inc"

// ---- the four launches that take an argument block (lnz_forward_args): forward (which = 0),
// input_grad (1), messages (2), gain_grad (3).  `in` holds the read-only device operands in the
// order of kIn below (None = NULL), `dims` the scalar fields in the order of kDim; the buffers a
// launch writes are separate, alias-annotated arguments.
enum { kNodeFeat, kNodeFeatF, kEmbedding, kMask, kLp, kV, kG, kWp, kBias, kWpHead, kBiasHead,
       kPlan, kNwg, kAct, kX0, kIdent, kRowOff, kStrips, kNStrips, kNumIn };
enum { dB, dN, dK, dNumLayer, dDin0, dDhid, dDout, dNLong, dNEdge, dNumAtom, dFilterKind, dGemmMode,
       dPlanCap, dBwdDin0, dMsgLayer, dDyCompactRows, dStripCap, kNumDim };
void fused_launch(int64_t which, const c10::List<c10::optional<Tensor>>& in, at::IntArrayRef dims,
                  at::IntArrayRef w_off, at::IntArrayRef b_off,
                  at::IntArrayRef short_dist, const c10::optional<Tensor>& score,
                  const c10::optional<Tensor>& state_out, const c10::optional<Tensor>& act_out,
                  const c10::optional<Tensor>& dy, const c10::optional<Tensor>& dx0,
                  const c10::optional<Tensor>& msg, const c10::optional<Tensor>& dgains,
                  const c10::optional<Tensor>& dy_compact, const c10::optional<Tensor>& dbias_part) {
  TORCH_CHECK((int)in.size() == kNumIn && (int)dims.size() == kNumDim && which >= 0 && which <= 3,
              "lanczosnet::fused_launch: ", (int)kNumIn, " operands and ", (int)kNumDim, " dims expected");
  TORCH_CHECK(w_off.size() <= 16 && b_off.size() <= 16 && short_dist.size() <= 8);
  auto opt = [&](int i) -> c10::optional<Tensor> { return in.get(i); };
  const c10::optional<Tensor> v = opt(kV);
  TORCH_CHECK(v.has_value() && v->dim() == 3, "lanczosnet::fused_launch: V [B,N,K] is required");
  lnz_forward_args a = {};
  a.B = dims[dB], a.N = dims[dN], a.K = dims[dK], a.num_layer = dims[dNumLayer];
  a.din0 = dims[dDin0], a.dhid = dims[dDhid], a.dout = dims[dDout];
  a.n_short = short_dist.size(), a.n_long = dims[dNLong], a.n_edge = dims[dNEdge];
  a.num_atom = dims[dNumAtom], a.filter_kind = dims[dFilterKind], a.gemm_mode = dims[dGemmMode];
  a.plan_wg_cap = dims[dPlanCap], a.bwd_din0 = dims[dBwdDin0], a.msg_layer = dims[dMsgLayer];
  TORCH_CHECK(v->size(0) == a.B && v->size(1) == a.N && v->size(2) == a.K && v->is_contiguous(),
              "lanczosnet::fused_launch: V does not match (B, N, K)");
  for (size_t i = 0; i < short_dist.size(); ++i) a.short_dist[i] = (int32_t)short_dist[i];
  for (size_t i = 0; i < w_off.size(); ++i) a.w_off[i] = w_off[i];
  for (size_t i = 0; i < b_off.size(); ++i) a.b_off[i] = b_off[i];
  a.node_feat = (const int64_t*)raw_ptr(opt(kNodeFeat), at::kLong, "node_feat");
  a.node_feat_f = (const float*)raw_ptr(opt(kNodeFeatF), at::kFloat, "node_feat_f");
  a.embedding = (const float*)raw_ptr(opt(kEmbedding), at::kFloat, "embedding");
  a.mask = (const uint8_t*)raw_ptr(opt(kMask), at::kByte, "mask");
  a.Lp = (const float*)(dims[dGemmMode] == 1 ? raw_2byte(opt(kLp), "Lp (float16 split pack, gemm_mode 1)")
                                              : raw_ptr(opt(kLp), at::kFloat, "Lp"));
  a.V = (const float*)raw_ptr(v, at::kFloat, "V");
  a.G = (const float*)raw_ptr(opt(kG), at::kFloat, "G");
  a.Wp = (const float*)raw_ptr(opt(kWp), at::kFloat, "Wp");
  a.bias = (const float*)raw_ptr(opt(kBias), at::kFloat, "bias");
  a.Wp_head = (const float*)raw_ptr(opt(kWpHead), at::kFloat, "Wp_head");
  a.bias_head = (const float*)raw_ptr(opt(kBiasHead), at::kFloat, "bias_head");
  a.plan = (const int32_t*)raw_ptr(opt(kPlan), at::kInt, "plan");
  a.n_wg = (const int32_t*)raw_ptr(opt(kNwg), at::kInt, "n_wg");
  a.act = (const float*)raw_ptr(opt(kAct), at::kFloat, "act");
  a.x0 = (const float*)raw_ptr(opt(kX0), at::kFloat, "x0");
  a.ident = (const uint32_t*)raw_ptr(opt(kIdent), at::kInt, "ident");
  a.row_off = (const int64_t*)raw_ptr(opt(kRowOff), at::kLong, "row_off");
  a.strips = (const int32_t*)raw_ptr(opt(kStrips), at::kInt, "strips");
  a.n_strips = (const int32_t*)raw_ptr(opt(kNStrips), at::kInt, "n_strips");
  a.strip_cap = dims[dStripCap];
  if (a.strips)
    TORCH_CHECK(a.n_strips && a.strip_cap > 0 &&
                    opt(kStrips)->numel() >= (int64_t)a.strip_cap * LNZ_STRIP_INTS,
                "lanczosnet::fused_launch: strips shorter than strip_cap entries, or n_strips missing");
  // shapes of the operands whose extents the kernels take on trust
  if (a.mask) TORCH_CHECK(opt(kMask)->numel() == (int64_t)a.B * a.N, "lanczosnet::fused_launch: mask is not [B,N]");
  if (a.node_feat) TORCH_CHECK(opt(kNodeFeat)->numel() == (int64_t)a.B * a.N, "lanczosnet::fused_launch: node_feat is not [B,N]");
  if (a.node_feat_f)
    TORCH_CHECK(opt(kNodeFeatF)->numel() == (int64_t)a.B * a.N * a.din0, "lanczosnet::fused_launch: node_feat is not [B,N,din0]");
  if (a.G)
    TORCH_CHECK(opt(kG)->numel() == (int64_t)a.num_layer * a.B * a.n_long * a.K * (a.filter_kind == 1 ? a.K : 1) ||
                    opt(kG)->numel() == (int64_t)a.num_layer * a.B * a.n_long * a.K,
                "lanczosnet::fused_launch: G does not match (num_layer, B, n_long, K[, K])");
  if (a.plan) TORCH_CHECK(opt(kPlan)->numel() >= 12 * (int64_t)a.plan_wg_cap, "lanczosnet::fused_launch: plan shorter than 12 * cap");
  a.score = (float*)raw_ptr(score, at::kFloat, "score");
  a.state_out = (float*)raw_ptr(state_out, at::kFloat, "state_out");
  a.act_out = (float*)raw_ptr(act_out, at::kFloat, "act_out");
  a.dy = (float*)raw_ptr(dy, at::kFloat, "dy");
  a.dx0 = (float*)raw_ptr(dx0, at::kFloat, "dx0");
  a.msg = (float*)raw_ptr(msg, at::kFloat, "msg");
  a.dgains = (float*)raw_ptr(dgains, at::kFloat, "dgains");
  a.dy_compact = (float*)raw_ptr(dy_compact, at::kFloat, "dy_compact");
  a.dy_compact_rows = dims[dDyCompactRows];
  a.dbias_part = (float*)raw_ptr(dbias_part, at::kFloat, "dbias_part");
  if (a.dy_compact)
    TORCH_CHECK(dy_compact->numel() >= (int64_t)a.num_layer * a.dy_compact_rows * a.dhid && a.row_off,
                "lanczosnet::fused_launch: dy_compact needs [num_layer, rows, dhid] and row_off");
  if (a.dbias_part) {
    TORCH_CHECK(dbias_part->numel() >= 2 * (int64_t)(a.plan ? a.plan_wg_cap : (a.B + 3) / 4) * a.num_layer * a.dhid,
                "lanczosnet::fused_launch: dbias_part shorter than [2 * workgroups, num_layer, dhid]");
    a.dbias_part_cap = (int32_t)(dbias_part->numel() / ((int64_t)a.num_layer * a.dhid));
  }
  const c10::DeviceGuard guard(v->device());
  int rc;
  switch (which) {
    case 0: rc = lnz_lanczosnet_forward(&a, cur_stream()); break;
    case 1: rc = lnz_lanczosnet_input_grad(&a, cur_stream()); break;
    case 2: rc = lnz_lanczosnet_messages(&a, cur_stream()); break;
    default: rc = lnz_lanczosnet_gain_grad(&a, cur_stream()); break;
  }
  check(rc, which == 0 ? "lanczosnet_forward" : which == 1 ? "lanczosnet_input_grad"
                                             : which == 2 ? "lanczosnet_messages" : "lanczosnet_gain_grad");
}

// ---- R1 ------------------------------------------------------------------------------------
Tensor laplacian_l4(const Tensor& adjs, const Tensor& n_nodes) {
  need(adjs, at::kFloat, "adjs");
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(adjs.dim() == 4 && adjs.size(1) == adjs.size(2) && n_nodes.numel() == adjs.size(0));
  const c10::DeviceGuard guard(adjs.device());
  const int B = adjs.size(0), N = adjs.size(1), E = adjs.size(3);
  Tensor L = at::empty({B, N, N, E + 1}, adjs.options());
  check(lnz_laplacian_l4(adjs.data_ptr<float>(), n_nodes.data_ptr<int32_t>(), B, N, E,
                         L.data_ptr<float>(), cur_stream()), "laplacian_l4");
  return L;
}

// ---- R2 + R6 -------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> lanczos_ritz(const Tensor& A, const Tensor& n_nodes, int64_t K) {
  need(A, at::kFloat, "A", /*contiguous=*/false);  // any strides: channel 0 of a channels-last L
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(A.dim() == 3 && A.size(1) == A.size(2) && n_nodes.numel() == A.size(0) && K > 0);
  const c10::DeviceGuard guard(A.device());
  const int B = A.size(0), N = A.size(1);
  Tensor D = at::empty({B, K}, A.options());
  Tensor V = at::empty({B, N, K}, A.options());
  Tensor info = at::empty({B}, n_nodes.options());
  const int64_t need_ws = lnz_lanczos_ritz_workspace_bytes(B, N);
  if (N > 32) {
    Tensor ws = at::empty({need_ws > 0 ? need_ws : 1}, A.options().dtype(at::kByte));
    check(lnz_lanczos_ritz_ws(A.data_ptr<float>(), A.stride(0), A.stride(1), A.stride(2),
                              n_nodes.data_ptr<int32_t>(), B, N, (int)K, D.data_ptr<float>(),
                              V.data_ptr<float>(), info.data_ptr<int32_t>(),
                              need_ws > 0 ? ws.data_ptr() : nullptr, need_ws, 0, cur_stream()),
          "lanczos_ritz");
  } else {
    check(lnz_lanczos_ritz(A.data_ptr<float>(), A.stride(0), A.stride(1), A.stride(2),
                           n_nodes.data_ptr<int32_t>(), B, N, (int)K, D.data_ptr<float>(),
                           V.data_ptr<float>(), info.data_ptr<int32_t>(), cur_stream()),
          "lanczos_ritz");
  }
  return {D, V, info};
}

// ---- batch preparation: pack + plan + Ritz pairs in one launch ------------------------------
// returns (Lp [B,C,4,64,4], ident [B], plan [12 cap + 2 + B K (+ scap LNZ_STRIP_INTS + 1)] = tile plan
//          | n_wg | n_rows | rows (| strip plan | n_strips, for B <= LNZ_STRIP_MAX_B), D [B,K], V [B,N,K])
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> prepare_batch(const Tensor& L, const Tensor& mask,
                                                                 const Tensor& n_nodes, int64_t K,
                                                                 int64_t n_cu, bool allow_pairs) {
  need(L, at::kFloat, "L", /*contiguous=*/false);
  need(mask, at::kByte, "mask");
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(L.dim() == 4 && L.size(1) == L.size(2) && mask.size(0) == L.size(0) &&
              mask.size(1) == L.size(1));
  const c10::DeviceGuard guard(L.device());
  const int B = L.size(0), N = L.size(1), C = L.size(3);
  const int cap = lnz_plan_wg_cap(B, (int)n_cu);
  auto iopt = n_nodes.options();
  Tensor Lp = at::empty({B, C, 4, 64, 4}, L.options());
  Tensor ident = at::empty({B}, iopt);
  const int scap = lnz_strip_cap(B);
  const int64_t soff = 12 * (int64_t)cap + 2 + (int64_t)B * K;
  Tensor plan = at::empty({soff + (scap ? (int64_t)scap * LNZ_STRIP_INTS + 1 : 0)}, iopt);
  Tensor D = at::empty({B, K}, L.options());
  Tensor V = at::empty({B, N, K}, L.options());
  int32_t* pb = plan.data_ptr<int32_t>();
  check(lnz_prepare_batch(L.data_ptr<float>(), L.stride(0), L.stride(1), L.stride(2), L.stride(3), B,
                          N, C, Lp.data_ptr<float>(), mask.data_ptr<uint8_t>(),
                          n_nodes.data_ptr<int32_t>(), (int)n_cu, allow_pairs ? 1 : 0, pb,
                          pb + 12 * cap, (int)K, pb + 12 * cap + 2, pb + 12 * cap + 1,
                          D.data_ptr<float>(), V.data_ptr<float>(), nullptr,
                          (uint32_t*)ident.data_ptr<int32_t>(), scap ? pb + soff : nullptr,
                          scap ? pb + soff + (int64_t)scap * LNZ_STRIP_INTS : nullptr, cur_stream()),
        "prepare_batch");
  return {Lp, ident, plan, D, V};
}

// plan + Ritz pairs without the pack (lnz_prepare_batch with Lp = NULL): the caller packs on a
// second stream, under the spectral-gains launch
std::tuple<Tensor, Tensor, Tensor> plan_ritz(const Tensor& L, const Tensor& mask, const Tensor& n_nodes,
                                             int64_t K, int64_t n_cu, bool allow_pairs) {
  need(L, at::kFloat, "L", /*contiguous=*/false);
  need(mask, at::kByte, "mask");
  need(n_nodes, at::kInt, "n_nodes");
  TORCH_CHECK(L.dim() == 4 && L.size(1) == L.size(2) && mask.size(0) == L.size(0) &&
              mask.size(1) == L.size(1));
  const c10::DeviceGuard guard(L.device());
  const int B = L.size(0), N = L.size(1), C = L.size(3);
  const int cap = lnz_plan_wg_cap(B, (int)n_cu);
  auto iopt = n_nodes.options();
  const int scap = lnz_strip_cap(B);
  const int64_t soff = 12 * (int64_t)cap + 2 + (int64_t)B * K;
  Tensor plan = at::empty({soff + (scap ? (int64_t)scap * LNZ_STRIP_INTS + 1 : 0)}, iopt);
  Tensor D = at::empty({B, K}, L.options());
  Tensor V = at::empty({B, N, K}, L.options());
  int32_t* pb = plan.data_ptr<int32_t>();
  check(lnz_prepare_batch(L.data_ptr<float>(), L.stride(0), L.stride(1), L.stride(2), L.stride(3), B,
                          N, C, nullptr, mask.data_ptr<uint8_t>(), n_nodes.data_ptr<int32_t>(),
                          (int)n_cu, allow_pairs ? 1 : 0, pb, pb + 12 * cap, (int)K, pb + 12 * cap + 2,
                          pb + 12 * cap + 1, D.data_ptr<float>(), V.data_ptr<float>(), nullptr, nullptr,
                          scap ? pb + soff : nullptr,
                          scap ? pb + soff + (int64_t)scap * LNZ_STRIP_INTS : nullptr, cur_stream()),
        "plan_ritz");
  return {plan, D, V};
}

// ---- R7a: spectral gains ---------------------------------------------------------------------
Tensor spectral_gains(const Tensor& D, at::IntArrayRef dist, int64_t num_layer,
                      const c10::optional<Tensor>& mlp_pack, const c10::optional<Tensor>& rows,
                      const c10::optional<Tensor>& n_rows, bool zero_fill,
                      const c10::optional<Tensor>& split_pack, const c10::optional<Tensor>& split_dst) {
  need(D, at::kFloat, "D");
  // split_pack: the batch's packed Laplacian (fp32), converted INTO split_dst (2-byte elements, the
  // same number of bytes) — the split-precision forward's form — by workgroups that ride along with
  // the MLP launch (lnz_spectral_gains_rows_split_to)
  TORCH_CHECK(split_pack.has_value() == split_dst.has_value(),
              "lanczosnet::spectral_gains: split_pack and split_dst come together");
  if (split_pack.has_value()) {
    need(*split_pack, at::kFloat, "split_pack");
    TORCH_CHECK(split_dst->is_cuda() && split_dst->is_contiguous() && split_dst->element_size() == 2 &&
                    split_dst->numel() == 2 * split_pack->numel(),
                "lanczosnet::spectral_gains: split_dst must be a contiguous 2-byte tensor of twice the pack's elements");
    TORCH_CHECK(mlp_pack.has_value() && split_pack->numel() % 4 == 0,
                "lanczosnet::spectral_gains: split_pack rides along with the MLP launch only");
  }
  TORCH_CHECK(D.dim() == 2 && !dist.empty());
  if (mlp_pack.has_value()) need(*mlp_pack, at::kFloat, "mlp_pack");
  const bool use_rows = rows.has_value() && n_rows.has_value() && mlp_pack.has_value();
  if (use_rows) {
    need(*rows, at::kInt, "rows");
    need(*n_rows, at::kInt, "n_rows");
  }
  const c10::DeviceGuard guard(D.device());
  const int B = D.size(0), K = D.size(1), S = dist.size();
  const int64_t n = num_layer * (int64_t)B * S * K;
  // + 64 B of slack: the split-precision forward reads gains as whole dwordx4 groups
  Tensor buf = (use_rows && zero_fill) ? at::zeros({n + 16}, D.options()) : at::empty({n + 16}, D.options());
  Tensor G = buf.narrow(0, 0, n).view({num_layer, B, S, K});
  std::vector<int32_t> d32(dist.begin(), dist.end());
  check(lnz_spectral_gains_rows_split_to(D.data_ptr<float>(), B, K, d32.data(), S, (int)num_layer,
                                         mlp_pack.has_value() ? 0 : 1, (const float*)optr(mlp_pack),
                                         use_rows ? rows->data_ptr<int32_t>() : nullptr,
                                         use_rows ? n_rows->data_ptr<int32_t>() : nullptr, G.data_ptr<float>(),
                                         split_pack.has_value() ? split_pack->data_ptr<float>() : nullptr,
                                         split_dst.has_value() ? (uint16_t*)split_dst->data_ptr() : nullptr,
                                         split_pack.has_value() ? split_pack->numel() : 0, cur_stream()),
        "spectral_gains");
  return G;
}

// ---- R7b + R9 + R10: the fused forward (exact-fp32 kernel) -------------------------------------
// dims = [num_layer, din0, dhid, dout, n_long, n_edge, filter_kind (, gemm_mode: 0, or 1 = split-
// precision GEMM1 on strips with Wp from lnz_pack_rows_k8_split)]
Tensor forward(const Tensor& node_feat, const c10::optional<Tensor>& embedding, const Tensor& Lp,
               const c10::optional<Tensor>& ident, const Tensor& V, const c10::optional<Tensor>& G,
               const Tensor& mask, const Tensor& Wp, const Tensor& bias, at::IntArrayRef w_off,
               at::IntArrayRef b_off, const Tensor& Wp_head, const Tensor& bias_head,
               const c10::optional<Tensor>& plan, int64_t plan_cap, at::IntArrayRef dims,
               at::IntArrayRef short_dist, const c10::optional<Tensor>& strips, int64_t strip_cap) {
  TORCH_CHECK(dims.size() == 7 || dims.size() == 8,
              "lanczosnet::forward: dims = [num_layer, din0, dhid, dout, n_long, n_edge, filter_kind(, gemm_mode)]");
  TORCH_CHECK(dims.size() == 7 || dims[7] == 0 || dims[7] == 1, "lanczosnet::forward: gemm_mode 0 or 1");
  // the pack's element type carries its format: fp32 fragments for the exact kernel, fp16 hi | lo
  // pieces (lnz_split_laplacian_pack_to) for gemm_mode 1
  if (dims.size() > 7 && dims[7] == 1) {
    TORCH_CHECK(Lp.is_cuda() && Lp.is_contiguous() && Lp.scalar_type() == at::kHalf,
                "lanczosnet::forward: gemm_mode 1 reads a float16 (split) Laplacian pack, got ", Lp.scalar_type());
  } else {
    need(Lp, at::kFloat, "Lp");
  }
  need(V, at::kFloat, "V");
  need(mask, at::kByte, "mask");
  need(Wp, at::kFloat, "Wp");
  need(bias, at::kFloat, "bias");
  need(Wp_head, at::kFloat, "Wp_head");
  need(bias_head, at::kFloat, "bias_head");
  TORCH_CHECK(V.dim() == 3 && (int64_t)w_off.size() >= dims[0] && (int64_t)b_off.size() >= dims[0] &&
              dims[0] <= 16 && short_dist.size() <= 8);
  TORCH_CHECK(mask.numel() == V.size(0) * V.size(1), "lanczosnet::forward: mask is not [B,N]");
  TORCH_CHECK(node_feat.size(0) == V.size(0) && node_feat.size(1) == V.size(1),
              "lanczosnet::forward: node_feat leading dims do not match V's");
  if (G.has_value())
    TORCH_CHECK(G->dim() >= 4 && G->size(0) == dims[0] && G->size(1) == V.size(0) && G->size(2) == dims[4] &&
                    G->size(3) == V.size(2) && (G->dim() == 4 || (G->dim() == 5 && G->size(4) == V.size(2))),
                "lanczosnet::forward: G must be [num_layer, B, n_long, K] or [..., K, K]");
  if (plan.has_value())
    TORCH_CHECK(plan->numel() >= 12 * plan_cap + 1, "lanczosnet::forward: plan shorter than 12 * cap + 1");
  TORCH_CHECK(bias.numel() >= b_off[dims[0] - 1] + dims[2], "lanczosnet::forward: bias shorter than b_off + dhid");
  const c10::DeviceGuard guard(V.device());
  lnz_forward_args a = {};
  a.B = V.size(0), a.N = V.size(1), a.K = V.size(2);
  a.num_layer = dims[0], a.din0 = dims[1], a.dhid = dims[2], a.dout = dims[3];
  a.n_short = short_dist.size(), a.n_long = dims[4], a.n_edge = dims[5];
  a.filter_kind = dims[6];
  a.gemm_mode = dims.size() > 7 ? (int32_t)dims[7] : 0;
  for (size_t i = 0; i < short_dist.size(); ++i) a.short_dist[i] = (int32_t)short_dist[i];
  if (node_feat.scalar_type() == at::kLong) {
    need(node_feat, at::kLong, "node_feat");
    TORCH_CHECK(embedding.has_value(), "lanczosnet::forward: atom ids need the embedding table");
    need(*embedding, at::kFloat, "embedding");
    a.node_feat = node_feat.data_ptr<int64_t>();
    a.embedding = embedding->data_ptr<float>();
    a.num_atom = embedding->size(0);
  } else {
    need(node_feat, at::kFloat, "node_feat");
    TORCH_CHECK(node_feat.size(-1) == a.din0, "lanczosnet::forward: float features must be padded "
                                              "to din0 columns");
    a.node_feat_f = node_feat.data_ptr<float>();
  }
  a.mask = mask.data_ptr<uint8_t>();
  a.V = V.data_ptr<float>();
  a.Lp = (const float*)Lp.data_ptr();
  if (ident.has_value()) {
    need(*ident, at::kInt, "ident");
    a.ident = (const uint32_t*)ident->data_ptr<int32_t>();
  }
  if (G.has_value()) {
    need(*G, at::kFloat, "G");
    a.G = G->data_ptr<float>();
  }
  a.Wp = Wp.data_ptr<float>(), a.bias = bias.data_ptr<float>();
  for (int i = 0; i < a.num_layer; ++i) a.w_off[i] = w_off[i], a.b_off[i] = b_off[i];
  a.Wp_head = Wp_head.data_ptr<float>(), a.bias_head = bias_head.data_ptr<float>();
  if (plan.has_value()) {
    need(*plan, at::kInt, "plan");
    a.plan = plan->data_ptr<int32_t>();
    a.n_wg = plan->data_ptr<int32_t>() + 12 * plan_cap;
    a.plan_wg_cap = (int)plan_cap;
  }
  if (strips.has_value()) {  // [strip_cap * LNZ_STRIP_INTS + 1]: the strip plan, then n_strips
    need(*strips, at::kInt, "strips");
    TORCH_CHECK(strip_cap > 0 && strips->numel() >= strip_cap * LNZ_STRIP_INTS + 1,
                "lanczosnet::forward: strips shorter than strip_cap entries + 1");
    a.strips = strips->data_ptr<int32_t>();
    a.n_strips = strips->data_ptr<int32_t>() + strip_cap * LNZ_STRIP_INTS;
    a.strip_cap = (int)strip_cap;
  }
  Tensor score = at::empty({a.B, a.dout}, V.options());
  a.score = score.data_ptr<float>();
  check(lnz_lanczosnet_forward(&a, cur_stream()), "forward");
  return score;
}

// ---- R12: the reference's native op surface, on tensors ----------------------------------------
Tensor segment_sum_forward(const Tensor& data, const Tensor& segment_ids, int64_t num_segments) {
  need(data, at::kFloat, "data");
  need(segment_ids, at::kLong, "segment_ids");
  TORCH_CHECK(data.dim() == 3 && segment_ids.dim() == 2 && segment_ids.size(0) == data.size(0) &&
              segment_ids.size(1) == data.size(1));
  const c10::DeviceGuard guard(data.device());
  Tensor out = at::zeros({data.size(0), num_segments, data.size(2)}, data.options());
  check(lnz_unsorted_segment_sum_forward(data.data_ptr<float>(), segment_ids.data_ptr<int64_t>(),
                                         data.size(0), data.size(1), data.size(2), (int)num_segments,
                                         out.data_ptr<float>(), cur_stream()),
        "unsorted_segment_sum_forward");
  return out;
}

Tensor segment_sum_backward(const Tensor& grad_out, const Tensor& segment_ids, int64_t dim1) {
  need(grad_out, at::kFloat, "grad_out");
  need(segment_ids, at::kLong, "segment_ids");
  TORCH_CHECK(grad_out.dim() == 3 && segment_ids.dim() == 2 && segment_ids.size(0) == grad_out.size(0) &&
                  segment_ids.size(1) == dim1,
              "lanczosnet::unsorted_segment_sum_backward: segment_ids must be [B, dim1] for grad_out [B, S, D]");
  const c10::DeviceGuard guard(grad_out.device());
  Tensor gd = at::zeros({grad_out.size(0), dim1, grad_out.size(2)}, grad_out.options());
  check(lnz_unsorted_segment_sum_backward(grad_out.data_ptr<float>(), segment_ids.data_ptr<int64_t>(),
                                          grad_out.size(0), (int)dim1, grad_out.size(2),
                                          grad_out.size(1), gd.data_ptr<float>(), cur_stream()),
        "unsorted_segment_sum_backward");
  return gd;
}

}  // namespace

TORCH_LIBRARY(lanczosnet, m) {
  m.def("abi_version() -> int", []() -> int64_t { return lnz_abi_version(); });
  m.def("last_kernel() -> str", []() -> std::string { return std::string(lnz_last_kernel()); });
  // a HIP stream confined to compute units [first_cu, end_cu) of `device`: its handle, for
  // torch.cuda.ExternalStream (lanczosnet_amd/utils/streams.py)
  m.def("cu_masked_stream(int first_cu, int end_cu, int device) -> int",
        [](int64_t first_cu, int64_t end_cu, int64_t device) -> int64_t {
          c10::DeviceGuard guard(c10::Device(c10::DeviceType::CUDA, (c10::DeviceIndex)device));
          lnz_stream_t st = nullptr;
          const int rc = lnz_stream_create_cu_masked((int)first_cu, (int)end_cu, &st);
          TORCH_CHECK(rc == 0, "lanczosnet_hip error ", rc, ": ", lnz_last_error());
          return (int64_t)reinterpret_cast<intptr_t>(st);
        });
  m.def("laplacian_l4(Tensor adjs, Tensor n_nodes) -> Tensor");
  m.def("lanczos_ritz(Tensor A, Tensor n_nodes, int K) -> (Tensor, Tensor, Tensor)");
  m.def("prepare_batch(Tensor L, Tensor mask, Tensor n_nodes, int K, int n_cu, bool allow_pairs) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("plan_ritz(Tensor L, Tensor mask, Tensor n_nodes, int K, int n_cu, bool allow_pairs) -> "
        "(Tensor, Tensor, Tensor)");
  m.def("spectral_gains(Tensor D, int[] dist, int num_layer, Tensor? mlp_pack, Tensor? rows, "
        "Tensor? n_rows, bool zero_fill, Tensor? split_pack, Tensor(a!)? split_dst) -> Tensor");
  m.def("forward(Tensor node_feat, Tensor? embedding, Tensor Lp, Tensor? ident, Tensor V, Tensor? G, "
        "Tensor mask, Tensor Wp, Tensor bias, int[] w_off, int[] b_off, Tensor Wp_head, "
        "Tensor bias_head, Tensor? plan, int plan_cap, int[] dims, int[] short_dist, Tensor? strips, "
        "int strip_cap) -> Tensor");
  m.def("unsorted_segment_sum_forward(Tensor data, Tensor segment_ids, int num_segments) -> Tensor");
  m.def("unsorted_segment_sum_backward(Tensor grad_out, Tensor segment_ids, int dim1) -> Tensor");
  m.def("fused_launch(int which, Tensor?[] operands, int[] dims, int[] w_off, int[] b_off, "
        "int[] short_dist, Tensor(a!)? score, Tensor(b!)? state_out, Tensor(c!)? act_out, Tensor(d!)? dy, "
        "Tensor(e!)? dx0, Tensor(f!)? msg, Tensor(g!)? dgains, Tensor(h!)? dy_compact, Tensor(i!)? dbias_part) -> ()");
  LNZ_RAW_DEFS(m)
}

TORCH_LIBRARY_IMPL(lanczosnet, CUDA, m) {  // the CUDA dispatch key is HIP on a ROCm build
  m.impl("laplacian_l4", laplacian_l4);
  m.impl("lanczos_ritz", lanczos_ritz);
  m.impl("prepare_batch", prepare_batch);
  m.impl("plan_ritz", plan_ritz);
  m.impl("spectral_gains", spectral_gains);
  m.impl("forward", forward);
  m.impl("unsorted_segment_sum_forward", segment_sum_forward);
  m.impl("unsorted_segment_sum_backward", segment_sum_backward);
  m.impl("fused_launch", fused_launch);
  LNZ_RAW_IMPLS_DEVICE(m)
}

// host-side size queries of the C ABI (no tensor arguments)
TORCH_LIBRARY_IMPL(lanczosnet, CompositeExplicitAutograd, m) {
  LNZ_RAW_IMPLS_HOST(m)
}
