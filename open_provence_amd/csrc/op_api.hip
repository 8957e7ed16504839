// op_api.hip -- C ABI of libopenprovence_hip.so (see include/open_provence_hip.h for the contract and
// the reference interfaces each entry point replaces).  Host side only: handle, weight re-packing,
// workspace carving, chunking of the packed batch and the launch sequence of one forward.
#include "../../include/open_provence_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define OPK_PACK_KERNELS 1
#include "op_internal.h"
#include "opk_small.hip.h"
#include "opk_tiled.hip.h"

using namespace opk;
using opl::Policy;

namespace {

thread_local std::string g_last_error;

enum ProfileKind {
  PK_ROWMAP = 0,
  PK_EMBED_LN,
  PK_LN,
  PK_GEMM_QK_ROPE,
  PK_GEMM_V_T,
  PK_ATTN_GLOBAL,
  PK_ATTN_LOCAL,
  PK_GEMM_ATTN_OUT,
  PK_GEMM_WI_GEGLU,
  PK_GEMM_MLP_OUT,
  PK_FINAL_LN_PRUNE,
  PK_RANK_HEAD,
  PK_CAPTURE,
  PK_ROW_QKV,
  PK_ROW_ATTN_OUT,
  PK_ROW_WI_GEGLU,
  PK_KSTREAM_MLP_OUT,
  PK_FUSED_ATTN_OUT_WI,
  PK_FUSED_MLP_OUT_QKV,
  PK_CLEAR_LO,
  PK_FUSED_LAYER,
  PK_GEMM_QKV_ROPE,
  PK_COUNT
};
const char* kProfileNames[PK_COUNT] = {"rowmap",        "embed_ln",      "layer_norm",    "gemm_qk_rope", "gemm_v_t",
                                       "attn_global",   "attn_local",    "gemm_attn_out", "gemm_wi_geglu",
                                       "gemm_mlp_out",  "final_ln_prune", "rank_head",    "capture",
                                       "rowgemm_ln_qkv_rope", "rowgemm_attn_out", "rowgemm_ln_wi_geglu",
                                       "kstream_mlp_out", "fused_attnout_ln_wi_geglu", "fused_mlpout_ln_qkv_rope",
                                       "clear_lo_planes", "fused_layer_attnout_mlp_qkv", "gemm_qkv_rope"};

struct LayerWeights {
  float* attn_norm = nullptr;  // absent on layer 0
  float* mlp_norm = nullptr;
  u16 *wqkv_hi = nullptr, *wqkv_lo = nullptr;
  u16 *wo_hi = nullptr, *wo_lo = nullptr;
  u16 *wi_hi = nullptr, *wi_lo = nullptr;
  u16 *wo2_hi = nullptr, *wo2_lo = nullptr;
  // row-stationary layouts (hidden <= 256): chunk-major, fragment-ordered, hi/lo planes interleaved per k-step
  u16 *wqkv_pk = nullptr, *wi_pk = nullptr;
  u16* wo2_pk = nullptr;  // k-streamed layouts (output features permuted): MLP output projection ...
  u16* wo_ks = nullptr;   // ... and attention output projection (fused kernels)
  // hi planes in the fragment order of the 32x32x16 whole-layer kernel (opk_layer32.hip.h), hidden = 256 only
  u16 *wo_p32 = nullptr, *wi_p32 = nullptr, *wo2_p32 = nullptr, *wqkv_p32 = nullptr;
  // the wave-pair whole-layer kernel (opk_layer16p.hip.h), hidden = 256: one plane in its fragment order, [0] bf16 / [1] fp16 values
  u16 *wo_pp[2] = {nullptr, nullptr}, *wi_pp[2] = {nullptr, nullptr}, *wo2_pp[2] = {nullptr, nullptr}, *wqkv_pp[2] = {nullptr, nullptr};
  // "f16 + fp8" kernel set (opk_common.hip.h): chunks of [fp16 plane | e4m3 plane] (Wqkv, Wi), fp16 k-streamed slabs
  // (attention Wo, MLP Wo) and the e4m3 K = 128 slabs of the attention Wo
  u16 *wqkv_f8 = nullptr, *wi_f8 = nullptr, *wo_f16 = nullptr, *wo_f8 = nullptr, *wo2_f16 = nullptr;
  // the same kernel sets on the panel path: per weight fp16 slabs + e4m3 slabs (w, then lo(w)) (pack_panel_f8_kernel)
  u16 *wqkv_p16 = nullptr, *wqkv_p8 = nullptr, *wo_p16 = nullptr, *wo_p8 = nullptr, *wi_p16 = nullptr, *wi_p8 = nullptr,
      *wo2_p16 = nullptr, *wo2_p8 = nullptr;
  // kernel set "f16" (single-pass fp16 operands): the layouts of wqkv_pk / wi_pk / wo2_pk / wo_ks with fp16 values in the hi plane
  u16 *wqkv_h16 = nullptr, *wi_h16 = nullptr, *wo2_h16 = nullptr, *wo_h16 = nullptr;
};

struct ProfileEvent {
  int kind;
  hipEvent_t start, stop;
};

}  // namespace

struct op_handle {
  op_config cfg;
  int H = 0, I = 0, N = 0, nh = 0, V = 0, nl = 0, max_pos = 0;
  Policy req = opl::kPolicies[0];  // requested term masks (op_config.precision / terms)
  Policy eff = opl::kPolicies[0];  // evaluated term masks: req minus weight-lo terms that are identically zero
  int pi = 0;                      // curated kernel set that runs `eff` (index into opl::kPolicies)
  bool emulate = false;            // eff is not curated: kernel set 0 with the unused lo operands cleared
  bool resolved = false;           // eff / pi / emulate are valid (set by op_weights_ready)
  float* f16_fit_dev = nullptr;    // [2] lost / total weight energy of the tensor being packed for the "f16 + fp8" sets
  int* any_lo_dev = nullptr;       // [OP_FAM_COUNT] device flags: some weight of the family has a non-zero lo element
                                   // [OP_FAM_COUNT]: some GEMM weight is not exactly an fp16 value (no "f16 + fp8" set)
  bool f8_packs = false;           // the "f16 + fp8" weight packs exist (row path, hidden a multiple of 128)
  bool f8_off = false;             // op_set_compact_operands(h, 0): keep the (hi, lo) bf16 sets although the packs exist
  bool h16_packs = false;          // the fp16 single-plane weight packs of kernel set "f16" exist
  int forced_set = -1;             // op_select_kernel_set / op_calibrate: run this kernel set (op_kernel_set numbering), -1 = the default selection
  int default_set = 0;             // the kernel set the default selection gives for this checkpoint (valid once resolved)
  bool f16_unfit = false;          // some weight TENSOR sits on fp16's subnormal grid: no kernel set with an fp16 weight plane
  bool wi_f8 = false;              // panel path, OP_FLAG_PANEL_F8_WI: the Wi GEMM (and its LayerNorm) in the fp16 + e4m3 format
  bool mlp_f8 = false;             // panel path, kernel sets 8 / 9: pi = PI_F16 for the attention side, the whole MLP in the fp16 + e4m3 format
  bool mlp_wlo = false;            // ... with the weights' lo part (set 8)
  uint64_t mlp_layers = ~0ull;     // sets 8 / 9: the layers whose MLP runs in that format (bit li); the others run the "f16" set's MLP
  uint64_t forced_mlp_layers = ~0ull;  // ... as pinned with forced_set (op_calibrate's per-layer search, op_select_mlp_correction_layers)
  bool attn_f16 = false;           // panel path, kernel sets 10 / 11: pi = PI_F16_F8_W / PI_F16_F8 with the attention on set 7's fp16 kernels
  bool row_path = false;    // hidden <= 256: row-stationary GEMMs with fused LayerNorm
  bool panel_path = false;  // hidden % 256 == 0, intermediate % 128 == 0: k-streamed panel GEMMs, fragment-packed operands
  int n_cus = 256;          // compute units of the device (hipDeviceProp multiProcessorCount)
  int chunk_rows = 0;
  float* emb = nullptr;
  float* emb_norm = nullptr;
  float* final_norm = nullptr;
  float* dense_t = nullptr;
  float* head_norm = nullptr;
  float* cls_w = nullptr;
  float* cls_b = nullptr;
  float* prune_w = nullptr;
  float* prune_b = nullptr;
  float* rope_cos[2] = {nullptr, nullptr};  // [0] = local theta, [1] = global theta
  float* rope_sin[2] = {nullptr, nullptr};
  std::vector<LayerWeights> layers;
  std::vector<void*> allocations;
  std::vector<std::string> missing;  // weight names not loaded yet
  float* capture = nullptr;
  bool profiling = false;
  std::vector<ProfileEvent> events;
  double prof_ms[PK_COUNT] = {0};
  int prof_launches[PK_COUNT] = {0};
  std::string err;
};

namespace {

int fail(op_handle* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  g_last_error = buf;
  return code;
}

#define OP_HIP(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) return fail(h, OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

template <typename T>
int dev_alloc(op_handle* h, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return fail(h, OP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  h->allocations.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return OP_OK;
}

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

void erase_missing(op_handle* h, const std::string& name) {
  auto it = std::find(h->missing.begin(), h->missing.end(), name);
  if (it != h->missing.end()) h->missing.erase(it);
}

// RoPE tables exactly as HF builds them (modeling_modernbert.py:117-163): inv_freq in fp32,
// angle = fp32(pos) * inv_freq in fp32, cos/sin of that fp32 angle.
void build_rope_host(float theta, int max_pos, std::vector<float>& cs, std::vector<float>& sn) {
  cs.resize((size_t)max_pos * ROPE_HALF);
  sn.resize((size_t)max_pos * ROPE_HALF);
  float inv_freq[ROPE_HALF];
  for (int k = 0; k < ROPE_HALF; ++k) {
    const float expo = (float)(2 * k) / (float)HEAD_DIM;
    inv_freq[k] = (float)(1.0 / std::pow((double)theta, (double)expo));
  }
  for (int p = 0; p < max_pos; ++p) {
    for (int k = 0; k < ROPE_HALF; ++k) {
      const float ang = (float)p * inv_freq[k];
      cs[(size_t)p * ROPE_HALF + k] = (float)std::cos((double)ang);
      sn[(size_t)p * ROPE_HALF + k] = (float)std::sin((double)ang);
    }
  }
}

struct Launcher {
  op_handle* h;
  hipStream_t stream;
  int begin(int kind) {
    if (!h->profiling) return OP_OK;
    ProfileEvent ev;
    ev.kind = kind;
    OP_HIP(h, hipEventCreate(&ev.start));
    OP_HIP(h, hipEventCreate(&ev.stop));
    OP_HIP(h, hipEventRecord(ev.start, stream));
    h->events.push_back(ev);
    return OP_OK;
  }
  int end() {
    OP_HIP(h, hipGetLastError());
    if (!h->profiling) return OP_OK;
    OP_HIP(h, hipEventRecord(h->events.back().stop, stream));
    return OP_OK;
  }
};

#define OP_TRY(expr)             \
  do {                           \
    int _rc = (expr);            \
    if (_rc != OP_OK) return _rc; \
  } while (0)

int drain_profile(op_handle* h) {
  for (auto& ev : h->events) {
    OP_HIP(h, hipEventSynchronize(ev.stop));
    float ms = 0.f;
    OP_HIP(h, hipEventElapsedTime(&ms, ev.start, ev.stop));
    h->prof_ms[ev.kind] += ms;
    h->prof_launches[ev.kind] += 1;
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  h->events.clear();
  return OP_OK;
}

struct Workspace {
  float* x;
  u16 *ln_hi, *ln_lo;
  u16 *q_hi, *q_lo, *k_hi, *k_lo;
  u16 *vt_hi, *vt_lo;
  u16 *o_hi, *o_lo;
  u16 *h_hi, *h_lo;
  int32_t *row_seq, *row_pos, *row_tok, *roff, *qboff, *qboff_l;
  float* cls;
  int* range_flag;  // panel path, fp16 + e4m3 format: "an fp16 operand was out of range" (PanelParams::range_flag)
  size_t bytes;
};

// rows the largest chunk can hold (before the +64 slack / 128 rounding)
int chunk_row_capacity(const op_handle* h, int n_seqs, int total_tokens, int max_seqlen) {
  const long upper = (long)total_tokens + (long)(ROW_ALIGN - 1) * n_seqs;
  const long all_rows = align_up((int)std::min<long>(upper, 1L << 30), ROW_ALIGN);
  const int one_seq = align_up(std::max(max_seqlen, 1), ROW_ALIGN);
  const long cap = std::max(h->chunk_rows, one_seq);
  return (int)std::min<long>(all_rows, cap);
}

void carve(const op_handle* h, char* base, int cap_rows_pad, int n_seqs, Workspace& ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up_sz(bytes, 256);
    return p;
  };
  const size_t R = (size_t)cap_rows_pad;
  const size_t H = (size_t)h->H, I = (size_t)h->I;
  ws.x = (float*)take(R * H * 4);
  ws.ln_hi = (u16*)take(R * H * 2);
  ws.ln_lo = (u16*)take(R * H * 2);
  ws.q_hi = (u16*)take(R * H * 2);
  ws.q_lo = (u16*)take(R * H * 2);
  ws.k_hi = (u16*)take(R * H * 2);
  ws.k_lo = (u16*)take(R * H * 2);
  ws.vt_hi = (u16*)take(R * H * 2);
  ws.vt_lo = (u16*)take(R * H * 2);
  ws.o_hi = (u16*)take(R * H * 2);
  ws.o_lo = (u16*)take(R * H * 2);
  ws.h_hi = (u16*)take(R * I * 2);
  ws.h_lo = (u16*)take(R * I * 2);
  ws.row_seq = (int32_t*)take(R * 4);
  ws.row_pos = (int32_t*)take(R * 4);
  ws.row_tok = (int32_t*)take(R * 4);
  ws.roff = (int32_t*)take(((size_t)n_seqs + 1) * 4);
  ws.qboff = (int32_t*)take(((size_t)n_seqs + 1) * 4);
  ws.qboff_l = (int32_t*)take(((size_t)n_seqs + 1) * 4);
  ws.cls = (float*)take(std::max<size_t>((size_t)n_seqs, 1) * H * 4);
  ws.range_flag = (int*)take(sizeof(int));
  ws.bytes = off;
}

template <int EPI>
int launch_gemm(Launcher& L, int kind, const GemmParams& p, bool split) {
  OP_TRY(L.begin(kind));
  const dim3 grid((unsigned)(p.n_tiles * p.m_tiles));
  if (split)
    hipLaunchKernelGGL((gemm_kernel<EPI, true>), grid, dim3(256), 0, L.stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<EPI, false>), grid, dim3(256), 0, L.stream, p);
  return L.end();
}

// Clear the lo plane of a fragment-packed tensor: a policy without that operand's lo term, evaluated on the
// all-terms kernel set (the extra MFMA pass then adds exact zeros).
int clear_lo(Launcher& L, u16* base, int unit, size_t n_pairs) {
  OP_TRY(L.begin(PK_CLEAR_LO));
  const size_t n = n_pairs * (size_t)(unit / 8);
  hipLaunchKernelGGL(zero_odd_units_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, L.stream, base, unit, n_pairs);
  return L.end();
}

struct AttnPlan {
  int waves_g;  // waves per block of the full-attention layers (8 or 4); sliding-window layers always use 4
  int items_g;  // work items (sequence, query block) of a full-attention layer
  int items_l;  // ... of a sliding-window layer (128-query blocks)
};

// One chunk of sequences through the whole encoder: the launch sequence of a forward.  The arguments and what every path
// derives from them are members; prologue() (row map, padding rows, embeddings), one *_layer() per path and layer, heads().
struct ChunkPass {
  op_handle* h;
  Launcher& L;
  const Workspace& ws;
  const int32_t* ids_dev;
  const int32_t* cu_dev;
  int s0, ns, rows, max_len, total_tokens;
  const AttnPlan& plan;
  float *prune_out, *rank_out, *keep_prob;

  int H, I;
  bool fp_layout;  // fragment-packed activations, attn_fp_kernel
  int r_pad, m_tiles;
  unsigned row_blocks;
  hipStream_t st;
  Policy E, V;  // evaluated policy, instantiated kernel set (V has every term of E)
  bool clr_q, clr_k, clr_v, clr_o, clr_h, clr_ln_attn, clr_ln_mlp, zero_p_lo, split;
  bool o_f8, f16, attn16, range_flagged, embed_in_qkv0;
  size_t plane_bytes;  // one row-major plane (tiled path)
  int q_tiles;
  bool small_blocks;
  unsigned row_grid;
  const char* no_kernel = "internal: no row-stationary kernel for hidden %d";
  bool layer_fused, head_in_last_layer;
  bool pair_layers;  // whole-layer launches as wave pairs on 32x32x16 MFMAs (opk_layer16p.hip.h)
  bool head_done = false;

  ChunkPass(op_handle* h_, Launcher& L_, const Workspace& ws_, const int32_t* ids_dev_, const int32_t* cu_dev_, int s0_, int ns_, int rows_,
            int max_len_, int total_tokens_, const AttnPlan& plan_, float* prune_out_, float* rank_out_, float* keep_prob_)
      : h(h_), L(L_), ws(ws_), ids_dev(ids_dev_), cu_dev(cu_dev_), s0(s0_), ns(ns_), rows(rows_), max_len(max_len_),
        total_tokens(total_tokens_), plan(plan_), prune_out(prune_out_), rank_out(rank_out_), keep_prob(keep_prob_) {
    H = h->H;
    I = h->I;
    // Row path: exactly the computed rows, rounded to the 128-row block -- no slack rows: a 131072-row batch is 1024
    // blocks = two full rounds of 2 blocks per CU; two extra (empty) blocks would cost a third round.  The tiled path
    // keeps 64 slack rows for its attention kernel's tile over-read.
    fp_layout = h->row_path || h->panel_path;
    r_pad = fp_layout ? align_up(rows, ROW_BM) : align_up(rows + 64, 256);
    m_tiles = r_pad / GEMM_BM;
    row_blocks = (unsigned)((r_pad + 3) / 4);
    st = L.stream;
    // Evaluated policy E, instantiated kernel set V (V has every term of E).  Where V multiplies by a lo operand that
    // E does not have, that operand is cleared: weights at load time (op_load_weight), activations in the launch sequence.
    E = h->eff;
    V = opl::kPolicies[h->pi];
    auto extra = [](int v, int e, int bit) { return (v & bit) != 0 && (e & bit) == 0; };
    clr_q = extra(V.qk, E.qk, 1);
    clr_k = extra(V.qk, E.qk, 2);
    clr_v = extra(V.pv, E.pv, 2);
    clr_o = extra(V.attn_out, E.attn_out, 1);
    clr_h = extra(V.mlp_out, E.mlp_out, 1);
    clr_ln_attn = extra(V.wqkv, E.wqkv, 1);
    clr_ln_mlp = extra(V.wi, E.wi, 1);
    zero_p_lo = extra(V.pv, E.pv, 1);
    // tiled path: two kernel sets only (single pass / all terms)
    split = (E.wqkv | E.qk | E.pv | E.attn_out | E.wi | E.mlp_out) != 0;
    o_f8 = (h->pi == opl::PI_F16_F8 || h->pi == opl::PI_F16_F8_W) && !h->emulate;  // o = fp16 pieces (ws.o_hi) + e4m3 pieces (ws.o_lo)
    f16 = h->pi == opl::PI_F16 && !h->emulate;  // kernel set "f16": set 2's layouts, fp16 values, the fp16 weight packs
    // kernel sets 10 / 11 (panel path): q / k / v^T as single-plane fp16, attention on set 7's kernels with o in sets 3 / 4's format
    attn16 = o_f8 && h->attn_f16 && h->panel_path;
    // Panel path, fp16 + e4m3 kernels (sets 3 / 4 / 8 - 11): they convert under MODE.FP16_OVFL = 1 (an h beyond fp16's range is
    // clamped) and their fp16 MFMAs take a NaN operand for a finite number, so an out-of-range activation cannot travel to the
    // outputs as Inf / NaN by itself: the kernels raise ws.range_flag and the head kernels write NaN logits.
    range_flagged = h->panel_path && !h->emulate && (o_f8 || (f16 && h->mlp_f8));
    // Row path without hidden-state capture: the layer-0 q / k / v kernel gathers and normalises the embeddings itself
    // (RowGemmParams::emb_table) -- no embedding launch, the residual rows are written once and not read back.
    embed_in_qkv0 = h->row_path && !h->capture && !(h->cfg.flags & OP_FLAG_NO_HEAD_FUSION);
    plane_bytes = (size_t)r_pad * H * sizeof(u16);
    q_tiles = (max_len + ATT_BQ - 1) / ATT_BQ;
    // 4 waves x 32 rows = 128-row blocks, two per CU.  Small batches (at most one such block per CU) use 4 waves x
    // 16 rows = 64-row blocks instead: twice the blocks, so a latency-bound request spreads over twice the CUs.
    small_blocks = (r_pad / ROW_BM) <= h->n_cus && !(h->cfg.flags & OP_FLAG_NO_SMALL_BLOCKS);
    row_grid = (unsigned)(r_pad / (small_blocks ? 64 : ROW_BM));
    // Kernel sets whose GEMM weights are single-plane run a whole layer (attention output projection, MLP, next q/k/v
    // projection) as ONE kernel with h kept on chip; the all-terms set keeps the two fused kernels per layer.
    layer_fused = h->row_path && !h->emulate && opl::has_row_layer_fused(h->pi) && !(h->cfg.flags & OP_FLAG_NO_LAYER_FUSION);
    head_in_last_layer = layer_fused && h->cfg.pooling != OP_POOL_MEAN && !h->capture && !(h->cfg.flags & OP_FLAG_NO_HEAD_FUSION);
    pair_layers = layer_fused && H == 256 && I % 64 == 0 && (f16 || h->pi == 2) && !h->capture &&
                  !(h->cfg.flags & (OP_FLAG_NO_LAYER_PAIRS | OP_FLAG_LAYER_8X16 | OP_FLAG_LAYER_M32)) &&
                  h->layers[0].wo_pp[f16 ? 1 : 0] != nullptr;
  }

  // row map, the range flag, padding rows of the attention output, embeddings
  int prologue() {
    OP_TRY(L.begin(PK_ROWMAP));
    hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, cu_dev, s0, ns, ROW_ALIGN, ROW_ALIGN, ws.roff);
    if (fp_layout) {
      hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, cu_dev, s0, ns, plan.waves_g * 32, 1, ws.qboff);
      hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, cu_dev, s0, ns, 128, 1, ws.qboff_l);
    }
    hipLaunchKernelGGL(row_map_kernel, dim3((unsigned)((r_pad + 255) / 256)), dim3(256), 0, st, cu_dev, s0, ns, ws.roff,
                       r_pad, ws.row_seq, ws.row_pos, ws.row_tok);
    OP_TRY(L.end());

    // rows >= `rows` are never produced by the attention kernel: keep its output finite there
    if (range_flagged) OP_HIP(h, hipMemsetAsync(ws.range_flag, 0, sizeof(int), st));
    if (fp_layout && o_f8) {
      const size_t n16 = (size_t)((r_pad - rows) / 16);
      if (n16) OP_HIP(h, hipMemsetAsync(ws.o_hi + (size_t)(rows / 16) * (H / 32) * 512, 0, n16 * (H / 32) * 512 * sizeof(u16), st));
      if (n16) OP_HIP(h, hipMemsetAsync(ws.o_lo + (size_t)(rows / 16) * h->nh * 512, 0, n16 * h->nh * 512 * sizeof(u16), st));
    } else if (fp_layout) {  // fragment-packed o: pieces of 16 rows, hi/lo interleaved, o_hi + o_lo are one buffer
      const size_t tail_off = (size_t)(rows / 16) * (H / 32) * 2 * 512;
      const size_t tail_bytes = (size_t)((r_pad - rows) / 16) * (H / 32) * 2 * 512 * sizeof(u16);
      if (tail_bytes) OP_HIP(h, hipMemsetAsync(ws.o_hi + tail_off, 0, tail_bytes, st));  // (a zero-byte node fails stream capture)
    } else {
      const size_t tail_off = (size_t)rows * H;
      const size_t tail_bytes = (size_t)(r_pad - rows) * H * sizeof(u16);
      if (tail_bytes) OP_HIP(h, hipMemsetAsync(ws.o_hi + tail_off, 0, tail_bytes, st));
      if (tail_bytes && split) OP_HIP(h, hipMemsetAsync(ws.o_lo + tail_off, 0, tail_bytes, st));
    }

    if (!embed_in_qkv0) {
      OP_TRY(L.begin(PK_EMBED_LN));
      if (split)
        hipLaunchKernelGGL((embed_ln_kernel<true>), dim3(row_blocks), dim3(256), 0, st, ids_dev, ws.row_tok, h->emb,
                           h->emb_norm, h->cfg.norm_eps, H, r_pad, h->V, ws.x, ws.ln_hi, ws.ln_lo);
      else
        hipLaunchKernelGGL((embed_ln_kernel<false>), dim3(row_blocks), dim3(256), 0, st, ids_dev, ws.row_tok, h->emb,
                           h->emb_norm, h->cfg.norm_eps, H, r_pad, h->V, ws.x, ws.ln_hi, ws.ln_lo);
      OP_TRY(L.end());
    }
    return OP_OK;
  }

  int capture(int index) {
    if (!h->capture) return OP_OK;
    OP_TRY(L.begin(PK_CAPTURE));
    hipLaunchKernelGGL(capture_rows_kernel, dim3(row_blocks), dim3(256), 0, st, ws.x, ws.row_tok, H, r_pad,
                       h->capture + (size_t)index * total_tokens * H);
    return L.end();
  }

  int attention(bool is_global) {
    OP_TRY(L.begin(is_global ? PK_ATTN_GLOBAL : PK_ATTN_LOCAL));
    const int window = is_global ? -1 : h->cfg.local_attention / 2;
    if (fp_layout) {
      // one block per (sequence, query block, head) work item: 256- or 128-query blocks on 64-key tiles for the
      // full-attention layers, 128-query blocks on 32-key tiles for the sliding-window layers
      AttnFpParams ap;  // q_hi/q_lo (k, vt, o likewise) are adjacent: together they hold the fragment-packed tensor
      ap.q_fp = ws.q_hi;
      ap.k_fp = ws.k_hi;
      ap.vt_fp = ws.vt_hi;
      ap.o_fp = ws.o_hi;
      ap.o_lo8 = ws.o_lo;
      ap.cu = cu_dev;
      ap.s0 = s0;
      ap.roff = ws.roff;
      ap.qboff = is_global ? ws.qboff : ws.qboff_l;
      ap.ns = ns;
      ap.H = H;
      ap.r_pad = r_pad;
      ap.window = window;
      ap.n_items = is_global ? plan.items_g : plan.items_l;
      ap.n_heads = h->nh;
      // XCD-aware block map (see attn_fp_kernel): panel path; round 3: also the row path under the fp16 + e4m3 kernel sets
      // (same-box A/B: sliding-window attention -5 %, whole forward +0.7 %; with the (hi, lo) bf16 whole-layer kernel
      // it measured -1 % in round 2 and stays off there unless OP_FLAG_ATTN_XCD_GROUP asks for it)
      ap.xcd_group = (h->panel_path || o_f8 || (h->cfg.flags & OP_FLAG_ATTN_XCD_GROUP)) ? 1 : 0;
      ap.range_flag = range_flagged ? ws.range_flag : nullptr;
      const unsigned item_span = 8u * opk::ATT_ITEM_GROUP;
      const dim3 grid((ap.xcd_group ? ((unsigned)ap.n_items + item_span - 1) / item_span * item_span : (unsigned)ap.n_items) *
                      (unsigned)h->nh);
      if (!opl::launch_attn(st, ap, is_global ? plan.waves_g : 4, is_global ? 2 : 1, h->pi, zero_p_lo && !attn16, grid, attn16))
        return fail(h, OP_ERR_UNSUPPORTED, "internal: no attention kernel for this configuration");
    } else {
      const dim3 grid((unsigned)q_tiles, (unsigned)h->nh, (unsigned)ns);
      AttnParams ap;
      ap.q_hi = ws.q_hi;
      ap.q_lo = ws.q_lo;
      ap.k_hi = ws.k_hi;
      ap.k_lo = ws.k_lo;
      ap.vt_hi = ws.vt_hi;
      ap.vt_lo = ws.vt_lo;
      ap.o_hi = ws.o_hi;
      ap.o_lo = ws.o_lo;
      ap.cu = cu_dev;
      ap.s0 = s0;
      ap.roff = ws.roff;
      ap.H = H;
      ap.r_pad = r_pad;
      ap.window = window;
      ap.zero_p_lo = (E.pv & 1) ? 0 : 1;
      if (split)
        hipLaunchKernelGGL((attn_kernel<true>), grid, dim3(256), 0, st, ap);
      else
        hipLaunchKernelGGL((attn_kernel<false>), grid, dim3(256), 0, st, ap);
    }
    OP_TRY(L.end());
    if (fp_layout && clr_o) OP_TRY(clear_lo(L, ws.o_hi, 512, (size_t)(r_pad / 16) * (H / 32)));
    if (!fp_layout && split && !(E.attn_out & 1)) OP_HIP(h, hipMemsetAsync(ws.o_lo, 0, plane_bytes, st));
    return OP_OK;
  }
  // fragment-packed q / k / v^T just written by a q/k/v projection: clear what the policy does not carry
  int clear_qkv() {
    if (clr_q) OP_TRY(clear_lo(L, ws.q_hi, 512, (size_t)(r_pad / 16) * (H / 32)));
    if (clr_k) OP_TRY(clear_lo(L, ws.k_hi, 512, (size_t)(r_pad / 16) * (H / 32)));
    if (clr_v) OP_TRY(clear_lo(L, ws.vt_hi, 2048, (size_t)h->nh * (r_pad / 32)));
    return OP_OK;
  }
  int clear_h() {
    if (clr_h) OP_TRY(clear_lo(L, ws.h_hi, 512, (size_t)(r_pad / 16) * (I / 32)));
    return OP_OK;
  }

  // parameters of a q/k/v projection of layer `li` (row-stationary kernels)
  RowGemmParams qkv_params(int li) {
    const LayerWeights& lw = h->layers[li];
    const bool is_global = h->cfg.layer_is_global[li] != 0;
    RowGemmParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.eps = h->cfg.norm_eps;
    rp.hidden = H;
    rp.r_pad = r_pad;
    rp.row_pos = ws.row_pos;
    rp.rope_cos = h->rope_cos[is_global ? 1 : 0];
    rp.rope_sin = h->rope_sin[is_global ? 1 : 0];
    rp.max_pos = h->max_pos;
    rp.x_in = ws.x;
    rp.ln_w = lw.attn_norm;
    rp.wp = f16 ? lw.wqkv_h16 : lw.wqkv_pk;
    rp.n_chunks = 3 * H / ROW_CHUNK;
    rp.n_swapped = 2 * H / ROW_CHUNK;
    rp.o0_hi = ws.q_hi;
    rp.o1_hi = ws.k_hi;
    rp.o2_hi = ws.vt_hi;
    rp.ld_out = H;
    rp.zero_a_lo = clr_ln_attn ? 1 : 0;
    return rp;
  }

  // one layer on the row path: (layer 0: q / k / v) -> attention -> the whole-layer kernel, or the two fused kernels of set 0
  int row_layer(int li) {
    const LayerWeights& lw = h->layers[li];
    const bool is_global = h->cfg.layer_is_global[li] != 0;
    // ---- row-stationary path (hidden <= 256): three launches per layer ------------------------------
    if (li == 0) {  // layer 0 has no attn_norm: split x0 directly
      RowGemmParams rp = qkv_params(0);
      if (embed_in_qkv0) {
        rp.emb_table = h->emb;
        rp.emb_ids = ids_dev;
        rp.emb_vocab = h->V;
        rp.row_tok = ws.row_tok;
        rp.ln_w = h->emb_norm;
        rp.x_io = ws.x;
      }
      OP_TRY(L.begin(PK_ROW_QKV));
      if (!opl::launch_row_qkv0(st, rp, H / 32, small_blocks, h->pi, row_grid)) return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
      OP_TRY(L.end());
      OP_TRY(clear_qkv());
    }  // else: q/k/v of this layer were produced by the fused kernel that closed layer li-1
    OP_TRY(attention(is_global));

    if (layer_fused) {
      // x += o Wo^T ; x += GeGLU(LN(x) Wi^T) Wo^T ; q, k, v^T of the NEXT layer -- one kernel, h stays on chip
      const bool with_qkv = li + 1 < h->N;
      if ((h->cfg.flags & OP_FLAG_LAYER_M32) && lw.wo_p32 && opl::has_layer32(h->pi) && I % 64 == 0) {
        // hidden = 256, on request: the same launch on the 32x32x16 MFMA shape (opk_layer32.hip.h) -- 6 % fewer
        // cycles, but that shape draws more power per flop and the chip clocks lower under it
        Layer32Params lp;
        memset(&lp, 0, sizeof(lp));
        const LayerWeights& nx = h->layers[with_qkv ? li + 1 : li];
        lp.o_fp = ws.o_hi;
        lp.x_io = ws.x;
        lp.ln_mlp = lw.mlp_norm;
        lp.ln_next = with_qkv ? nx.attn_norm : nullptr;
        lp.eps = h->cfg.norm_eps;
        lp.wo_p = lw.wo_p32;
        lp.wi_p = lw.wi_p32;
        lp.wo2_p = lw.wo2_p32;
        lp.wqkv_p = nx.wqkv_p32;
        lp.n_pairs = I / 32;
        lp.q_fp = ws.q_hi;
        lp.k_fp = ws.k_hi;
        lp.vt_fp = ws.vt_hi;
        lp.r_pad = r_pad;
        lp.row_pos = ws.row_pos;
        const int gl = h->cfg.layer_is_global[with_qkv ? li + 1 : li] ? 1 : 0;
        lp.rope_cos = h->rope_cos[gl];
        lp.rope_sin = h->rope_sin[gl];
        lp.max_pos = h->max_pos;
        OP_TRY(L.begin(PK_FUSED_LAYER));
        if (!opl::launch_layer32(st, lp, h->pi, with_qkv, (unsigned)(r_pad / ROW_BM))) return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
        OP_TRY(L.end());
        return OP_OK;
      }
      // Single-pass kernel sets ("f16" / "bf16"), hidden = 256: the wave-pair kernel (opk_layer16p.hip.h), at every batch size --
      // below one 128-row block per CU too, where the other launches switch to 64-row blocks (small_blocks): 4 - 8 % faster
      // forwards from 512 to 32 k tokens than the 8 x 16 kernel's 64-row form (profiles/r06_small_request.txt).  The last layer keeps the 8 x 16 kernel (it ends with final_norm + the pruning head).  Between two
      // wave-pair launches the residual stream is TILED (coalesced 1 KiB loads / stores): the first one reads rows, the last
      // one writes rows.  Hidden-state capture reads rows after every layer: it keeps the 8 x 16 kernel.
      if (pair_layers && with_qkv) {
        Layer32Params lp;
        memset(&lp, 0, sizeof(lp));
        const LayerWeights& nx = h->layers[li + 1];
        lp.o_fp = ws.o_hi;
        lp.x_io = ws.x;
        lp.ln_mlp = lw.mlp_norm;
        lp.ln_next = nx.attn_norm;
        lp.eps = h->cfg.norm_eps;
        lp.wo_p = lw.wo_pp[f16 ? 1 : 0];
        lp.wi_p = lw.wi_pp[f16 ? 1 : 0];
        lp.wo2_p = lw.wo2_pp[f16 ? 1 : 0];
        lp.wqkv_p = nx.wqkv_pp[f16 ? 1 : 0];
        lp.n_pairs = I / 32;
        lp.q_fp = ws.q_hi;
        lp.k_fp = ws.k_hi;
        lp.vt_fp = ws.vt_hi;
        lp.r_pad = r_pad;
        lp.row_pos = ws.row_pos;
        const int gl = h->cfg.layer_is_global[li + 1] ? 1 : 0;
        lp.rope_cos = h->rope_cos[gl];
        lp.rope_sin = h->rope_sin[gl];
        lp.max_pos = h->max_pos;
        OP_TRY(L.begin(PK_FUSED_LAYER));
        if (!opl::launch_layer16p(st, lp, f16, true, /*xin_t=*/li > 0, /*xout_t=*/li + 2 < h->N, (unsigned)(r_pad / ROW_BM)))
          return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
        OP_TRY(L.end());
        return OP_OK;
      }
      RowGemmParams rl = qkv_params(with_qkv ? li + 1 : li);
      rl.a1_fp = ws.o_hi;
      rl.w1p = f16 ? lw.wo_h16 : lw.wo_ks;
      rl.k1_steps = H / 32;
      rl.x_io = ws.x;
      rl.ln_w_mlp = lw.mlp_norm;
      rl.wi_pk = f16 ? lw.wi_h16 : lw.wi_pk;
      rl.wo2_ks = f16 ? lw.wo2_h16 : lw.wo2_pk;
      rl.n_pairs = I / 32;
      if (o_f8) {  // the "f16 + fp8" packs of the same weights
        rl.a1_lo8 = ws.o_lo;
        rl.w1p = lw.wo_f16;
        rl.w1p8 = lw.wo_f8;
        rl.wi_pk = lw.wi_f8;
        rl.wo2_ks = lw.wo2_f16;
        rl.wp = h->layers[with_qkv ? li + 1 : li].wqkv_f8;
      }
      if (!with_qkv && head_in_last_layer) {
        // the last layer's rows go straight through final_norm + the pruning head (no write-back of x, no
        // final_ln_prune launch); mean pooling and hidden-state capture need all normalised rows and keep the kernel
        rl.fin_ln = h->final_norm;
        rl.fin_pw = h->prune_w;
        rl.fin_pb = h->prune_b;
        rl.row_tok = ws.row_tok;
        rl.row_seq = ws.row_seq;
        rl.fin_prune = prune_out;
        rl.fin_keep = keep_prob;
        rl.fin_cls = ws.cls;
        rl.fin_pre_norm = h->cfg.prune_pre_final_norm ? 1 : 0;
        head_done = true;
      }
      OP_TRY(L.begin(PK_FUSED_LAYER));
      if (!opl::launch_row_layer_fused(st, rl, H / 32, h->pi, with_qkv, (unsigned)(r_pad / ROW_BM),
                                       (h->cfg.flags & OP_FLAG_LAYER_8X16) != 0 || (opl::kPolicies[h->pi].wi & 1) == 0))
        return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
      OP_TRY(L.end());
      return OP_OK;
    }

    // x += o Wo^T ; h = GeGLU(LN(x) Wi^T)   -- one kernel, the hidden state stays in registers in between
    RowGemmParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.eps = h->cfg.norm_eps;
    rp.hidden = H;
    rp.r_pad = r_pad;
    rp.ln_w = lw.mlp_norm;
    rp.wp = lw.wi_pk;
    rp.n_chunks = 2 * I / ROW_CHUNK;
    rp.o0_hi = ws.h_hi;  // fragment-packed h (h_hi + h_lo are one buffer)
    rp.ld_out = I;
    rp.a1_fp = ws.o_hi;
    rp.w1p = lw.wo_ks;
    rp.k1_steps = H / 32;
    rp.x_io = ws.x;
    rp.zero_a_lo = clr_ln_mlp ? 1 : 0;
    OP_TRY(L.begin(PK_FUSED_ATTN_OUT_WI));
    if (!opl::launch_row_geglu_fused(st, rp, H / 32, small_blocks, h->pi, row_grid)) return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
    OP_TRY(L.end());
    OP_TRY(clear_h());

    if (li + 1 < h->N) {
      // x += h Wo^T ; q, k, v^T of the NEXT layer = RoPE / transpose of LN(x) Wqkv^T
      RowGemmParams rq = qkv_params(li + 1);
      rq.a1_fp = ws.h_hi;
      rq.w1p = lw.wo2_pk;
      rq.k1_steps = I / 32;
      rq.x_io = ws.x;
      OP_TRY(L.begin(PK_FUSED_MLP_OUT_QKV));
      if (!opl::launch_row_qkv_fused(st, rq, H / 32, small_blocks, h->pi, row_grid)) return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
      OP_TRY(L.end());
      OP_TRY(clear_qkv());
    } else {
      KStreamParams kp;
      kp.a_fp = ws.h_hi;
      kp.wp = lw.wo2_pk;
      kp.n_ksteps = I / 32;
      kp.x = ws.x;
      OP_TRY(L.begin(PK_KSTREAM_MLP_OUT));
      if (!opl::launch_kstream(st, kp, H / 16, h->pi, (unsigned)(r_pad / ROW_BM))) return fail(h, OP_ERR_UNSUPPORTED, no_kernel, H);
      OP_TRY(L.end());
    }
    return OP_OK;
  }

  // one layer on the panel path: LayerNorm -> q / k / v -> attention -> output projection -> LayerNorm -> Wi + GeGLU -> MLP output
  int panel_layer(int li) {
    const LayerWeights& lw = h->layers[li];
    const bool is_global = h->cfg.layer_is_global[li] != 0;
    // ---- panel path (hidden % 256 == 0): LayerNorm -> fragment-packed planes, k-streamed panel GEMMs ----
    const dim3 ln_grid((unsigned)(r_pad / 16));
    const bool pf8 = o_f8;  // kernel sets 3 / 4: activations as fp16 pieces + e4m3 pieces (x 2^12) of their lo part
    // kernel sets 8 / 9: the attention side on the "f16" kernels (pi = PI_F16), the MLP -- LayerNorm(mlp_norm), Wi + GeGLU
    // with h as fp16 + e4m3 pieces, MLP output projection -- on the fp16 + e4m3 kernels of sets 4 / 3
    // -- layer by layer: op_calibrate keeps the fp16 + e4m3 MLP only in the layers the tolerance needs it in (mlp_layers)
    const bool mlp8 = f16 && h->mlp_f8 && (li >= 64 || ((h->mlp_layers >> li) & 1ull) != 0);
    const bool wlo8 = h->pi == opl::PI_F16_F8_W || (h->wi_f8 && h->pi == opl::PI_ALL_TERMS) || (mlp8 && h->mlp_wlo);
    // OP_FLAG_PANEL_F8_WI: the format in the Wi GEMM alone -- its LayerNorm writes fp16 + e4m3 pieces, its epilogue
    // writes h as the (hi, lo) bf16 pieces the MLP output projection's kernel reads
    auto layer_norm_fp = [&](const float* w, bool with_lo, bool clear, bool f8_here = false) -> int {
      OP_TRY(L.begin(PK_LN));
      if (pf8 || f8_here) {
        hipLaunchKernelGGL(ln_fp8_kernel, ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad, w ? 1 : 0, ws.ln_hi,
                           ws.ln_lo);
        return L.end();
      }
      if (with_lo)
        hipLaunchKernelGGL((ln_fp_kernel<true>), ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                           w ? 1 : 0, ws.ln_hi);
      else if (f16)
        hipLaunchKernelGGL((ln_fp_kernel<false, true>), ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                           w ? 1 : 0, ws.ln_hi);
      else
        hipLaunchKernelGGL((ln_fp_kernel<false>), ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                           w ? 1 : 0, ws.ln_hi);
      OP_TRY(L.end());
      if (clear) OP_TRY(clear_lo(L, ws.ln_hi, 512, (size_t)(r_pad / 16) * (H / 32)));
      return OP_OK;
    };
    auto panel = [&](int kind, int epi, const PanelParams& pp, int n_tiles, bool f8_here = false, bool f8_full = false) -> int {
      OP_TRY(L.begin(kind));
      PanelParams q = pp;
      q.n_tiles = n_tiles;
      q.row_group = 8;  // sweep 1 / 2 / 4 / 8 / 16 on base and en-gte: flat within 2 %, 8 best on the Wi GEMM (-5 %)
      const unsigned per_xcd = ((unsigned)(r_pad / ROW_BM) + 7) / 8;  // row blocks each XCD owns
      const unsigned groups = (per_xcd + q.row_group - 1) / q.row_group;
      const dim3 grid(8u * groups * (unsigned)q.row_group * (unsigned)n_tiles);  // XCD-aware block map: see panel_gemm_kernel
      const bool ok = f8_here ? opl::launch_panel_f8(st, q, 103, wlo8, grid)
                      : f8_full ? opl::launch_panel_f8(st, q, epi, wlo8, grid)
                      : pf8   ? (epi == 102 ? opl::launch_panel_f8_qkv(st, q, wlo8, attn16, grid) : opl::launch_panel_f8(st, q, epi, wlo8, grid))
                              : (epi == 102 ? opl::launch_panel_qkv(st, q, h->pi, grid) : opl::launch_panel(st, q, epi, h->pi, grid));
      if (!ok) return fail(h, OP_ERR_UNSUPPORTED, "internal: no panel kernel");
      return L.end();
    };
    // layer 0: attn_norm is Identity -> plain split
    OP_TRY(layer_norm_fp(li != 0 ? lw.attn_norm : nullptr, (V.wqkv & 1) != 0, clr_ln_attn));
    PanelParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.r_pad = r_pad;
    pp.hidden = H;
    pp.range_flag = range_flagged ? ws.range_flag : nullptr;
    pp.row_pos = ws.row_pos;
    pp.rope_cos = h->rope_cos[is_global ? 1 : 0];
    pp.rope_sin = h->rope_sin[is_global ? 1 : 0];
    pp.max_pos = h->max_pos;
    pp.a_fp = ws.ln_hi;
    pp.a_lo8 = ws.ln_lo;
    pp.n_ksteps = H / 32;
    pp.wp = pf8 ? lw.wqkv_p16 : (f16 ? lw.wqkv_h16 : lw.wqkv_pk);
    pp.wp8 = lw.wqkv_p8;
    pp.w8_lo_off = (size_t)3 * H * H / 2;  // u16 elements: the tensor's e4m3(w) slabs, then those of lo(w)
    pp.o0 = ws.q_hi;
    pp.o1 = ws.k_hi;
    if (pf8 || !(h->cfg.flags & OP_FLAG_NO_LAYER_FUSION)) {  // q, k, v^T in one launch
      pp.o2 = ws.vt_hi;
      pp.n_qk_tiles = 2 * H / 256;
      OP_TRY(panel(PK_GEMM_QKV_ROPE, 102, pp, 3 * H / 256));
    } else {
      OP_TRY(panel(PK_GEMM_QK_ROPE, PE_QK, pp, 2 * H / 256));
      pp.wp = (f16 ? lw.wqkv_h16 : lw.wqkv_pk) + (size_t)(2 * H / 256) * (H / 32) * 2 * 8192;
      pp.o0 = ws.vt_hi;
      OP_TRY(panel(PK_GEMM_V_T, PE_V, pp, H / 256));
    }
    if (!attn16) {  // (sets 10 / 11 neither write nor read a lo plane of q / k / v^T)
      OP_TRY(clear_qkv());
    }
    OP_TRY(attention(is_global));
    pp.a_fp = ws.o_hi;
    pp.a_lo8 = ws.o_lo;
    pp.wp = pf8 ? lw.wo_p16 : (f16 ? lw.wo_h16 : lw.wo_ks);
    pp.wp8 = lw.wo_p8;
    pp.w8_lo_off = (size_t)H * H / 2;
    pp.x = ws.x;
    pp.ld_out = H;
    OP_TRY(panel(PK_GEMM_ATTN_OUT, 100, pp, H / 256));
    const bool wi8 = h->wi_f8 && !pf8;
    OP_TRY(layer_norm_fp(lw.mlp_norm, (V.wi & 1) != 0, clr_ln_mlp && !wi8, wi8 || mlp8));
    pp.a_fp = ws.ln_hi;
    pp.a_lo8 = ws.ln_lo;
    pp.wp = (pf8 || wi8 || mlp8) ? lw.wi_p16 : (f16 ? lw.wi_h16 : lw.wi_pk);
    pp.wp8 = lw.wi_p8;
    pp.w8_lo_off = (size_t)2 * I * H / 2;
    pp.o0 = ws.h_hi;
    pp.o0_lo8 = ws.h_lo;
    pp.ld_out = I;
    OP_TRY(panel(PK_GEMM_WI_GEGLU, PE_GEGLU, pp, I / 128, wi8, mlp8));
    OP_TRY(clear_h());
    pp.a_fp = ws.h_hi;
    pp.a_lo8 = ws.h_lo;
    pp.n_ksteps = I / 32;
    pp.wp = (pf8 || mlp8) ? lw.wo2_p16 : (f16 ? lw.wo2_h16 : lw.wo2_pk);
    pp.wp8 = lw.wo2_p8;
    pp.w8_lo_off = (size_t)H * I / 2;
    pp.ld_out = H;
    OP_TRY(panel(PK_GEMM_MLP_OUT, 101, pp, H / 256, false, mlp8));
    return OP_OK;
  }

  // one layer on the tiled path (generic fallback)
  int tiled_layer(int li) {
    const LayerWeights& lw = h->layers[li];
    const bool is_global = h->cfg.layer_is_global[li] != 0;
    // ---- tiled path (any hidden % 128 == 0): separate LayerNorm kernels, 128 x 128 x 32 tiles, row-major planes.
    // Two kernel sets (single pass / all terms); a narrower policy clears the lo planes it does not carry. ----
    auto layer_norm = [&](const float* w, int term_mask) -> int {
      if (w) {
        OP_TRY(L.begin(PK_LN));
        if (split)
          hipLaunchKernelGGL((ln_kernel<true>), dim3(row_blocks), dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                             ws.ln_hi, ws.ln_lo);
        else
          hipLaunchKernelGGL((ln_kernel<false>), dim3(row_blocks), dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                             ws.ln_hi, ws.ln_lo);
        OP_TRY(L.end());
      }
      if (split && !(term_mask & 1)) OP_HIP(h, hipMemsetAsync(ws.ln_lo, 0, plane_bytes, st));
      return OP_OK;
    };
    OP_TRY(layer_norm(li != 0 ? lw.attn_norm : nullptr, E.wqkv));  // layer 0: embed_ln wrote the planes
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.a_hi = ws.ln_hi;
    p.a_lo = ws.ln_lo;
    p.K = H;
    p.m_tiles = m_tiles;
    p.hidden = H;
    p.row_pos = ws.row_pos;
    p.rope_cos = h->rope_cos[is_global ? 1 : 0];
    p.rope_sin = h->rope_sin[is_global ? 1 : 0];
    p.max_pos = h->max_pos;
    // q, k = RoPE(x Wq^T), RoPE(x Wk^T)
    p.w_hi = lw.wqkv_hi;
    p.w_lo = lw.wqkv_lo;
    p.n_tiles = 2 * H / GEMM_BN;
    p.o0_hi = ws.q_hi;
    p.o0_lo = ws.q_lo;
    p.o1_hi = ws.k_hi;
    p.o1_lo = ws.k_lo;
    p.ld_out = H;
    OP_TRY(launch_gemm<EPI_QK_ROPE>(L, PK_GEMM_QK_ROPE, p, split));
    // v^T
    p.w_hi = lw.wqkv_hi + (size_t)2 * H * H;
    p.w_lo = lw.wqkv_lo + (size_t)2 * H * H;
    p.n_tiles = H / GEMM_BN;
    p.o0_hi = ws.vt_hi;
    p.o0_lo = ws.vt_lo;
    p.ld_out = r_pad;
    OP_TRY(launch_gemm<EPI_V_T>(L, PK_GEMM_V_T, p, split));
    if (split && !(E.qk & 1)) OP_HIP(h, hipMemsetAsync(ws.q_lo, 0, plane_bytes, st));
    if (split && !(E.qk & 2)) OP_HIP(h, hipMemsetAsync(ws.k_lo, 0, plane_bytes, st));
    if (split && !(E.pv & 2)) OP_HIP(h, hipMemsetAsync(ws.vt_lo, 0, plane_bytes, st));

    OP_TRY(attention(is_global));

    // x += attn Wo^T
    p.a_hi = ws.o_hi;
    p.a_lo = ws.o_lo;
    p.w_hi = lw.wo_hi;
    p.w_lo = lw.wo_lo;
    p.K = H;
    p.n_tiles = H / GEMM_BN;
    p.x = ws.x;
    p.ld_out = H;
    OP_TRY(launch_gemm<EPI_RESIDUAL>(L, PK_GEMM_ATTN_OUT, p, split));
    // x += (gelu(a) * g) Wo^T,  (a, g) = LN(x) Wi^T
    OP_TRY(layer_norm(lw.mlp_norm, E.wi));
    p.a_hi = ws.ln_hi;
    p.a_lo = ws.ln_lo;
    p.w_hi = lw.wi_hi;
    p.w_lo = lw.wi_lo;
    p.K = H;
    p.n_tiles = 2 * I / GEMM_BN;
    p.o0_hi = ws.h_hi;
    p.o0_lo = ws.h_lo;
    p.ld_out = I;
    OP_TRY(launch_gemm<EPI_GEGLU>(L, PK_GEMM_WI_GEGLU, p, split));
    if (split && !(E.mlp_out & 1)) OP_HIP(h, hipMemsetAsync(ws.h_lo, 0, (size_t)r_pad * I * sizeof(u16), st));
    p.a_hi = ws.h_hi;
    p.a_lo = ws.h_lo;
    p.w_hi = lw.wo2_hi;
    p.w_lo = lw.wo2_lo;
    p.K = I;
    p.n_tiles = H / GEMM_BN;
    p.x = ws.x;
    p.ld_out = H;
    OP_TRY(launch_gemm<EPI_RESIDUAL>(L, PK_GEMM_MLP_OUT, p, split));
    return OP_OK;
  }

  // final_norm + pruning head (unless the last whole-layer launch did it), ranking head
  int heads() {
    const int mean_pool = h->cfg.pooling == OP_POOL_MEAN ? 1 : 0;
    if (!head_done) {
      OP_TRY(L.begin(PK_FINAL_LN_PRUNE));
      hipLaunchKernelGGL(final_ln_prune_kernel, dim3(row_blocks), dim3(256), 0, st, ws.x, h->final_norm, h->cfg.norm_eps, H,
                         r_pad, ws.row_tok, ws.row_seq, ws.row_pos, h->prune_w, h->prune_b, prune_out, keep_prob,
                         h->cfg.prune_pre_final_norm ? 1 : 0, mean_pool, ws.cls,
                         h->capture ? h->capture + (size_t)h->N * total_tokens * H : nullptr,
                         range_flagged ? ws.range_flag : nullptr);
      OP_TRY(L.end());
    }
    OP_TRY(L.begin(PK_RANK_HEAD));
    hipLaunchKernelGGL(rank_head_kernel, dim3((unsigned)ns), dim3(256), 0, st, ws.cls, ws.x, cu_dev, s0, ws.roff, mean_pool,
                       H, h->nl, h->dense_t, h->head_norm, h->cfg.norm_eps, h->cls_w, h->cls_b, rank_out,
                       range_flagged ? ws.range_flag : nullptr);
    OP_TRY(L.end());
    return OP_OK;
  }

  int run() {
    OP_TRY(prologue());
    for (int li = 0; li < h->N; ++li) {
      OP_TRY(capture(li));
      if (h->row_path) OP_TRY(row_layer(li));
      else if (h->panel_path) OP_TRY(panel_layer(li));
      else OP_TRY(tiled_layer(li));
    }
    return heads();
  }
};

int forward_chunk(op_handle* h, Launcher& L, const Workspace& ws, const int32_t* ids_dev, const int32_t* cu_dev, int s0,
                  int ns, int rows, int max_len, int total_tokens, const AttnPlan& plan, float* prune_out, float* rank_out,
                  float* keep_prob) {
  return ChunkPass(h, L, ws, ids_dev, cu_dev, s0, ns, rows, max_len, total_tokens, plan, prune_out, rank_out, keep_prob).run();
}

}  // namespace

extern "C" {

int op_abi_version(void) { return OP_ABI_VERSION; }

const char* op_last_error(const op_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

const char* op_profile_kind_name(int kind) { return (kind >= 0 && kind < PK_COUNT) ? kProfileNames[kind] : "?"; }

int op_device_count(int* count) {
  if (!count) return fail(nullptr, OP_ERR_INVALID, "op_device_count: count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, OP_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return OP_OK;
}

int op_create(const op_config* cfg, op_handle** out) {
  if (!cfg || !out) return fail(nullptr, OP_ERR_INVALID, "op_create: NULL argument");
  *out = nullptr;
  if (cfg->struct_bytes != sizeof(op_config))
    return fail(nullptr, OP_ERR_INVALID, "op_create: op_config.struct_bytes=%u, library expects %zu (ABI mismatch)",
                cfg->struct_bytes, sizeof(op_config));
  const int H = cfg->hidden_size, I = cfg->intermediate_size, N = cfg->num_layers, nh = cfg->num_heads;
  if (H <= 0 || I <= 0 || N <= 0 || nh <= 0 || cfg->vocab_size <= 0 || cfg->num_labels <= 0)
    return fail(nullptr, OP_ERR_INVALID, "op_create: non-positive dimension");
  if (N > OP_MAX_LAYERS) return fail(nullptr, OP_ERR_UNSUPPORTED, "num_layers %d > %d", N, OP_MAX_LAYERS);
  if (H % nh != 0 || H / nh != HEAD_DIM)
    return fail(nullptr, OP_ERR_UNSUPPORTED, "head_dim must be 64 (hidden_size %d / num_heads %d)", H, nh);
  if (H % GEMM_BN != 0 || H > 1024)
    return fail(nullptr, OP_ERR_UNSUPPORTED, "hidden_size %d must be a multiple of 128 and <= 1024", H);
  if (I % 64 != 0) return fail(nullptr, OP_ERR_UNSUPPORTED, "intermediate_size %d must be a multiple of 64", I);
  if (cfg->num_labels > 64) return fail(nullptr, OP_ERR_UNSUPPORTED, "num_labels %d > 64", cfg->num_labels);
  Policy req;
  switch (cfg->precision) {
    case OP_PRECISION_BF16X3: req = opl::kPolicies[0]; break;
    case OP_PRECISION_BF16X2: req = opl::kPolicies[1]; break;
    case OP_PRECISION_BF16: req = opl::kPolicies[2]; break;
    case OP_PRECISION_CUSTOM:
      for (int f = 0; f < OP_FAM_COUNT; ++f)
        if (cfg->terms[f] > 3) return fail(nullptr, OP_ERR_INVALID, "terms[%d] = %d is not a term mask (0..3)", f, cfg->terms[f]);
      req = Policy{cfg->terms[OP_FAM_WQKV], cfg->terms[OP_FAM_QK], cfg->terms[OP_FAM_PV], cfg->terms[OP_FAM_ATTN_OUT],
                   cfg->terms[OP_FAM_WI], cfg->terms[OP_FAM_MLP_OUT]};
      break;
    default: return fail(nullptr, OP_ERR_INVALID, "unknown precision %d", cfg->precision);
  }
  if (cfg->pooling != OP_POOL_CLS && cfg->pooling != OP_POOL_MEAN)
    return fail(nullptr, OP_ERR_INVALID, "unknown pooling %d", cfg->pooling);
  if (cfg->local_attention < 0 || cfg->max_position_embeddings <= 0)
    return fail(nullptr, OP_ERR_INVALID, "bad local_attention / max_position_embeddings");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, OP_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (cfg->device_id < 0 || cfg->device_id >= ndev)
    return fail(nullptr, OP_ERR_INVALID, "device_id %d out of range (count %d)", cfg->device_id, ndev);

  op_handle* h = new (std::nothrow) op_handle();
  if (!h) return fail(nullptr, OP_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  h->H = H;
  h->I = I;
  h->N = N;
  h->nh = nh;
  h->V = cfg->vocab_size;
  h->nl = cfg->num_labels;
  h->max_pos = cfg->max_position_embeddings;
  h->req = h->eff = req;
  h->chunk_rows = cfg->chunk_rows > 0 ? align_up(cfg->chunk_rows, ROW_ALIGN) : 262144;  // grids must cover the chip several times over
  h->layers.resize(N);
  // H % 64 and I % 32: every row-GEMM streams an EVEN number of 32-feature chunks on each side of the q/k -> v
  // boundary (H/16 q/k chunks, H/32 v chunks, H/32 out-projection chunks, I/16 Wi chunks) -- rowgemm_kernel's
  // two-stage loop is unrolled by two.
  const bool force_tiled = (cfg->flags & OP_FLAG_FORCE_TILED) != 0;
  h->row_path = (H <= 256) && (H % 64 == 0) && (I % 32 == 0) && !force_tiled;
  // panels of 256 output features (4 heads; 128 GeGLU input + 128 gate columns), an even number of k-steps
  h->panel_path = !h->row_path && (H % 256 == 0) && (I % 128 == 0) && !force_tiled;

#define OP_CREATE_TRY(expr)  \
  do {                       \
    int _rc = (expr);        \
    if (_rc != OP_OK) {      \
      g_last_error = h->err; \
      op_destroy(h);         \
      return _rc;            \
    }                        \
  } while (0)
#define OP_CREATE_HIP(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      int _rc = fail(h, OP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));          \
      op_destroy(h);                                                                         \
      return _rc;                                                                            \
    }                                                                                        \
  } while (0)

  OP_CREATE_HIP(hipSetDevice(cfg->device_id));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess && prop.multiProcessorCount > 0)
      h->n_cus = prop.multiProcessorCount;
  }
  const size_t HH = (size_t)H * H;
  OP_CREATE_TRY(dev_alloc(h, &h->any_lo_dev, OP_FAM_COUNT + 3));  // + the fp16-fit flags of the "f16 + fp8" packs (note_f16_fit)
  OP_CREATE_HIP(hipMemset(h->any_lo_dev, 0, (OP_FAM_COUNT + 3) * sizeof(int)));
  OP_CREATE_TRY(dev_alloc(h, &h->f16_fit_dev, 2));
  OP_CREATE_HIP(hipMemset(h->f16_fit_dev, 0, 2 * sizeof(float)));
  // panel path: opt-in (OP_FLAG_PANEL_F8) -- at the depth of the published models (19-25 layers) the format's error
  // reaches 0.45-1.0e-3 on logits, the (hi, lo) bf16 sets stay at 0.2-0.5e-3 (scripts/f8_depth_check.py)
  // (panel path: the packs are always built -- the Wi GEMM takes the format by default for fp32-valued weights, see
  // resolve_policy; OP_FLAG_PANEL_F8 extends it to every GEMM of the layer)
  h->f8_packs = ((h->row_path && (H / 32) % 4 == 0) || h->panel_path) && !(cfg->flags & OP_FLAG_NO_F8);
  // kernel set "f16": whole-layer kernel shapes (hidden 128 / 256) or the panel path; never with the A/B flags that pick
  // another layer structure
  h->h16_packs = ((h->row_path && (H == 128 || H == 256)) || h->panel_path) &&
                 !(cfg->flags & (OP_FLAG_NO_F8 | OP_FLAG_NO_LAYER_FUSION | OP_FLAG_NO_POLICY_KERNELS));
  OP_CREATE_TRY(dev_alloc(h, &h->emb, (size_t)h->V * H));
  OP_CREATE_TRY(dev_alloc(h, &h->emb_norm, H));
  OP_CREATE_TRY(dev_alloc(h, &h->final_norm, H));
  OP_CREATE_TRY(dev_alloc(h, &h->dense_t, HH));
  OP_CREATE_TRY(dev_alloc(h, &h->head_norm, H));
  OP_CREATE_TRY(dev_alloc(h, &h->cls_w, (size_t)h->nl * H));
  OP_CREATE_TRY(dev_alloc(h, &h->cls_b, h->nl));
  OP_CREATE_TRY(dev_alloc(h, &h->prune_w, (size_t)2 * H));
  OP_CREATE_TRY(dev_alloc(h, &h->prune_b, 2));
  h->missing = {"model.embeddings.tok_embeddings.weight", "model.embeddings.norm.weight", "model.final_norm.weight",
                "head.dense.weight", "head.norm.weight", "classifier.weight", "classifier.bias",
                "pruning_head.classifier.weight", "pruning_head.classifier.bias"};
  for (int i = 0; i < N; ++i) {
    LayerWeights& lw = h->layers[i];
    const std::string pre = "model.layers." + std::to_string(i) + ".";
    if (i != 0) {
      OP_CREATE_TRY(dev_alloc(h, &lw.attn_norm, H));
      h->missing.push_back(pre + "attn_norm.weight");
    }
    OP_CREATE_TRY(dev_alloc(h, &lw.mlp_norm, H));
    if (!h->row_path && !h->panel_path) {  // tiled path: row-major hi / lo planes
      OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_hi, 3 * HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_lo, 3 * HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo_hi, HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo_lo, HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wi_hi, (size_t)2 * I * H));
      OP_CREATE_TRY(dev_alloc(h, &lw.wi_lo, (size_t)2 * I * H));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo2_hi, (size_t)H * I));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo2_lo, (size_t)H * I));
    } else {  // fragment-ordered layouts (hi and lo planes interleaved): chunk-major (row path) or panel-major
      OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_pk, 2 * 3 * HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wi_pk, (size_t)2 * 2 * I * H));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo2_pk, (size_t)2 * H * I));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo_ks, 2 * HH));
      if (h->h16_packs) {
        OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_h16, 2 * 3 * HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wi_h16, (size_t)2 * 2 * I * H));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo2_h16, (size_t)2 * H * I));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_h16, 2 * HH));
      }
      if (h->f8_packs && h->panel_path) {  // 4 bytes per weight element: fp16 + e4m3(w) + e4m3(lo(w))
        OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_p16, 3 * HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_p8, 3 * HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_p16, HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_p8, HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wi_p16, (size_t)2 * I * H));
        OP_CREATE_TRY(dev_alloc(h, &lw.wi_p8, (size_t)2 * I * H));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo2_p16, (size_t)H * I));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo2_p8, (size_t)H * I));
      }
      if (h->f8_packs && h->row_path) {  // 4 bytes per weight element: fp16 + e4m3 + e4m3 of the lo part (fp16 for the MLP's Wo)
        OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_f8, 3 * HH * 2));
        OP_CREATE_TRY(dev_alloc(h, &lw.wi_f8, (size_t)2 * I * H * 2));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_f16, HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_f8, HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo2_f16, (size_t)2 * H * I));
      }
      if (h->row_path && H == 256) {
        OP_CREATE_TRY(dev_alloc(h, &lw.wo_p32, HH));
        OP_CREATE_TRY(dev_alloc(h, &lw.wi_p32, (size_t)2 * I * H));
        OP_CREATE_TRY(dev_alloc(h, &lw.wo2_p32, (size_t)H * I));
        OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_p32, 3 * HH));
        for (int f = 0; f < (h->h16_packs ? 2 : 1); ++f) {
          OP_CREATE_TRY(dev_alloc(h, &lw.wo_pp[f], HH));
          OP_CREATE_TRY(dev_alloc(h, &lw.wi_pp[f], (size_t)2 * I * H));
          OP_CREATE_TRY(dev_alloc(h, &lw.wo2_pp[f], (size_t)H * I));
          OP_CREATE_TRY(dev_alloc(h, &lw.wqkv_pp[f], 3 * HH));
        }
      }
    }
    h->missing.push_back(pre + "mlp_norm.weight");
    h->missing.push_back(pre + "attn.Wqkv.weight");
    h->missing.push_back(pre + "attn.Wo.weight");
    h->missing.push_back(pre + "mlp.Wi.weight");
    h->missing.push_back(pre + "mlp.Wo.weight");
  }
  for (int t = 0; t < 2; ++t) {
    std::vector<float> cs, sn;
    build_rope_host(t == 1 ? cfg->global_rope_theta : cfg->local_rope_theta, h->max_pos, cs, sn);
    OP_CREATE_TRY(dev_alloc(h, &h->rope_cos[t], cs.size()));
    OP_CREATE_TRY(dev_alloc(h, &h->rope_sin[t], sn.size()));
    OP_CREATE_HIP(hipMemcpy(h->rope_cos[t], cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
    OP_CREATE_HIP(hipMemcpy(h->rope_sin[t], sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = h;
  return OP_OK;
}

void op_destroy(op_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device_id);
  for (auto& ev : h->events) {
    (void)hipEventDestroy(ev.start);
    (void)hipEventDestroy(ev.stop);
  }
  for (void* p : h->allocations) (void)hipFree(p);
  delete h;
}

int op_load_weight(op_handle* h, const char* name_c, const void* data, int dtype, const int64_t* shape, int ndim) {
  if (!h || !name_c || !data || !shape) return fail(h, OP_ERR_INVALID, "op_load_weight: NULL argument");
  if (dtype != OP_DTYPE_F32 && dtype != OP_DTYPE_BF16 && dtype != OP_DTYPE_F16)
    return fail(h, OP_ERR_INVALID, "op_load_weight(%s): unknown dtype %d", name_c, dtype);
  if (ndim < 1 || ndim > 2) return fail(h, OP_ERR_INVALID, "op_load_weight(%s): ndim %d not in {1,2}", name_c, ndim);
  std::string name(name_c);
  const std::string legacy_prefix = "ranking_model.";
  if (name.compare(0, legacy_prefix.size(), legacy_prefix) == 0) name = name.substr(legacy_prefix.size());
  const int64_t d0 = shape[0], d1 = ndim == 2 ? shape[1] : 1;
  const int H = h->H, I = h->I;

  enum Kind { F32_COPY, F32_TRANSPOSE, PLANES, PLANES_GEGLU };
  u16* dst_pk = nullptr;
  u16* dst_ks = nullptr;  // additional k-streamed packing (attention Wo)
  u16* dst_p32 = nullptr; // additional packing for the 32x32x16 whole-layer kernel
  u16* const* dst_pp = nullptr;  // the wave-pair kernel's packs ([0] bf16, [1] fp16)
  u16 *dst_f8a = nullptr, *dst_f8b = nullptr;  // "f16 + fp8" packs: chunked (a) or k-streamed fp16 (a) + e4m3 (b)
  u16 *dst_p16 = nullptr, *dst_p8 = nullptr;   // ... on the panel path: fp16 slabs + e4m3 slabs
  u16 *dst_pk_h16 = nullptr, *dst_ks_h16 = nullptr;  // kernel set "f16": the layouts of dst_pk / dst_ks with fp16 values
  int p32_mode = 0, p32_kmajor = 0;
  int pk_mode = -1;
  int family = -1;        // op_gemm_family of a GEMM weight
  Kind kind = F32_COPY;
  float* dst_f32 = nullptr;
  u16 *dst_hi = nullptr, *dst_lo = nullptr;
  int64_t want0 = 0, want1 = 1;
  auto expect = [&](int64_t a, int64_t b) {
    want0 = a;
    want1 = b;
  };

  if (name == "model.embeddings.tok_embeddings.weight") {
    dst_f32 = h->emb; expect(h->V, H);
  } else if (name == "model.embeddings.norm.weight") {
    dst_f32 = h->emb_norm; expect(H, 1);
  } else if (name == "model.final_norm.weight") {
    dst_f32 = h->final_norm; expect(H, 1);
  } else if (name == "head.dense.weight") {
    dst_f32 = h->dense_t; kind = F32_TRANSPOSE; expect(H, H);
  } else if (name == "head.norm.weight") {
    dst_f32 = h->head_norm; expect(H, 1);
  } else if (name == "classifier.weight") {
    dst_f32 = h->cls_w; expect(h->nl, H);
  } else if (name == "classifier.bias") {
    dst_f32 = h->cls_b; expect(h->nl, 1);
  } else if (name == "pruning_head.classifier.weight") {
    dst_f32 = h->prune_w; expect(2, H);
  } else if (name == "pruning_head.classifier.bias") {
    dst_f32 = h->prune_b; expect(2, 1);
  } else {
    int li = -1;
    char tail[64] = {0};
    if (sscanf(name.c_str(), "model.layers.%d.%63s", &li, tail) != 2 || li < 0 || li >= h->N)
      return fail(h, OP_ERR_INVALID, "op_load_weight: unknown tensor name '%s'", name_c);
    LayerWeights& lw = h->layers[li];
    const std::string t(tail);
    if (t == "attn_norm.weight" && li != 0) {
      dst_f32 = lw.attn_norm; expect(H, 1);
    } else if (t == "mlp_norm.weight") {
      dst_f32 = lw.mlp_norm; expect(H, 1);
    } else if (t == "attn.Wqkv.weight") {
      kind = PLANES; dst_hi = lw.wqkv_hi; dst_lo = lw.wqkv_lo; expect(3 * H, H);
      dst_pk = lw.wqkv_pk; pk_mode = RE_QKV; family = OP_FAM_WQKV;
      dst_p32 = lw.wqkv_p32; dst_pp = lw.wqkv_pp; p32_mode = L32_QKV; p32_kmajor = 0;
      dst_f8a = lw.wqkv_f8;
      dst_p16 = lw.wqkv_p16; dst_p8 = lw.wqkv_p8;
      dst_pk_h16 = lw.wqkv_h16;
    } else if (t == "attn.Wo.weight") {
      kind = PLANES; dst_hi = lw.wo_hi; dst_lo = lw.wo_lo; expect(H, H);
      dst_pk = nullptr; pk_mode = 101; dst_ks = lw.wo_ks; family = OP_FAM_ATTN_OUT;
      dst_p32 = lw.wo_p32; dst_pp = lw.wo_pp; p32_mode = L32_RESID; p32_kmajor = 1;
      dst_f8a = lw.wo_f16; dst_f8b = lw.wo_f8;
      dst_p16 = lw.wo_p16; dst_p8 = lw.wo_p8;
      dst_ks_h16 = lw.wo_h16;
    } else if (t == "mlp.Wi.weight") {
      kind = PLANES_GEGLU; dst_hi = lw.wi_hi; dst_lo = lw.wi_lo; expect(2 * I, H);
      dst_pk = lw.wi_pk; pk_mode = RE_GEGLU; family = OP_FAM_WI;
      dst_p32 = lw.wi_p32; dst_pp = lw.wi_pp; p32_mode = L32_GEGLU; p32_kmajor = 0;
      dst_f8a = lw.wi_f8;
      dst_p16 = lw.wi_p16; dst_p8 = lw.wi_p8;
      dst_pk_h16 = lw.wi_h16;
    } else if (t == "mlp.Wo.weight") {
      kind = PLANES; dst_hi = lw.wo2_hi; dst_lo = lw.wo2_lo; expect(H, I);
      dst_pk = lw.wo2_pk; pk_mode = 100; family = OP_FAM_MLP_OUT;  // k-streamed
      dst_p32 = lw.wo2_p32; dst_pp = lw.wo2_pp; p32_mode = L32_RESID; p32_kmajor = 1;
      dst_f8a = lw.wo2_f16;
      dst_p16 = lw.wo2_p16; dst_p8 = lw.wo2_p8;
      dst_pk_h16 = lw.wo2_h16;
    } else {
      return fail(h, OP_ERR_INVALID, "op_load_weight: unknown tensor name '%s'", name_c);
    }
  }
  if (d0 != want0 || d1 != want1)
    return fail(h, OP_ERR_INVALID, "op_load_weight(%s): shape [%lld, %lld] but the model needs [%lld, %lld]", name_c,
                (long long)d0, (long long)d1, (long long)want0, (long long)want1);

  OP_HIP(h, hipSetDevice(h->cfg.device_id));
  const size_t count = (size_t)d0 * (size_t)d1;
  const size_t esz = dtype == OP_DTYPE_F32 ? 4 : 2;
  void* raw = nullptr;
  float* f32 = nullptr;
  OP_HIP(h, hipMalloc(&raw, count * esz));
  hipError_t e = hipMemcpy(raw, data, count * esz, hipMemcpyDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&f32, count * sizeof(float));
  if (e != hipSuccess) {
    (void)hipFree(raw);
    return fail(h, OP_ERR_HIP, "op_load_weight(%s): staging failed: %s", name_c, hipGetErrorString(e));
  }
  const unsigned blocks = (unsigned)((count + 255) / 256);
  hipLaunchKernelGGL(convert_to_f32_kernel, dim3(blocks), dim3(256), 0, 0, raw, dtype, count, f32);
  // A requested policy without the hi x lo(weight) term stores zeros in the lo plane (so that any kernel set gives
  // that policy's numerics); `any_lo` records whether the tensor had a non-zero lo element at all.
  int zero_lo = 0;
  int* any_lo = h->any_lo_dev;
  if (family >= 0) {
    const int req_mask[OP_FAM_COUNT] = {h->req.wqkv, h->req.qk, h->req.pv, h->req.attn_out, h->req.wi, h->req.mlp_out};
    zero_lo = (req_mask[family] & OP_TERM_RIGHT_LO) ? 0 : 1;
    any_lo = h->any_lo_dev + family;
    h->resolved = false;
    // a kernel set pinned by op_select_kernel_set or measured by op_calibrate belongs to the weights it was chosen on: new
    // GEMM weights start from the default selection again (and from the compact formats, if op_set_compact_operands left them)
    h->forced_set = -1;
    h->forced_mlp_layers = ~0ull;
    h->f8_off = false;
  }
  switch (kind) {
    case F32_COPY:
      e = hipMemcpyAsync(dst_f32, f32, count * sizeof(float), hipMemcpyDeviceToDevice, 0);
      break;
    case F32_TRANSPOSE:
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, dst_f32);
      break;
    case PLANES:
      if (dst_hi)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 0, dst_hi, dst_lo,
                           zero_lo, any_lo);
      break;
    case PLANES_GEGLU:
      if (dst_hi)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, I, dst_hi, dst_lo,
                           zero_lo, any_lo);
      break;
  }
  if (h->panel_path && (dst_pk || dst_ks)) {
    // panel-major packing: [panel][k-step][plane][16 fragments][512], rows permuted per consumer (panel_source_row)
    const int K = (int)d1;
    auto pack = [&](int n_tiles, int mode, u16* dst) {
      const size_t total = (size_t)n_tiles * 256 * K;
      hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, f32, n_tiles, K, mode,
                         H, I, dst, zero_lo, any_lo);
    };
    if (dst_p16) {  // the fp16 + e4m3 packs of the same panels (kernel sets 3 / 4)
      int* not_f16 = h->any_lo_dev + OP_FAM_COUNT;
      const size_t lo_off = count;  // bytes of the whole tensor's e4m3(w) slabs: the lo(w) slabs follow
      auto pack8 = [&](int n_tiles, int mode, int tile0) {
        const size_t total = (size_t)n_tiles * 256 * K;
        hipLaunchKernelGGL(pack_panel_f8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, f32, n_tiles, K, mode, H, I,
                           dst_p16 + (size_t)tile0 * 256 * K, dst_p8 + (size_t)tile0 * 256 * K / 2, lo_off, zero_lo, not_f16, h->f16_fit_dev);
      };
      if (pk_mode == RE_QKV) {
        pack8(2 * H / 256, PE_QK, 0);
        pack8(H / 256, PE_V, 2 * H / 256);
      } else if (pk_mode == RE_GEGLU) {
        pack8(I / 128, PE_GEGLU, 0);
      } else {
        pack8(H / 256, PE_RESIDUAL, 0);
      }
      hipLaunchKernelGGL(weight_energy_kernel, dim3(256), dim3(256), 0, 0, f32, count, h->f16_fit_dev);
      hipLaunchKernelGGL(f16_fit_close_tensor_kernel, dim3(1), dim3(1), 0, 0, not_f16, h->f16_fit_dev);
    }
    if (pk_mode == RE_QKV) {
      pack(2 * H / 256, PE_QK, dst_pk);
      pack(H / 256, PE_V, dst_pk + (size_t)(2 * H / 256) * (K / 32) * 2 * 8192);
    } else if (pk_mode == 101) {
      pack(H / 256, PE_RESIDUAL, dst_ks);  // attention output projection
    } else if (pk_mode == RE_GEGLU) {
      pack(I / 128, PE_GEGLU, dst_pk);
    } else {
      pack(H / 256, PE_RESIDUAL, dst_pk);  // MLP output projection
    }
    if (dst_pk_h16 || dst_ks_h16) {  // kernel set "f16": the same panels with fp16 values
      auto pack16 = [&](int n_tiles, int mode, u16* dst) {
        const size_t total = (size_t)n_tiles * 256 * K;
        hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, f32, n_tiles, K, mode,
                           H, I, dst, 1, any_lo, 1);
      };
      if (pk_mode == RE_QKV) {
        pack16(2 * H / 256, PE_QK, dst_pk_h16);
        pack16(H / 256, PE_V, dst_pk_h16 + (size_t)(2 * H / 256) * (K / 32) * 2 * 8192);
      } else if (pk_mode == 101) {
        pack16(H / 256, PE_RESIDUAL, dst_ks_h16);
      } else if (pk_mode == RE_GEGLU) {
        pack16(I / 128, PE_GEGLU, dst_pk_h16);
      } else {
        pack16(H / 256, PE_RESIDUAL, dst_pk_h16);
      }
    }
  }
  if ((dst_pk || dst_ks) && h->row_path) {
    if (dst_pk && pk_mode == 100)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_pk, zero_lo, any_lo);
    else if (dst_pk)
      hipLaunchKernelGGL(pack_rowgemm_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, pk_mode, H, I, dst_pk,
                         zero_lo, any_lo);
    if (dst_ks)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_ks, zero_lo, any_lo);
    // kernel set "f16": the same layouts with fp16 values in the hi plane
    if (dst_pk_h16 && pk_mode == 100)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_pk_h16, 1, any_lo, 1);
    else if (dst_pk_h16)
      hipLaunchKernelGGL(pack_rowgemm_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, pk_mode, H, I, dst_pk_h16, 1, any_lo, 1);
    if (dst_ks_h16)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_ks_h16, 1, any_lo, 1);
    if (dst_f8a) {  // "f16 + fp8" kernel set; raises the not-fp16 flag (any_lo_dev[OP_FAM_COUNT]) for a weight it cannot hold exactly
      int* not_f16 = h->any_lo_dev + OP_FAM_COUNT;
      if (pk_mode == 100 || pk_mode == 101)
        hipLaunchKernelGGL(pack_kstream_f8_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_f8a, dst_f8b, zero_lo, not_f16, h->f16_fit_dev);
      else
        hipLaunchKernelGGL(pack_rowgemm_f8_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, pk_mode, H, I, dst_f8a, zero_lo, not_f16, h->f16_fit_dev);
      hipLaunchKernelGGL(weight_energy_kernel, dim3(256), dim3(256), 0, 0, f32, count, h->f16_fit_dev);
      hipLaunchKernelGGL(f16_fit_close_tensor_kernel, dim3(1), dim3(1), 0, 0, not_f16, h->f16_fit_dev);
    }
    if (dst_p32)  // the 32x32x16 whole-layer kernel's order (hi plane; that kernel runs only when the lo planes are zero)
      hipLaunchKernelGGL(pack_layer32_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, p32_mode, p32_kmajor, H, I,
                         dst_p32);
    for (int f = 0; dst_pp && f < 2; ++f)  // (the modes of Layer32Pack and Layer16pPack are the same numbers)
      if (dst_pp[f])
        hipLaunchKernelGGL(pack_layer16p_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, p32_mode, p32_kmajor, H, I, dst_pp[f], f);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(0);
  (void)hipFree(raw);
  (void)hipFree(f32);
  if (e != hipSuccess) return fail(h, OP_ERR_HIP, "op_load_weight(%s): %s", name_c, hipGetErrorString(e));
  erase_missing(h, name);
  return OP_OK;
}

}  // extern "C"

namespace {
// op_kernel_set number of the handle's current selection (op_effective_policy)
int public_set(const op_handle* h) {
  if (h->emulate) return -1;
  if (h->pi == opl::PI_F16 && h->mlp_f8) return h->mlp_wlo ? OP_KS_F16_MLP_F8_W : OP_KS_F16_MLP_F8;
  if (h->pi == opl::PI_F16) return OP_KS_F16;
  if (h->attn_f16 && h->pi == opl::PI_F16_F8_W) return OP_KS_F16_F8_W_ATTN_F16;
  if (h->attn_f16 && h->pi == opl::PI_F16_F8) return OP_KS_F16_F8_ATTN_F16;
  return h->wi_f8 ? 5 + h->pi : h->pi;  // 5 / 6: sets 0 / 1 with the Wi GEMM in the fp16 + e4m3 format
}

// Can this handle run kernel set `set` (op_kernel_set numbering)?  Packs present, a path that has the kernels, and no
// weight tensor below the reach of an fp16 plane for the sets that carry one.
bool set_available(const op_handle* h, int set) {
  const bool fast_path = h->row_path || h->panel_path;
  const bool row_layer_ok = !h->row_path || !(h->cfg.flags & (OP_FLAG_NO_LAYER_FUSION | OP_FLAG_LAYER_8X16 | OP_FLAG_LAYER_M32));
  if (h->cfg.flags & OP_FLAG_NO_POLICY_KERNELS) return set == OP_KS_BF16X3;
  switch (set) {
    case OP_KS_BF16X3: return true;
    case OP_KS_BF16_WEIGHTS:
    case OP_KS_BF16: return fast_path;
    case OP_KS_F16_F8:
    case OP_KS_F16_F8_W: return fast_path && h->f8_packs && !h->f16_unfit && row_layer_ok;
    case OP_KS_BF16X3_WI_F8:
    case OP_KS_BF16_WEIGHTS_WI_F8: return h->panel_path && h->f8_packs && !h->f16_unfit;
    case OP_KS_F16: return h->h16_packs && !h->f16_unfit && row_layer_ok;
    case OP_KS_F16_MLP_F8_W:
    case OP_KS_F16_MLP_F8: return h->panel_path && h->h16_packs && h->f8_packs && !h->f16_unfit;
    case OP_KS_F16_F8_W_ATTN_F16:
    case OP_KS_F16_F8_ATTN_F16: return h->panel_path && h->f8_packs && !h->f16_unfit;
    default: return false;
  }
}

// Run kernel set `set`: the evaluated terms become the set's own (a set with fewer terms than the checkpoint carries is
// an approximation -- op_calibrate measures it before choosing it).
void apply_set(op_handle* h, int set) {
  h->wi_f8 = set == OP_KS_BF16X3_WI_F8 || set == OP_KS_BF16_WEIGHTS_WI_F8;
  h->mlp_f8 = set == OP_KS_F16_MLP_F8_W || set == OP_KS_F16_MLP_F8;
  h->mlp_wlo = set == OP_KS_F16_MLP_F8_W;
  h->mlp_layers = (h->mlp_f8 && set == h->forced_set) ? h->forced_mlp_layers : ~0ull;
  h->attn_f16 = set == OP_KS_F16_F8_W_ATTN_F16 || set == OP_KS_F16_F8_ATTN_F16;
  h->pi = (set == OP_KS_F16 || h->mlp_f8) ? opl::PI_F16 : (h->wi_f8 ? set - 5 : set);
  if (h->attn_f16) h->pi = set == OP_KS_F16_F8_W_ATTN_F16 ? opl::PI_F16_F8_W : opl::PI_F16_F8;
  h->eff = opl::kPolicies[h->pi];
  if (h->attn_f16) h->eff.qk = h->eff.pv = 0;
  h->emulate = false;
}

// Evaluated policy = requested policy minus the hi x lo(weight) terms whose lo planes are identically zero for this
// checkpoint (bf16 weights: dropping the term changes no bit of the result), then the curated kernel set that has
// exactly those terms -- or kernel set 0 with the unused lo operands cleared.
int resolve_policy(op_handle* h) {
  if (h->resolved) return OP_OK;
  // [OP_FAM_COUNT]: some GEMM weight is not exactly an fp16 value; [+ 2]: some weight TENSOR sits on fp16's subnormal grid
  // (note_f16_fit in opk_common.hip.h): the "f16 + fp8" sets cannot represent it
  int any_lo[OP_FAM_COUNT + 3] = {0};
  OP_HIP(h, hipSetDevice(h->cfg.device_id));
  OP_HIP(h, hipMemcpy(any_lo, h->any_lo_dev, sizeof(any_lo), hipMemcpyDeviceToHost));
  Policy e = h->req;
  if (!any_lo[OP_FAM_WQKV]) e.wqkv &= ~OP_TERM_RIGHT_LO;
  if (!any_lo[OP_FAM_ATTN_OUT]) e.attn_out &= ~OP_TERM_RIGHT_LO;
  if (!any_lo[OP_FAM_WI]) e.wi &= ~OP_TERM_RIGHT_LO;
  if (!any_lo[OP_FAM_MLP_OUT]) e.mlp_out &= ~OP_TERM_RIGHT_LO;
  h->eff = e;
  h->pi = 0;
  h->emulate = true;
  h->wi_f8 = false;
  h->mlp_f8 = h->mlp_wlo = false;
  h->mlp_layers = ~0ull;
  h->attn_f16 = false;
  if (!(h->cfg.flags & OP_FLAG_NO_POLICY_KERNELS)) {
    for (int i = 0; i < opl::N_POLICIES; ++i)
      if (opl::kPolicies[i] == e) {
        h->pi = i;
        h->emulate = false;
        break;
      }
    // bf16-valued weights that are also exact fp16 values, on the whole-layer kernel's shapes: the "f16 + fp8" kernel
    // set evaluates the same terms at 1.5 instead of 2 MFMA units per product (op_internal.h)
    const bool f8_ok = !h->emulate && h->f8_packs && !h->f8_off && !any_lo[OP_FAM_COUNT + 2] &&
                       !(h->cfg.flags & (OP_FLAG_NO_LAYER_FUSION | OP_FLAG_LAYER_8X16 | OP_FLAG_LAYER_M32));
    // Panel path (hidden 512 / 768).  The whole layer in the format (OP_FLAG_PANEL_F8) costs 0.45-1.0e-3 on logits at the
    // published depths: opt-in.  The Wi GEMM ALONE -- 52 % of the GEMM FLOPs -- costs 1-1.5e-4 (base / large / en-gte at
    // 19-25 layers: <= 5.4e-4 against the oracle where the (hi, lo) bf16 sets give <= 4.7e-4; scripts/f8_depth_check.py,
    // profiles/r04_f8_depth_check.txt) and takes the fp32-valued Wi GEMM from 3 to 2 MFMA units per product: +5.6 %
    // pairs/s on base -- the DEFAULT for fp32-valued weights.  For bf16-valued weights (2 -> 1.5 units in a GEMM that
    // is not pipe-bound there) it measures +0.7 %: only on request (OP_FLAG_PANEL_F8_WI).
    const bool full_f8 = h->row_path || (h->cfg.flags & OP_FLAG_PANEL_F8);
    const bool wi_only = h->panel_path && !full_f8;
    h->wi_f8 = wi_only && f8_ok &&
               (h->pi == opl::PI_ALL_TERMS || ((h->cfg.flags & OP_FLAG_PANEL_F8_WI) && h->pi == opl::PI_BF16_WEIGHTS && !any_lo[OP_FAM_COUNT]));
    if (!wi_only) {
      if (f8_ok && h->pi == opl::PI_BF16_WEIGHTS && !any_lo[OP_FAM_COUNT]) h->pi = opl::PI_F16_F8;
      // every term requested and carried (fp32-valued weights): the same format with the weights' lo part as a third
      // plane -- one kernel per layer at 2 MFMA units per product instead of two kernels at 3
      if (f8_ok && h->pi == opl::PI_ALL_TERMS) h->pi = opl::PI_F16_F8_W;
    }
  } else if (opl::kPolicies[0] == e) {
    h->emulate = false;
  }
  h->f16_unfit = any_lo[OP_FAM_COUNT + 2] != 0;
  h->default_set = public_set(h);
  // a kernel set pinned by op_select_kernel_set / chosen by op_calibrate replaces the default selection
  if (h->forced_set >= 0 && set_available(h, h->forced_set)) apply_set(h, h->forced_set);
  h->resolved = true;
  return OP_OK;
}

// ---- calibration: the cheapest kernel set whose outputs stay within `tolerance` of the (hi, lo) bf16 kernels' ----------
// MFMA pipe time per algorithmic product of each kernel set, in 16-bit units (the panel path's sets 4
// and 5 measured in that order on base: 5.15 k vs 4.80 k pairs/s).  Candidates are tried in this order.
float set_cost(const op_handle* h, int set) {
  switch (set) {
    case OP_KS_F16: return 1.0f;
    case OP_KS_BF16: return 1.01f;  // same MFMA count as "f16", 8 instead of 11 significant bits: tried second
    case OP_KS_F16_MLP_F8: return 1.375f;
    case OP_KS_F16_F8_ATTN_F16: return 1.45f;
    case OP_KS_F16_F8: return 1.5f;
    case OP_KS_F16_MLP_F8_W: return 1.74f;
    case OP_KS_BF16_WEIGHTS_WI_F8: return 1.75f;
    case OP_KS_F16_F8_W_ATTN_F16: return 1.9f;
    case OP_KS_BF16_WEIGHTS: return 2.0f;
    case OP_KS_F16_F8_W: return h->panel_path ? 2.1f : 1.99f;
    case OP_KS_BF16X3_WI_F8: return 2.5f;
    default: return 3.0f;
  }
}

// deterministic calibration batch: 24 rows of min(512, max_pos) tokens + 14 ragged rows, ids uniform over the vocabulary
// (specials avoided as bench.py does: [1000, V - 1000) when the vocabulary is that large)
void synthetic_calibration_rows(const op_handle* h, std::vector<int32_t>& ids, std::vector<int32_t>& cu) {
  const int full = std::min(512, h->max_pos);
  // (the forward fuzz on calibrated sets finds its worst rows among the 1-3 token ones: several of them are in)
  const int ragged[14] = {1, 2, 2, 3, 5, 9, 17, 33, 64, 96, 130, 257, 333, 511};
  std::vector<int> lens(24, full);
  for (int r : ragged) lens.push_back(std::min(r, h->max_pos));
  const int lo = h->V > 4000 ? 1000 : 0, span = h->V > 4000 ? h->V - 2000 : h->V;
  uint64_t state = 0x9E3779B97F4A7C15ull;
  cu.assign(1, 0);
  ids.clear();
  for (int len : lens) {
    for (int i = 0; i < len; ++i) {
      state += 0x9E3779B97F4A7C15ull;  // splitmix64
      uint64_t z = state;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      ids.push_back(lo + (int32_t)(z % (uint64_t)span));
    }
    cu.push_back((int32_t)ids.size());
  }
}
}  // namespace

extern "C" {

int op_weights_ready(op_handle* h) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_weights_ready: NULL handle");
  if (h->missing.empty()) return resolve_policy(h);
  std::string msg = "missing weights:";
  size_t shown = 0;
  for (const auto& m : h->missing) {
    if (shown++ == 8) {
      msg += " ...";
      break;
    }
    msg += " " + m;
  }
  msg += " (" + std::to_string(h->missing.size()) + " total)";
  return fail(h, OP_ERR_STATE, "%s", msg.c_str());
}

int op_set_compact_operands(op_handle* h, int enabled, int* changed) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_set_compact_operands: NULL handle");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  const int before = public_set(h);
  h->f8_off = enabled == 0;
  if (!enabled) {  // a pinned / calibrated compact set goes too
    h->forced_set = -1;
    h->forced_mlp_layers = ~0ull;
  }
  h->resolved = false;
  rc = resolve_policy(h);
  if (changed) *changed = (rc == OP_OK && public_set(h) != before) ? 1 : 0;
  return rc;
}

int op_select_kernel_set(op_handle* h, int kernel_set) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_select_kernel_set: NULL handle");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  if (kernel_set != OP_KS_AUTO && !set_available(h, kernel_set))
    return fail(h, OP_ERR_UNSUPPORTED, "op_select_kernel_set: kernel set %d cannot run on this handle (shape, flags or weights)", kernel_set);
  h->forced_set = kernel_set == OP_KS_AUTO ? -1 : kernel_set;
  h->forced_mlp_layers = ~0ull;
  h->resolved = false;
  return resolve_policy(h);
}

int op_mlp_correction_layers(op_handle* h, uint64_t* layer_mask) {
  if (!h || !layer_mask) return fail(h, OP_ERR_INVALID, "op_mlp_correction_layers: NULL argument");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  const int n = h->cfg.num_layers;
  const uint64_t all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  *layer_mask = h->mlp_f8 ? (h->mlp_layers & all) : 0ull;
  return OP_OK;
}

int op_select_mlp_correction_layers(op_handle* h, uint64_t layer_mask) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_select_mlp_correction_layers: NULL handle");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  if (h->forced_set != OP_KS_F16_MLP_F8 && h->forced_set != OP_KS_F16_MLP_F8_W)
    return fail(h, OP_ERR_STATE, "op_select_mlp_correction_layers: pin kernel set %d or %d first (op_select_kernel_set / op_calibrate)", OP_KS_F16_MLP_F8_W,
                OP_KS_F16_MLP_F8);
  if (h->cfg.num_layers > 64) return fail(h, OP_ERR_UNSUPPORTED, "op_select_mlp_correction_layers: more than 64 layers");
  h->forced_mlp_layers = layer_mask;
  h->resolved = false;
  return resolve_policy(h);
}

int op_calibrate(op_handle* h, float tolerance, const int32_t* ids_host, const int32_t* cu_seqlens_host, int n_seqs,
                 op_calibration* report) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_calibrate: NULL handle");
  if (report && report->struct_bytes != sizeof(op_calibration))
    return fail(h, OP_ERR_INVALID, "op_calibrate: op_calibration.struct_bytes=%u, library expects %zu", report->struct_bytes, sizeof(op_calibration));
  if (!(tolerance >= 0.f)) return fail(h, OP_ERR_INVALID, "op_calibrate: tolerance must be >= 0");
  if ((ids_host == nullptr) != (cu_seqlens_host == nullptr) || (ids_host && n_seqs <= 0))
    return fail(h, OP_ERR_INVALID, "op_calibrate: ids_host / cu_seqlens_host / n_seqs must be given together");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  // the caller's batch is checked BEFORE the handle's state is touched: a refused call leaves a pinned set where it was
  if (ids_host) {
    if (cu_seqlens_host[0] != 0 || cu_seqlens_host[n_seqs] <= 0)
      return fail(h, OP_ERR_INVALID, "op_calibrate: cu_seqlens must start at 0 and hold at least one token");
    for (int s = 0; s < n_seqs; ++s) {
      const int len = cu_seqlens_host[s + 1] - cu_seqlens_host[s];
      if (len < 0 || len > h->cfg.max_position_embeddings)
        return fail(h, OP_ERR_INVALID, "op_calibrate: row %d has %d tokens (cu_seqlens must not decrease; max_position_embeddings is %d)", s, len,
                    h->cfg.max_position_embeddings);
    }
    for (int i = 0; i < cu_seqlens_host[n_seqs]; ++i)
      if (ids_host[i] < 0 || ids_host[i] >= h->cfg.vocab_size)
        return fail(h, OP_ERR_INVALID, "op_calibrate: token id %d at position %d is outside the embedding table (vocab_size %d)", ids_host[i], i,
                    h->cfg.vocab_size);
  }
  const bool full_report = report && (report->flags & OP_CAL_FULL_REPORT) != 0;
  OP_HIP(h, hipSetDevice(h->cfg.device_id));

  // the default selection and its (hi, lo) bf16 realisation = the reference of the comparison
  const bool f8_off_before = h->f8_off;
  const int forced_before = h->forced_set;
  const uint64_t forced_mlp_before = h->forced_mlp_layers;
  h->forced_set = -1;
  h->forced_mlp_layers = ~0ull;
  h->resolved = false;
  OP_TRY(resolve_policy(h));
  const int default_set = public_set(h);
  h->f8_off = true;
  h->resolved = false;
  OP_TRY(resolve_policy(h));
  const int reference_set = public_set(h);
  h->f8_off = f8_off_before;
  h->resolved = false;
  OP_TRY(resolve_policy(h));

  op_calibration rep;
  memset(&rep, 0, sizeof(rep));
  rep.struct_bytes = sizeof(rep);
  rep.tolerance = tolerance;
  rep.reference_set = reference_set;
  rep.default_set = default_set;
  rep.chosen_set = default_set;
  rep.flags = report ? report->flags : 0u;
  auto finish = [&]() -> int {
    if (report) *report = rep;
    return OP_OK;
  };
  // nothing to measure: whatever was pinned before the call stays pinned
  auto finish_unchanged = [&]() -> int {
    h->forced_set = forced_before;
    h->forced_mlp_layers = forced_mlp_before;
    h->resolved = false;
    OP_TRY(resolve_policy(h));
    rep.chosen_set = public_set(h);
    return finish();
  };
  if (default_set < 0 || reference_set < 0) return finish_unchanged();  // a custom policy on the all-terms kernels: nothing cheaper is defined

  // candidates: every available kernel set cheaper than the default one, cheapest first
  std::vector<int> cand;
  for (int set = 0; set < OP_KS_COUNT; ++set)
    if (set != default_set && set_available(h, set) && set_cost(h, set) < set_cost(h, default_set)) cand.push_back(set);
  std::sort(cand.begin(), cand.end(), [&](int a, int b) { return set_cost(h, a) < set_cost(h, b); });
  if (cand.size() > 16) cand.resize(16);
  if (cand.empty()) return finish_unchanged();

  std::vector<int32_t> ids, cu;
  if (ids_host) {
    cu.assign(cu_seqlens_host, cu_seqlens_host + n_seqs + 1);
    ids.assign(ids_host, ids_host + cu[n_seqs]);
  } else {
    synthetic_calibration_rows(h, ids, cu);
    n_seqs = (int)cu.size() - 1;
  }
  const int total = cu[n_seqs];
  int max_len = 0;
  for (int s = 0; s < n_seqs; ++s) max_len = std::max(max_len, cu[s + 1] - cu[s]);
  rep.n_rows = n_seqs;
  rep.n_tokens = total;

  const size_t ws_bytes = op_workspace_bytes(h, n_seqs, total, max_len);
  const size_t n_prune = (size_t)total * 2, n_rank = (size_t)n_seqs * h->nl;
  int32_t *ids_dev = nullptr, *cu_dev = nullptr;
  float* out_dev = nullptr;
  void* ws_dev = nullptr;
  hipStream_t st = nullptr;
  auto release = [&]() {
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(ids_dev);
    (void)hipFree(cu_dev);
    (void)hipFree(out_dev);
    (void)hipFree(ws_dev);
  };
  hipError_t e = hipMalloc((void**)&ids_dev, ids.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&cu_dev, cu.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&out_dev, (n_prune + n_rank) * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&ws_dev, ws_bytes + 256);
  if (e == hipSuccess) e = hipStreamCreate(&st);
  if (e == hipSuccess) e = hipMemcpy(ids_dev, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(cu_dev, cu.data(), cu.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    release();
    return fail(h, OP_ERR_HIP, "op_calibrate: staging failed: %s", hipGetErrorString(e));
  }
  void* ws_aligned = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws_dev) + 255) / 256 * 256);
  std::vector<float> ref(n_prune + n_rank), got(n_prune + n_rank);
  const bool profiling = h->profiling;
  float* const capture = h->capture;
  h->profiling = false;
  h->capture = nullptr;
  auto run = [&](int set, std::vector<float>& out, uint64_t mlp_layers = ~0ull) -> int {
    h->forced_set = set;
    h->forced_mlp_layers = mlp_layers;
    h->resolved = false;
    OP_TRY(resolve_policy(h));
    OP_TRY(op_forward_packed(h, ids_dev, cu_dev, cu.data(), n_seqs, total, max_len, out_dev, out_dev + n_prune, nullptr, ws_aligned,
                             ws_bytes, st));
    OP_HIP(h, hipStreamSynchronize(st));
    OP_HIP(h, hipMemcpy(out.data(), out_dev, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    return OP_OK;
  };
  rc = run(reference_set, ref);
  bool ref_finite = true;
  for (float v : ref) ref_finite = ref_finite && std::isfinite(v);
  int chosen = -1;
  auto max_diff = [&]() {
    float err = 0.f;
    for (size_t i = 0; i < got.size(); ++i) {
      const float d = std::fabs(got[i] - ref[i]);
      err = (d <= err) ? err : d;  // a NaN difference propagates: (NaN <= err) is false
    }
    return std::isfinite(err) ? err : INFINITY;
  };
  if (rc == OP_OK && ref_finite && default_set != reference_set) {
    rc = run(default_set, got);
    if (rc == OP_OK) rep.default_err = max_diff();
  }
  if (rc == OP_OK && ref_finite) {
    for (size_t c = 0; c < cand.size() && rc == OP_OK; ++c) {
      rc = run(cand[c], got);
      if (rc != OP_OK) break;
      const float err = max_diff();
      rep.candidate_set[rep.n_candidates] = cand[c];
      rep.candidate_err[rep.n_candidates] = err;
      rep.n_candidates += 1;
      if (chosen < 0 && err <= tolerance) chosen = cand[c];
      if (chosen >= 0 && !full_report) break;  // cheapest first: the first one that holds is the answer (OP_CAL_FULL_REPORT measures them all)
    }
  }
  // Kernel sets 8 / 9 chosen (the "f16" set alone is beyond the tolerance, the MLP in the fp16 + e4m3 format brings it back):
  // the correction is a per-layer choice (panel_layer): drop it layer by layer for as long as the batch stays within the
  // tolerance.  The error is a maximum over logits, not a sum over layers, so the order of the walk decides how many layers
  // go: three walks (first to last, last to first, and by the price of dropping each layer alone), the one that keeps the
  // fewest layers wins.  About four forwards of the calibration batch per layer, at load.
  const int n_layers = h->cfg.num_layers;
  const uint64_t all_layers = n_layers >= 64 ? ~0ull : ((1ull << n_layers) - 1ull);
  uint64_t mlp_layers = all_layers;
  if (rc == OP_OK && (chosen == OP_KS_F16_MLP_F8 || chosen == OP_KS_F16_MLP_F8_W) && n_layers <= 64 && !(rep.flags & OP_CAL_WHOLE_DEPTH)) {
    std::vector<std::pair<float, int>> alone;
    for (int li = 0; li < n_layers && rc == OP_OK; ++li) {
      rc = run(chosen, got, all_layers & ~(1ull << li));
      if (rc == OP_OK) alone.emplace_back(max_diff(), li);
    }
    std::vector<std::vector<int>> walks(3);
    for (int li = 0; li < n_layers; ++li) {
      walks[0].push_back(li);
      walks[1].push_back(n_layers - 1 - li);
    }
    std::vector<std::pair<float, int>> by_price = alone;
    std::stable_sort(by_price.begin(), by_price.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    for (const auto& pr : by_price) walks[2].push_back(pr.second);
    int best_kept = n_layers + 1;
    for (size_t w = 0; w < walks.size() && rc == OP_OK && alone.size() == (size_t)n_layers; ++w) {
      uint64_t mask = all_layers;
      float mask_err = 0.f;
      for (int li : walks[w]) {
        const uint64_t trial = mask & ~(1ull << li);
        if (trial == 0ull) break;  // (= the "f16" set, measured above)
        float err = alone[li].first;
        if (trial != (all_layers & ~(1ull << li))) {
          rc = run(chosen, got, trial);
          if (rc != OP_OK) break;
          err = max_diff();
        }
        if (err <= tolerance) {
          mask = trial;
          mask_err = err;
        }
      }
      const int kept = __builtin_popcountll(mask);
      if (rc == OP_OK && kept < best_kept) {
        best_kept = kept;
        mlp_layers = mask;
        rep.mlp_layers_err = mask_err;
      }
    }
  }
  h->profiling = profiling;
  h->capture = capture;
  release();
  // nothing cheaper holds: the default stays -- unless it cannot even represent this batch (fp16 range), or is itself further
  // from the (hi, lo) bf16 kernels than TEN times the tolerance (1e-3 at the default tolerance: the path's own bar; the O(1)
  // worst-case weights put it at 6e-4): a checkpoint with outlier channels at 30 - 100 x puts the fp16 + e4m3 default 4e-2
  // from them (scripts/trained_like_probe.py) -- then the reference set runs, the most exact arithmetic this library has
  if (rc == OP_OK && chosen < 0 && ref_finite && default_set != reference_set &&
      (!std::isfinite(rep.default_err) || rep.default_err > 10.0f * tolerance))
    chosen = reference_set;
  h->forced_set = (rc == OP_OK && chosen >= 0) ? chosen : -1;
  h->forced_mlp_layers = (rc == OP_OK && chosen >= 0) ? (mlp_layers | ~all_layers) : ~0ull;
  h->resolved = false;
  const int rc2 = resolve_policy(h);
  if (rc != OP_OK) return rc;
  if (rc2 != OP_OK) return rc2;
  rep.chosen_set = public_set(h);
  rep.mlp_layers = h->mlp_f8 ? (h->mlp_layers & all_layers) : 0ull;
  return finish();
}

int op_effective_policy(op_handle* h, uint8_t* terms_out, int* kernel_set) {
  if (!h || !terms_out || !kernel_set) return fail(h, OP_ERR_INVALID, "op_effective_policy: NULL argument");
  int rc = op_weights_ready(h);
  if (rc != OP_OK) return rc;
  terms_out[OP_FAM_WQKV] = (uint8_t)h->eff.wqkv;
  terms_out[OP_FAM_QK] = (uint8_t)h->eff.qk;
  terms_out[OP_FAM_PV] = (uint8_t)h->eff.pv;
  terms_out[OP_FAM_ATTN_OUT] = (uint8_t)h->eff.attn_out;
  terms_out[OP_FAM_WI] = (uint8_t)h->eff.wi;
  terms_out[OP_FAM_MLP_OUT] = (uint8_t)h->eff.mlp_out;
  *kernel_set = public_set(h);
  return OP_OK;
}

size_t op_workspace_bytes(const op_handle* h, int n_seqs, int total_tokens, int max_seqlen) {
  if (!h || n_seqs < 0 || total_tokens < 0 || max_seqlen < 0) return 0;
  const int cap = chunk_row_capacity(h, n_seqs, total_tokens, max_seqlen);
  const int cap_pad = align_up(cap + 64, 256);
  Workspace ws;
  carve(h, nullptr, cap_pad, n_seqs, ws);
  return ws.bytes;
}

int op_debug_capture_hidden(op_handle* h, float* hidden_dev) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_debug_capture_hidden: NULL handle");
  h->capture = hidden_dev;
  return OP_OK;
}

int op_profile_enable(op_handle* h, int enabled) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_profile_enable: NULL handle");
  h->profiling = enabled != 0;
  return OP_OK;
}

int op_segment_means(op_handle* h, const float* keep_prob_dev, int n_values, const int32_t* seg_dev, int n_seg,
                     float* out_dev, void* hip_stream) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_segment_means: NULL handle");
  if (n_seg < 0 || n_values < 0 || (n_seg > 0 && (!keep_prob_dev || !seg_dev || !out_dev)))
    return fail(h, OP_ERR_INVALID, "op_segment_means: NULL buffer or negative count");
  if (n_seg == 0) return OP_OK;
  OP_HIP(h, hipSetDevice(h->cfg.device_id));
  hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)((n_seg + 127) / 128)), dim3(128), 0, (hipStream_t)hip_stream,
                     keep_prob_dev, seg_dev, n_seg, n_values, out_dev);
  OP_HIP(h, hipGetLastError());
  return OP_OK;
}

int op_debug_clock_probe(op_handle* h, int spin_us, unsigned long long* out_dev, void* hip_stream) {
  if (!h || !out_dev || spin_us <= 0) return fail(h, OP_ERR_INVALID, "op_debug_clock_probe: bad argument");
  OP_HIP(h, hipSetDevice(h->cfg.device_id));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, (unsigned long long)spin_us * 100ull, out_dev);
  OP_HIP(h, hipGetLastError());
  return OP_OK;
}

int op_profile_reset(op_handle* h) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_profile_reset: NULL handle");
  int rc = drain_profile(h);
  for (int k = 0; k < PK_COUNT; ++k) {
    h->prof_ms[k] = 0;
    h->prof_launches[k] = 0;
  }
  return rc;
}

int op_profile_read(op_handle* h, op_profile_entry* entries, int max_entries) {
  if (!h || (!entries && max_entries > 0)) return fail(h, OP_ERR_INVALID, "op_profile_read: NULL argument");
  int rc = drain_profile(h);
  if (rc != OP_OK) return rc;
  int n = 0;
  for (int k = 0; k < PK_COUNT && n < max_entries; ++k) {
    if (h->prof_launches[k] == 0) continue;
    entries[n].kind = k;
    entries[n].launches = h->prof_launches[k];
    entries[n].total_ms = h->prof_ms[k];
    ++n;
  }
  return n;
}

int op_forward_packed(op_handle* h, const int32_t* ids_dev, const int32_t* cu_dev, const int32_t* cu_host_in, int n_seqs,
                      int total_tokens, int max_seqlen, float* prune_out, float* rank_out, float* keep_prob,
                      void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_forward_packed: NULL handle");
  if (n_seqs < 0 || total_tokens < 0 || max_seqlen < 0) return fail(h, OP_ERR_INVALID, "negative size");
  if (n_seqs == 0 || total_tokens == 0) {
    if (n_seqs > 0 && rank_out) {
      hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
      OP_HIP(h, hipMemsetAsync(rank_out, 0, (size_t)n_seqs * h->nl * sizeof(float), st));
    }
    return OP_OK;
  }
  if (!ids_dev || !cu_dev || !prune_out || !rank_out || !workspace)
    return fail(h, OP_ERR_INVALID, "op_forward_packed: NULL buffer");
  if (op_weights_ready(h) != OP_OK) return OP_ERR_STATE;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0)
    return fail(h, OP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  const size_t need = op_workspace_bytes(h, n_seqs, total_tokens, max_seqlen);
  if (workspace_bytes < need)
    return fail(h, OP_ERR_WORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, need);

  OP_HIP(h, hipSetDevice(h->cfg.device_id));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);

  std::vector<int32_t> cu_copy;
  const int32_t* cu = cu_host_in;
  if (!cu) {
    cu_copy.resize((size_t)n_seqs + 1);
    OP_HIP(h, hipMemcpyAsync(cu_copy.data(), cu_dev, cu_copy.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    OP_HIP(h, hipStreamSynchronize(stream));
    cu = cu_copy.data();
  }
  if (cu[0] != 0 || cu[n_seqs] != total_tokens)
    return fail(h, OP_ERR_INVALID, "cu_seqlens must start at 0 and end at total_tokens (%d): got %d..%d", total_tokens,
                cu[0], cu[n_seqs]);
  for (int s = 0; s < n_seqs; ++s) {
    const int len = cu[s + 1] - cu[s];
    if (len < 0) return fail(h, OP_ERR_INVALID, "cu_seqlens not monotone at %d", s);
    if (len > max_seqlen) return fail(h, OP_ERR_INVALID, "sequence %d has %d tokens > max_seqlen %d", s, len, max_seqlen);
    if (len > h->max_pos)
      return fail(h, OP_ERR_UNSUPPORTED, "sequence %d has %d tokens > max_position_embeddings %d", s, len, h->max_pos);
  }

  const int cap = chunk_row_capacity(h, n_seqs, total_tokens, max_seqlen);
  const int cap_pad = align_up(cap + 64, 256);
  Workspace ws;
  carve(h, reinterpret_cast<char*>(workspace), cap_pad, n_seqs, ws);

  Launcher L{h, stream};
  int s0 = 0;
  while (s0 < n_seqs) {
    int rows = 0, s1 = s0, max_len = 0;
    while (s1 < n_seqs) {
      const int len = cu[s1 + 1] - cu[s1];
      const int r = align_up(len, ROW_ALIGN);
      if (s1 > s0 && rows + r > cap) break;
      rows += r;
      max_len = std::max(max_len, len);
      ++s1;
    }
    if (rows > cap) return fail(h, OP_ERR_WORKSPACE, "internal: chunk of %d rows exceeds capacity %d", rows, cap);
    // attention work items of the full-attention layers: 256-query blocks (8 waves) once a sequence is longer than
    // 128 tokens, else 128 -- unless that leaves CUs idle (small request): then 128-query blocks, twice as many work
    // items.  Sliding-window layers always use 128-query blocks.
    auto count_items = [&](int waves) {
      int items = 0;
      for (int s = s0; s < s1; ++s) items += (cu[s + 1] - cu[s] + waves * 32 - 1) / (waves * 32);
      return items;
    };
    const bool forced = (h->cfg.flags & (OP_FLAG_ATT_WAVES_4 | OP_FLAG_ATT_WAVES_8)) != 0;
    AttnPlan plan;
    plan.waves_g = forced ? ((h->cfg.flags & OP_FLAG_ATT_WAVES_4) ? 4 : 8) : (max_len > 128 ? 8 : 4);
    plan.items_g = count_items(plan.waves_g);
    if (!forced && plan.waves_g == 8 && (long)plan.items_g * h->nh <= h->n_cus) {
      plan.waves_g = 4;
      plan.items_g = count_items(4);
    }
    plan.items_l = count_items(4);
    if (rows > 0) {
      OP_TRY(forward_chunk(h, L, ws, ids_dev, cu_dev, s0, s1 - s0, rows, max_len, total_tokens, plan, prune_out, rank_out,
                           keep_prob));
    } else {
      // only empty sequences in this chunk
      OP_HIP(h, hipMemsetAsync(rank_out + (size_t)s0 * h->nl, 0, (size_t)(s1 - s0) * h->nl * sizeof(float), stream));
    }
    s0 = s1;
  }
  return OP_OK;
}

}  // extern "C"
