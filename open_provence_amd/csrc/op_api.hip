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

#include "op_kernels.hip.h"

using namespace opk;

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
  PK_COUNT
};
const char* kProfileNames[PK_COUNT] = {"rowmap",        "embed_ln",      "layer_norm",    "gemm_qk_rope", "gemm_v_t",
                                       "attn_global",   "attn_local",    "gemm_attn_out", "gemm_wi_geglu",
                                       "gemm_mlp_out",  "final_ln_prune", "rank_head",    "capture",
                                       "rowgemm_ln_qkv_rope", "rowgemm_attn_out", "rowgemm_ln_wi_geglu",
                                       "kstream_mlp_out", "fused_attnout_ln_wi_geglu", "fused_mlpout_ln_qkv_rope"};

struct LayerWeights {
  float* attn_norm = nullptr;  // absent on layer 0
  float* mlp_norm = nullptr;
  u16 *wqkv_hi = nullptr, *wqkv_lo = nullptr;
  u16 *wo_hi = nullptr, *wo_lo = nullptr;
  u16 *wi_hi = nullptr, *wi_lo = nullptr;
  u16 *wo2_hi = nullptr, *wo2_lo = nullptr;
  // row-stationary layouts (hidden <= 256): chunk-major, fragment-ordered, hi/lo planes interleaved per k-step
  u16 *wqkv_pk = nullptr, *wo_pk = nullptr, *wi_pk = nullptr;
  u16* wo2_pk = nullptr;  // k-streamed layouts (output features permuted): MLP output projection ...
  u16* wo_ks = nullptr;   // ... and attention output projection (fused kernels)
};

struct ProfileEvent {
  int kind;
  hipEvent_t start, stop;
};

}  // namespace

struct op_handle {
  op_config cfg;
  int H = 0, I = 0, N = 0, nh = 0, V = 0, nl = 0, max_pos = 0;
  bool split = true;
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
  int32_t *row_seq, *row_pos, *row_tok, *roff, *qboff;
  float* cls;
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
  ws.cls = (float*)take(std::max<size_t>((size_t)n_seqs, 1) * H * 4);
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


template <int EPI, int PRO>
int launch_rowgemm(Launcher& L, int kind, const RowGemmParams& p, int hidden, int r_pad, bool split) {
  OP_TRY(L.begin(kind));
  // 4 waves x 32 rows = 128-row blocks, two per CU.  Small batches (at most one such block per CU) use 4 waves x
  // 16 rows = 64-row blocks instead: twice the blocks, so a latency-bound request spreads over twice the CUs.
  // (rowgemm_kernel also compiles as 8 waves x 16 rows = 4 waves per SIMD at <= 128 VGPRs; measured on MI355X it is
  // equal on the q/k/v kernel and 10 % slower on the GeGLU kernel, which spills.)
  const bool small = (r_pad / ROW_BM) <= L.h->n_cus && getenv("OPEN_PROVENCE_NO_SMALL_BLOCKS") == nullptr;
  const dim3 grid((unsigned)(r_pad / (small ? 64 : ROW_BM)));
  const dim3 block(256);
  const int ks = hidden / 32;
#define OPK_ROW_LAUNCH(KS_)                                                                              \
  do {                                                                                                   \
    if (split && small)                                                                                  \
      hipLaunchKernelGGL((rowgemm_kernel<KS_, EPI, PRO, true, 4, 1>), grid, block, 0, L.stream, p);      \
    else if (split)                                                                                      \
      hipLaunchKernelGGL((rowgemm_kernel<KS_, EPI, PRO, true, 4, 2>), grid, block, 0, L.stream, p);      \
    else if (small)                                                                                      \
      hipLaunchKernelGGL((rowgemm_kernel<KS_, EPI, PRO, false, 4, 1>), grid, block, 0, L.stream, p);     \
    else                                                                                                 \
      hipLaunchKernelGGL((rowgemm_kernel<KS_, EPI, PRO, false, 4, 2>), grid, block, 0, L.stream, p);     \
  } while (0)
  if (ks == 4) OPK_ROW_LAUNCH(4);
  else if (ks == 8) OPK_ROW_LAUNCH(8);
  else return fail(L.h, OP_ERR_UNSUPPORTED, "row-stationary GEMM supports hidden 128 or 256, got %d", hidden);
#undef OPK_ROW_LAUNCH
  return L.end();
}

int launch_kstream(Launcher& L, int kind, const KStreamParams& p, int hidden, int r_pad, bool split) {
  OP_TRY(L.begin(kind));
  const dim3 grid((unsigned)(r_pad / ROW_BM));
  const dim3 block(256);
  const int nf = hidden / 16;
#define OPK_KS_LAUNCH(NF_)                                                                        \
  do {                                                                                            \
    if (split)                                                                                    \
      hipLaunchKernelGGL((kstream_gemm_kernel<NF_, true, 4>), grid, block, 0, L.stream, p);       \
    else                                                                                          \
      hipLaunchKernelGGL((kstream_gemm_kernel<NF_, false, 4>), grid, block, 0, L.stream, p);      \
  } while (0)
  if (nf == 8) OPK_KS_LAUNCH(8);
  else if (nf == 16) OPK_KS_LAUNCH(16);
  else return fail(L.h, OP_ERR_UNSUPPORTED, "k-streamed GEMM supports hidden 128 or 256, got %d", hidden);
#undef OPK_KS_LAUNCH
  return L.end();
}

int forward_chunk(op_handle* h, Launcher& L, const Workspace& ws, const int32_t* ids_dev, const int32_t* cu_dev, int s0,
                  int ns, int rows, int max_len, int total_tokens, int att_waves, int att_items, float* prune_out,
                  float* rank_out) {
  const int H = h->H, I = h->I;
  // Row path: exactly the computed rows, rounded to the 128-row block -- no slack rows: a 131072-row batch is 1024
  // blocks = two full rounds of 2 blocks per CU; two extra (empty) blocks would cost a third round.  The tiled path
  // keeps 64 slack rows for its attention kernel's tile over-read.
  const bool fp_layout = h->row_path || h->panel_path;  // fragment-packed activations, attn_fp_kernel
  const int r_pad = fp_layout ? align_up(rows, ROW_BM) : align_up(rows + 64, 256);
  const int m_tiles = r_pad / GEMM_BM;
  const unsigned row_blocks = (unsigned)((r_pad + 3) / 4);
  const bool split = h->split;
  hipStream_t st = L.stream;

  OP_TRY(L.begin(PK_ROWMAP));
  hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, cu_dev, s0, ns, ROW_ALIGN, ROW_ALIGN, ws.roff);
  if (fp_layout) hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, st, cu_dev, s0, ns, att_waves * 32, 1, ws.qboff);
  hipLaunchKernelGGL(row_map_kernel, dim3((unsigned)((r_pad + 255) / 256)), dim3(256), 0, st, cu_dev, s0, ns, ws.roff,
                     r_pad, ws.row_seq, ws.row_pos, ws.row_tok);
  OP_TRY(L.end());

  // rows >= `rows` are never produced by the attention kernel: keep its output finite there
  if (fp_layout) {  // fragment-packed o: pieces of 16 rows, hi/lo interleaved, o_hi + o_lo are one buffer
    const size_t tail_off = (size_t)(rows / 16) * (H / 32) * 2 * 512;
    const size_t tail_bytes = (size_t)((r_pad - rows) / 16) * (H / 32) * 2 * 512 * sizeof(u16);
    OP_HIP(h, hipMemsetAsync(ws.o_hi + tail_off, 0, tail_bytes, st));
  } else {
    const size_t tail_off = (size_t)rows * H;
    const size_t tail_bytes = (size_t)(r_pad - rows) * H * sizeof(u16);
    OP_HIP(h, hipMemsetAsync(ws.o_hi + tail_off, 0, tail_bytes, st));
    if (split) OP_HIP(h, hipMemsetAsync(ws.o_lo + tail_off, 0, tail_bytes, st));
  }

  OP_TRY(L.begin(PK_EMBED_LN));
  if (split)
    hipLaunchKernelGGL((embed_ln_kernel<true>), dim3(row_blocks), dim3(256), 0, st, ids_dev, ws.row_tok, h->emb,
                       h->emb_norm, h->cfg.norm_eps, H, r_pad, h->V, ws.x, ws.ln_hi, ws.ln_lo);
  else
    hipLaunchKernelGGL((embed_ln_kernel<false>), dim3(row_blocks), dim3(256), 0, st, ids_dev, ws.row_tok, h->emb,
                       h->emb_norm, h->cfg.norm_eps, H, r_pad, h->V, ws.x, ws.ln_hi, ws.ln_lo);
  OP_TRY(L.end());

  auto capture = [&](int index) -> int {
    if (!h->capture) return OP_OK;
    OP_TRY(L.begin(PK_CAPTURE));
    hipLaunchKernelGGL(capture_rows_kernel, dim3(row_blocks), dim3(256), 0, st, ws.x, ws.row_tok, H, r_pad,
                       h->capture + (size_t)index * total_tokens * H);
    return L.end();
  };
  auto layer_norm = [&](const float* w) -> int {
    OP_TRY(L.begin(PK_LN));
    if (split)
      hipLaunchKernelGGL((ln_kernel<true>), dim3(row_blocks), dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                         ws.ln_hi, ws.ln_lo);
    else
      hipLaunchKernelGGL((ln_kernel<false>), dim3(row_blocks), dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                         ws.ln_hi, ws.ln_lo);
    return L.end();
  };

  const int q_tiles = (max_len + ATT_BQ - 1) / ATT_BQ;

  const bool fuse = getenv("OPEN_PROVENCE_NO_FUSE") == nullptr;

  auto attention = [&](bool is_global) -> int {
    OP_TRY(L.begin(is_global ? PK_ATTN_GLOBAL : PK_ATTN_LOCAL));
    // fragment-packed attention: one block per (sequence, block of att_waves * 32 queries, head) work item
    const dim3 grid(fp_layout ? (unsigned)att_items : (unsigned)q_tiles, (unsigned)h->nh, fp_layout ? 1u : (unsigned)ns);
    const int window = is_global ? -1 : h->cfg.local_attention / 2;
    if (fp_layout) {
      AttnFpParams ap;  // q_hi/q_lo (k, vt, o likewise) are adjacent: together they hold the fragment-packed tensor
      ap.q_fp = ws.q_hi;
      ap.k_fp = ws.k_hi;
      ap.vt_fp = ws.vt_hi;
      ap.o_fp = ws.o_hi;
      ap.cu = cu_dev;
      ap.s0 = s0;
      ap.roff = ws.roff;
      ap.qboff = ws.qboff;
      ap.ns = ns;
      ap.H = H;
      ap.r_pad = r_pad;
      ap.window = window;
      if (split && att_waves == 8)
        hipLaunchKernelGGL((attn_fp_kernel<true, 8>), grid, dim3(512), 0, st, ap);
      else if (split)
        hipLaunchKernelGGL((attn_fp_kernel<true, 4>), grid, dim3(256), 0, st, ap);
      else if (att_waves == 8)
        hipLaunchKernelGGL((attn_fp_kernel<false, 8>), grid, dim3(512), 0, st, ap);
      else
        hipLaunchKernelGGL((attn_fp_kernel<false, 4>), grid, dim3(256), 0, st, ap);
    } else {
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
      if (split)
        hipLaunchKernelGGL((attn_kernel<true>), grid, dim3(256), 0, st, ap);
      else
        hipLaunchKernelGGL((attn_kernel<false>), grid, dim3(256), 0, st, ap);
    }
    return L.end();
  };

  // parameters of a q/k/v projection of layer `li` (row-stationary kernels)
  auto qkv_params = [&](int li) {
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
    rp.wp = lw.wqkv_pk;
    rp.n_chunks = 3 * H / ROW_CHUNK;
    rp.n_swapped = 2 * H / ROW_CHUNK;
    rp.o0_hi = ws.q_hi;
    rp.o1_hi = ws.k_hi;
    rp.o2_hi = ws.vt_hi;
    rp.ld_out = H;
    return rp;
  };

  for (int li = 0; li < h->N; ++li) {
    const LayerWeights& lw = h->layers[li];
    const bool is_global = h->cfg.layer_is_global[li] != 0;
    OP_TRY(capture(li));

    if (h->row_path) {
      // ---- row-stationary path (hidden <= 256) --------------------------------------------------------
      if (li == 0) {  // layer 0 has no attn_norm: split x0 directly
        RowGemmParams rp = qkv_params(0);
        OP_TRY((launch_rowgemm<RE_QKV, RP_SPLIT>(L, PK_ROW_QKV, rp, H, r_pad, split)));
      } else if (!fuse) {
        RowGemmParams rp = qkv_params(li);
        OP_TRY((launch_rowgemm<RE_QKV, RP_LN>(L, PK_ROW_QKV, rp, H, r_pad, split)));
      }  // else: q/k/v of this layer were produced by the fused kernel that closed layer li-1
      OP_TRY(attention(is_global));

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
      if (fuse) {
        // x += o Wo^T ; h = GeGLU(LN(x) Wi^T)   -- one kernel, the hidden state stays in registers in between
        rp.a1_fp = ws.o_hi;
        rp.w1p = lw.wo_ks;
        rp.k1_steps = H / 32;
        rp.x_io = ws.x;
        OP_TRY((launch_rowgemm<RE_GEGLU, RP_KSTREAM>(L, PK_FUSED_ATTN_OUT_WI, rp, H, r_pad, split)));
      } else {
        RowGemmParams ro;
        memset(&ro, 0, sizeof(ro));
        ro.hidden = H;
        ro.r_pad = r_pad;
        ro.a_hi = ws.o_hi;
        ro.wp = lw.wo_pk;
        ro.n_chunks = H / ROW_CHUNK;
        ro.x = ws.x;
        ro.ld_out = H;
        OP_TRY((launch_rowgemm<RE_RESIDUAL, RP_PLANES>(L, PK_ROW_ATTN_OUT, ro, H, r_pad, split)));
        rp.x_in = ws.x;
        OP_TRY((launch_rowgemm<RE_GEGLU, RP_LN>(L, PK_ROW_WI_GEGLU, rp, H, r_pad, split)));
      }

      if (fuse && li + 1 < h->N) {
        // x += h Wo^T ; q, k, v^T of the NEXT layer = RoPE / transpose of LN(x) Wqkv^T
        RowGemmParams rq = qkv_params(li + 1);
        rq.a1_fp = ws.h_hi;
        rq.w1p = lw.wo2_pk;
        rq.k1_steps = I / 32;
        rq.x_io = ws.x;
        OP_TRY((launch_rowgemm<RE_QKV, RP_KSTREAM>(L, PK_FUSED_MLP_OUT_QKV, rq, H, r_pad, split)));
      } else {
        KStreamParams kp;
        kp.a_fp = ws.h_hi;
        kp.wp = lw.wo2_pk;
        kp.n_ksteps = I / 32;
        kp.x = ws.x;
        OP_TRY(launch_kstream(L, PK_KSTREAM_MLP_OUT, kp, H, r_pad, split));
      }
      continue;
    }

    if (h->panel_path) {
      // ---- panel path (hidden % 256 == 0): LayerNorm -> fragment-packed planes, k-streamed panel GEMMs ----
      const dim3 ln_grid((unsigned)(r_pad / 16));
      auto layer_norm_fp = [&](const float* w) -> int {
        OP_TRY(L.begin(PK_LN));
        if (split)
          hipLaunchKernelGGL((ln_fp_kernel<true>), ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                             w ? 1 : 0, ws.ln_hi);
        else
          hipLaunchKernelGGL((ln_fp_kernel<false>), ln_grid, dim3(256), 0, st, ws.x, w, h->cfg.norm_eps, H, r_pad,
                             w ? 1 : 0, ws.ln_hi);
        return L.end();
      };
      auto panel = [&](int kind, int epi, const PanelParams& pp, int n_tiles) -> int {
        OP_TRY(L.begin(kind));
        const dim3 grid((unsigned)(r_pad / ROW_BM), (unsigned)n_tiles);
#define OPK_PANEL(EPI_)                                                                         \
  do {                                                                                          \
    if (split)                                                                                  \
      hipLaunchKernelGGL((panel_gemm_kernel<EPI_, true>), grid, dim3(256), 0, st, pp);          \
    else                                                                                        \
      hipLaunchKernelGGL((panel_gemm_kernel<EPI_, false>), grid, dim3(256), 0, st, pp);         \
  } while (0)
        if (epi == PE_RESIDUAL) OPK_PANEL(PE_RESIDUAL);
        else if (epi == PE_QK) OPK_PANEL(PE_QK);
        else if (epi == PE_V) OPK_PANEL(PE_V);
        else OPK_PANEL(PE_GEGLU);
#undef OPK_PANEL
        return L.end();
      };
      OP_TRY(layer_norm_fp(li != 0 ? lw.attn_norm : nullptr));  // layer 0: attn_norm is Identity -> plain split
      PanelParams pp;
      memset(&pp, 0, sizeof(pp));
      pp.r_pad = r_pad;
      pp.hidden = H;
      pp.row_pos = ws.row_pos;
      pp.rope_cos = h->rope_cos[is_global ? 1 : 0];
      pp.rope_sin = h->rope_sin[is_global ? 1 : 0];
      pp.max_pos = h->max_pos;
      pp.a_fp = ws.ln_hi;
      pp.n_ksteps = H / 32;
      pp.wp = lw.wqkv_pk;
      pp.o0 = ws.q_hi;
      pp.o1 = ws.k_hi;
      OP_TRY(panel(PK_GEMM_QK_ROPE, PE_QK, pp, 2 * H / 256));
      pp.wp = lw.wqkv_pk + (size_t)(2 * H / 256) * (H / 32) * 2 * 8192;
      pp.o0 = ws.vt_hi;
      OP_TRY(panel(PK_GEMM_V_T, PE_V, pp, H / 256));
      OP_TRY(attention(is_global));
      pp.a_fp = ws.o_hi;
      pp.wp = lw.wo_ks;
      pp.x = ws.x;
      pp.ld_out = H;
      OP_TRY(panel(PK_GEMM_ATTN_OUT, PE_RESIDUAL, pp, H / 256));
      OP_TRY(layer_norm_fp(lw.mlp_norm));
      pp.a_fp = ws.ln_hi;
      pp.wp = lw.wi_pk;
      pp.o0 = ws.h_hi;
      pp.ld_out = I;
      OP_TRY(panel(PK_GEMM_WI_GEGLU, PE_GEGLU, pp, I / 128));
      pp.a_fp = ws.h_hi;
      pp.n_ksteps = I / 32;
      pp.wp = lw.wo2_pk;
      pp.ld_out = H;
      OP_TRY(panel(PK_GEMM_MLP_OUT, PE_RESIDUAL, pp, H / 256));
      continue;
    }

    // ---- tiled path (any hidden % 128 == 0): separate LayerNorm kernels, 128 x 128 x 32 tiles, row-major planes ----
    if (li != 0) OP_TRY(layer_norm(lw.attn_norm));
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
    OP_TRY(layer_norm(lw.mlp_norm));
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
    p.a_hi = ws.h_hi;
    p.a_lo = ws.h_lo;
    p.w_hi = lw.wo2_hi;
    p.w_lo = lw.wo2_lo;
    p.K = I;
    p.n_tiles = H / GEMM_BN;
    p.x = ws.x;
    p.ld_out = H;
    OP_TRY(launch_gemm<EPI_RESIDUAL>(L, PK_GEMM_MLP_OUT, p, split));
  }

  const int mean_pool = h->cfg.pooling == OP_POOL_MEAN ? 1 : 0;
  OP_TRY(L.begin(PK_FINAL_LN_PRUNE));
  hipLaunchKernelGGL(final_ln_prune_kernel, dim3(row_blocks), dim3(256), 0, st, ws.x, h->final_norm, h->cfg.norm_eps, H,
                     r_pad, ws.row_tok, ws.row_seq, ws.row_pos, h->prune_w, h->prune_b, prune_out, mean_pool, ws.cls,
                     h->capture ? h->capture + (size_t)h->N * total_tokens * H : nullptr);
  OP_TRY(L.end());
  OP_TRY(L.begin(PK_RANK_HEAD));
  hipLaunchKernelGGL(rank_head_kernel, dim3((unsigned)ns), dim3(256), 0, st, ws.cls, ws.x, cu_dev, s0, ws.roff, mean_pool,
                     H, h->nl, h->dense_t, h->head_norm, h->cfg.norm_eps, h->cls_w, h->cls_b, rank_out);
  OP_TRY(L.end());
  return OP_OK;
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
  if (cfg->precision != OP_PRECISION_BF16X3 && cfg->precision != OP_PRECISION_BF16)
    return fail(nullptr, OP_ERR_INVALID, "unknown precision %d", cfg->precision);
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
  h->split = cfg->precision == OP_PRECISION_BF16X3;
  h->chunk_rows = cfg->chunk_rows > 0 ? align_up(cfg->chunk_rows, ROW_ALIGN) : 262144;  // grids must cover the chip several times over
  h->layers.resize(N);
  // H % 64 and I % 32: every row-GEMM streams an EVEN number of 32-feature chunks on each side of the q/k -> v
  // boundary (H/16 q/k chunks, H/32 v chunks, H/32 out-projection chunks, I/16 Wi chunks) -- rowgemm_kernel's
  // two-stage loop is unrolled by two.
  const bool force_tiled = getenv("OPEN_PROVENCE_FORCE_TILED") != nullptr;
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
      if (h->row_path) OP_CREATE_TRY(dev_alloc(h, &lw.wo_pk, 2 * HH));
      OP_CREATE_TRY(dev_alloc(h, &lw.wi_pk, (size_t)2 * 2 * I * H));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo2_pk, (size_t)2 * H * I));
      OP_CREATE_TRY(dev_alloc(h, &lw.wo_ks, 2 * HH));
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
  int pk_mode = -1;
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
      dst_pk = lw.wqkv_pk; pk_mode = RE_QKV;
    } else if (t == "attn.Wo.weight") {
      kind = PLANES; dst_hi = lw.wo_hi; dst_lo = lw.wo_lo; expect(H, H);
      dst_pk = lw.wo_pk; pk_mode = RE_RESIDUAL; dst_ks = lw.wo_ks;
    } else if (t == "mlp.Wi.weight") {
      kind = PLANES_GEGLU; dst_hi = lw.wi_hi; dst_lo = lw.wi_lo; expect(2 * I, H);
      dst_pk = lw.wi_pk; pk_mode = RE_GEGLU;
    } else if (t == "mlp.Wo.weight") {
      kind = PLANES; dst_hi = lw.wo2_hi; dst_lo = lw.wo2_lo; expect(H, I);
      dst_pk = lw.wo2_pk; pk_mode = 100;  // k-streamed
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
  switch (kind) {
    case F32_COPY:
      e = hipMemcpyAsync(dst_f32, f32, count * sizeof(float), hipMemcpyDeviceToDevice, 0);
      break;
    case F32_TRANSPOSE:
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, dst_f32);
      break;
    case PLANES:
      if (dst_hi)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 0, dst_hi, dst_lo);
      break;
    case PLANES_GEGLU:
      if (dst_hi)
        hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, I, dst_hi, dst_lo);
      break;
  }
  if (h->panel_path && (dst_pk || dst_ks)) {
    // panel-major packing: [panel][k-step][plane][16 fragments][512], rows permuted per consumer (panel_source_row)
    const int K = (int)d1;
    auto pack = [&](int n_tiles, int mode, u16* dst) {
      const size_t total = (size_t)n_tiles * 256 * K;
      hipLaunchKernelGGL(pack_panel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, f32, n_tiles, K, mode,
                         H, I, dst);
    };
    if (pk_mode == RE_QKV) {
      pack(2 * H / 256, PE_QK, dst_pk);
      pack(H / 256, PE_V, dst_pk + (size_t)(2 * H / 256) * (K / 32) * 2 * 8192);
    } else if (pk_mode == RE_RESIDUAL) {
      pack(H / 256, PE_RESIDUAL, dst_ks);  // attention output projection
    } else if (pk_mode == RE_GEGLU) {
      pack(I / 128, PE_GEGLU, dst_pk);
    } else {
      pack(H / 256, PE_RESIDUAL, dst_pk);  // MLP output projection
    }
  }
  if (dst_pk && h->row_path) {
    if (pk_mode == 100)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_pk);
    else
      hipLaunchKernelGGL(pack_rowgemm_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, pk_mode, H, I, dst_pk);
    if (dst_ks)
      hipLaunchKernelGGL(pack_kstream_kernel, dim3(blocks), dim3(256), 0, 0, f32, (int)d0, (int)d1, 1, dst_ks);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(0);
  (void)hipFree(raw);
  (void)hipFree(f32);
  if (e != hipSuccess) return fail(h, OP_ERR_HIP, "op_load_weight(%s): %s", name_c, hipGetErrorString(e));
  erase_missing(h, name);
  return OP_OK;
}

int op_weights_ready(op_handle* h) {
  if (!h) return fail(nullptr, OP_ERR_INVALID, "op_weights_ready: NULL handle");
  if (h->missing.empty()) return OP_OK;
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
                      int total_tokens, int max_seqlen, float* prune_out, float* rank_out, void* workspace,
                      size_t workspace_bytes, void* hip_stream) {
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
    // attention work items: 256-query blocks (8 waves) once a sequence is longer than 128 tokens, else 128
    // -- unless that leaves CUs idle (small request): then 128-query blocks, twice as many work items
    const char* aw = getenv("OPEN_PROVENCE_ATT_WAVES");
    auto count_items = [&](int waves) {
      int items = 0;
      for (int s = s0; s < s1; ++s) items += (cu[s + 1] - cu[s] + waves * 32 - 1) / (waves * 32);
      return items;
    };
    int att_waves = aw ? (atoi(aw) == 4 ? 4 : 8) : (max_len > 128 ? 8 : 4);
    int att_items = count_items(att_waves);
    if (!aw && att_waves == 8 && (long)att_items * h->nh <= h->n_cus) {
      att_waves = 4;
      att_items = count_items(4);
    }
    if (rows > 0) {
      OP_TRY(forward_chunk(h, L, ws, ids_dev, cu_dev, s0, s1 - s0, rows, max_len, total_tokens, att_waves, att_items,
                           prune_out, rank_out));
    } else {
      // only empty sequences in this chunk
      OP_HIP(h, hipMemsetAsync(rank_out + (size_t)s0 * h->nl, 0, (size_t)(s1 - s0) * h->nl * sizeof(float), stream));
    }
    s0 = s1;
  }
  return OP_OK;
}

}  // extern "C"
