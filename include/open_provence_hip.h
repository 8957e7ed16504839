/*
 * open_provence_hip.h -- C ABI of libopenprovence_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of hotchpotch/open_provence: the batched (query, context) cross-encoder
 * forward.  Each entry point names the reference interface it replaces (paths relative to the
 * reference tree; "standalone.py" = open_provence/modeling_open_provence_standalone.py; "HF" = the
 * third-party transformers ModernBERT the reference instantiates at standalone.py:1341).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns OP_OK (0) or a negative OP_ERR_* code and never throws;
 *     the message is available from op_last_error(handle) (or op_last_error(NULL) when no
 *     handle exists yet);
 *   - one handle per device; calls on one handle must be serialised by the caller (host side); the
 *     forward is asynchronous with respect to the given HIP stream, and forwards enqueued on DIFFERENT
 *     streams with DIFFERENT workspaces may run side by side on the device (the weights are read-only;
 *     profiling and hidden-state capture off) -- HipEncoder.forward_packed_on does that on two streams
 *     bound to disjoint halves of the CUs;
 *   - the caller owns every I/O buffer and the workspace; the library owns only its re-packed weights.
 */
#ifndef OPEN_PROVENCE_HIP_H
#define OPEN_PROVENCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OP_ABI_VERSION 9 /* 9: kernel sets 8 / 9 layer by layer (op_calibration.mlp_layers / mlp_layers_err, OP_CAL_WHOLE_DEPTH, op_mlp_correction_layers, op_select_mlp_correction_layers); 8: op_calibration.flags (OP_CAL_FULL_REPORT; the search stops at the first candidate that holds otherwise), op_calibrate validates its batch before it touches the handle, op_load_weight of a GEMM weight drops a pinned / calibrated kernel set; 7: kernel sets 10 / 11 (fp16 attention inside the fp16 + e4m3 sets), op_calibration holds 16 candidates; 6: op_select_kernel_set, op_calibrate, kernel set 7 ("f16": single-pass fp16 operands); 5: op_set_compact_operands (run-time fallback to the (hi, lo) bf16 kernel sets); 4: kernel set 3 (fp16 + e4m3 operands), flag NO_F8; 3: op_segment_means, flags LAYER_M32 / NO_HEAD_FUSION (struct layouts as in 2) */
#define OP_MAX_LAYERS 128

typedef struct op_handle op_handle;

enum op_status {
  OP_OK = 0,
  OP_ERR_INVALID = -1,     /* bad argument                                   */
  OP_ERR_UNSUPPORTED = -2, /* configuration the kernels do not implement     */
  OP_ERR_HIP = -3,         /* HIP runtime failure (message has the HIP text) */
  OP_ERR_STATE = -4,       /* call order (e.g. forward before all weights)   */
  OP_ERR_WORKSPACE = -5,   /* workspace too small / misaligned               */
  OP_ERR_NOMEM = -6
};

enum op_dtype { OP_DTYPE_F32 = 0, OP_DTYPE_BF16 = 1, OP_DTYPE_F16 = 2 };

/* Arithmetic of the MFMA contractions (accumulation is always fp32; LayerNorm, softmax, GELU,
 * RoPE and the residual stream are always fp32).  Every operand can be carried as a (hi, lo) bf16
 * pair (hi = RNE(v), lo = RNE(v - hi), ~16 mantissa bits together); a contraction left x right then
 * evaluates hi*hi plus the optional terms of its mask:
 *   OP_TERM_LEFT_LO   lo(left) * hi(right)        OP_TERM_RIGHT_LO   hi(left) * lo(right)
 * "left" is the activation-side operand.  One mask per contraction family (enum op_gemm_family):
 *   OP_PRECISION_BF16X3  all masks 3: the mode that meets the 1e-3 parity bar against the fp32 CPU
 *                        reference for any checkpoint;
 *   OP_PRECISION_BF16X2  weights rounded to bf16 (no hi*lo(weight) term in the four weight GEMMs),
 *                        activations (hi, lo), attention all terms;
 *   OP_PRECISION_BF16    all masks 0: single-pass bf16 operands (what the reference itself computes
 *                        with on a GPU: standalone.py:219-233 picks bf16), ~1e-2 on logits;
 *   OP_PRECISION_CUSTOM  op_config.terms[] as given.
 * Independently of the mode, the hi*lo(weight) term of a family is dropped -- without changing a bit
 * of the result -- when every lo element of every weight of that family is zero, i.e. for bf16
 * checkpoints (detected in op_load_weight, applied in op_weights_ready; see op_effective_policy). */
enum op_precision { OP_PRECISION_BF16X3 = 0, OP_PRECISION_BF16 = 1, OP_PRECISION_BF16X2 = 2, OP_PRECISION_CUSTOM = 3 };
enum op_term { OP_TERM_LEFT_LO = 1, OP_TERM_RIGHT_LO = 2 };
enum op_gemm_family {
  OP_FAM_WQKV = 0,     /* LN(x) x Wqkv        HF :272      */
  OP_FAM_QK = 1,       /* q x k               HF :166-185  */
  OP_FAM_PV = 2,       /* softmax(..) x v                  */
  OP_FAM_ATTN_OUT = 3, /* attention out x Wo  HF :299      */
  OP_FAM_WI = 4,       /* LN(x) x Wi          HF :90       */
  OP_FAM_MLP_OUT = 5,  /* GeGLU out x Wo      HF :91       */
  OP_FAM_COUNT = 6
};

/* op_config.flags */
enum op_flags {
  OP_FLAG_FORCE_TILED = 1,      /* generic 128x128x32 tiled kernels for every shape (test hook)             */
  OP_FLAG_NO_SMALL_BLOCKS = 2,  /* never use the 64-row GEMM blocks of the latency regime (measurement hook) */
  OP_FLAG_ATT_WAVES_4 = 4,      /* 128-query attention blocks for full-attention layers (measurement hook)   */
  OP_FLAG_ATT_WAVES_8 = 8,      /* 256-query attention blocks for full-attention layers (measurement hook)   */
  OP_FLAG_NO_POLICY_KERNELS = 16, /* always run the all-terms kernels with cleared lo operands (test hook)   */
  OP_FLAG_NO_LAYER_FUSION = 32,   /* two fused kernels per layer instead of the whole-layer kernel (A/B hook) */
  OP_FLAG_LAYER_8X16 = 64,        /* whole-layer kernel as 8 waves x 16 rows (two waves per SIMD) (A/B hook)   */
  OP_FLAG_NO_HEAD_FUSION = 256,   /* embedding + LayerNorm and final_norm + pruning head as their own launches on the row path too (A/B hook) */
  OP_FLAG_LAYER_M32 = 128,        /* whole-layer kernel on 32x32x16 MFMAs (hidden = 256): fewer cycles, more power per flop -- slower under the power limit (A/B hook) */
  OP_FLAG_ATTN_XCD_GROUP = 1024,  /* row path too: the XCD-grouped attention block map (the query blocks of a sequence-head follow each other on one XCD; default on the panel path only) (A/B hook) */
  OP_FLAG_PANEL_F8 = 2048,        /* hidden 512 / 768 (panel GEMMs): select the fp16 + e4m3 kernel sets there too.  OFF by default: through 19-25 layers their error against the fp32 reference reaches 0.45-1.0e-3 on logits (the (hi, lo) bf16 sets: 0.2-0.5e-3), too close to the 1e-3 bar of the path for +2.5 % (bf16 checkpoint) / +14 % (fp32) pairs/s.  On this opt-in path the MLP activation h is an fp16 operand converted with saturation: a value beyond 65504 is CLAMPED, not reported (the row path's sets turn it into NaN and fall back; the default panel sets keep h as (hi, lo) bf16 pairs with fp32's range) */
  OP_FLAG_PANEL_F8_WI = 4096,     /* hidden 512 / 768: the fp16 + e4m3 format in the Wi GEMM alone (52 % of the GEMM FLOPs; LayerNorm(mlp_norm) written in that format, h still as (hi, lo) bf16 pieces for the MLP output projection): the cheapest place for the format's error.  DEFAULT for fp32-valued weights (+5.6 % pairs/s on base, <= 5.4e-4 at 19-25 layers); this flag requests it for bf16-valued weights too (+0.7 %).  OP_FLAG_NO_F8 switches it off */
  OP_FLAG_NO_LAYER_PAIRS = 8192,  /* hidden = 256, single-pass kernel sets ("f16" / "bf16"): keep the 8 waves x 16 rows whole-layer kernel instead of the wave-pair kernel (opk_layer16p.hip.h) (A/B and test hook) */
  OP_FLAG_NO_F8 = 512             /* never select kernel set 3 (fp16 hi + e4m3 lo operands in the whole-layer kernel): keep the (hi, lo) bf16 kernel sets (A/B and bit-identity test hook) */
};

enum op_pooling { OP_POOL_CLS = 0, OP_POOL_MEAN = 1 };

/* Mirrors the fields of HF ModernBertConfig that change the arithmetic
 * (transformers/models/modernbert/configuration_modernbert.py:77-162) plus
 * OpenProvenceConfig.num_labels (standalone.py:1279). */
typedef struct op_config {
  uint32_t struct_bytes; /* = sizeof(op_config), checked */
  int32_t device_id;
  int32_t vocab_size;
  int32_t hidden_size;
  int32_t intermediate_size;
  int32_t num_layers;
  int32_t num_heads; /* head_dim = hidden_size / num_heads must be 64 */
  int32_t num_labels;
  int32_t local_attention; /* full window; keys with |q-k| <= local_attention/2 are visible */
  int32_t max_position_embeddings;
  int32_t pooling;   /* enum op_pooling */
  int32_t precision; /* enum op_precision */
  float norm_eps;
  float global_rope_theta;
  float local_rope_theta;
  int32_t chunk_rows; /* rows of the packed batch processed per pass (0 = library default) */
  uint8_t layer_is_global[OP_MAX_LAYERS]; /* 1 = full attention, 0 = sliding window */
  uint8_t terms[8];   /* OP_PRECISION_CUSTOM: term mask per op_gemm_family (entries >= OP_FAM_COUNT unused) */
  uint32_t flags;     /* enum op_flags */
  /* Which hidden state feeds the pruning head (standalone.py:1695 takes outputs.hidden_states[-1]):
   * 0 = the final_norm output (transformers >= 5 ties hidden_states[-1] to last_hidden_state,
   * utils/output_capturing.py:269-277 -- what the reference computes in this image), 1 = the last
   * layer's output BEFORE final_norm (what transformers 4.x ModernBertModel.forward appended, i.e.
   * what the reference computes under its own uv.lock pin 4.57.1).  The rank head always sees the
   * normalised state. */
  int32_t prune_pre_final_norm;
} op_config;

int op_abi_version(void);

/* Replaces: OpenProvencePreTrainedModel.__init__ building the backbone from
 * config.base_model_config (standalone.py:1340-1342, 1354-1375). */
int op_create(const op_config* cfg, op_handle** out);

/* Replaces: PreTrainedModel.load_state_dict on the reference's checkpoint keys
 * (standalone.py:1448-1464; key list in SURVEY.md section 8b).  `name` is the checkpoint key with or
 * without the "ranking_model." prefix (legacy checkpoints omit it).  `data` may be a host or a
 * device pointer; the tensor is copied and re-packed (bf16 hi/lo planes, GeGLU row interleave),
 * the caller keeps ownership of the source. */
int op_load_weight(op_handle* h, const char* name, const void* data, int dtype, const int64_t* shape, int ndim);

/* OP_OK when every tensor of the model has been loaded; otherwise OP_ERR_STATE and
 * op_last_error() lists the missing keys. */
int op_weights_ready(op_handle* h);

/* Bytes of caller-provided device workspace needed by op_forward_packed for a batch of `n_seqs`
 * sequences, `total_tokens` tokens, longest sequence `max_seqlen`. */
size_t op_workspace_bytes(const op_handle* h, int n_seqs, int total_tokens, int max_seqlen);

/* After op_weights_ready: the term masks actually evaluated (requested policy minus the weight-lo
 * terms that are identically zero for this checkpoint), terms_out[OP_FAM_COUNT], and *kernel_set =
 * index of the curated kernel set that implements them with exactly those MFMA passes (0 = all
 * terms, 1 = bf16 weights, 2 = single pass, 3 = the terms of 1 with the whole-layer kernel's
 * operands carried as fp16 hi + e4m3 lo -- 1.5 instead of 2 MFMA units per product, 4 = the terms
 * of 0 in that format with the weights' lo part as a second e4m3 plane -- 2 units instead of 3 and
 * one kernel per layer instead of two), or -1 when the policy runs on the all-terms kernels with
 * cleared lo operands (same numerics, no speed-up); 5 / 6 = sets 0 / 1 on the panel path with the Wi GEMM alone in the
 * format of sets 4 / 3 (OP_FLAG_PANEL_F8_WI); 7 = "f16", single-pass fp16 operands (only ever chosen by op_calibrate /
 * op_select_kernel_set: then terms_out holds that set's masks, not the checkpoint's).  Sets 3 / 4 replace 1 / 0 for hidden <= 256
 * (hidden 512 / 768: with OP_FLAG_PANEL_F8) unless OP_FLAG_NO_F8 is set, set 3 needs every GEMM
 * weight to be exactly an fp16 value, and neither is taken for a checkpoint with a weight TENSOR
 * scaled into fp16's subnormal range (checked at load time).  Their fp16 operand plane has fp16's
 * range: an MLP activation beyond it turns the outputs into NaN (on purpose: not clamped; on the panel path through a
 * range flag the kernels raise in the workspace and the head kernels turn into NaN logits -- under MODE.FP16_OVFL = 1 a
 * conversion clamps and the fp16 MFMA takes a NaN operand for a finite number). */
int op_effective_policy(op_handle* h, uint8_t* terms_out, int* kernel_set);

/* Kernel sets (the numbering of op_effective_policy).  MFMA pipe time per algorithmic product in 16-bit units:
 * BF16X3 3, BF16X3_WI_F8 ~2.5 (panel path), F16_F8_W 2, BF16_WEIGHTS 2, BF16_WEIGHTS_WI_F8 ~1.75, F16_F8 1.5, BF16 1,
 * F16 1.  F16 (round 5) = the kernels and layouts of BF16 with every operand carried as fp16 (11 significant bits
 * instead of 8; fp16's range: an activation beyond 65504 turns the outputs into NaN, as on sets 3 / 4). */
enum op_kernel_set {
  OP_KS_AUTO = -1, /* op_select_kernel_set: back to the default selection of op_weights_ready */
  OP_KS_BF16X3 = 0,
  OP_KS_BF16_WEIGHTS = 1,
  OP_KS_BF16 = 2,
  OP_KS_F16_F8 = 3,
  OP_KS_F16_F8_W = 4,
  OP_KS_BF16X3_WI_F8 = 5,
  OP_KS_BF16_WEIGHTS_WI_F8 = 6,
  OP_KS_F16 = 7,
  /* panel path (hidden 512 / 768) only: the attention side of a layer (q / k / v projection, attention, output projection)
   * on set 7, its MLP (LayerNorm(mlp_norm), Wi GEMM + GeGLU, MLP output projection) in the fp16 + e4m3 format of set 4
   * (F16_MLP_F8_W: the weights' lo part too) or of set 3 (F16_MLP_F8).  On weights of a trained checkpoint's scale the
   * MLP's two contractions carry ~6 x the error of the other four families together (scripts/family_error_probe.py), and
   * they are 75 % of the linear FLOPs: ~1.75 / ~1.4 MFMA units per product. */
  OP_KS_F16_MLP_F8_W = 8,
  OP_KS_F16_MLP_F8 = 9,
  /* panel path only: sets 4 / 3 with the ATTENTION itself (q x k, p x v) single pass on fp16 operands -- the q / k / v
   * projection writes single-plane fp16 q, k, v^T, attention runs set 7's kernels and writes o as fp16 + e4m3 pieces; the
   * four weight GEMMs keep every term of sets 4 / 3.  Attention is 18 - 21 % of large 64 x 2048 / en-gte varlen on sets
   * 4 / 3 (three MFMA passes per product), and q x k / p x v are the two families whose single-pass error is smallest
   * (scripts/family_error_probe.py). */
  OP_KS_F16_F8_W_ATTN_F16 = 10,
  OP_KS_F16_F8_ATTN_F16 = 11,
  OP_KS_COUNT = 12
};

/* Pin the kernel set the forward runs on (OP_KS_AUTO: un-pin).  A set with fewer product terms than the loaded
 * checkpoint carries (e.g. OP_KS_F16 on any checkpoint, OP_KS_F16_F8 on fp32-valued weights) is an APPROXIMATION of the
 * requested policy: op_calibrate is the call that measures it first.  OP_ERR_UNSUPPORTED when the handle cannot run the
 * set (shape without those kernels, packs not built because of OP_FLAG_NO_F8, a weight tensor below fp16's reach).
 * Replaces: the reference choosing its arithmetic from the device and what loads (standalone.py:219-244, 1589-1615). */
int op_select_kernel_set(op_handle* h, int kernel_set);

/* Choose the arithmetic from the checkpoint that is loaded.  The default selection of op_weights_ready is safe for ANY
 * weights (it only drops terms that are identically zero) -- and therefore priced for the worst case: weights of O(1)
 * scale, where every dropped term costs >= 7e-3 on a logit.  A trained checkpoint's weights are ~0.02: there most of
 * the correction terms are below the parity bar by orders of magnitude.  op_calibrate runs one batch (ids_host /
 * cu_seqlens_host / n_seqs: the caller's sample of real inputs; NULL / NULL / 0: 24 rows x min(512, max positions) + 14
 * ragged rows of uniform token ids) through the (hi, lo) bf16 realisation of the requested policy (`reference_set`) and
 * through every available kernel set CHEAPER than the default one, cheapest first, and keeps the first whose
 * max |logit difference| (pruning and ranking logits) to the reference outputs is <= tolerance and finite; if none is,
 * the default selection stays (unless its own outputs on that batch are not finite: then the reference set runs).  The north_star bar against the fp32 CPU reference is 1e-3; 1e-4 is the tolerance the
 * Python layer uses.  Synchronous (load-time call; its temporary device buffers are freed before it returns).
 * Replaces: standalone.py:219-244 / 1589-1615 / 1631-1642 (dtype and attention implementation chosen per device and
 * checkpoint, with a retry) -- the reference decides its arithmetic at load time too. */
typedef struct op_calibration {
  uint32_t struct_bytes; /* = sizeof(op_calibration), checked when a report is requested */
  float tolerance;
  int32_t reference_set; /* the (hi, lo) bf16 kernel set the candidates were compared with */
  int32_t default_set;   /* what op_weights_ready selects for this checkpoint */
  int32_t chosen_set;    /* what the forward runs on from now on */
  int32_t n_candidates;
  int32_t candidate_set[16];
  float candidate_err[16]; /* max |difference| to the reference outputs; +inf: non-finite outputs */
  int32_t n_rows, n_tokens; /* the calibration batch */
  float default_err;        /* the default set's own difference to the reference outputs on that batch.  It is not held to
                             * the tolerance (it is the set the parity tests stand on), but when it is NOT FINITE -- an
                             * activation beyond fp16's range on sets 3 - 6 -- or beyond 10 x tolerance, and no candidate
                             * passes, the reference set is chosen right away (ABI 8: the bound on a finite default_err) */
  uint32_t flags;           /* IN: OP_CAL_* bits */
  float mlp_layers_err;     /* chosen_set 8 / 9 with layers dropped: the difference of what runs now (0 when no layer was dropped) */
  uint64_t mlp_layers;      /* chosen_set 8 / 9 (the "f16" set with the MLP in the fp16 + e4m3 format): bit li = layer li keeps that
                             * MLP; in the other layers the batch stayed within the tolerance on the "f16" set's MLP (ABI 9).
                             * 0 for every other chosen_set */
} op_calibration;
#define OP_CAL_FULL_REPORT 1u /* measure every candidate (the report lists them all); default: cheapest first, stop at the first that holds */
#define OP_CAL_WHOLE_DEPTH 2u /* sets 8 / 9 for the whole depth or not at all (no per-layer search) */
int op_calibrate(op_handle* h, float tolerance, const int32_t* ids_host, const int32_t* cu_seqlens_host, int n_seqs,
                 op_calibration* report);

/* Kernel sets 8 / 9 layer by layer: bit li of *layer_mask = layer li runs its MLP in the fp16 + e4m3 format (0 when another
 * set runs).  op_select_mlp_correction_layers pins a mask on a handle whose set 8 / 9 was pinned by op_select_kernel_set or
 * chosen by op_calibrate (OP_ERR_STATE otherwise; at most 64 layers) -- what a process group uses to give every rank the
 * mask one rank measured.  Replaces: nothing (see op_calibrate). */
int op_mlp_correction_layers(op_handle* h, uint64_t* layer_mask);
int op_select_mlp_correction_layers(op_handle* h, uint64_t layer_mask);

/* Run-time switch between the fp16 + e4m3 kernel sets (3 / 4) and the (hi, lo) bf16 sets (1 / 0) they replace: both
 * weight packs of a handle stay resident, so this only re-runs the selection of op_weights_ready (with enabled = 0 as if
 * OP_FLAG_NO_F8 were set).  The Python layer calls it when a forward on sets 3 / 4 returns non-finite outputs -- an MLP
 * activation beyond fp16's range -- and repeats that forward, so that process() returns what the reference returns
 * instead of raising.  Replaces: nothing; precedent for a silent, correct retry: the reference's own fallback from an
 * unsupported dtype / attention implementation at load time (standalone.py:1631-1642).  Returns OP_OK; *changed (may
 * be NULL) = 1 when the evaluated kernel set is different afterwards (sets 5 / 6 and a pinned / calibrated set count:
 * enabled = 0 also drops a set chosen by op_select_kernel_set / op_calibrate).  The workspace size may change: query
 * op_workspace_bytes again. */
int op_set_compact_operands(op_handle* h, int enabled, int* changed);

/* Replaces: OpenProvenceModel.forward (standalone.py:1666-1739) = HF
 * ModernBertForSequenceClassification.forward + OpenProvenceHead.forward (standalone.py:434-448),
 * on the UNPADDED batch: ids_dev[total_tokens] are the attention_mask==1 tokens of all rows laid
 * end to end, cu_seqlens[n_seqs+1] their prefix offsets.  cu_seqlens_host may be NULL (the library
 * then reads cu_seqlens_dev back, which synchronises the stream once).
 *   prune_logits_dev [total_tokens, 2]  fp32   (pruning_logits at attention_mask==1 positions)
 *   rank_logits_dev  [n_seqs, num_labels] fp32 (ranking_logits)
 *   keep_prob_dev    [total_tokens] fp32 or NULL: softmax(pruning_logits, -1)[:, 1] evaluated as
 *                    sigmoid(l1 - l0) in the same kernel (replaces standalone.py:2918-2924)
 * Work is enqueued on hip_stream (a hipStream_t, NULL = default stream). */
int op_forward_packed(op_handle* h, const int32_t* ids_dev, const int32_t* cu_seqlens_dev,
                      const int32_t* cu_seqlens_host, int n_seqs, int total_tokens, int max_seqlen,
                      float* prune_logits_dev, float* rank_logits_dev, float* keep_prob_dev,
                      void* workspace_dev, size_t workspace_bytes, void* hip_stream);

/* Replaces: the per-fragment `float(block_probs[start:end].mean())` of the reference's post-processing
 * (standalone.py:3075-3082), evaluated on the device on the keep-probabilities a forward left there:
 *   seg_dev [n_seg, 2] int32  token ranges [start, end) into keep_prob_dev (clamped to [0, n_values))
 *   out_dev [n_seg]    fp32   mean of the range in numpy's float32 pairwise order, bit for bit; 1.0 for
 *                             an empty range (ref :3081)
 * so that process() copies back 4 bytes per fragment instead of 4 per token.  Enqueued on hip_stream. */
int op_segment_means(op_handle* h, const float* keep_prob_dev, int n_values, const int32_t* seg_dev, int n_seg,
                     float* out_dev, void* hip_stream);

/* Test hook.  Replaces: output_hidden_states=True of the reference forward (standalone.py:1689,
 * 1727).  When `hidden_dev` is non-NULL the next forwards also write the (num_layers+1) hidden
 * states, fp32 [num_layers+1, total_tokens, hidden]; entry num_layers is the post-final_norm
 * tensor, as in HF (transformers >= 5).  Pass NULL to switch capturing off. */
int op_debug_capture_hidden(op_handle* h, float* hidden_dev);

/* Measurement hook: when enabled every kernel launch of the forward is bracketed by HIP events
 * on the launch stream; op_profile_read returns accumulated milliseconds and launch counts per
 * kernel kind (names via op_profile_kind_name) and resets nothing; op_profile_reset clears. */
typedef struct op_profile_entry {
  int32_t kind;
  int32_t launches;
  double total_ms;
} op_profile_entry;
int op_profile_enable(op_handle* h, int enabled);
int op_profile_read(op_handle* h, op_profile_entry* entries, int max_entries); /* returns count, syncs */
int op_profile_reset(op_handle* h);
const char* op_profile_kind_name(int kind);

/* Measurement hook: one wave spins for `spin_us` microseconds (of the constant 100 MHz counter) on `hip_stream` --
 * a stream of its own, next to the timed work -- and writes out_dev[0] = shader cycles elapsed (s_memtime), out_dev[1] =
 * 100 MHz ticks elapsed (s_memrealtime): out[0] / out[1] / 10 = the shader clock in GHz the chip HELD while the timed
 * loop ran (the power limit, not the nominal 2.4 GHz, sets it: DESIGN.md section 4).  out_dev: two uint64. */
int op_debug_clock_probe(op_handle* h, int spin_us, unsigned long long* out_dev, void* hip_stream);

/* Replaces: nothing in the reference (it has no multi-GPU path); helper for SURVEY.md section 8e. */
int op_device_count(int* count);

void op_destroy(op_handle* h);
const char* op_last_error(const op_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* OPEN_PROVENCE_HIP_H */
