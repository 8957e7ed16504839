// opk_rowgemm_pack.hip.h -- row-stationary GEMMs (hidden <= 256): launch parameters, weight row order, pack kernels
// (first part of what opk_rowgemm.hip.h provides; include that one)
#pragma once

#include <type_traits>
#include <utility>

#include "opk_common.hip.h"

// Measurement hook kept in this header: OPK_TIMING / OPK_SEG_TIMING -- cycle stamps of wave 0 at the phase boundaries
// (microbench/rowgemm_ablate.hip compiles the header with them; the library never defines them).  The ablation switches
// that price one component of a loop with wrong results (no DMA, no barrier, no GeGLU arithmetic, the q / k / v^T pair
// stream's parts, one chunk per barrier) are NOT here: microbench/experiments/rowgemm_ablation_hooks.patch adds them to a
// scratch copy of csrc/ (scripts/ablate_x.sh); their numbers are in DESIGN.md section 4 and profiles/r0*.
namespace opk {

// ----------------------------------------------------------------------------------------------
// Row-stationary GEMM for K = hidden <= 256 (the three projections whose input is the hidden state):
//   C[m, n] = sum_k A[m, k] W[n, k],  A = 128 rows per block kept IN REGISTERS as MFMA fragments
//   (each wave owns 32 rows = 2 fragments x K/32 k-steps x hi/lo), W streamed through LDS in chunks of
//   32 output features, double-buffered, one barrier per chunk (96 MFMAs per wave between barriers).
// Why: the activations are the big operand (read exactly once, never staged through LDS); the weights
// are small, L2-resident and STATIC, so they are pre-packed at load time in exactly the order the MFMA
// X/Y fragments want them ([chunk][k-step][plane][frag][k-group][row][8]) -- every LDS fragment read
// is a lane-linear, conflict-free 1 KiB ds_read_b128, every global->LDS copy a linear memcpy.
// Prologues fuse what used to be separate kernels: LayerNorm (+ hi/lo split) of the fp32 residual
// stream is computed in registers directly in fragment layout (a row lives in 4 lanes).
// ----------------------------------------------------------------------------------------------

struct RowGemmParams {
#ifdef OPK_TIMING
  unsigned long long* dbg;  // [blocks][16] cycle stamps of wave 0 (microbench/rowgemm_ablate.hip only)
#endif
  const float* x_in;  // RP_SPLIT: fp32 [r_pad][K]
  const float* ln_w;  // LayerNorm weight in front of the chunk loop (RP_KSTREAM, RP_MLP)
  float eps;
  const u16* wp;  // packed weights of the chunk loop, n_chunks x (K/32) x 2 planes x 2 frags x 512 elements
  int n_chunks;
  int n_swapped;  // RE_QKV: chunks [0, n_swapped) are q/k (RoPE), the rest v (transposed store)
  u16* o0_hi;     // RE_QKV: q   RE_GEGLU: h      (fragment-packed, hi / lo planes interleaved per piece)
  u16* o1_hi;     // RE_QKV: k
  u16* o2_hi;     // RE_QKV: v^T pieces [head][r_pad/32][plane][4][512]
  int ld_out;     // RE_GEGLU: I   RE_QKV: H
  int hidden;
  int r_pad;
  const int32_t* row_pos;
  const float* rope_cos;
  const float* rope_sin;
  int max_pos;
  // RP_KSTREAM / RP_MLP: x_new = x + A1 W1^T first, A1 fragment-packed [r_pad/16][k1_steps][2][512], W1 packed
  // by pack_kstream_kernel with permuted output features; then LayerNorm(x_new) feeds the next GEMM.
  const u16* a1_fp;
  const u16* w1p;
  int k1_steps;
  // F8 kernel set: a1_fp = fp16 pieces [r_pad/16][k1_steps][512] (no plane interleave), a1_lo8 = e4m3 pieces
  // [r_pad/16][k1_steps/2][512] (one 1 KiB half-fragment per head, opk_common.hip.h); w1p = fp16 slabs [k-step][NF1][512],
  // w1p8 = e4m3 slabs [K-step of 128][NF1][2 halves][512]; wi_pk / wp = chunks of [fp16 plane | e4m3 plane] pieces
  // (pack_rowgemm_f8_kernel), wo2_ks = fp16 slabs [k-step][NF1][512]
  const u16* a1_lo8;
  const u16* w1p8;
  // F8 = 2: the e4m3 slabs of lo(w) of the attention output weight are at w1p8 + k1_steps * 32 * hidden / 2; wo2_ks then
  // holds [k-step][plane][NF1][512] with plane 1 = fp16(lo(w)); F8 = 1 reads plane 0 of the same pack
  float* x_io;
  int zero_a_lo;  // clear the lo fragments of the in-register (LayerNorm / split) operand: see rowgemm_kernel
  // RP_MLP: the whole MLP between phase 1 and the chunk loop, h kept on chip
  const float* ln_w_mlp;  // this layer's mlp_norm weight (ln_w is then the NEXT layer's attn_norm)
  const u16* wi_pk;       // Wi, chunk-major pack (as wp of the RE_GEGLU kernel)
  const u16* wo2_ks;      // MLP Wo, k-streamed pack (as w1p of the RE_QKV / RP_KSTREAM kernel)
  int n_pairs;            // intermediate / 32 (even)
  // RP_MLP + RE_NONE (the last layer), optional (fin_ln != nullptr): the rows this launch ends with go straight through
  // final_norm and the pruning head (final_ln_prune_kernel's work: logits per token, keep-probability, the normalised
  // CLS row of every sequence for the ranking head) and are NOT written back -- the residual stream's last round trip
  // (4 H bytes per token each way) and one launch are gone.
  const float* fin_ln;       // final_norm weight [H]
  const float* fin_pw;       // pruning head weight [2][H]
  const float* fin_pb;       // pruning head bias [2]
  const int32_t* row_tok;    // packed row -> token index (< 0: alignment row)
  const int32_t* row_seq;    // packed row -> sequence index
  float* fin_prune;          // [tokens][2]
  float* fin_keep;           // [tokens] or nullptr
  float* fin_cls;            // [sequences][H]: normalised row of position 0
  int fin_pre_norm;          // the head reads the row before final_norm (transformers 4.x convention)
  // RP_SPLIT (layer 0), optional (emb_table != nullptr): the rows are built here -- embedding gather + embeddings.norm
  // (embed_ln_kernel's work, HF :52-71) -- written to x_io once and split straight into the q / k / v operand: the
  // residual stream is not read back and the embedding launch (with its unused row-major operand planes) is gone.
  // ln_w = embeddings.norm weight, row_tok as above.
  const float* emb_table;    // [vocab][H] fp32
  const int32_t* emb_ids;    // [tokens]
  int emb_vocab;
};

// source row of packed row `pr` (0..31) of chunk `c`
__device__ __forceinline__ int rowgemm_source_row(int mode, int c, int pr, int H, int I) {
  const int nf = pr >> 4, i = pr & 15;
  if (mode == RE_QKV) {
    const int per_block = H / ROW_CHUNK;  // chunks in each of q, k, v
    const int blk = c / per_block, cc = c % per_block;
    const int head = cc >> 1, j = cc & 1;
    // q / k: the chunk pair (j = 0, 1) of a head leaves lane slot i = 4g + r with d = 8g + 4j + r (fragment 0)
    // and its RoPE partner d + 32 (fragment 1): after the pair a lane owns 8 consecutive d of both k-steps.
    if (blk < 2) return blk * H + head * HEAD_DIM + 32 * nf + 8 * (i >> 2) + 4 * j + (i & 3);
    // v: fragment nf of chunk j becomes piece n = 2j + nf of the transposed layout, whose row i is
    // d = 32j + 8(i>>2) + 4nf + (i&3) -- the order that makes the attention output lane-contiguous.
    return 2 * H + head * HEAD_DIM + 32 * j + 8 * (i >> 2) + 4 * nf + (i & 3);
  }
  if (mode == RE_GEGLU) {  // chunk pair (2t, 2t+1): lane slot i = 4g + r -> h-column 32t + 8g + 4u + r
    const int col = 32 * (c >> 1) + 8 * (i >> 2) + 4 * (c & 1) + (i & 3);
    return nf == 0 ? col : I + col;  // input column | matching gate column
  }
  return c * ROW_CHUNK + pr;
}

#ifdef OPK_PACK_KERNELS  // weight re-packing runs in op_api.hip only
// dst[chunk][ks][plane][nf][g][i][e] <- src[source_row(chunk, nf*16+i)][ks*32 + g*8 + e]
// f16 = 1 (kernel set "f16"): the hi plane holds RNE_fp16(w), the lo plane zeros; any_lo is not touched
__global__ void pack_rowgemm_kernel(const float* __restrict__ src, int n_rows, int K, int mode, int H, int I,
                                    u16* __restrict__ dst, int zero_lo, int* __restrict__ any_lo, int f16 = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * K;
  if (idx >= total) return;
  const int KS = K / 32;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t & 1); t >>= 1;
  const int ks = (int)(t % KS);
  const int c = (int)(t / KS);
  const int srow = rowgemm_source_row(mode, c, nf * 16 + i, H, I);
  const float v = src[(size_t)srow * K + ks * 32 + g * 8 + e];
  const size_t base = (((size_t)c * KS + ks) * 2) * 1024 + (size_t)nf * 512 + (size_t)g * 128 + i * 8 + e;
  if (f16) {
    dst[base] = f2h(v);
    dst[base + 1024] = (u16)0;
    return;
  }
  const u16 h = f2bf(v);
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  dst[base] = h;
  dst[base + 1024] = zero_lo ? (u16)0 : l;
}

// "f16 + fp8" kernel sets: chunk c = [fp16 pieces (ks, nf)][e4m3 pieces (nf, K-step S, half hh)][e4m3 pieces of the
// WEIGHT's lo part lo(w) = (w - fp16(w)) x 2^12, same order], 1 KiB each (F8Chunk
// below); the same source-row permutations as pack_rowgemm_kernel.  *not_f16 is raised when a weight of magnitude
// >= 2^-14 is not exactly an fp16 value (then this kernel set would drop bits of the weight and the library keeps the
// bf16 sets; every bf16 value in [2^-14, 65504] is an fp16 value).  Smaller weights land on the fp16 subnormal grid:
// absolute error <= 2^-25 per weight, ~1e-6 on a logit.
__global__ void pack_rowgemm_f8_kernel(const float* __restrict__ src, int n_rows, int K, int mode, int H, int I,
                                       u16* __restrict__ dst, int round_bf16, int* __restrict__ not_f16, float* __restrict__ fit) {
  set_saturating_conversions();
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * K;
  if (idx >= total) return;
  const int KS = K / 32, NS8 = KS / 4, CP = 2 * KS + 8 * NS8;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t & 1); t >>= 1;
  const int ks = (int)(t % KS);
  const int c = (int)(t / KS);
  const int srow = rowgemm_source_row(mode, c, nf * 16 + i, H, I);
  float v = src[(size_t)srow * K + ks * 32 + g * 8 + e];
  if (round_bf16) v = bf2f(f2bf(v));  // a policy without the hi x lo(weight) term: the weight IS its bf16 rounding
  const _Float16 hv = (_Float16)v;
  note_f16_fit(v, hv, not_f16, fit);
  dst[((size_t)c * CP + ks * 2 + nf) * 512 + g * 128 + i * 8 + e] = __builtin_bit_cast(u16, hv);
  const int s8 = ks >> 2, hh = (ks & 3) >> 1, pbyte = 8 * (ks & 1) + e;
  unsigned char* d8 = reinterpret_cast<unsigned char*>(dst + ((size_t)c * CP + 2 * KS + (nf * NS8 + s8) * 2 + hh) * 512);
  d8[(g * 16 + i) * 16 + pbyte] = f2e4m3(v * (float)(1 << F8_W_SHIFT));
  d8[(size_t)4 * NS8 * 1024 + (g * 16 + i) * 16 + pbyte] = f2e4m3((v - (float)hv) * (float)(1 << (F8_LO_SHIFT + F8_W_SHIFT)));
}

// k-streamed weights of the "f16 + fp8" kernel sets.  dst8 != nullptr (attention output projection, K = hidden): fp16
// slabs dst16[ks][nf][512], e4m3 slabs dst8[K-step S][nf][half][1 KiB] and the same of lo(w) x 2^12 behind them
// (dst8 + N K / 2).  dst8 == nullptr (MLP output projection, streamed 32 k at a time): dst16[ks][plane][nf][512] with
// plane 0 = fp16(w), plane 1 = fp16(lo(w)), UNSCALED: lo(w) ~ 2^-12 |w| sits in fp16's subnormal range for |w| < 0.25,
// where the grid is 2^-24 -- an absolute error <= 2^-25 per weight, the bound this format accepts for small weights
// anyway -- and the MFMA takes subnormal operands at full rate (default denormal mode); it multiplies the fp16 hi
// fragment of h the main product uses.  `permute` as pack_kstream_kernel.
__global__ void pack_kstream_f8_kernel(const float* __restrict__ src, int N, int K, int permute, u16* __restrict__ dst16,
                                       u16* __restrict__ dst8, int round_bf16, int* __restrict__ not_f16, float* __restrict__ fit) {
  set_saturating_conversions();
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * K) return;
  const int NF = N / 16;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t % NF);
  const int ks = (int)(t / NF);
  const int row = permute ? (32 * (nf >> 1) + 8 * (i >> 2) + 4 * (nf & 1) + (i & 3)) : (nf * 16 + i);
  float v = src[(size_t)row * K + ks * 32 + g * 8 + e];
  if (round_bf16) v = bf2f(f2bf(v));
  const _Float16 hv = (_Float16)v;
  note_f16_fit(v, hv, not_f16, fit);
  const float wlo = v - (float)hv;
  if (dst8 != nullptr) {
    dst16[((size_t)ks * NF + nf) * 512 + g * 128 + i * 8 + e] = __builtin_bit_cast(u16, hv);
    const int s8 = ks >> 2, hh = (ks & 3) >> 1, pbyte = 8 * (ks & 1) + e;
    unsigned char* d8 = reinterpret_cast<unsigned char*>(dst8 + (((size_t)s8 * NF + nf) * 2 + hh) * 512);
    d8[(g * 16 + i) * 16 + pbyte] = f2e4m3(v * (float)(1 << F8_W_SHIFT));
    d8[(size_t)N * K + (g * 16 + i) * 16 + pbyte] = f2e4m3(wlo * (float)(1 << (F8_LO_SHIFT + F8_W_SHIFT)));
  } else {
    dst16[((size_t)ks * 2 * NF + nf) * 512 + g * 128 + i * 8 + e] = __builtin_bit_cast(u16, hv);
    dst16[((size_t)(ks * 2 + 1) * NF + nf) * 512 + g * 128 + i * 8 + e] = f2h(wlo);
  }
}
#endif


}  // namespace opk
