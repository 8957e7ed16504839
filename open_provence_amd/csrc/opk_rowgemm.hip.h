// opk_rowgemm.hip.h -- row-stationary GEMMs (hidden <= 256) and the k-streamed output projection
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

// (hand-placed LDS fragment reads, static_for and the fragment streams: opk_common.hip.h)

// LDS byte offsets of the fused MLP's macro-iteration stream inside one stage:
//   [Wi chunk 2t : KS k-steps x 2 fragments][Wi chunk 2t+1][MLP-Wo slab t-1 : NF1 fragments]
// steps 0..KS-1 = chunk 2t, then (SLAB) NF1/2 steps of the slab, then KS steps of chunk 2t+1.
template <int KS, int NF1, bool SLAB>
struct MlpStreamOff {
  static constexpr int WI = KS * 2048;
  static constexpr int NS = SLAB ? NF1 / 2 : 0;
  static constexpr int at(int s, int j) {
    if (s < KS) return s * 2048 + j * 1024;
    if (s < KS + NS) return 2 * WI + ((s - KS) * 2 + j) * 1024;
    return WI + (s - KS - NS) * 2048 + j * 1024;
  }
};

// One weight chunk (32 output features x K) against this wave's 32 rows: 2 x 2 accumulators, K/32 k-steps of
// 4 MFMAs per product term, weight fragments two k-steps deep in registers (see above).  `lds_addr` = LDS byte
// address of this lane's 16 bytes in piece 0 of stage 0; STAGE_BYTES = compile-time offset of the stage to read.
template <int KS, int MF, int T, bool SWAPPED, int STAGE_BYTES, bool PIN_AGPR = false, int DEPTH = 1, bool H16 = false>
__device__ __forceinline__ void rowgemm_chunk_mfma(uint32_t lds_addr, const bf16x8 (&a_hi)[MF][KS],
                                                   const bf16x8 (&a_lo)[MF][KS], f32x4 (&acc)[2][MF]) {
  constexpr bool W_LO = (T & T_RIGHT_LO) != 0, A_LO = (T & T_LEFT_LO) != 0;
  constexpr int PLANES = W_LO ? 2 : 1;
  constexpr int STEP_DS = 2 * PLANES;  // fragment reads per k-step
  // DEPTH k-steps of fragment reads stay in flight behind the one being multiplied, in DEPTH + 1 rotating register
  // sets.  DEPTH = 1 everywhere: 2 measured no faster in the one-wave-per-SIMD layer kernel (48.0 k vs 49.1 k cycles
  // for its q / k / v^T loop) -- the loop is not waiting for LDS.
  constexpr int SETS = DEPTH + 1;
  bf16x8 wh[SETS][2], wl[SETS][2];  // [register set][fragment]
  auto read_step = [&](auto ks_tag, auto pinned_tag) {
    constexpr int ks = decltype(ks_tag)::value;
    constexpr int S = ks % SETS;
    constexpr bool PINNED = decltype(pinned_tag)::value;
    constexpr int base = STAGE_BYTES + (ks * PLANES) * 2048;
    // the first read of the group is ordered behind every MFMA of the k-step whose set it re-uses (and ahead of the next's)
    if (!PINNED) wh[S][0] = lds_read_frag<base>(lds_addr);
    else if (MF == 2) wh[S][0] = lds_read_frag_after<base, PIN_AGPR>(lds_addr, acc[0][0], acc[0][MF - 1], acc[1][0], acc[1][MF - 1]);
    else wh[S][0] = lds_read_frag_after<base, PIN_AGPR>(lds_addr, acc[0][0], acc[1][0]);
    wh[S][1] = lds_read_frag<base + 1024>(lds_addr);
    if (W_LO) {
      wl[S][0] = lds_read_frag<base + 2048>(lds_addr);
      wl[S][1] = lds_read_frag<base + 2048 + 1024>(lds_addr);
    }
  };
  auto mfma_step = [&](auto ks_tag) {
    constexpr int ks = decltype(ks_tag)::value;
    constexpr int S = ks % SETS;
    // The product terms are issued term-major over the four accumulators: an accumulator is touched every
    // fourth MFMA, so no MFMA waits for the result of the previous one.
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      if ((term == 0 && !W_LO) || (term == 1 && !A_LO)) continue;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const bf16x8 w = term == 0 ? wl[S][nf] : wh[S][nf];
          const bf16x8 a = term == 1 ? a_lo[mf][ks] : a_hi[mf][ks];
          acc[nf][mf] = SWAPPED ? mfma16x<H16>(w, a, acc[nf][mf]) : mfma16x<H16>(a, w, acc[nf][mf]);
        }
      }
    }
  };
  const std::true_type yes{};
  const std::false_type no{};
  static_for<(SETS < KS ? SETS : KS)>([&](auto t) { read_step(t, no); });
  static_for<KS>([&](auto t) {
    constexpr int ks = decltype(t)::value;
    constexpr int S = ks % SETS;
    constexpr int ahead = (KS - 1 - ks) < DEPTH ? (KS - 1 - ks) : DEPTH;  // k-steps whose reads may stay in flight
    if (W_LO) lds_wait4<STEP_DS * ahead>(wh[S][0], wh[S][1], wl[S][0], wl[S][1]);
    else lds_wait2<STEP_DS * ahead>(wh[S][0], wh[S][1]);
    mfma_step(t);
    if constexpr (ks + SETS < KS) read_step(std::integral_constant<int, ks + SETS>{}, yes);
  });
}

// ---- "f16 + fp8" kernel set: a weight chunk (32 output features x K) is CHUNK = 2 KS fp16 pieces [ks][fragment]
// followed by 4 (KS / 4) e4m3 pieces [fragment][K-step of 128][half].  Its fragment stream is KS / 4 groups of
//   4 fp16 steps (both fragments of one k-step: 2 x MF MFMAs of 16 cycles) + 2 e4m3 steps (the two halves of ONE
//   fragment of the group's K-step: MF MFMAs of 32 cycles)
// so every step reads two 1 KiB pieces and keeps the matrix pipe busy for 64 cycles (MF = 2).
// WLO (fp32-valued weights): every group gets 2 more e4m3 steps, e4m3(left hi) x e4m3(lo(weight) x 2^12), from a third
// region of 4 (KS / 4) pieces [fragment][K-step][half].
template <int KS, bool WLO = false>
struct F8Chunk {
  static constexpr int NS8 = KS / 4;
  static constexpr int G = WLO ? 8 : 6;  // steps per group of 4 k-steps
  static constexpr int STEPS = (G * KS) / 4;
  static constexpr int PIECES = 2 * KS + (WLO ? 8 : 4) * NS8;
  static constexpr int SRC_PIECES = 2 * KS + 8 * NS8;  // the packed chunk always carries the weight-lo region
  static constexpr int BYTES = PIECES * 1024;
  static constexpr bool is_f8(int cs) { return cs % G >= 4; }
  static constexpr bool is_wlo(int cs) { return cs % G >= 6; }        // e4m3 step on the weight's lo part
  static constexpr int ks(int cs) { return 4 * (cs / G) + cs % G; }   // fp16 step: k-step
  static constexpr int nf(int cs) { return (cs % G - 4) & 1; }        // e4m3 step: fragment
  static constexpr int s8(int cs) { return cs / G; }                  //            K-step of 128
  static constexpr int off(int cs, int j) {
    return is_f8(cs) ? (2 * KS + (is_wlo(cs) ? 4 * NS8 : 0) + (nf(cs) * NS8 + s8(cs)) * 2 + j) * 1024 : (ks(cs) * 2 + j) * 1024;
  }
};
// LDS byte offsets of the fused MLP's macro-iteration stream, F8 form: [Wi chunk 2t][Wi chunk 2t+1][MLP-Wo slab t-1]
template <int KS, int NF1, bool SLAB>
struct MlpStreamOff8 {
  using C = F8Chunk<KS>;
  static constexpr int NS = SLAB ? NF1 / 2 : 0;
  // stream order (F8): chunk 2t, chunk 2t+1, slab t-1
  static constexpr int at(int s, int j) {
    if (s < C::STEPS) return C::off(s, j);
    if (s < 2 * C::STEPS) return C::BYTES + C::off(s - C::STEPS, j);
    return 2 * C::BYTES + ((s - 2 * C::STEPS) * 2 + j) * 1024;
  }
};

// One F8 weight chunk against this wave's 32 rows (the q / k / v^T loop of the whole-layer kernel): the same
// discipline as rowgemm_chunk_mfma -- reads one step ahead in two rotating register sets, each read group pinned
// behind the MFMAs of the step whose set it re-uses.
// DEPTH steps of reads stay in flight behind the one being multiplied (a step is 64 pipe cycles: one step ahead does
// not cover the LDS latency under load).
template <int KS, int MF, bool SWAPPED, bool PIN_AGPR, bool WLO = false, int DEPTH = 3>
__device__ __forceinline__ void rowgemm_chunk_mfma_f8(uint32_t lds_addr, const bf16x8 (&a_hi)[MF][KS], const i32x8 (&a_lo8)[MF][KS / 4],
                                                      const i32x8 (&a_h8)[MF][KS / 4], f32x4 (&acc)[2][MF]) {
  using C = F8Chunk<KS, WLO>;
  static_assert(MF == 2, "32 rows per wave");
  constexpr int SETS = DEPTH + 1;
  bf16x8 w[SETS][2];
  auto read_step = [&](auto cs_tag, auto pinned_tag) {
    constexpr int cs = decltype(cs_tag)::value;
    constexpr int S = cs % SETS;
    if constexpr (!decltype(pinned_tag)::value) w[S][0] = lds_read_frag<C::off(cs, 0)>(lds_addr);
    else w[S][0] = lds_read_frag_after<C::off(cs, 0), PIN_AGPR>(lds_addr, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
    w[S][1] = lds_read_frag<C::off(cs, 1)>(lds_addr);
  };
  auto mfma_step = [&](auto cs_tag) {
    constexpr int cs = decltype(cs_tag)::value;
    constexpr int S = cs % SETS;
    if constexpr (!C::is_f8(cs)) {
      constexpr int ks = C::ks(cs);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = SWAPPED ? mfma16h(w[S][nf], a_hi[mf][ks], acc[nf][mf]) : mfma16h(a_hi[mf][ks], w[S][nf], acc[nf][mf]);
    } else {
      constexpr int nf = C::nf(cs), s8 = C::s8(cs);
      const i32x8 w8 = f8_frag(w[S][0], w[S][1]);
      if constexpr (C::is_wlo(cs)) {  // e4m3(activation) x lo(weight): the scaled operand is the weight
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = SWAPPED ? mfma8w<false>(w8, a_h8[mf][s8], acc[nf][mf]) : mfma8w<true>(a_h8[mf][s8], w8, acc[nf][mf]);
      } else {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = SWAPPED ? mfma8<true>(w8, a_lo8[mf][s8], acc[nf][mf]) : mfma8<false>(a_lo8[mf][s8], w8, acc[nf][mf]);
      }
    }
  };
  const std::true_type yes{};
  const std::false_type no{};
  static_for<(SETS < C::STEPS ? SETS : C::STEPS)>([&](auto t) { read_step(t, no); });
  static_for<C::STEPS>([&](auto t) {
    constexpr int cs = decltype(t)::value;
    constexpr int ahead = (C::STEPS - 1 - cs) < DEPTH ? (C::STEPS - 1 - cs) : DEPTH;
    lds_wait2<2 * ahead>(w[cs % SETS][0], w[cs % SETS][1]);
    mfma_step(t);
    if constexpr (cs + SETS < C::STEPS) read_step(std::integral_constant<int, cs + SETS>{}, yes);
  });
}

// T2 = term mask of the chunk loop's GEMM (left = this block's rows, right = the streamed weight), T1 = term mask of
// the fused phase-1 GEMM (RP_KSTREAM only), OLO = which outputs also get a lo plane (bit 0: o0 = q / h, bit 1: o1 = k,
// bit 2: o2 = v^T).
// RP_MLP only: TW = term mask of LN(x) x Wi, TM = of h x Wo(mlp); T1 is then the attention output projection and T2
// the next layer's q/k/v projection.  Both weights must be single-plane (no hi x lo(weight) term): two LDS stages of
// [two Wi chunks | one Wo slab] are 96 KiB then.
// F8 (whole-layer kernel only): the "f16 + fp8" kernel set (opk_common.hip.h) -- every hi operand is fp16, the lo
// operand of the three K = hidden contractions (attention output projection, Wi, next q / k / v) is an e4m3 plane
// multiplied at twice the rate on the K = 128 block-scaled MFMA, h (K = 32 per step) keeps a 16-bit lo fragment.
// F8 = 2 (fp32-valued weights, kernel set 4): additionally every weight carries its lo part lo(w) = w - fp16(w) --
// e4m3 x 2^12 for the K = hidden contractions (multiplied against e4m3(activation): 0.5 MFMA units), unscaled fp16 for the
// MLP output projection (against the fp16 hi fragment of h) -- 2 MFMA units per product where the (hi, lo) bf16 kernels need 3; one Wi chunk +
// half a Wo slab per LDS stage (a stage of two chunks + a slab with their lo planes would be 96 KiB).
// H16 (kernel set "f16", round 5): the single-pass instantiation (every term mask 0) with fp16 operands -- weights from the
// fp16 packs (pack_*_kernel with f16 = 1), activations converted with v_cvt_pk_f16_f32, products on v_mfma_f32_16x16x32_f16.
template <int KS, int EPI, int PRO, int T1, int T2, int OLO, int WAVES, int MF = 2, int TW = 0, int TM = 0, int F8 = 0, bool H16 = false>
__global__ __launch_bounds__(WAVES * 64, PRO == RP_MLP ? (WAVES == 8 ? 2 : 1) : ((MF == 1 && WAVES == 8) ? 4 : 2)) void rowgemm_kernel(RowGemmParams p) {
  constexpr bool WLO = F8 == 2;
  static_assert(!H16 || (F8 == 0 && T1 == 0 && T2 == 0 && OLO == 0 && TW == 0 && TM == 0 && PRO != RP_KSTREAM && EPI != RE_GEGLU),
                "fp16 operands: the single-pass whole-layer / layer-0 q/k/v kernels only");
  constexpr int TF8 = WLO ? 3 : T_LEFT_LO;  // the term masks this instantiation stands for
  static_assert(!F8 || (PRO == RP_MLP && WAVES == 4 && MF == 2 && KS % 4 == 0 && T1 == TF8 && TW == TF8 && TM == TF8 &&
                        (EPI == RE_NONE || T2 == TF8)),
                "f16 + fp8 kernel sets: whole-layer kernel, 4 waves x 32 rows, every activation lo term (+ F8 = 2: every weight lo term)");
  constexpr int NS8 = F8 ? KS / 4 : 1;  // K = 128 steps of the fp8 lo product
  // A block is WAVES x MF x 16 rows; the library launches 4 waves x 2 fragments = 128 rows, two blocks per CU, and
  // 4 waves x 1 fragment = 64 rows for small batches (fewer than one 128-row block per CU-slot: twice the blocks, so
  // twice the CUs work on a latency-bound request).
  // Measured alternatives on MI355X (xsmall, 256 x 512): 8 waves x 2 (256 rows, one block per CU, half the DMA
  // instructions and L2 -> LDS traffic per row) is within +-2 % on both fused kernels; 8 waves x 1 (16 rows per wave,
  // <= 128 VGPRs, 4 waves per SIMD, twice the fragment reads per MFMA) is equal on q/k/v and 10 % slower on GeGLU.
  static_assert(MF == 1 || MF == 2, "one or two 16-row fragments per wave");
  // (F8: the weight's lo part is not a second bf16 plane -- W_LO / W_LO1 describe the bf16 kernels' LDS layout only)
  constexpr bool W_LO = !F8 && (T2 & T_RIGHT_LO) != 0, A_LO = (T2 & T_LEFT_LO) != 0;
  constexpr bool PHASE1 = PRO == RP_KSTREAM || PRO == RP_MLP;
  constexpr bool W_LO1 = !F8 && PHASE1 && (T1 & T_RIGHT_LO) != 0, A_LO1 = PHASE1 && (T1 & T_LEFT_LO) != 0;
  // RP_MLP: 4 waves x 32 rows, ONE wave per SIMD with the 512-register budget: the normalised rows (2 x 64 registers
  // with their lo plane) and the 256 x 32 output accumulators (128) are both resident for the whole MLP; two waves of
  // 16 rows per SIMD (256 registers each) spill.  The fragment streams prefetch by hand, so latency is covered
  // without a partner wave.
  static_assert(PRO != RP_MLP || (((WAVES == 4 && MF == 2) || (WAVES == 8 && MF == 1)) && (WLO || ((TW & T_RIGHT_LO) == 0 && (TM & T_RIGHT_LO) == 0))),
                "fused MLP: 4 waves x 32 rows or 8 waves x 16 rows, single-plane Wi and Wo");
  constexpr int PLANES = W_LO ? 2 : 1;
  constexpr int PLANES1 = W_LO1 ? 2 : 1;
  constexpr int K = KS * 32;
  // F8: a chunk is [fp16 plane: KS k-steps x 2 fragments | e4m3 plane: 2 fragments x KS/4 K-steps x 2 halves] 1 KiB pieces
  // (WLO: + the e4m3 plane of the weight's lo part; the packed chunk in global memory always carries it)
  constexpr int CHUNK_PIECES8 = F8Chunk<KS, WLO>::PIECES;
  constexpr int CHUNK_SRC = F8 ? F8Chunk<KS, WLO>::SRC_PIECES * 512 : KS * 2 * 1024;  // elements per packed chunk in global memory
  constexpr int STAGE = F8 ? CHUNK_PIECES8 * 512 : KS * PLANES * 1024;  // elements per LDS stage
  // two Wi chunks + one Wo slab (2 KS fragments), one plane; F8: the chunks carry their e4m3 plane; WLO: ONE chunk +
  // half a slab (KS fragments, fp16 + bf16-lo planes) per stage
  constexpr int MLP_UNIT = PRO == RP_MLP ? (WLO ? (CHUNK_PIECES8 + 2 * KS) * 512 : F8 ? (2 * CHUNK_PIECES8 + 2 * KS) * 512 : 3 * KS * 1024) : 0;
  // phase 1 slabs: 2 KS fragments per plane; F8: two fp16 slabs + half an e4m3 K = 128 slab per stage (WLO: + the
  // same half slab of the weight's lo part)
  constexpr int STAGE_GEMM = WLO ? 8 * KS * 512 : F8 ? 6 * KS * 512 : KS * (PLANES > PLANES1 ? PLANES : PLANES1) * 1024;
  constexpr int STAGE_ALLOC = STAGE_GEMM > MLP_UNIT ? STAGE_GEMM : MLP_UNIT;
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE_ALLOC];
  // Whole-layer kernel: its two LayerNorm weight vectors (this layer's mlp_norm, the next layer's attn_norm) live in
  // LDS.  Read from global memory inside the LayerNorm they were 32 L2 round trips per block issued just in time
  // (the compiler cannot hoist them over the asm fences), each behind an in-order vmcnt wait that also waited for the
  // write acknowledgements of the residual rows stored just before: 16 k of a block's 252 k cycles per LayerNorm.
  constexpr bool LN_V2 = PRO == RP_MLP;
  static_assert(!F8 || LN_V2, "f16 + fp8 kernel set: LayerNorm phases of the whole-layer kernel");
  constexpr bool FIN_HEAD = LN_V2 && EPI == RE_NONE;  // (run-time switch: p.fin_ln)
  __shared__ __attribute__((aligned(16))) float sLn[LN_V2 ? (FIN_HEAD ? 4 : 2) * KS * 32 : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * (WAVES * 16 * MF) + wave * (16 * MF);
  // requested before anything else, written to LDS in front of the first block barrier (index clamped, no branch:
  // a load under a branch is drained at the join)
  const int ln_i = tid < KS * 32 ? tid : KS * 32 - 1;
  float ln_fill0 = 0.f, ln_fill1 = 0.f, ln_fill2 = 0.f, ln_fill3 = 0.f;
  if constexpr (LN_V2) {
    ln_fill0 = p.ln_w_mlp[ln_i];
    if (EPI != RE_NONE) ln_fill1 = p.ln_w[ln_i];
    if (FIN_HEAD && p.fin_ln != nullptr) {  // block-uniform; these few loads are the first of the kernel
      ln_fill1 = p.fin_ln[ln_i];
      ln_fill2 = p.fin_pw[ln_i];
      ln_fill3 = p.fin_pw[KS * 32 + ln_i];
    }
  }
#ifdef OPK_SEG_TIMING
#define OPK_SEG_DUMP() for (int i_ = 0; i_ < 3; ++i_) p.dbg[(size_t)blockIdx.x * 16 + 11 + i_] = opk_seg[i_];
#else
#define OPK_SEG_DUMP()
#endif
#ifdef OPK_TIMING
  unsigned long long opk_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, opk_wait = 0, opk_wait2 = 0, opk_wait1 = 0, opk_x[4] = {0, 0, 0, 0};
#ifdef OPK_SEG_TIMING
  unsigned long long opk_seg[3] = {0, 0, 0}, opk_seg_t = 0;
#endif
#define OPK_STAMP(i) opk_ts[i] = __builtin_readcyclecounter()
#define OPK_DUMP()                                                                              \
  do {                                                                                          \
    if (threadIdx.x == 0) {                                                                     \
      for (int i_ = 0; i_ < 8; ++i_) p.dbg[(size_t)blockIdx.x * 16 + i_] = opk_ts[i_];          \
      p.dbg[(size_t)blockIdx.x * 16 + 8] = opk_wait;                                            \
      p.dbg[(size_t)blockIdx.x * 16 + 9] = opk_wait2;                                           \
      p.dbg[(size_t)blockIdx.x * 16 + 10] = opk_wait1;                                          \
      for (int i_ = 0; i_ < 4; ++i_) p.dbg[(size_t)blockIdx.x * 16 + 11 + i_] = opk_x[i_];     \
      OPK_SEG_DUMP()                                                                            \
      p.dbg[(size_t)blockIdx.x * 16 + 15] = wall_clock64() - opk_rt0;                           \
    }                                                                                           \
  } while (0)
  const unsigned long long opk_rt0 = wall_clock64();  // constant 100 MHz: shader clock = cycle stamps / this
  OPK_STAMP(0);
#else
#define OPK_STAMP(i)
#define OPK_DUMP()
#endif

  // ---- weight streaming: global -> LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave-instruction) --
  // Stage layout = [ks][plane][frag][512] = a sequence of 1 KiB pieces; wave w copies pieces w, w+4, ...
  // The copy is linear (the packing kernel already wrote fragment order), so the lane-linear LDS
  // destination the DMA imposes is exactly the layout the fragment reads want.  No staging VGPRs, and
  // the request is in flight while the MFMAs of the current chunk run.
  constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  auto stage_chunk = [&](int chunk, int stage) {
    const u16* src = p.wp + (size_t)chunk * CHUNK_SRC;
    if constexpr (F8 != 0) {
      // groups of GS consecutive pieces per wave: one pointer and one M0 per group, the rest through the DMA's immediate
      // offset (it moves both addresses; the stage mirrors the packed chunk) -- see stage_piece of the MLP loop
      constexpr int PIECES = STAGE / 512;
      constexpr int GS = (PIECES / 4) % WAVES == 0 ? 4 : ((PIECES / 2) % WAVES == 0 ? 2 : 1);
      static_for<WAVE_PIECES>([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        const int piece0 = GS * (wave + WAVES * (u / GS));  // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece0 * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(&sW[stage][piece0 * 512]), 16, (u % GS) * 1024, 0);
      });
      return;
    }
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + WAVES * u;              // wave-uniform
      const int elem = piece * 512;                    // position inside the LDS stage
      const int ks = elem / (PLANES * 1024);
      const int rem = elem % (PLANES * 1024);
      const int src_elem = F8 ? elem : ks * 2048 + rem;  // source keeps both planes per k-step (F8: packed as staged)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + src_elem + lane * 8),
          (__attribute__((address_space(3))) void*)(&sW[stage][elem]), 16, 0, 0);
    }
  };
  // Kernel set "f16" (round 5): the q / k / v^T loop of the whole-layer kernel takes TWO chunks per LDS stage and barrier
  // (pair_iteration below); chunk `chunk` copied to element offset `elem_off` of stage `stage`
  constexpr bool QKV2 = PRO == RP_MLP && EPI == RE_QKV && !F8 && H16 && 2 * STAGE <= STAGE_ALLOC;
  auto stage_chunk_at = [&](int chunk, int stage, int elem_off) {
    const u16* src = p.wp + (size_t)chunk * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int elem = (wave + WAVES * u) * 512;                // position inside the chunk's single plane
      const int src_elem = (elem / 1024) * 2048 + elem % 1024;  // the source keeps both planes per k-step
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_elem + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][elem_off + elem]), 16, 0, 0);
    }
  };
  // F8 kernel sets, q / k / v^T projection: the chunks are streamed in PAIRS (qkv_pairs below), a stage holds chunks
  // 2t and 2t+1 back to back.  Instruction u of a wave copies piece u % GS of its group u / GS (GS consecutive pieces
  // through one pointer / M0 and the DMA's immediate offset); the chunk a group belongs to is a wave-uniform integer
  // select, never control flow (a DMA under a branch is drained at the join).
  constexpr bool QKV_PAIRS = F8 != 0 && EPI == RE_QKV;
  constexpr int FRAG_ILV = 4;  // fragment groups in flight where the reads sit between the MFMAs (frag_stream2i); two measured slower in the forward
  constexpr int PAIR_DMA = QKV_PAIRS ? 2 * CHUNK_PIECES8 / WAVES : 1;  // DMA instructions per wave and pair
  constexpr int PAIR_GS = (2 * CHUNK_PIECES8 / 4) % WAVES == 0 ? 4 : 2;  // pieces per group (hidden 128 / 384: 2)
  static_assert(!QKV_PAIRS || (CHUNK_PIECES8 % PAIR_GS == 0 && (2 * CHUNK_PIECES8 / PAIR_GS) % WAVES == 0 && 2 * STAGE <= STAGE_ALLOC),
                "a chunk pair is whole groups of pieces per wave and fits one LDS stage");
  auto stage_pair_piece = [&](auto u_tag, int pair, int stage) {
    constexpr int u = decltype(u_tag)::value;
    const int piece0 = PAIR_GS * (wave + WAVES * (u / PAIR_GS));  // within the pair's stage
    const int second = piece0 >= CHUNK_PIECES8 ? 1 : 0;           // group lies in chunk 2t+1
    const u16* src = p.wp + (size_t)(2 * pair + second) * CHUNK_SRC + (piece0 - second * CHUNK_PIECES8) * 512 + lane * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(&sW[0][0] + stage * STAGE_ALLOC + piece0 * 512), 16,
                                     (u % PAIR_GS) * 1024, 0);
  };
  auto stage_pair = [&](int pair, int stage) { static_for<PAIR_DMA>([&](auto u_tag) { stage_pair_piece(u_tag, pair, stage); }); };
  bf16x8 a_hi[MF][KS], a_lo[MF][KS];
  i32x8 a_lo8[MF][NS8];  // F8: e4m3 lo plane of the in-register operand, one K = 128 fragment per 4 k-steps
  i32x8 a_h8[MF][NS8];   // WLO: e4m3 of the operand itself (multiplied against the weights' lo part)
  if constexpr (F8) set_saturating_conversions();
  // RE_QKV: RoPE rows of this lane's tokens, cos/sin [pos][8g + 4j .. +3] for half-head j.  Two-wave kernels fetch the
  // half-head of the chunk whose (deferred) epilogue runs in an iteration at the top of that iteration (the partner wave
  // covers the latency).  The one-wave-per-SIMD layer kernel (RP_MLP) has nobody to cover it -- the loads sat behind a
  // full s_waitcnt vmcnt(0) in front of each epilogue, ~1000 cycles per chunk -- so it fetches both half-heads once,
  // while the LayerNorm in front of the chunk loop runs (ROPE_PRELOAD).
  constexpr bool ROPE_PRELOAD = EPI == RE_QKV && PRO == RP_MLP;
  const float* rope_c_row[MF];
  const float* rope_s_row[MF];
  f32x4 rope_c[MF], rope_s[MF];
  f32x4 rope_cc[MF][2], rope_ss[MF][2];
  auto rope_rows = [&]() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int pos = p.row_pos[m0 + mf * 16 + l15];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      rope_c_row[mf] = p.rope_cos + (size_t)pos * ROPE_HALF + g * 8;
      rope_s_row[mf] = p.rope_sin + (size_t)pos * ROPE_HALF + g * 8;
      rope_c[mf] = rope_s[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto rope_preload = [&]() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rope_cc[mf][j] = *reinterpret_cast<const f32x4*>(rope_c_row[mf] + j * 4);
        rope_ss[mf][j] = *reinterpret_cast<const f32x4*>(rope_s_row[mf] + j * 4);
      }
  };
  if (ROPE_PRELOAD) rope_rows();  // the position index load flies during phase 1
  // LDS byte address of this lane's 16 bytes in piece 0 of each stage (the hand-placed fragment reads add immediates)
  uint32_t lds_stage[2];
  lds_stage[0] = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0][0]) + (uint32_t)lane * 16u;
  lds_stage[1] = lds_stage[0] + (uint32_t)(STAGE_ALLOC * 2);
  if (PHASE1) {
    // ---- fused phase 1: x_new[32 rows, H] = x + A1[32 rows, K1] W1[H, K1]^T, K1 streamed ----------------
    // Same structure as kstream_gemm_kernel (one [H x 32] weight slab per k-step by DMA, A1 fragments straight
    // from the fragment-packed activation, prefetched one k-step ahead), all H outputs of the 32 rows in
    // accumulators.  W1's output features were permuted at load time so that accumulator fragments (2s, 2s+1)
    // are exactly lane slot g of k-step s of THIS kernel's chunk loop: residual add, LayerNorm and the hi/lo
    // split happen in registers and the hidden state makes one fp32 round trip (read + write) per block.
    constexpr int NF1 = 2 * KS;
    constexpr int SLAB_SRC = F8 ? NF1 * 512 : NF1 * 2 * 512;
    constexpr int SLAB_PIECES = (NF1 * PLANES1) / WAVES;
    static_assert((NF1 * PLANES1) % WAVES == 0, "slab must split evenly over the waves");
    auto stage_slab = [&](int ks1, int stage) {
      const u16* src = p.w1p + (size_t)ks1 * SLAB_SRC;
#pragma unroll
      for (int u = 0; u < SLAB_PIECES; ++u) {
        const int piece = wave + WAVES * u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                         (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
      }
    };
    const int nks1 = p.k1_steps;
    const u16* a_base0 = p.a1_fp + ((size_t)(m0 >> 4) * nks1 * 2) * 512 + lane * 8;
    const size_t a_block = (size_t)nks1 * 2 * 512;  // elements per 16-row block of A1
    bf16x8 an_hi[MF], an_lo[MF];
    auto load_a1 = [&](int ks1) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        an_hi[mf] = load_stream_frag(a_base0 + mf * a_block + (size_t)ks1 * 1024);
        an_lo[mf] = A_LO1 ? load_stream_frag(a_base0 + mf * a_block + (size_t)ks1 * 1024 + 512) : an_hi[mf];
      }
    };
    f32x4 acc1[NF1][MF];
#pragma unroll
    for (int nf = 0; nf < NF1; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // RP_MLP (one wave per SIMD, nobody to hide a load behind): everything phase 1 reads from HBM is requested up
    // front -- all K1 / 32 = KS fragment pairs of A1 (into a_hi / a_lo, which only the LayerNorm below overwrites) and
    // the residual rows x of row fragment 0 -- and the weight slabs come two k-steps per LDS stage (half the block
    // barriers, each DMA issued two k-steps ahead of its use).  Measured per 128-row block before / after: phase 1
    // 26k -> 12k cycles (8.3k are MFMAs), residual + LayerNorm 29k -> 11k (eight serialized HBM round trips gone).
    float4 xq0[(PRO == RP_MLP) ? NF1 : 1];
    if constexpr (PRO == RP_MLP) {
      static_assert(PLANES1 == 1, "whole-layer kernel: single-plane attention output weight");
      // DMA instructions per wave per stage: two k-steps of fp16 slabs (+ F8: NF1 / 2 fragments = NF1 half-pieces of the
      // e4m3 slab of K-step j / 2: its first half of the output features in the even stage, the second in the odd one)
      constexpr int PAIR_PIECES = (WLO ? 4 : F8 ? 3 : 2) * NF1 / WAVES;
      static_assert((WLO ? 4 : F8 ? 3 : 2) * NF1 * 512 <= STAGE_ALLOC, "two slabs per LDS stage");
      auto stage_pair = [&](int j, int stage) {
#pragma unroll
        for (int u = 0; u < PAIR_PIECES; ++u) {
          const int piece = wave + WAVES * u;  // [k-step 2j | 2j+1][nf]
          const u16* src = p.w1p + (size_t)(2 * j + piece / NF1) * SLAB_SRC + (piece % NF1) * 512;
          if (F8 && u >= 2 * NF1 / WAVES)
            src = p.w1p8 + ((size_t)(j >> 1) * NF1 + (NF1 / 2) * (j & 1)) * 1024 + (piece - 2 * NF1) * 512;
          if (WLO && u >= 3 * NF1 / WAVES)  // the same half slab of lo(w): a second array right behind the first
            src = p.w1p8 + (size_t)nks1 * 16 * K + ((size_t)(j >> 1) * NF1 + (NF1 / 2) * (j & 1)) * 1024 + (piece - 3 * NF1) * 512;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                           (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
        }
      };
      if constexpr (F8) {
        const u16* h_base = p.a1_fp + (size_t)(m0 >> 4) * nks1 * 512 + lane * 8;
        const u16* l_base = p.a1_lo8 + (size_t)(m0 >> 4) * (nks1 >> 1) * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) a_hi[mf][ks] = load_stream_frag(h_base + ((size_t)mf * nks1 + ks) * 512);
#pragma unroll
        for (int s8 = 0; s8 < NS8; ++s8)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            a_lo8[mf][s8] = f8_frag(load_stream_frag(l_base + ((size_t)mf * (nks1 >> 1) + 2 * s8) * 512),
                                    load_stream_frag(l_base + ((size_t)mf * (nks1 >> 1) + 2 * s8 + 1) * 512));
      } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          a_hi[mf][ks] = load_stream_frag(a_base0 + mf * a_block + (size_t)ks * 1024);
          a_lo[mf][ks] = A_LO1 ? load_stream_frag(a_base0 + mf * a_block + (size_t)ks * 1024 + 512) : a_hi[mf][ks];
        }
      }
      stage_pair(0, 0);
      __builtin_amdgcn_sched_barrier(0);  // the residual rows are requested last and are not waited for here
      {
        const float* xrow = p.x_io + (size_t)(m0 + l15) * K + g * 8;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) xq0[nf] = load_stream_f4(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      // vmcnt retires in order: everything but the NF1 residual-row loads (the A operand and this wave's share of the
      // first weight stage) has landed; then all waves meet
      if constexpr (LN_V2) {
        sLn[ln_i] = ln_fill0;
        sLn[KS * 32 + ln_i] = ln_fill1;
        if constexpr (FIN_HEAD) {
          sLn[2 * KS * 32 + ln_i] = ln_fill2;
          sLn[3 * KS * 32 + ln_i] = ln_fill3;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the raw barrier below publishes them
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF1) : "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (WLO) {  // e4m3(o) for the product with the weights' lo part: from the fp16 fragments that just landed
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const uint4 h = as_u4(a_hi[mf][ks]);
            const int d0 = 4 * ((ks % 4) / 2) + 2 * (ks % 2);
            a_h8[mf][ks / 4][d0] = (int)f16x4_to_e4m3(h.x, h.y);
            a_h8[mf][ks / 4][d0 + 1] = (int)f16x4_to_e4m3(h.z, h.w);
          }
      }
      static_for<KS / 2>([&](auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int cur = j & 1;
        if constexpr (j + 1 < KS / 2) stage_pair(j + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // one fragment stream per stage (fragment reads placed by hand, two steps ahead -- see frag_stream2): step
        // (kk, nf / 2) = the two weight fragments nf, nf + 1 of k-step 2j + kk against both row fragments
        struct P1Off {
          static constexpr int at(int st, int jj) { return (2 * st + jj) * 1024; }
        };
        frag_stream2<(WLO ? 2 * NF1 : F8 ? NF1 + NF1 / 2 : NF1), (F8 ? 4 : 2), P1Off>(lds_stage[cur], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
          constexpr int st = decltype(step_tag)::value;
          if constexpr (F8) {
            if constexpr (st < NF1) {  // fp16: fragments nf, nf + 1 of k-step ks
              constexpr int ks = 2 * j + st / (NF1 / 2), nf = (st % (NF1 / 2)) * 2;
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w0, a_hi[mf][ks], acc1[nf][mf]);
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16h(w1, a_hi[mf][ks], acc1[nf + 1][mf]);
            } else if constexpr (st < NF1 + NF1 / 2) {  // e4m3: the two halves of fragment nf8, K-step j / 2
              constexpr int nf8 = (NF1 / 2) * (j & 1) + (st - NF1);
              const i32x8 w8 = f8_frag(w0, w1);
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) acc1[nf8][mf] = mfma8<true>(w8, a_lo8[mf][j >> 1], acc1[nf8][mf]);
            } else {  // WLO: e4m3(o) x lo(w) of the same fragments
              constexpr int nf8 = (NF1 / 2) * (j & 1) + (st - NF1 - NF1 / 2);
              const i32x8 w8 = f8_frag(w0, w1);
#pragma unroll
              for (int mf = 0; mf < MF; ++mf) acc1[nf8][mf] = mfma8w<false>(w8, a_h8[mf][j >> 1], acc1[nf8][mf]);
            }
            return;
          }
          constexpr int ks = 2 * j + (st % NF1) / (NF1 / 2), nf = (st % (NF1 / 2)) * 2;
          if (A_LO1) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16(w0, a_lo[mf][ks], acc1[nf][mf]);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16(w1, a_lo[mf][ks], acc1[nf + 1][mf]);
          }
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16x<H16>(w0, a_hi[mf][ks], acc1[nf][mf]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16x<H16>(w1, a_hi[mf][ks], acc1[nf + 1][mf]);
        });
        __syncthreads();
      });
    } else {
      stage_slab(0, 0);
      load_a1(0);
  #pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        asm volatile("" : "+v"(an_hi[mf]));
        asm volatile("" : "+v"(an_lo[mf]));
      }
      __syncthreads();
      auto slab_step = [&](int ks1, auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        const int kn = ks1 + 1 < nks1 ? ks1 + 1 : ks1;
        stage_slab(kn, cur ^ 1);
        bf16x8 c_hi[MF], c_lo[MF];
  #pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          c_hi[mf] = an_hi[mf];
          c_lo[mf] = an_lo[mf];
        }
        load_a1(kn);
        __builtin_amdgcn_sched_barrier(0);
        // two weight fragments at a time, the three product terms issued term-major over their 2 x MF accumulators:
        // no MFMA reads the accumulator the previous one wrote (a dependent pair stalls the pipe)
  #pragma unroll
        for (int nf = 0; nf < NF1; nf += 2) {
          bf16x8 wh[2], wl[2];
  #pragma unroll
          for (int j = 0; j < 2; ++j) {
            wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
            wl[j] = W_LO1 ? lds_frag(&sW[cur][(NF1 + nf + j) * 512 + lane * 8]) : wh[j];
          }
  #pragma unroll
          for (int term = 0; term < 3; ++term) {
            if ((term == 0 && !W_LO1) || (term == 1 && !A_LO1)) continue;
  #pragma unroll
            for (int j = 0; j < 2; ++j)
  #pragma unroll
              for (int mf = 0; mf < MF; ++mf)
                acc1[nf + j][mf] = mfma16(term == 0 ? wl[j] : wh[j], term == 1 ? c_lo[mf] : c_hi[mf], acc1[nf + j][mf]);
          }
        }
        __syncthreads();
      };
      for (int k0 = 0; k0 < nks1; k0 += 2) {  // even number of k-steps (checked on the host)
        slab_step(k0, std::integral_constant<int, 0>{});
        slab_step(k0 + 1, std::integral_constant<int, 1>{});
      }
    }
    OPK_STAMP(1);
    // ---- transition: residual add, (store the new hidden state,) LayerNorm, split -> fragments -------------------
    // LOAD: acc1 += x rows from memory; STORE: write the rows back; then LayerNorm with `lnw` into a_hi / a_lo.
    auto residual_ln = [&](auto load_tag, auto store_tag, auto lo_tag, const float* __restrict__ lnw) {
      constexpr bool LOAD = decltype(load_tag)::value, STORE = decltype(store_tag)::value, LO = decltype(lo_tag)::value;
      // RP_MLP: row fragment 0 of x was requested at the top of the kernel, fragment 1 is requested here and arrives
      // while fragment 0 is normalised
      constexpr bool XPRE = LOAD && PRO == RP_MLP;
      float4 xq1[(XPRE && MF > 1) ? NF1 : 1];
      if (XPRE && MF > 1) {
        const float* xrow1 = p.x_io + (size_t)(m0 + 16 + l15) * K + g * 8;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) xq1[nf] = load_stream_f4(xrow1 + 32 * (nf >> 1) + 4 * (nf & 1));
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        float* xrow = p.x_io + (size_t)(m0 + mf * 16 + l15) * K + g * 8;
        float sum = 0.f;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
          float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
          float4 r4 = make_float4(acc1[nf][mf][0], acc1[nf][mf][1], acc1[nf][mf][2], acc1[nf][mf][3]);
          if (LOAD) {
            const float4 x4 = XPRE ? (mf == 0 ? xq0[nf] : xq1[(XPRE && MF > 1) ? nf : 0]) : load_stream_f4(reinterpret_cast<const float*>(px));
            r4.x += x4.x;
            r4.y += x4.y;
            r4.z += x4.z;
            r4.w += x4.w;
            acc1[nf][mf] = f32x4{r4.x, r4.y, r4.z, r4.w};
          }
          if (STORE) store_stream16(reinterpret_cast<float*>(px), r4);
          sum += (r4.x + r4.y) + (r4.z + r4.w);
          // keep the scheduler from hoisting all 16 row loads (64 more registers) on top of the accumulators
          if (((LOAD && !XPRE) || STORE) && (nf & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (lnw == nullptr) continue;  // RE_NONE: the residual stream is all the last layer leaves behind
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum / (float)K;
        float q = 0.f;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc1[nf][mf][r] - mean;
            q += d * d;
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q / (float)K + p.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 w0 = *reinterpret_cast<const float4*>(lnw + ks * 32 + g * 8);
          const float4 w1 = *reinterpret_cast<const float4*>(lnw + ks * 32 + g * 8 + 4);
          const float v[8] = {(acc1[2 * ks][mf][0] - mean) * rstd * w0.x,     (acc1[2 * ks][mf][1] - mean) * rstd * w0.y,
                              (acc1[2 * ks][mf][2] - mean) * rstd * w0.z,     (acc1[2 * ks][mf][3] - mean) * rstd * w0.w,
                              (acc1[2 * ks + 1][mf][0] - mean) * rstd * w1.x, (acc1[2 * ks + 1][mf][1] - mean) * rstd * w1.y,
                              (acc1[2 * ks + 1][mf][2] - mean) * rstd * w1.z, (acc1[2 * ks + 1][mf][3] - mean) * rstd * w1.w};
          pack8<LO>(v, a_hi[mf][ks], a_lo[mf][ks]);
        }
      }
    };
    // ---- the same transition in the whole-layer kernel (one wave per SIMD: a vector-only phase runs at one
    // instruction per ~4.9 cycles, 8 for the conversion / accumulator-move class and for an instruction that needs
    // the result of the one before it).  The row's 64 values per lane are read from the accumulators once, sums run in
    // four independent chains of packed instructions, the LayerNorm weights come from LDS (sLn, above) and the
    // write-back of the rows is left to store_rows(): a store in the middle puts every later counted vmcnt wait behind
    // its write acknowledgement.
    auto layer_ln = [&](auto load_tag, auto lo_tag, int which) {
      constexpr bool LOAD = decltype(load_tag)::value, LO = decltype(lo_tag)::value;
      float4 xq1[(LOAD && MF > 1) ? NF1 : 1];
      if (LOAD && MF > 1) {  // row fragment 1 of x arrives while fragment 0 is normalised
        const float* xrow1 = p.x_io + (size_t)(m0 + 16 + l15) * K + g * 8;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) xq1[nf] = load_stream_f4(xrow1 + 32 * (nf >> 1) + 4 * (nf & 1));
      }
      // this lane's columns 32 ks + 8 g .. + 7 of the weight vector.  The offset is made opaque HERE: otherwise the
      // compiler hoists the second LayerNorm's 64 weight values above the MLP loop and spills them across it.
      // (read a step ahead by hand, lds_read_f4: a compiler-placed LDS read behind the in-flight weight DMA is always
      // followed by lgkmcnt(0))
      const uint32_t ln_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)&sLn[0]) + (uint32_t)(which * K + g * 8) * 4u;
      f32x4 wq[2][2];
      auto ln_read = [&](auto ks_tag) {
        constexpr int ks = decltype(ks_tag)::value;
        wq[ks & 1][0] = lds_read_f4<ks * 128>(ln_addr);
        wq[ks & 1][1] = lds_read_f4<ks * 128 + 16>(ln_addr);
      };
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        f32x2 v[2 * NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
          f32x4 a = acc1[nf][mf];
          if (LOAD) {
            // (scalar adds: with packed ones feeding the accumulators the register allocator permutes all 128 of them
            // through scratch around the MLP loop)
            const float4 x4 = mf == 0 ? xq0[nf] : xq1[(LOAD && MF > 1) ? nf : 0];
            a = f32x4{a[0] + x4.x, a[1] + x4.y, a[2] + x4.z, a[3] + x4.w};
            acc1[nf][mf] = a;
          }
          v[2 * nf] = f32x2{a[0], a[1]};
          v[2 * nf + 1] = f32x2{a[2], a[3]};
        }
        f32x2 s4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int i = 4; i < 2 * NF1; ++i) s4[i & 3] = pk_add(s4[i & 3], v[i]);
        const f32x2 st = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
        float sum = st.x + st.y;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / (float)K);
        const f32x2 m2 = f32x2{mean, mean};
        f32x2 q4[4];
#pragma unroll
        for (int i = 0; i < 2 * NF1; ++i) {
          v[i] = pk_sub(v[i], m2);
          q4[i & 3] = i < 4 ? pk_mul(v[i], v[i]) : pk_fma(v[i], v[i], q4[i & 3]);
        }
        const f32x2 qt = pk_add(pk_add(q4[0], q4[1]), pk_add(q4[2], q4[3]));
        float q = qt.x + qt.y;
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)K) + p.eps);
        const f32x2 r2 = f32x2{rstd, rstd};
        ln_read(std::integral_constant<int, 0>{});
        if constexpr (KS > 1) ln_read(std::integral_constant<int, 1>{});
        static_for<KS>([&](auto ks_tag) {
          constexpr int ks = decltype(ks_tag)::value;
          f32x4& w0 = wq[ks & 1][0];
          f32x4& w1 = wq[ks & 1][1];
          lds_wait_f4<(ks + 1 < KS ? 2 : 0)>(w0, w1);
          const f32x2 lw[4] = {f32x2{w0[0], w0[1]}, f32x2{w0[2], w0[3]}, f32x2{w1[0], w1[1]}, f32x2{w1[2], w1[3]}};
          if constexpr (F8) {
            // fp16 hi fragment + the 8 e4m3 lo bytes of this k-step inside the K = 128 fragment ks / 4 (opk_common.hip.h)
            f32x2 y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = pk_mul(pk_mul(v[4 * ks + j], r2), lw[j]);
            const float va[4] = {y[0].x, y[0].y, y[1].x, y[1].y}, vb[4] = {y[2].x, y[2].y, y[3].x, y[3].y};
            uint2 h0, h1;
            uint32_t l0, l1;
            split4_f8(va, h0, l0);
            split4_f8(vb, h1, l1);
            a_hi[mf][ks] = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
            constexpr int d0 = 4 * ((ks % 4) / 2) + 2 * (ks % 2);
            a_lo8[mf][ks / 4][d0] = (int)l0;
            a_lo8[mf][ks / 4][d0 + 1] = (int)l1;
            if constexpr (WLO) {
              a_h8[mf][ks / 4][d0] = (int)f32x4_to_e4m3(va);
              a_h8[mf][ks / 4][d0 + 1] = (int)f32x4_to_e4m3(vb);
            }
            if constexpr (LOAD && (ks % 4) == 3) {  // parked where the MLP wants them
              asm volatile("" : "+a"(a_lo8[mf][ks / 4]));
              if constexpr (WLO) asm volatile("" : "+a"(a_h8[mf][ks / 4]));
            }
          } else {
          uint32_t h[4], l[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) split2x_pk<LO, H16>(pk_mul(pk_mul(v[4 * ks + j], r2), lw[j]), h[j], l[j]);
          a_hi[mf][ks] = as_frag(make_uint4(h[0], h[1], h[2], h[3]));
          a_lo[mf][ks] = as_frag(make_uint4(l[0], l[1], l[2], l[3]));
          if constexpr (LO && LOAD && MF == 2) asm volatile("" : "+a"(a_lo[mf][ks]));  // parked where the MLP wants it (see below)
          }
          if constexpr (ks + 2 < KS) ln_read(std::integral_constant<int, ks + 2>{});
        });
#ifdef OPK_TIMING
        if (mf == 0) opk_x[LOAD ? 0 : 1] = __builtin_readcyclecounter();
#endif
      }
    };
    // The residual rows as they stand in the accumulators, in one burst behind the LayerNorm's arithmetic.  Measured
    // (stamps): the arithmetic takes 5.5 k cycles, the 32 stores 5.7 k to ISSUE -- every CU of the chip writes its
    // 128 KB at the same moment -- and spread between the arithmetic instructions they cost more (LayerNorm phase
    // 11.2 k as a burst, 15 - 17.7 k interleaved).
    auto store_rows = [&]() {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        float* xrow = p.x_io + (size_t)(m0 + mf * 16 + l15) * K + g * 8;
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf)
          store_stream16(xrow + 32 * (nf >> 1) + 4 * (nf & 1),
                         make_float4(acc1[nf][mf][0], acc1[nf][mf][1], acc1[nf][mf][2], acc1[nf][mf][3]));
      }
    };
    // final_norm + pruning head on the rows in the accumulators (see RowGemmParams::fin_ln).  Same arithmetic per row
    // as layer_ln; the head's two dot products ride on the normalised values (or, fin_pre_norm, on the raw row), the
    // four lanes of a row are summed, the lane of column group 0 writes the token's logits and keep-probability.
    auto final_head = [&]() {
      const float* lw_s = &sLn[K + g * 8];
      const float* p0_s = &sLn[2 * K + g * 8];
      const float* p1_s = &sLn[3 * K + g * 8];
      const float b0 = p.fin_pb[0], b1 = p.fin_pb[1];
      const bool pre = p.fin_pre_norm != 0;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int row = m0 + mf * 16 + l15;
        const int tok = p.row_tok[row];
        const bool is_cls = tok >= 0 && p.row_pos[row] == 0;
        f32x2 v[2 * NF1];
#pragma unroll
        for (int nf = 0; nf < NF1; ++nf) {
          const f32x4 a = acc1[nf][mf];
          v[2 * nf] = f32x2{a[0], a[1]};
          v[2 * nf + 1] = f32x2{a[2], a[3]};
        }
        f32x2 s4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int i = 4; i < 2 * NF1; ++i) s4[i & 3] = pk_add(s4[i & 3], v[i]);
        const f32x2 st = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
        float sum = st.x + st.y;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / (float)K);
        const f32x2 m2 = f32x2{mean, mean};
        f32x2 q4[4], d0[2], d1[2];
        d0[0] = d0[1] = d1[0] = d1[1] = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2 * NF1; ++i) {
          const f32x2 c = pk_sub(v[i], m2);
          q4[i & 3] = i < 4 ? pk_mul(c, c) : pk_fma(c, c, q4[i & 3]);
        }
        const f32x2 qt = pk_add(pk_add(q4[0], q4[1]), pk_add(q4[2], q4[3]));
        float q = qt.x + qt.y;
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)K) + p.eps);
        const f32x2 r2 = f32x2{rstd, rstd};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 w0 = *reinterpret_cast<const float4*>(lw_s + ks * 32), w1 = *reinterpret_cast<const float4*>(lw_s + ks * 32 + 4);
          const float4 a0 = *reinterpret_cast<const float4*>(p0_s + ks * 32), a1 = *reinterpret_cast<const float4*>(p0_s + ks * 32 + 4);
          const float4 c0 = *reinterpret_cast<const float4*>(p1_s + ks * 32), c1 = *reinterpret_cast<const float4*>(p1_s + ks * 32 + 4);
          const f32x2 lw[4] = {f32x2{w0.x, w0.y}, f32x2{w0.z, w0.w}, f32x2{w1.x, w1.y}, f32x2{w1.z, w1.w}};
          const f32x2 pa[4] = {f32x2{a0.x, a0.y}, f32x2{a0.z, a0.w}, f32x2{a1.x, a1.y}, f32x2{a1.z, a1.w}};
          const f32x2 pc[4] = {f32x2{c0.x, c0.y}, f32x2{c0.z, c0.w}, f32x2{c1.x, c1.y}, f32x2{c1.z, c1.w}};
          f32x2 y[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            y[j] = pk_mul(pk_mul(pk_sub(v[4 * ks + j], m2), r2), lw[j]);
            const f32x2 src = pre ? v[4 * ks + j] : y[j];
            d0[j & 1] = pk_fma(src, pa[j], d0[j & 1]);
            d1[j & 1] = pk_fma(src, pc[j], d1[j & 1]);
          }
          if (is_cls) {  // one row in a sequence: the ranking head's input
            float* dst = p.fin_cls + (size_t)p.row_seq[row] * K + ks * 32 + g * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
          }
        }
        const f32x2 e0 = pk_add(d0[0], d0[1]), e1 = pk_add(d1[0], d1[1]);
        float l0 = e0.x + e0.y, l1 = e1.x + e1.y;
        l0 += __shfl_xor(l0, 16, 64);
        l0 += __shfl_xor(l0, 32, 64);
        l1 += __shfl_xor(l1, 16, 64);
        l1 += __shfl_xor(l1, 32, 64);
        if (g == 0 && tok >= 0) {
          l0 += b0;
          l1 += b1;
          p.fin_prune[(size_t)tok * 2 + 0] = l0;
          p.fin_prune[(size_t)tok * 2 + 1] = l1;
          if (p.fin_keep) p.fin_keep[tok] = 1.0f / (1.0f + expf(l0 - l1));
        }
      }
    };
    const std::true_type yes_{};
    const std::false_type no_{};
    if constexpr (PRO == RP_KSTREAM) {
      stage_chunk(0, 0);  // first weight chunk of phase 2 flies while the LayerNorm below runs
      residual_ln(yes_, yes_, std::integral_constant<bool, A_LO>{}, p.ln_w);
    } else {
      // ---- fused MLP (RP_MLP): x_mid = x + o Wo^T stays in the accumulators acc1; for every pair t of Wi chunks
      //   h[:, 32t .. 32t+31] = GeGLU(LN(x_mid) Wi^T) is produced as ONE MFMA operand fragment per lane (the packing
      //   of Wi puts a lane's 8 consecutive h columns in the two chunks' accumulators) and immediately multiplied into
      //   acc1 += h_t Wo[:, 32t..]^T.  h never leaves the registers: the 2 x 4 I bytes per token of the h round trip
      //   -- the largest HBM stream of the layer -- and the x round trip between the two fused kernels are gone.
      // One LDS stage = [Wi chunk 2t | Wi chunk 2t+1 | Wo slab t-1]; macro-iteration t runs, as one fragment stream,
      //   chunk 2t (with the GeGLU of chunk 2t-1 between its MFMAs), slab t-1 (GeGLU of chunk 2t), chunk 2t+1.
      constexpr bool A_LOW = (TW & T_LEFT_LO) != 0;  // lo(LN(x)) x hi(Wi)
      constexpr bool H_LO = (TM & T_LEFT_LO) != 0;   // lo(h) x hi(Wo)
      constexpr int UNIT_PIECES = MLP_UNIT / 512;
      constexpr int WI_PIECES = F8 ? 2 * CHUNK_PIECES8 : 2 * KS * 2;  // both chunks
      static_assert(UNIT_PIECES % WAVES == 0 && WI_PIECES % WAVES == 0, "stage regions must split over the waves");
      const int n_pairs = p.n_pairs;
      constexpr int UNIT_DMA = UNIT_PIECES / WAVES;  // DMA instructions per wave per stage
      // piece u of this wave's share of stage `stage` for macro-iteration t.  Chunk / slab indices are clamped into
      // range: the first stage has no slab yet, the last one no chunks any more (their pieces are copied again,
      // harmlessly) -- a DMA under a branch would be drained at the join.
      auto stage_piece = [&](auto u_tag, int t, int stage) {
        constexpr int u = decltype(u_tag)::value;
        const int tc = t < n_pairs ? t : n_pairs - 1;
        const int ts = t > 0 ? t - 1 : 0;
        const int piece = wave + WAVES * u;  // wave-uniform
        const u16* src;
        if constexpr (F8 != 0) {  // the chunk's fp16 + e4m3 pieces (not its weight-lo region), then plane 0 (fp16) of the slab
          // A wave copies GROUPS of four consecutive 1 KiB pieces: one base pointer and one M0 per group, the other three
          // pieces through the instruction's immediate offset -- it moves the global AND the LDS address
          // (microbench/dma_offset_probe.hip), and the stage mirrors the packed chunk piece for piece.  Per DMA that is
          // no scalar instruction instead of three (s_add_u32 / s_addc_u32 on the pointer, s_add_i32 on M0): the loop
          // issued 45 of them per iteration.  The region of a group (chunk 2t, chunk 2t+1, slab) is a compile-time fact
          // when all waves' groups of that round fall into the same one, else a wave-uniform select.
          constexpr auto fits = [](int gs) {
            return CHUNK_PIECES8 % gs == 0 && (UNIT_PIECES - WI_PIECES) % gs == 0 && (UNIT_PIECES / gs) % WAVES == 0;
          };
          constexpr int GS = fits(4) ? 4 : (fits(2) ? 2 : 1);  // pieces per group (4 on the hidden-256 kernels)
          constexpr int j = u / GS, q = u % GS;
          const int piece0 = GS * (wave + WAVES * j);  // wave-uniform
          constexpr int lo_piece = GS * (WAVES * j), hi_piece = GS * (WAVES - 1 + WAVES * j) + GS - 1;  // over the waves
          constexpr auto region = [](int pc) { return pc < CHUNK_PIECES8 ? 0 : (pc < WI_PIECES ? 1 : 2); };
          // (integer selects on wave-uniform values: a pointer chosen by control flow would put the DMA under a branch,
          // and the compiler drains a DMA issued under a branch at the join)
          constexpr int r_lo = region(lo_piece), r_hi = region(hi_piece);
          const int in_chunk1 = r_lo == r_hi ? (r_lo == 1 ? 1 : 0) : (piece0 >= CHUNK_PIECES8 ? 1 : 0);
          const size_t off_wi = (size_t)(2 * tc + in_chunk1) * CHUNK_SRC + (size_t)(piece0 - in_chunk1 * CHUNK_PIECES8) * 512;
          const size_t off_slab = (size_t)ts * (2 * NF1 * 512) + (size_t)(piece0 - WI_PIECES) * 512;
          const uintptr_t a_wi = reinterpret_cast<uintptr_t>(p.wi_pk) + 2 * off_wi, a_slab = reinterpret_cast<uintptr_t>(p.wo2_ks) + 2 * off_slab;
          uintptr_t a0;
          if constexpr (r_hi <= 1) a0 = a_wi;
          else if constexpr (r_lo == 2) a0 = a_slab;
          else a0 = piece0 < WI_PIECES ? a_wi : a_slab;
          const u16* src0 = reinterpret_cast<const u16*>(a0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + lane * 8),
                                           (__attribute__((address_space(3))) void*)(&sW[stage][piece0 * 512]), 16, q * 1024, 0);
          return;
        } else if constexpr (u < WI_PIECES / WAVES) {  // [chunk 0..1][ks][frag]: hi pieces of the chunk-major pack
          const int c = piece / (2 * KS), within = piece % (2 * KS);
          src = p.wi_pk + (size_t)(2 * tc + c) * CHUNK_SRC + (within >> 1) * 2048 + (within & 1) * 512;
        } else {
          src = p.wo2_ks + (size_t)ts * (NF1 * 2 * 512) + (piece - WI_PIECES) * 512;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                         (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
      };
      auto stage_unit = [&](int t, int stage) { static_for<UNIT_DMA>([&](auto u) { stage_piece(u, t, stage); }); };
      // WLO: a stage is the half-unit of chunk c: [Wi chunk c with both e4m3 planes][half (c & 1) of the Wo slab of pair
      // c / 2 - 1: NF1 / 2 fragments of plane 0 (fp16), then the same fragments of plane 1 (bf16 of the weight's lo part)]
      auto stage_piece_w = [&](auto u_tag, int c, int stage) {
        constexpr int u = decltype(u_tag)::value;
        const int cc = c < 2 * n_pairs ? c : 2 * n_pairs - 1;  // clamped: the tail stages carry only a slab half
        const int ts = (c >> 1) > 0 ? (c >> 1) - 1 : 0;
        // groups of four consecutive pieces per wave, as stage_piece: the chunk (CHUNK_PIECES8 pieces), then the two runs
        // of NF1 / 2 slab fragments (plane 0, plane 1) -- every run is a multiple of four pieces
        constexpr auto fits = [](int gs) { return CHUNK_PIECES8 % gs == 0 && (NF1 / 2) % gs == 0 && (UNIT_PIECES / gs) % WAVES == 0; };
        constexpr int GS = fits(4) ? 4 : (fits(2) ? 2 : 1);
        constexpr int j = u / GS, q4 = u % GS;
        const int piece0 = GS * (wave + WAVES * j);  // wave-uniform
        constexpr int lo_piece = GS * (WAVES * j), hi_piece = GS * (WAVES - 1 + WAVES * j) + GS - 1;
        constexpr bool all_chunk = hi_piece < CHUNK_PIECES8, all_slab = lo_piece >= CHUNK_PIECES8;
        const size_t off_chunk = (size_t)cc * CHUNK_SRC + (size_t)piece0 * 512;
        const int qs = piece0 - CHUNK_PIECES8, plane = qs / (NF1 / 2), n = qs % (NF1 / 2);  // (powers of two: shifts)
        const size_t off_slab = ((size_t)(ts * 2 + plane) * NF1 + (c & 1) * (NF1 / 2) + n) * 512;
        const uintptr_t a_chunk = reinterpret_cast<uintptr_t>(p.wi_pk) + 2 * off_chunk, a_slab = reinterpret_cast<uintptr_t>(p.wo2_ks) + 2 * off_slab;
        uintptr_t a0;
        if constexpr (all_chunk) a0 = a_chunk;
        else if constexpr (all_slab) a0 = a_slab;
        else a0 = piece0 < CHUNK_PIECES8 ? a_chunk : a_slab;  // (integer select: no DMA under a branch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const u16*>(a0) + lane * 8),
                                         (__attribute__((address_space(3))) void*)(&sW[stage][piece0 * 512]), 16, q4 * 1024, 0);
      };
      if constexpr (WLO) static_for<UNIT_DMA>([&](auto u) { stage_piece_w(u, 0, 0); });
      else stage_unit(0, 0);  // flies while the LayerNorm below runs
      if constexpr (LN_V2) layer_ln(yes_, std::integral_constant<bool, A_LOW>{}, 0);
      else residual_ln(yes_, no_, std::integral_constant<bool, A_LOW>{}, p.ln_w_mlp);
      // The lo fragments of the normalised rows live in AGPRs from here on (an MFMA takes its A / B operands from
      // either file): the 256 architectural VGPRs were short by about that much, and the compiler's own answer was to
      // park fragments in AGPRs and move them back in front of each use -- ~8 issue cycles per v_accvgpr move.
      if (A_LOW && MF == 2 && !F8) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(a_lo[mf][ks]));
      }
      // bf16-valued weights: the fp16 fragments of the normalised rows go to the accumulator file as well (an MFMA takes
      // A / B from either): 64 more VGPRs for the riders of the loop (MLP loop 122.6 k -> 119.8 k cycles per tile).  The
      // fp32-valued kernel has no room for them there (its e4m3 copies of the rows live in AGPRs already).
      if constexpr (F8 == 1) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(a_hi[mf][ks]));
      }

      f32x4 acc_b[2][MF];  // accumulators of the pair's second chunk: [input | gate] fragment x row fragment
      uint2 hold_hi[MF], hold_lo[MF];
      bf16x8 h_hi[MF], h_lo[MF];
      float g_prev[MF][4], g_cur[MF][4];  // GeGLU values of the chunk finished last / of this pair's first chunk
      float gx[MF * 4], gq[MF * 4];       // GeGLU in flight: inputs and the running polynomial / exponential / result
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        hold_hi[mf] = hold_lo[mf] = make_uint2(0u, 0u);
        h_hi[mf] = h_lo[mf] = as_frag(make_uint4(0u, 0u, 0u, 0u));
        acc_b[0][mf] = acc_b[1][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // The GeGLU epilogues are cut into slices that ride along with the fragment stream's steps (8 MFMAs each):
      // value i of a chunk = gelu(input) * gate of (row fragment i / 4, slot i % 4); VPS values per step; a row
      // fragment is split / packed as soon as its four values exist.
      constexpr int NV = MF * 4;
      constexpr int VPS = (NV + KS - 1) / KS;
      auto geglu_slice = [&](const f32x4 (&av)[2][MF], float (&gv)[MF][4], auto slice_tag, auto&& pack) {
        constexpr int sl = decltype(slice_tag)::value;
        // Stage-major: slice sl advances ALL NV values of the chunk by 8 / KS stages of gelu(input) * gate (five
        // polynomial FMAs, exp2, the final FMA, the product with the gate + split / pack), so consecutive vector
        // instructions belong to different values: no instruction waits for the one issued just before it.
        constexpr int STAGES_PER_SLICE = 8 / KS;
        static_assert(KS == 8 || KS == 4, "eight GeGLU stages over the k-steps of a chunk");
        static_for<STAGES_PER_SLICE>([&](auto u_tag) {
          constexpr int st = sl * STAGES_PER_SLICE + decltype(u_tag)::value;
          static_for<NV>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            constexpr int mf = i >> 2, r = i & 3;
            if constexpr (st == 0) {
              gx[i] = av[0][mf][r];
              gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
            } else if constexpr (st < 5) {
              gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), st);
            } else if constexpr (st == 5) {
              gq[i] = __builtin_amdgcn_exp2f(gq[i]);
            } else if constexpr (st == 6) {
              gq[i] = gelu_erf_finish(gq[i], gx[i]);
            } else {
              gv[mf][r] = gq[i] * av[1][mf][r];
            }
          });
          if constexpr (st == 7) static_for<MF>([&](auto mf_tag) { pack(mf_tag); });
        });
      };
      // The same GeGLU as micro-operations [B, E) of its stage-major list (operation o = stage o / NV of value o % NV, 8 NV
      // in all; the pack follows the last one): the F8 stream spreads a chunk's GeGLU evenly over ALL steps of the next
      // chunk -- fp16 and e4m3 steps last 64 pipe cycles each and hide about as many vector instructions.
      auto geglu_ops = [&](const f32x4 (&av)[2][MF], float (&gv)[MF][4], auto begin_tag, auto end_tag, auto&& pack) {
        constexpr int B = decltype(begin_tag)::value, E = decltype(end_tag)::value;
        static_for<E - B>([&](auto o_tag) {
          constexpr int o = B + decltype(o_tag)::value;
          constexpr int st = o / NV, i = o % NV, mf = i >> 2, r = i & 3;
          if constexpr (st == 0) {
            gx[i] = av[0][mf][r];
            gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
          } else if constexpr (st < 5) {
            gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), st);
          } else if constexpr (st == 5) {
            gq[i] = __builtin_amdgcn_exp2f(gq[i]);
          } else if constexpr (st == 6) {
            gq[i] = gelu_erf_finish(gq[i], gx[i]);
          } else {
            gv[mf][r] = gq[i] * av[1][mf][r];
          }
          if constexpr (o == 8 * NV - 1) static_for<MF>([&](auto mf_tag) { pack(mf_tag); });
        });
      };
      auto pack_h = [&](auto mf_tag) {  // second half of a pair -> the pair's h fragment of this row fragment
        constexpr int mf = decltype(mf_tag)::value;
        uint2 h2, l2;
        if constexpr (F8) split4_f16(g_prev[mf], h2, l2);
        else split4x<H_LO, H16>(g_prev[mf], h2, l2);
        h_hi[mf] = as_frag(make_uint4(hold_hi[mf].x, hold_hi[mf].y, h2.x, h2.y));
        h_lo[mf] = as_frag(make_uint4(hold_lo[mf].x, hold_lo[mf].y, l2.x, l2.y));
      };
      auto pack_hold = [&](auto mf_tag) {
        constexpr int mf = decltype(mf_tag)::value;
        if constexpr (F8) split4_f16(g_cur[mf], hold_hi[mf], hold_lo[mf]);
        else split4x<H_LO, H16>(g_cur[mf], hold_hi[mf], hold_lo[mf]);
      };
      // The GeGLU of one chunk + the split / pack of its values as ONE list of RB_OPS = 9 NV micro-operations, so that it
      // can be cut anywhere (OPK_X_REBAL spreads it over the steps that follow the chunk in proportion to their MFMA pipe
      // time): operations [0, 7 NV) = stages 0..6 of value o % NV; then per row fragment mf: 4 x (stage 7 = the product
      // with the gate) and 4 pack sub-operations (hi pair | lo parts 0, 1 | lo parts 2, 3 | lo pair + commit).
      // IS_H: the chunk closes the pair (pack_h: h fragments from the held first half), else pack_hold.
      constexpr int RB_OPS = 9 * MF * 4;
      uint2 pk_hi[MF];
      float pk_d[MF][4];
      auto geglu_ops72 = [&](const f32x4 (&av)[2][MF], float (&gv)[MF][4], auto begin_tag, auto end_tag, auto is_h_tag) {
        constexpr int B = decltype(begin_tag)::value, E = decltype(end_tag)::value;
        constexpr bool IS_H = decltype(is_h_tag)::value;
        static_for<(E > B ? E - B : 0)>([&](auto o_tag) {
          constexpr int o = B + decltype(o_tag)::value;
          if constexpr (o < 7 * NV) {
            constexpr int st = o / NV, i = o % NV, mf = i >> 2, r = i & 3;
            if constexpr (st == 0) {
              gx[i] = av[0][mf][r];
              gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
            } else if constexpr (st < 5) {
              gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), st);
            } else if constexpr (st == 5) {
              gq[i] = __builtin_amdgcn_exp2f(gq[i]);
            } else {
              gq[i] = gelu_erf_finish(gq[i], gx[i]);
            }
          } else {
            constexpr int q = o - 7 * NV, mf = q / 8, u = q % 8;
            if constexpr (u < 4) {
              gv[mf][u] = gq[4 * mf + u] * av[1][mf][u];
            } else if constexpr (u == 4) {
              pk_hi[mf].x = pack_f16x2(gv[mf][0], gv[mf][1]);
              pk_hi[mf].y = pack_f16x2(gv[mf][2], gv[mf][3]);
            } else if constexpr (u == 5) {
              pk_d[mf][0] = sub_f16_half<0>(gv[mf][0], pk_hi[mf].x);
              pk_d[mf][1] = sub_f16_half<1>(gv[mf][1], pk_hi[mf].x);
            } else if constexpr (u == 6) {
              pk_d[mf][2] = sub_f16_half<0>(gv[mf][2], pk_hi[mf].y);
              pk_d[mf][3] = sub_f16_half<1>(gv[mf][3], pk_hi[mf].y);
            } else {
              const uint2 lo = make_uint2(pack_f16x2(pk_d[mf][0], pk_d[mf][1]), pack_f16x2(pk_d[mf][2], pk_d[mf][3]));
              if constexpr (IS_H) {
                h_hi[mf] = as_frag(make_uint4(hold_hi[mf].x, hold_hi[mf].y, pk_hi[mf].x, pk_hi[mf].y));
                h_lo[mf] = as_frag(make_uint4(hold_lo[mf].x, hold_lo[mf].y, lo.x, lo.y));
              } else {
                hold_hi[mf] = pk_hi[mf];
                hold_lo[mf] = lo;
              }
            }
          }
        });
      };
      auto no_rd = [](auto, f32x4&) {};
      auto chunk_step = [&](f32x4 (&acc)[2][MF], auto ks_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
        constexpr int ks = decltype(ks_tag)::value;
        // the first MFMA of an accumulator takes the constant 0 as its C operand (no zero-fill of the registers)
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F8) {  // fp16 product only; the e4m3 lo product follows in chunk_step8
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[0][mf] = mfma16h(w0, a_hi[mf][ks], ks == 0 ? zero : acc[0][mf]);
          rd(std::integral_constant<int, 0>{}, acc[0][0]);
          acc[1][0] = mfma16h(w1, a_hi[0][ks], ks == 0 ? zero : acc[1][0]);
          rd(std::integral_constant<int, 1>{}, acc[0][MF - 1]);
#pragma unroll
          for (int mf = 1; mf < MF; ++mf) acc[1][mf] = mfma16h(w1, a_hi[mf][ks], ks == 0 ? zero : acc[1][mf]);
          return;
        }
        if (A_LOW) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[0][mf] = mfma16(w0, a_lo[mf][ks], ks == 0 ? zero : acc[0][mf]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc[1][mf] = mfma16(w1, a_lo[mf][ks], ks == 0 ? zero : acc[1][mf]);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[0][mf] = mfma16x<H16>(w0, a_hi[mf][ks], (ks == 0 && !A_LOW) ? zero : acc[0][mf]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[1][mf] = mfma16x<H16>(w1, a_hi[mf][ks], (ks == 0 && !A_LOW) ? zero : acc[1][mf]);
      };
      // F8: lo(LN(x)) x Wi as e4m3, fragment nf of the chunk, K-step s8: (w0, w1) are the fragment's two halves
      auto chunk_step8 = [&](f32x4 (&acc)[2][MF], auto nf_tag, auto s8_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
        constexpr int nf = decltype(nf_tag)::value, s8 = decltype(s8_tag)::value;
        const i32x8 w8 = f8_frag(w0, w1);
        acc[nf][0] = mfma8<true>(w8, a_lo8[0][s8 < NS8 ? s8 : 0], acc[nf][0]);
        rd(std::integral_constant<int, 0>{}, acc[nf][0]);
        rd(std::integral_constant<int, 1>{}, acc[nf][0]);
#pragma unroll
        for (int mf = 1; mf < MF; ++mf) acc[nf][mf] = mfma8<true>(w8, a_lo8[mf][s8 < NS8 ? s8 : 0], acc[nf][mf]);
      };
      auto slab_pair = [&](auto nf_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
        constexpr int nf = decltype(nf_tag)::value;
        if constexpr (F8) {  // h: (hi, lo) fp16 pair, both on the fp16 shape (K = 32 per step is too short for the fp8 one)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w0, h_lo[mf], acc1[nf][mf]);
          rd(std::integral_constant<int, 0>{}, acc1[nf][0]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16h(w1, h_lo[mf], acc1[nf + 1][mf]);
          rd(std::integral_constant<int, 1>{}, acc1[nf + 1][0]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w0, h_hi[mf], acc1[nf][mf]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16h(w1, h_hi[mf], acc1[nf + 1][mf]);
          return;
        }
        if (H_LO) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16(w0, h_lo[mf], acc1[nf][mf]);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16(w1, h_lo[mf], acc1[nf + 1][mf]);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16x<H16>(w0, h_hi[mf], acc1[nf][mf]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[nf + 1][mf] = mfma16x<H16>(w1, h_hi[mf], acc1[nf + 1][mf]);
      };
      // WLO: e4m3(LN(x)) x lo(Wi), and one output fragment of the slab with all three terms:
      //   lo(h) x Wo, h x lo(Wo) (w1 = that plane's fragment: unscaled fp16, see pack_kstream_f8_kernel) and h x Wo, all
      //   on the fp16 shape
      auto chunk_step8w = [&](f32x4 (&acc)[2][MF], auto nf_tag, auto s8_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
        constexpr int nf = decltype(nf_tag)::value, s8 = decltype(s8_tag)::value;
        const i32x8 w8 = f8_frag(w0, w1);
        acc[nf][0] = mfma8w<false>(w8, a_h8[0][s8 < NS8 ? s8 : 0], acc[nf][0]);
        rd(std::integral_constant<int, 0>{}, acc[nf][0]);
        rd(std::integral_constant<int, 1>{}, acc[nf][0]);
#pragma unroll
        for (int mf = 1; mf < MF; ++mf) acc[nf][mf] = mfma8w<false>(w8, a_h8[mf][s8 < NS8 ? s8 : 0], acc[nf][mf]);
      };
      auto slab_one = [&](auto nf_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
        constexpr int nf = decltype(nf_tag)::value;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w0, h_lo[mf], acc1[nf][mf]);
        rd(std::integral_constant<int, 0>{}, acc1[nf][0]);
        rd(std::integral_constant<int, 1>{}, acc1[nf][MF - 1]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w1, h_hi[mf], acc1[nf][mf]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = mfma16h(w0, h_hi[mf], acc1[nf][mf]);
      };
      // one MFMA : up to three vector instructions inside a step (the slice's VALU spread between its MFMAs)
      auto interleave_step = [&]() {
        constexpr int STEP_MFMA = 2 * MF * ((A_LOW || H_LO) ? 2 : 1);
#pragma unroll
        for (int i = 0; i < STEP_MFMA; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      };
      // F8: a step holds n_mfma MFMAs (4 fp16 or 2 e4m3 of a chunk: 64 pipe cycles; 8 fp16 of a slab) and up to
      // n_mfma x per vector instructions of the GeGLU slice riding on it
      auto interleave_n = [&](auto n_tag, auto per_tag) {
#pragma unroll
        for (int i = 0; i < decltype(n_tag)::value; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, decltype(per_tag)::value, 0);
        }
      };
      constexpr int DEPTH = 2;  // fragment groups in flight ahead of the one being consumed (a step = 8 MFMAs)
      constexpr int DEPTH8 = 4;  // F8: a chunk step is 4 fp16 / 2 e4m3 MFMAs = half the pipe time, twice the steps ahead
      // The stage index is a RUNTIME value here (one copy of the loop body): the fragment reads are inline asm with
      // the stage's base address in a register, so the compiler has no DMA-vs-read aliasing to resolve, and a body
      // unrolled by two would permute the 128 accumulator registers between its copies on every back edge.
      // Riders (round 4).  A wave that is alone on its SIMD issues ONE instruction per four cycles, in order: a 16-cycle
      // MFMA leaves three slots behind it for everything else of the step (vector instructions, fragment reads, the DMA,
      // waits), a 32-cycle e4m3 MFMA seven.  So the GeGLU of a chunk is cut into RB_OPS micro-operations (geglu_ops72) and
      // spread in proportion to MFMA pipe time over the units of 64 cycles that FOLLOW the chunk.  bf16-valued weights
      // (20 units) -- chunk 2t: the 12 steps of chunk 2t+1 and the first half of the slab steps; chunk 2t+1: the second half and the 12
      // steps of the NEXT iteration's chunk 2t+2 (its accumulators cross the back edge as VGPR values, nbv): 3.6 operations
      // per unit where the stream had 0 on chunk 2t, 5.3 + the accumulator reads on chunk 2t+1 and 4 + the pack on the
      // slab; MLP loop 125.5 k -> 118.0 k cycles per tile.  fp32-valued weights (28 units): see half_iter.
      // (hidden 256: 12 chunk steps + 8 slab steps of two units = 20 units, fp32-valued weights 16 + 8 x 1.5 = 28; hidden
      // 128: half of each)
      constexpr int RB_CS = F8 ? F8Chunk<KS, WLO>::STEPS : KS;  // chunk steps = units per chunk
      constexpr int RB_NSL = NF1 / 2;                            // slab steps (per half-iteration with fp32-valued weights)
      constexpr int RB_UNITS = WLO ? RB_CS + 3 * RB_NSL / 2 : RB_CS + RB_NSL;
      f32x4 nbv[2][MF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) nbv[0][mf] = nbv[1][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto macro = [&](int t, int cur, auto slab_tag) {
        constexpr bool SLAB = decltype(slab_tag)::value;  // false only for t = 0
        using Off = MlpStreamOff<KS, NF1, SLAB>;
        constexpr int NS = Off::NS;
        // The next stage's DMA instructions are spread over the first steps of the stream, one per step: issued in
        // one burst at the top they cost this (only) wave of the SIMD their full issue time with no MFMA in flight.
        // na: chunk 2t, written by its first k-step (C operand = 0).  Chunk 2t+1 accumulates straight into acc_b: the
        // GeGLU of chunk 2t-1 (the last reader of acc_b's old contents) is over after the first KS steps.
        f32x4 na[2][MF];
        if constexpr (F8) {
          // F8 stream order: chunk 2t -> na | chunk 2t+1 -> nb | slab t-1 (h of pair t-1), riders as above.  A finished
          // chunk leaves the accumulator file ONCE ("+v"): left to the register allocator, the MFMA destinations of
          // the next chunk landed on tiles of acc1 and those were saved and restored through VGPRs on every iteration
          // (64 v_accvgpr moves per iteration, 32 of them shuffles; 32 now).
          using Off8 = MlpStreamOff8<KS, NF1, SLAB>;
          using C8 = F8Chunk<KS>;
          constexpr int CS = C8::STEPS;
          f32x4 nb[2][MF];
          constexpr int NSTEPS8 = 2 * CS + NS;
          auto unit_body = [&](auto step_tag, bf16x8& w0, bf16x8& w1, auto&&... rd_opt) {
            constexpr int s = decltype(step_tag)::value;
            auto&& rd = rd_or(no_rd, rd_opt...);
#ifdef OPK_SEG_TIMING  // cycles of the three segments of an iteration (chunk 2t | chunk 2t+1 + GeGLU | slab + GeGLU)
            if constexpr (s == 0 || s == CS || s == 2 * CS) {
              const unsigned long long now = __builtin_readcyclecounter();
              if constexpr (s > 0) opk_seg[s == CS ? 0 : 1] += now - opk_seg_t;
              opk_seg_t = now;
            }
#endif
            if constexpr (s < UNIT_DMA) stage_piece(step_tag, t + 1, cur ^ 1);
            if constexpr (s < 2 * CS) {  // a chunk step
              constexpr bool FIRST_CHUNK = s < CS;
              constexpr int cs = FIRST_CHUNK ? s : s - CS;
              auto& acc = *(FIRST_CHUNK ? &na : &nb);
              if constexpr (!C8::is_f8(cs)) chunk_step(acc, std::integral_constant<int, C8::ks(cs)>{}, w0, w1, rd);
              else chunk_step8(acc, std::integral_constant<int, C8::nf(cs)>{}, std::integral_constant<int, C8::s8(cs)>{}, w0, w1, rd);
              if constexpr (!FIRST_CHUNK && cs == CS - 1) {  // chunk 2t+1 is complete: it crosses the back edge in VGPRs
#pragma unroll
                for (int nf_ = 0; nf_ < 2; ++nf_)
#pragma unroll
                  for (int mf_ = 0; mf_ < MF; ++mf_) {
                    nbv[nf_][mf_] = nb[nf_][mf_];
                    asm volatile("" : "+v"(nbv[nf_][mf_]));
                  }
              }
              if constexpr (FIRST_CHUNK && cs == CS - 1) {
#pragma unroll
                for (int nf_ = 0; nf_ < 2; ++nf_)
#pragma unroll
                  for (int mf_ = 0; mf_ < MF; ++mf_) asm volatile("" : "+v"(na[nf_][mf_]));
              }
              if constexpr (!FIRST_CHUNK) {  // GeGLU(2t): units 0..11 of its 20 (t = 0: all of it, there is no slab to ride on)
                constexpr int OB = SLAB ? cs * RB_OPS / RB_UNITS : cs * RB_OPS / CS, OE = SLAB ? (cs + 1) * RB_OPS / RB_UNITS : (cs + 1) * RB_OPS / CS;
                geglu_ops72(na, g_cur, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, no_);
              } else if constexpr (SLAB) {  // GeGLU(2t-1) of the previous iteration: units 8..19 of its 20
                constexpr int OB = (RB_NSL + cs) * RB_OPS / RB_UNITS, OE = (RB_NSL + cs + 1) * RB_OPS / RB_UNITS;
                geglu_ops72(nbv, g_prev, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, yes_);
              }
              if constexpr (!C8::is_f8(cs)) interleave_n(std::integral_constant<int, 2 * MF>{}, std::integral_constant<int, 2>{});
              else interleave_n(std::integral_constant<int, MF>{}, std::integral_constant<int, 4>{});
            } else {  // slab t-1: steps 0..3 carry units 12..19 of GeGLU(2t), steps 4..7 units 0..7 of GeGLU(2t+1)
              constexpr int i = s - 2 * CS;
              slab_pair(std::integral_constant<int, 2 * i>{}, w0, w1, rd);
              if constexpr (i < RB_NSL / 2) {
                constexpr int OB = (RB_CS + 2 * i) * RB_OPS / RB_UNITS, OE = (RB_CS + 2 * i + 2) * RB_OPS / RB_UNITS;
                geglu_ops72(na, g_cur, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, no_);
              } else {
                constexpr int OB = (2 * (i - RB_NSL / 2)) * RB_OPS / RB_UNITS, OE = (2 * (i - RB_NSL / 2) + 2) * RB_OPS / RB_UNITS;
                geglu_ops72(nbv, g_prev, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, yes_);
              }
              interleave_n(std::integral_constant<int, 4 * MF>{}, std::integral_constant<int, 2>{});
            }
          };
          // (reads between the MFMAs, frag_stream2i, are 5 % faster on the bare stream here too, but with the riders placed
          // by sched_group_barrier they measure +5 % SLOWER, and equal without the barriers: this form stays)
          frag_stream2<NSTEPS8, DEPTH8, Off8>(cur ? lds_stage[1] : lds_stage[0], unit_body);
#ifdef OPK_SEG_TIMING
          opk_seg[2] += __builtin_readcyclecounter() - opk_seg_t;
#endif
          if constexpr (!SLAB)  // first pair: the slab-borne units 0..7 of GeGLU(1) have no slab to ride on
            geglu_ops72(nbv, g_prev, std::integral_constant<int, 0>{}, std::integral_constant<int, RB_NSL * RB_OPS / RB_UNITS>{}, yes_);
        } else
        frag_stream2<2 * KS + NS, DEPTH, Off>(cur ? lds_stage[1] : lds_stage[0], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
          constexpr int s = decltype(step_tag)::value;
          if constexpr (s < UNIT_DMA) stage_piece(step_tag, t + 1, cur ^ 1);
          if constexpr (s < KS) {  // chunk 2t, with the GeGLU of chunk 2t-1 (-> h of pair t-1 ready for the slab)
            chunk_step(na, std::integral_constant<int, s>{}, w0, w1, no_rd);
            if constexpr (SLAB) geglu_slice(acc_b, g_prev, std::integral_constant<int, s>{}, pack_h);
          } else if constexpr (s < KS + NS) {  // slab t-1, with the GeGLU of chunk 2t
            slab_pair(std::integral_constant<int, 2 * (s - KS)>{}, w0, w1, no_rd);
            geglu_slice(na, g_cur, std::integral_constant<int, s - KS>{}, pack_hold);
          } else {  // chunk 2t+1 (t = 0: with the GeGLU of chunk 0)
            chunk_step(acc_b, std::integral_constant<int, s - KS - NS>{}, w0, w1, no_rd);
            if constexpr (!SLAB) geglu_slice(na, g_cur, std::integral_constant<int, s - KS>{}, pack_hold);
          }
          interleave_step();
        });
#ifdef OPK_TIMING
        const unsigned long long opk_w0 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next stage has landed (no other VMEM in this loop)
        __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
        opk_wait += __builtin_readcyclecounter() - opk_w0;
#endif
      };
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // stage 0 has landed
      OPK_STAMP(2);
      if constexpr (F8 != 0) set_overflowing_conversions();  // the MLP loop converts h to fp16 only: out of range = Inf, not 65504
      if constexpr (WLO) {
        // ---- fp32-valued weights: half-iterations.  Half hb of iteration t streams stage hb = [chunk 2t + hb | half hb
        // of slab t-1]: the chunk's KS fp16 steps + 2 x KS/2 e4m3 steps (lo(LN(x)) x Wi, LN(x) x lo(Wi)), then NF1 / 2
        // slab steps of ONE output fragment each with all three terms (6 MFMAs); meanwhile the DMA fills the other
        // stage with the next half-unit.  GeGLU(2t) rides on the slab steps of half 0 and the chunk steps of half 1,
        // GeGLU(2t+1) on the slab steps of half 1 and closes h of pair t.  Iteration 0 multiplies the slab by h = 0.
        using C8 = F8Chunk<KS, true>;
        constexpr int CS = C8::STEPS, NSH = NF1 / 2;
        struct OffW {
          static constexpr int at(int st, int j) { return st < CS ? C8::off(st, j) : C8::BYTES + (j * NSH + (st - CS)) * 1024; }
        };
        auto half_iter = [&](int c, auto hb_tag, auto with_chunk_tag, f32x4 (&na)[2][MF], f32x4 (&nb)[2][MF]) {
          constexpr int hb = decltype(hb_tag)::value;
          constexpr bool CHUNK = decltype(with_chunk_tag)::value;  // false: tail (slab steps only)
          constexpr int S0 = CHUNK ? 0 : CS;
          // tail: the stream is shorter than the DMA list -- the first tail half requests the last slab half up front,
          // the second one has nothing left to request
          if constexpr (!CHUNK && hb == 0) static_for<UNIT_DMA>([&](auto u) { stage_piece_w(u, c + 1, hb ^ 1); });
          frag_stream2i<CS + NSH - S0, FRAG_ILV, OffShift<OffW, S0>>(lds_stage[hb], [&](auto step_tag, bf16x8& w0, bf16x8& w1, auto&&... rd_opt) {
            auto&& rd = rd_or(no_rd, rd_opt...);
            constexpr int sr = decltype(step_tag)::value, st = sr + S0;
            if constexpr (CHUNK && sr < UNIT_DMA) stage_piece_w(step_tag, c + 1, hb ^ 1);
            if constexpr (st < CS) {
              auto& acc = *(hb == 0 ? &na : &nb);
              if constexpr (!C8::is_f8(st)) chunk_step(acc, std::integral_constant<int, C8::ks(st)>{}, w0, w1, rd);
              else if constexpr (!C8::is_wlo(st)) chunk_step8(acc, std::integral_constant<int, C8::nf(st)>{}, std::integral_constant<int, C8::s8(st)>{}, w0, w1, rd);
              else chunk_step8w(acc, std::integral_constant<int, C8::nf(st)>{}, std::integral_constant<int, C8::s8(st)>{}, w0, w1, rd);
              // Riders in proportion to pipe time: a chunk step is one unit of 64 cycles, a slab step (6 MFMAs) 1.5.  The
              // GeGLU of chunk 2t rides on the 12 + 16 units that follow it (slab half 0, chunk 2t+1): 2.6 operations per
              // unit where it had 4 per slab step and 3 per chunk step.  A finished chunk leaves the accumulator file once.
              if constexpr (st == CS - 1) {
#pragma unroll
                for (int nf_ = 0; nf_ < 2; ++nf_)
#pragma unroll
                  for (int mf_ = 0; mf_ < MF; ++mf_) {
                    if constexpr (hb == 1) nbv[nf_][mf_] = nb[nf_][mf_];
                    asm volatile("" : "+v"((hb == 0 ? na : nbv)[nf_][mf_]));
                  }
              }
              // Chunk 2t+1's GeGLU stays inside the iteration, all of it on slab half 1 (nothing crosses the back edge):
              // carried into the next iteration's chunk 2t+2, as the bf16-valued kernel does, it costs this kernel -- 256
              // VGPRs + 208 AGPRs in use -- 48 more accumulator shuffles per iteration than the emptier steps return
              // (MLP loop 166.2 k cycles per tile before, 161.6 k this way, 173.5 k carried).  With the fragment reads
              // between the MFMAs (frag_stream2i): 154.8 k two groups ahead, 157.2 k four ahead in the isolated launch --
              // and in the whole forward (same box, alternating runs, ten layers with their own weights) four ahead is the
              // faster one: 36.5 k pairs/s against 35.7 k two ahead and 36.1 k for the round-3 kernel.
              if constexpr (hb == 1) {
                constexpr int OB = (3 * RB_NSL / 2 + st) * RB_OPS / RB_UNITS, OE = (3 * RB_NSL / 2 + st + 1) * RB_OPS / RB_UNITS;
                geglu_ops72(na, g_cur, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, no_);
              }
            } else {
              constexpr int i = st - CS;
              slab_one(std::integral_constant<int, hb * NSH + i>{}, w0, w1, rd);
              if constexpr (CHUNK && hb == 0) {
                constexpr int OB = (3 * i / 2) * RB_OPS / RB_UNITS, OE = (3 * (i + 1) / 2) * RB_OPS / RB_UNITS;
                geglu_ops72(na, g_cur, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, no_);
              }
              if constexpr (CHUNK && hb == 1) {
                constexpr int OB = i * RB_OPS / NSH, OE = (i + 1) * RB_OPS / NSH;
                geglu_ops72(nbv, g_prev, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{}, yes_);
              }
            }
          });
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the other stage has landed
          __builtin_amdgcn_s_barrier();
        };
        const std::integral_constant<int, 0> h0{};
        const std::integral_constant<int, 1> h1{};
        int t = 0;
        do {
          f32x4 na[2][MF], nb[2][MF];
          half_iter(2 * t, h0, yes_, na, nb);
          half_iter(2 * t + 1, h1, yes_, na, nb);
        } while (++t < n_pairs);
        {  // tail: slab of the last pair, half by half
          f32x4 na[2][MF], nb[2][MF];
          half_iter(2 * n_pairs, h0, no_, na, nb);
          half_iter(2 * n_pairs + 1, h1, no_, na, nb);
        }
      } else {
      macro(0, 0, no_);
      {  // n_pairs is even (checked on the host): at least one more iteration, and the tail reads stage 0.  Written as
        // do-while: around a loop that may run zero times the compiler parks accumulator values in scratch.
        int t = 1;
        do {
          macro(t, t & 1, yes_);
        } while (++t < n_pairs);
      }
      }
      if constexpr (!WLO) {  // tail: the last pair's h fragments and their slab (stage 0 of the ring)
        if constexpr (!F8) static_for<KS>([&](auto sl) { geglu_slice(acc_b, g_prev, sl, pack_h); });
        if constexpr (F8 == 1)  // units 8..19 of the last chunk's GeGLU: no next iteration to ride on
          geglu_ops72(nbv, g_prev, std::integral_constant<int, RB_NSL * RB_OPS / RB_UNITS>{}, std::integral_constant<int, RB_OPS>{}, yes_);
        struct TailOff {
          static constexpr int at(int s, int j) { return (F8 ? 2 * F8Chunk<KS>::BYTES : 2 * KS * 2048) + (s * 2 + j) * 1024; }
        };
        frag_stream2<NF1 / 2, DEPTH, TailOff>(lds_stage[0], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
          slab_pair(std::integral_constant<int, 2 * decltype(step_tag)::value>{}, w0, w1, no_rd);
        });
      }
      __builtin_amdgcn_s_barrier();  // every wave is done with the ring: the chunk loop may reuse stage 0
      OPK_STAMP(3);
      if constexpr (F8 != 0) set_saturating_conversions();
      if constexpr (EPI == RE_NONE) {
        if (FIN_HEAD && p.fin_ln != nullptr) final_head();
        else residual_ln(no_, yes_, no_, nullptr);  // acc1 = x + o Wo^T + h Wo^T: the layer's output
        OPK_STAMP(4);
        OPK_DUMP();
        return;
      } else {
        if constexpr (QKV_PAIRS) stage_pair(0, 0);
        else stage_chunk(0, 0);
        if constexpr (QKV2) stage_chunk_at(1, 0, STAGE);  // the first PAIR of chunks
        if (ROPE_PRELOAD) rope_preload();
        if constexpr (LN_V2) {
          layer_ln(no_, std::integral_constant<bool, A_LO>{}, 1);
#ifdef OPK_TIMING
          opk_x[2] = __builtin_readcyclecounter();
#endif
          // chunk 0 and the RoPE rows have landed (they had the whole LayerNorm); only then the 2 x NF1 row stores,
          // which nothing waits for before the end of the first chunk
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef OPK_TIMING
          opk_x[3] = __builtin_readcyclecounter();
#endif
          store_rows();
        } else {
          residual_ln(no_, yes_, std::integral_constant<bool, A_LO>{}, p.ln_w);
        }
        OPK_STAMP(4);
      }
    }
  } else {
    stage_chunk(0, 0);
  }

  // ---- prologue (layer 0, RP_SPLIT): this wave's rows of x as fragments, no LayerNorm (attn_norm is Identity) ----
  if (PRO == RP_SPLIT && p.emb_table != nullptr) {
    // x0 = LayerNorm(E[id]) (alignment rows: zeros), packed arithmetic as in layer_ln; explicit instructions, so that
    // every instantiation (kernel set, rows per wave) produces the same bits for a row
    const float* lnw_g = p.ln_w + g * 8;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int row = m0 + mf * 16 + l15;
      const int tok = p.row_tok[row];
      int id = p.emb_ids[tok < 0 ? 0 : tok];
      id = id < 0 ? 0 : (id >= p.emb_vocab ? p.emb_vocab - 1 : id);
      const float* src = p.emb_table + (size_t)id * K + g * 8;
      const float live = tok < 0 ? 0.f : 1.f;
      const f32x2 live2 = f32x2{live, live};
      f32x2 v[4 * KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 f0 = *reinterpret_cast<const float4*>(src + ks * 32);
        const float4 f1 = *reinterpret_cast<const float4*>(src + ks * 32 + 4);
        v[4 * ks + 0] = pk_mul(f32x2{f0.x, f0.y}, live2);
        v[4 * ks + 1] = pk_mul(f32x2{f0.z, f0.w}, live2);
        v[4 * ks + 2] = pk_mul(f32x2{f1.x, f1.y}, live2);
        v[4 * ks + 3] = pk_mul(f32x2{f1.z, f1.w}, live2);
      }
      f32x2 s4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int i = 4; i < 4 * KS; ++i) s4[i & 3] = pk_add(s4[i & 3], v[i]);
      const f32x2 st = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
      float sum = st.x + st.y;
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / (float)K);
      const f32x2 m2 = f32x2{mean, mean};
      f32x2 q4[4];
#pragma unroll
      for (int i = 0; i < 4 * KS; ++i) {
        v[i] = pk_sub(v[i], m2);
        q4[i & 3] = i < 4 ? pk_mul(v[i], v[i]) : pk_fma(v[i], v[i], q4[i & 3]);
      }
      const f32x2 qt = pk_add(pk_add(q4[0], q4[1]), pk_add(q4[2], q4[3]));
      float q = qt.x + qt.y;
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = 1.0f / sqrtf(q * (1.0f / (float)K) + p.eps);
      const f32x2 r2 = f32x2{rstd, rstd};
      float* xrow = p.x_io + (size_t)row * K + g * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 w0 = *reinterpret_cast<const float4*>(lnw_g + ks * 32);
        const float4 w1 = *reinterpret_cast<const float4*>(lnw_g + ks * 32 + 4);
        const f32x2 lw[4] = {f32x2{w0.x, w0.y}, f32x2{w0.z, w0.w}, f32x2{w1.x, w1.y}, f32x2{w1.z, w1.w}};
        f32x2 y[4];
        uint32_t hb[4], lb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = pk_mul(pk_mul(v[4 * ks + j], r2), lw[j]);
          split2x_pk<A_LO, H16>(y[j], hb[j], lb[j]);
        }
        store_stream16(xrow + ks * 32, make_float4(y[0].x, y[0].y, y[1].x, y[1].y));
        store_stream16(xrow + ks * 32 + 4, make_float4(y[2].x, y[2].y, y[3].x, y[3].y));
        a_hi[mf][ks] = as_frag(make_uint4(hb[0], hb[1], hb[2], hb[3]));
        a_lo[mf][ks] = as_frag(make_uint4(lb[0], lb[1], lb[2], lb[3]));
      }
    }
  } else if (PRO == RP_SPLIT) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const size_t row = (size_t)(m0 + mf * 16 + l15);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 f0 = load_stream_f4(p.x_in + row * K + ks * 32 + g * 8);
        const float4 f1 = load_stream_f4(p.x_in + row * K + ks * 32 + g * 8 + 4);
        const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        pack8x<A_LO, H16>(v, a_hi[mf][ks], a_lo[mf][ks]);
      }
    }
  }
  // Numerics of a narrower policy on this (wider) instantiation: the lo fragments of the in-register operand are
  // ANDed with a launch-constant mask (all ones, or zero: their product term then adds exact zeros, bit-identical to
  // the kernel that omits the term).  Straight-line on purpose: a branch here makes the compiler keep two copies of
  // the 64 fragment registers and spill.
  if (A_LO && PRO != RP_MLP) {
    const unsigned keep = p.zero_a_lo ? 0u : 0xffffffffu;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FragU f;
        f.v = a_lo[mf][ks];
        f.u = make_uint4(f.u.x & keep, f.u.y & keep, f.u.z & keep, f.u.w & keep);
        a_lo[mf][ks] = f.v;
      }
  }
  if (EPI == RE_QKV && !ROPE_PRELOAD) rope_rows();
  if constexpr (LN_V2) __builtin_amdgcn_s_barrier();  // (this wave's share of chunk 0 was waited for above)
  else __syncthreads();  // chunk 0 has landed (the barrier's release waits for this wave's DMA: vmcnt(0))

  // ---- F8 kernel sets: the q / k / v^T projection as ONE fragment stream per chunk PAIR ---------------------------
  // (round 4) One chunk per barrier left this loop at 2.3 x its MFMA pipe time (55.7 k cycles per tile for 24.6 k of pipe,
  // fp32-valued weights) with next to nothing of it spent waiting for the DMA or the barrier: the wave is alone on its
  // SIMD, and the deferred epilogue placed by sched_group_barrier, a read pipeline refilled every 1024 pipe cycles and
  // 16 accumulator moves bunched behind every chunk is what it issued in between.  Here: a stage holds chunks 2t, 2t+1
  // (one RoPE head / one v^T head), the pair is one stream of 2 CS steps with the fragment reads between the MFMAs
  // (frag_stream2i), the next pair's DMA is one instruction per step over the first half, the epilogue of pair t-1 is
  // a list of micro-operations cut evenly over the steps (rope + scale of one value pair, or one (hi, lo) split of two
  // values), and its 8 stores go out behind the last DMA of the iteration -- the counted wait in front of the barrier
  // still leaves them in flight.  Values are bit-identical to the one-chunk loop (same operations in the same order).
  if constexpr (QKV_PAIRS) {
    using C8 = F8Chunk<KS, WLO>;
    constexpr int CS = C8::STEPS, NST = 2 * CS;
    static_assert(PAIR_DMA <= CS, "the pair's DMA instructions ride on the first chunk's steps");
    struct OffP {
      static constexpr int at(int st, int j) { return st < CS ? C8::off(st, j) : C8::BYTES + C8::off(st - CS, j); }
    };
    constexpr bool O0_LO = (OLO & 1) != 0, O1_LO = (OLO & 2) != 0, O2_LO = (OLO & 4) != 0;
    constexpr bool QK_LO = O0_LO || O1_LO;
    const std::true_type yes_{};
    const std::false_type no_{};
    const int pairs_q = (p.hidden / ROW_CHUNK) >> 1;  // heads: chunk pairs of q (and of k)
    const int pairs_qk = p.n_swapped >> 1, pairs_all = p.n_chunks >> 1;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    auto st16 = [&](u16* dst, const uint4& v) { store_stream16(dst, v); };

    // one step (64 pipe cycles) of a chunk: SW = weights as the left operand (q / k: C rows = features), else v
    auto step = [&](auto sw_tag, f32x4 (&acc)[2][MF], auto cs_tag, const bf16x8& w0, const bf16x8& w1, auto&& rd) {
      constexpr bool SW = decltype(sw_tag)::value;
      constexpr int cs = decltype(cs_tag)::value;
      if constexpr (!C8::is_f8(cs)) {
        constexpr int ks = C8::ks(cs);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[0][mf] = SW ? mfma16h(w0, a_hi[mf][ks], ks == 0 ? zero : acc[0][mf]) : mfma16h(a_hi[mf][ks], w0, ks == 0 ? zero : acc[0][mf]);
        rd(std::integral_constant<int, 0>{}, acc[0][0]);
        acc[1][0] = SW ? mfma16h(w1, a_hi[0][ks], ks == 0 ? zero : acc[1][0]) : mfma16h(a_hi[0][ks], w1, ks == 0 ? zero : acc[1][0]);
        rd(std::integral_constant<int, 1>{}, acc[0][MF - 1]);
#pragma unroll
        for (int mf = 1; mf < MF; ++mf)
          acc[1][mf] = SW ? mfma16h(w1, a_hi[mf][ks], ks == 0 ? zero : acc[1][mf]) : mfma16h(a_hi[mf][ks], w1, ks == 0 ? zero : acc[1][mf]);
      } else {
        constexpr int nf = C8::nf(cs), s8 = C8::s8(cs);
        const i32x8 w8 = f8_frag(w0, w1);
        auto one = [&](int mf) {
          if constexpr (C8::is_wlo(cs)) acc[nf][mf] = SW ? mfma8w<false>(w8, a_h8[mf][s8], acc[nf][mf]) : mfma8w<true>(a_h8[mf][s8], w8, acc[nf][mf]);
          else acc[nf][mf] = SW ? mfma8<true>(w8, a_lo8[mf][s8], acc[nf][mf]) : mfma8<false>(a_lo8[mf][s8], w8, acc[nf][mf]);
        };
        one(0);
        rd(std::integral_constant<int, 0>{}, acc[nf][0]);
        rd(std::integral_constant<int, 1>{}, acc[nf][0]);
#pragma unroll
        for (int mf = 1; mf < MF; ++mf) one(mf);
      }
    };

    // ---- the epilogue of a finished pair (pa = chunk 2t, pb = chunk 2t+1; VGPR values) as micro-operations ----
    f32x4 pa[2][MF], pb[2][MF];
    float rl[4], rh[4];
    uint2 hold[2][MF][4];  // q / k: [half-head j][mf][d < 32 hi, lo | d >= 32 hi, lo];  v: [chunk][nf][hi, lo of mf 0 | of mf 1]
    // q / k pair: 32 operations.  o = 16 j + 8 mf + v: v < 4 rotates value pair r = v of half-head j (chunk 2t + j),
    // v >= 4 splits two of the four rotated values into (hi, lo) bf16
    constexpr int QK_OPS = 16 * MF;
    auto qk_ops = [&](float qscale, auto begin_tag, auto end_tag) {
      constexpr int B = decltype(begin_tag)::value, E = decltype(end_tag)::value;
      static_for<(E > B ? E - B : 0)>([&](auto o_tag) {
        constexpr int o = B + decltype(o_tag)::value;
        constexpr int j = o / (8 * MF), mf = (o % (8 * MF)) / 8, v = o % 8;
        const f32x4 (&av)[2][MF] = *(j == 0 ? &pa : &pb);
        if constexpr (v < 4) {
          const float x1 = av[0][mf][v], x2 = av[1][mf][v];
          rl[v] = rope_lo(x1, x2, rope_cc[mf][j][v], rope_ss[mf][j][v]) * qscale;
          rh[v] = rope_hi(x1, x2, rope_cc[mf][j][v], rope_ss[mf][j][v]) * qscale;
        } else if constexpr (v == 4) {
          split2<QK_LO>(rl[0], rl[1], hold[j][mf][0].x, hold[j][mf][1].x);
        } else if constexpr (v == 5) {
          split2<QK_LO>(rl[2], rl[3], hold[j][mf][0].y, hold[j][mf][1].y);
        } else if constexpr (v == 6) {
          split2<QK_LO>(rh[0], rh[1], hold[j][mf][2].x, hold[j][mf][3].x);
        } else {
          split2<QK_LO>(rh[2], rh[3], hold[j][mf][2].y, hold[j][mf][3].y);
        }
      });
    };
    // the four pieces of row fragment mf: (hi, lo) x (d < 32, d >= 32) of head `pair` of q or k
    auto qk_store = [&](int pair, auto mf_tag) {
      constexpr int mf = decltype(mf_tag)::value;
      const bool is_q = pair < pairs_q;
      u16* out = is_q ? p.o0_hi : p.o1_hi;
      const size_t rb = (size_t)((m0 >> 4) + mf);
      const size_t kb = (size_t)(is_q ? pair : pair - pairs_q) * 2;  // k-step of d in [0, 32); d + 32 is the next one
      u16* sp = out + ((rb * (size_t)(p.hidden >> 5) + kb) * 2) * 512 + lane * 8;
      st16(sp, make_uint4(hold[0][mf][0].x, hold[0][mf][0].y, hold[1][mf][0].x, hold[1][mf][0].y));
      st16(sp + 1024, make_uint4(hold[0][mf][2].x, hold[0][mf][2].y, hold[1][mf][2].x, hold[1][mf][2].y));
      if (QK_LO && (O0_LO == O1_LO || (is_q ? O0_LO : O1_LO))) {  // q and k may differ: wave-uniform select
        st16(sp + 512, make_uint4(hold[0][mf][1].x, hold[0][mf][1].y, hold[1][mf][1].x, hold[1][mf][1].y));
        st16(sp + 1536, make_uint4(hold[0][mf][3].x, hold[0][mf][3].y, hold[1][mf][3].x, hold[1][mf][3].y));
      }
    };
    // v pair: 16 operations.  o = 8 c + 4 nf + 2 mf + half: (hi, lo) split of two of the four values of chunk 2t + c,
    // weight fragment nf, row fragment mf (C rows = tokens 4g + r of block mf, column = feature slot l15: the two row
    // blocks are the two halves of the 8 key slots of one v^T fragment lane)
    constexpr int V_OPS = 8 * MF;
    auto v_ops = [&](auto begin_tag, auto end_tag) {
      constexpr int B = decltype(begin_tag)::value, E = decltype(end_tag)::value;
      static_for<(E > B ? E - B : 0)>([&](auto o_tag) {
        constexpr int o = B + decltype(o_tag)::value;
        constexpr int c = o / (4 * MF), nf = (o % (4 * MF)) / (2 * MF), mf = (o % (2 * MF)) / 2, hf = o % 2;
        const f32x4 (&av)[2][MF] = *(c == 0 ? &pa : &pb);
        if constexpr (hf == 0) split2<O2_LO>(av[nf][mf][0], av[nf][mf][1], hold[c][nf][2 * mf].x, hold[c][nf][2 * mf + 1].x);
        else split2<O2_LO>(av[nf][mf][2], av[nf][mf][3], hold[c][nf][2 * mf].y, hold[c][nf][2 * mf + 1].y);
      });
    };
    auto v_store = [&](int pair, auto c_tag) {
      constexpr int c = decltype(c_tag)::value;
      const size_t head = (size_t)(pair - pairs_qk);
      const size_t tb = (size_t)(m0 >> 5);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        u16* sp = p.o2_hi + (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + (size_t)(c * 2 + nf)) * 512 + lane * 8;
        st16(sp, make_uint4(hold[c][nf][0].x, hold[c][nf][0].y, hold[c][nf][2].x, hold[c][nf][2].y));
        if (O2_LO) st16(sp + 2048, make_uint4(hold[c][nf][1].x, hold[c][nf][1].y, hold[c][nf][3].x, hold[c][nf][3].y));
      }
    };
    constexpr int N_ST_QK = MF * (2 + ((O0_LO && O1_LO) ? 2 : 0));  // stores the counted wait may leave in flight
    constexpr int N_ST_V = 4 * (1 + (O2_LO ? 1 : 0));

    // pair t: SW = q / k (else v); RIDE = 0: nothing rides (first pair), 1: the epilogue of a q / k pair, 2: of a v pair
    auto pair_stream = [&](int t, auto sw_tag, auto ride_tag) {
      constexpr int RIDE = decltype(ride_tag)::value;
      const int cur = t & 1;
      const int nxt = t + 1 < pairs_all ? t + 1 : t;  // unconditional DMA: the last pair re-copies itself into the idle stage
      const float qscale = (t - 1) < pairs_q ? 0.125f * 1.44269504088896340736f : 1.0f;  // head_dim^-0.5 * log2(e) on q
      f32x4 na[2][MF], nb[2][MF], va[2][MF], vb[2][MF];
      frag_stream2i<NST, FRAG_ILV, OffP>(lds_stage[0] + (uint32_t)cur * (uint32_t)(STAGE_ALLOC * 2), [&](auto step_tag, bf16x8& w0, bf16x8& w1, auto&& rd) {
        constexpr int s = decltype(step_tag)::value;
        if constexpr (s < PAIR_DMA) stage_pair_piece(step_tag, nxt, cur ^ 1);
        if constexpr (s < CS) step(sw_tag, na, step_tag, w0, w1, rd);
        else step(sw_tag, nb, std::integral_constant<int, s - CS>{}, w0, w1, rd);
        // a finished chunk leaves the accumulator file once
        if constexpr (s == CS - 1) {
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              va[nf][mf] = na[nf][mf];
              asm volatile("" : "+v"(va[nf][mf]));
            }
        }
        if constexpr (s == NST - 1) {
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
              vb[nf][mf] = nb[nf][mf];
              asm volatile("" : "+v"(vb[nf][mf]));
            }
        }
        // riders: chunk 2t-2's half of the list over the first CS steps (its values are dead when va is written), chunk
        // 2t-1's over the next CS - 1; the stores behind the iteration's last DMA (step PAIR_DMA - 1 < CS)
        if constexpr (RIDE == 1) {
          constexpr int H = QK_OPS / 2;
          constexpr int OB = s < CS ? s * H / CS : H + (s - CS) * H / (CS - 1);
          constexpr int OE = s < CS ? (s + 1) * H / CS : (s == NST - 1 ? QK_OPS : H + (s - CS + 1) * H / (CS - 1));
          qk_ops(qscale, std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{});
          // row fragment mf is complete after operation H + 8 mf + 7
          static_for<MF>([&](auto mf_tag) {
            constexpr int done = H + 8 * decltype(mf_tag)::value + 8;
            if constexpr (OB < done && OE >= done) qk_store(t - 1, mf_tag);
          });
        } else if constexpr (RIDE == 2) {
          constexpr int H = V_OPS / 2;
          constexpr int OB = s < CS ? s * H / CS : H + (s - CS) * H / (CS - 1);
          constexpr int OE = s < CS ? (s + 1) * H / CS : (s == NST - 1 ? V_OPS : H + (s - CS + 1) * H / (CS - 1));
          v_ops(std::integral_constant<int, OB>{}, std::integral_constant<int, OE>{});
          if constexpr (s == CS) v_store(t - 1, std::integral_constant<int, 0>{});
          if constexpr (s == NST - 1) v_store(t - 1, std::integral_constant<int, 1>{});
        }
      });
      // End of the iteration: this wave's share of the next pair must have landed, then all waves meet.  vmcnt retires in
      // order: everything but the stores issued behind the last DMA.
      constexpr int N_STORES = RIDE == 1 ? N_ST_QK : (RIDE == 2 ? N_ST_V : 0);
#ifdef OPK_TIMING
      const unsigned long long opk_w0 = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
#ifdef OPK_TIMING
      const unsigned long long opk_w1 = __builtin_readcyclecounter();
      opk_wait1 += opk_w1 - opk_w0;
#endif
      __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
      opk_wait2 += __builtin_readcyclecounter() - opk_w1;
#endif
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          pa[nf][mf] = va[nf][mf];
          pb[nf][mf] = vb[nf][mf];
        }
    };
    const std::integral_constant<int, 0> ride_none{};
    const std::integral_constant<int, 1> ride_qk{};
    const std::integral_constant<int, 2> ride_v{};
    pair_stream(0, yes_, ride_none);
    for (int t = 1; t < pairs_qk; ++t) pair_stream(t, yes_, ride_qk);
    pair_stream(pairs_qk, no_, ride_qk);  // first v pair; finishes the last k pair
    for (int t = pairs_qk + 1; t < pairs_all; ++t) pair_stream(t, no_, ride_v);
    v_ops(std::integral_constant<int, 0>{}, std::integral_constant<int, V_OPS>{});
    v_store(pairs_all - 1, std::integral_constant<int, 0>{});
    v_store(pairs_all - 1, std::integral_constant<int, 1>{});
    OPK_STAMP(5);
    OPK_DUMP();
    return;
  }

  // ---- stream the weight chunks ---------------------------------------------------------------
  uint2 hold_hi[MF], hold_lo[MF];  // RE_GEGLU: first half of a chunk pair
  uint2 qk_hold[MF][4];            // RE_QKV: first half-head of a q/k chunk pair: [mf][d<32 hi, lo, d>=32 hi, lo]
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int t = 0; t < 4; ++t) qk_hold[mf][t] = make_uint2(0u, 0u);
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) hold_hi[mf] = hold_lo[mf] = make_uint2(0u, 0u);
  // Epilogue of chunk `cc` (compile-time parity PP = cc & 1) from accumulators `av`.  It runs one iteration late,
  // inside the iteration that computes chunk cc+1, its VALU instructions scheduled between that chunk's MFMAs
  // (sched_group_barrier recipe below): vector instructions of all kinds share the SIMD's issue port, so the
  // epilogue costs its instruction count either way, but interleaved it no longer adds a serial VALU-only phase.
  constexpr bool O0_LO = (OLO & 1) != 0, O1_LO = (OLO & 2) != 0, O2_LO = (OLO & 4) != 0;
  constexpr bool QK_LO = O0_LO || O1_LO;
  // The epilogue is cut in two.  epilogue() is pure register work (RoPE / GeGLU, hi/lo split, packing) and leaves what
  // has to be written in st_v / st_p; epilogue_store() issues the stores and runs AFTER the chunk's MFMA stream.  The
  // hand-placed fragment reads are volatile asm: a store cannot move across them, so stores in front of the stream pin
  // every instruction that feeds them in front of it too -- the storing half of the iterations ran its whole epilogue
  // (~110 vector instructions) before its first MFMA instead of between them.
  uint4 st_v[2][4];  // [row fragment (q / k / h) or weight fragment (v^T)][store]
  u16* st_p[2];
  auto epilogue = [&](int cc, auto parity_tag, auto sw_tag, const f32x4 (&av)[2][MF]) {
    constexpr int PP = decltype(parity_tag)::value;
    constexpr bool sw = decltype(sw_tag)::value;  // q/k chunk ("swapped" MFMA orientation) or v chunk
    if (EPI == RE_NONE) {
    } else if (EPI == RE_GEGLU) {
      // Output = "fragment-packed" h (see hfp_offset): chunk 2t gives this lane h-columns 32t + 8g + (0..3),
      // chunk 2t+1 columns 32t + 8g + (4..7) (the Wi rows were permuted that way at load time), so after the
      // pair the lane owns the 8 consecutive k-values of ITS OWN fragment slot for k-step t of the next GEMM
      // and the wave stores one contiguous 1 KiB piece per (16-row block, plane).
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(av[0][mf][r]) * av[1][mf][r];
        uint2 h2, l2;
        split4<O0_LO>(v, h2, l2);
        if (PP == 0) {
          hold_hi[mf] = h2;
          hold_lo[mf] = l2;
        } else {
          const size_t rb = (size_t)((m0 >> 4) + mf);
          st_p[mf] = p.o0_hi + ((rb * (size_t)(p.ld_out >> 5) + (size_t)(cc >> 1)) * 2) * 512 + lane * 8;
          st_v[mf][0] = make_uint4(hold_hi[mf].x, hold_hi[mf].y, h2.x, h2.y);
          if (O0_LO) st_v[mf][1] = make_uint4(hold_lo[mf].x, hold_lo[mf].y, l2.x, l2.y);
        }
      }
    } else {  // RE_QKV: fragment-packed q, k (pieces [row/16][H/32][plane]) and v^T (pieces [head][row/32][plane][4])
      const int per_block = p.hidden / ROW_CHUNK;
      if (sw) {
        const bool is_q = cc < per_block;
        const int cq = is_q ? cc : cc - per_block;
        u16* out = is_q ? p.o0_hi : p.o1_hi;
        // head_dim^-0.5 * log2(e): the fragment-packed attention kernel exponentiates with exp2
        const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          // half-head index j = cq & 1 equals the chunk parity PP (even number of chunks per block)
          const f32x4 c4 = rope_c[mf];
          const f32x4 s4 = rope_s[mf];
          float lo_half[4], hi_half[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x1 = av[0][mf][r], x2 = av[1][mf][r];
            lo_half[r] = rope_lo(x1, x2, c4[r], s4[r]) * qscale;
            hi_half[r] = rope_hi(x1, x2, c4[r], s4[r]) * qscale;
          }
          uint2 h0, l0, h1, l1;
          split4x<QK_LO, H16>(lo_half, h0, l0);
          split4x<QK_LO, H16>(hi_half, h1, l1);
          if (PP == 0) {
            qk_hold[mf][0] = h0; qk_hold[mf][1] = l0; qk_hold[mf][2] = h1; qk_hold[mf][3] = l1;
          } else {
            const size_t rb = (size_t)((m0 >> 4) + mf);
            const size_t kb = (size_t)(cq >> 1) * 2;  // k-step of d in [0, 32); d + 32 is the next one
            st_p[mf] = out + ((rb * (size_t)(p.hidden >> 5) + kb) * 2) * 512 + lane * 8;
            st_v[mf][0] = make_uint4(qk_hold[mf][0].x, qk_hold[mf][0].y, h0.x, h0.y);
            st_v[mf][1] = make_uint4(qk_hold[mf][2].x, qk_hold[mf][2].y, h1.x, h1.y);
            if (QK_LO) {
              st_v[mf][2] = make_uint4(qk_hold[mf][1].x, qk_hold[mf][1].y, l0.x, l0.y);
              st_v[mf][3] = make_uint4(qk_hold[mf][3].x, qk_hold[mf][3].y, l1.x, l1.y);
            }
          }
        }
      } else {
        // C rows = tokens 4g + r of block mf, column = feature slot l15: the two 16-row blocks of a 32-row
        // group are the two halves of the 8 key slots of one v^T fragment lane.  With 32 rows per wave the lane
        // stores all 16 bytes, with 16 rows per wave the 8 bytes of its half.
        const int cv = cc - p.n_swapped;
        const size_t head = (size_t)(cv >> 1);
        const size_t tb = (size_t)(m0 >> 5);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const size_t n = (size_t)((cv & 1) * 2 + nf);
          const int half = MF == 2 ? 0 : ((m0 >> 4) & 1) * 4;
          st_p[nf] = p.o2_hi + (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + n) * 512 + lane * 8 + half;
          const float v0[4] = {av[nf][0][0], av[nf][0][1], av[nf][0][2], av[nf][0][3]};
          uint2 h0, l0;
          split4x<O2_LO, H16>(v0, h0, l0);
          if (MF == 2) {
            const float v1[4] = {av[nf][MF - 1][0], av[nf][MF - 1][1], av[nf][MF - 1][2], av[nf][MF - 1][3]};
            uint2 h1, l1;
            split4x<O2_LO, H16>(v1, h1, l1);
            st_v[nf][0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            if (O2_LO) st_v[nf][1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
          } else {
            st_v[nf][0] = make_uint4(h0.x, h0.y, 0u, 0u);
            if (O2_LO) st_v[nf][1] = make_uint4(l0.x, l0.y, 0u, 0u);
          }
        }
      }
    }
  };
  // the stores of epilogue(cc, parity, sw): same conditions, same order as the counted wait below expects
  auto st16 = [&](u16* dst, const uint4& v) { store_stream16(dst, v); };  // see store_stream16
  auto epilogue_store = [&](int cc, auto parity_tag, auto sw_tag) {
    constexpr int PP = decltype(parity_tag)::value;
    constexpr bool sw = decltype(sw_tag)::value;
    if (EPI == RE_GEGLU) {
      if (PP == 1) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          st16(st_p[mf], st_v[mf][0]);
          if (O0_LO) st16(st_p[mf] + 512, st_v[mf][1]);
        }
      }
    } else if (EPI == RE_QKV) {
      if (sw) {
        if (PP == 1) {
          const bool is_q = cc < p.hidden / ROW_CHUNK;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            st16(st_p[mf], st_v[mf][0]);
            st16(st_p[mf] + 1024, st_v[mf][1]);
            if (QK_LO && (O0_LO == O1_LO || (is_q ? O0_LO : O1_LO))) {  // q and k may differ: wave-uniform select
              st16(st_p[mf] + 512, st_v[mf][2]);
              st16(st_p[mf] + 1536, st_v[mf][3]);
            }
          }
        }
      } else {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          if (MF == 2) {
            st16(st_p[nf], st_v[nf][0]);
            if (O2_LO) st16(st_p[nf] + 2048, st_v[nf][1]);
          } else {
            *reinterpret_cast<uint2*>(st_p[nf]) = make_uint2(st_v[nf][0].x, st_v[nf][0].y);
            if (O2_LO) *reinterpret_cast<uint2*>(st_p[nf] + 2048) = make_uint2(st_v[nf][1].x, st_v[nf][1].y);
          }
        }
      }
    }
  };

  f32x4 acc_prev[2][MF];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc_prev[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Unrolled by two so that the LDS stage index is a compile-time constant in each copy: the compiler can
  // then tell the DMA into stage cur^1 from the fragment reads of stage cur and does NOT drain the DMA
  // (s_waitcnt vmcnt(0)) before the first ds_read -- the wait sits only in front of the barrier.
  // Every iteration is ONE basic block: the MFMA orientation of the chunk (SW) and the kind of the deferred
  // epilogue (SWP: q/k or v for RE_QKV) are compile-time tags, the chunk loop is split at the q/k -> v boundary.
  auto iteration = [&](int c, auto cur_tag, auto first_tag, auto sw_tag, auto swp_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr bool SW = decltype(sw_tag)::value;
    constexpr bool SWP = decltype(swp_tag)::value;
    // every wave passed the barrier that ended iteration c-1, so nobody reads stage cur^1 any more.
    // Unconditional (the last iteration harmlessly re-copies its own chunk into the idle stage): a DMA issued
    // under a branch makes the compiler drain it at the join, in front of the first fragment read.
    if (EPI == RE_QKV && SWP && !FIRST) {  // RoPE rows for the half-head (j = cur ^ 1) of the chunk finished last
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if (ROPE_PRELOAD) {
          rope_c[mf] = rope_cc[mf][cur ^ 1];
          rope_s[mf] = rope_ss[mf][cur ^ 1];
        } else {
          rope_c[mf] = *reinterpret_cast<const f32x4*>(rope_c_row[mf] + (cur ^ 1) * 4);
          rope_s[mf] = *reinterpret_cast<const f32x4*>(rope_s_row[mf] + (cur ^ 1) * 4);
        }
      }
    }
    stage_chunk(c + 1 < p.n_chunks ? c + 1 : c, cur ^ 1);
    // Nothing crosses this point: the RoPE loads stay ahead of the DMA, and the epilogue's stores stay BEHIND it --
    // the counted wait in front of the barrier below relies on that order.
    __builtin_amdgcn_sched_barrier(0);
    if (!FIRST) epilogue(c - 1, std::integral_constant<int, (cur ^ 1)>{}, swp_tag, acc_prev);

    f32x4 acc[2][MF];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (F8) rowgemm_chunk_mfma_f8<KS, MF, SW, true, WLO>(lds_stage[cur], a_hi, a_lo8, a_h8, acc);
    else rowgemm_chunk_mfma<KS, MF, T2, SW, 0, (PRO == RP_MLP), 1, H16>(lds_stage[cur], a_hi, a_lo, acc);
    if (!FIRST) epilogue_store(c - 1, std::integral_constant<int, (cur ^ 1)>{}, swp_tag);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc_prev[nf][mf] = acc[nf][mf];
    if (!FIRST) {
      // Scheduling recipe for this iteration: the first k-step's fragment reads, then per MFMA two (bf16x3) or five
      // (bf16) VALU instructions of the deferred epilogue and the fragment reads for the k-step ahead, spread evenly.
      // Measured (microbench/mfma_loop.hip and the GeLU rewrite): VALU work is NOT free beside MFMAs -- every vector
      // instruction costs its issue slot -- so the gain of the interleave is only that no wave sits in a VALU-only
      // phase while its partner waits for the same port; the lever that pays is fewer epilogue instructions.
      constexpr int NT = term_count(T2);
      constexpr int N_MFMA = F8 ? KS * 2 * MF + (WLO ? 4 : 2) * NS8 * MF : KS * 2 * MF * NT;
      constexpr int VALU_PER_MFMA = NT == 3 ? 2 : (NT == 2 ? 3 : 5);  // (2 or 4 for NT == 2: no change, measured)
#pragma unroll
      for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
      }
    }
    // End of the iteration: this wave's share of the next chunk must have landed in LDS, then all waves meet.
    // vmcnt retires in order, so waiting until only the N_STORES epilogue stores issued AFTER the DMA may still be
    // in flight covers the DMA without waiting for the stores' write acknowledgements (a __syncthreads() here is
    // fence + barrier = vmcnt(0): every iteration would wait for its own stores to reach L2).  Every fragment read
    // of stage `cur` has already returned (its MFMAs were issued), so the raw barrier is enough for the stage reuse.
    constexpr int PPREV = cur ^ 1;
    constexpr int N_STORES =
        FIRST ? 0
        : EPI == RE_GEGLU ? (PPREV == 1 ? MF * (1 + (O0_LO ? 1 : 0)) : 0)
        : EPI == RE_QKV ? (SWP ? (PPREV == 1 ? MF * (2 + ((O0_LO && O1_LO) ? 2 : 0)) : 0) : 2 * (1 + (O2_LO ? 1 : 0)))
                        : 0;
#ifdef OPK_TIMING
    const unsigned long long opk_w0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
#ifdef OPK_TIMING
    const unsigned long long opk_w1 = __builtin_readcyclecounter();
    opk_wait1 += opk_w1 - opk_w0;
#endif
    __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
    opk_wait2 += __builtin_readcyclecounter() - opk_w1;
#endif
  };
  // Even chunk counts on both sides of the q/k -> v boundary (checked on the host).  The first pair is peeled so
  // that the deferred epilogue is unconditional in the steady-state loops.
  const std::integral_constant<int, 0> even{};
  const std::integral_constant<int, 1> odd{};
  const std::true_type yes{};
  const std::false_type no{};
  const int n_sw = (EPI == RE_QKV) ? p.n_swapped : p.n_chunks;

  // ---- (round 5, kernel set "f16") the q / k / v^T loop of the whole-layer kernel with TWO chunks per LDS stage and barrier.
  // One chunk per barrier: 24 barriers of eight waves per block, each with its counted wait and a refill of the fragment-read
  // pipeline, around 256 pipe cycles of MFMAs per wave.  A stage now holds the chunk pair (2t, 2t+1) -- one RoPE head or one
  // v^T head --, the pair's two chunks run back to back with the deferred epilogue of the chunk before each riding on it,
  // and the eight waves meet once per pair.  Same operations in the same order per chunk: bit-identical values.
  if constexpr (QKV2) {
    static_assert(PLANES == 1, "chunk pairs: single-plane weights");
    auto set_rope = [&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        rope_c[mf] = rope_cc[mf][j];
        rope_s[mf] = rope_ss[mf][j];
      }
    };
    auto stores_of = [](bool sw_chunk, int parity) constexpr {
      return sw_chunk ? (parity == 1 ? MF * (2 + ((O0_LO && O1_LO) ? 2 : 0)) : 0) : 2 * (1 + (O2_LO ? 1 : 0));
    };
    auto recipe = [&]() {
      constexpr int N_MFMA = KS * 2 * MF;
#pragma unroll
      for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
    };
    auto pair_iteration = [&](int c0, auto stage_tag, auto first_tag, auto sw_tag, auto swp_tag) {
      constexpr int S = decltype(stage_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr bool SW = decltype(sw_tag)::value;    // this pair: q / k chunks (weights as the MFMA row operand) or v chunks
      constexpr bool SWP = decltype(swp_tag)::value;  // the chunk in front of the pair
      if (SWP && !FIRST) set_rope(odd);
      const int cn = c0 + 2 < p.n_chunks ? c0 + 2 : c0;  // (the last pair harmlessly re-copies itself: no DMA under a branch)
      stage_chunk_at(cn, S ^ 1, 0);
      stage_chunk_at(cn + 1, S ^ 1, STAGE);
      __builtin_amdgcn_sched_barrier(0);  // RoPE values ahead of the DMA, the epilogues' stores behind it (counted wait below)
      if (!FIRST) epilogue(c0 - 1, odd, swp_tag, acc_prev);
      f32x4 acc[2][MF];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
      rowgemm_chunk_mfma<KS, MF, T2, SW, 0, true, 1, H16>(lds_stage[S], a_hi, a_lo, acc);
      if (!FIRST) epilogue_store(c0 - 1, odd, swp_tag);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc_prev[nf][mf] = acc[nf][mf];
      if (!FIRST) recipe();
      __builtin_amdgcn_sched_barrier(0);
      // second chunk of the pair, the first one's epilogue riding on it
      if (SW) set_rope(even);
      epilogue(c0, even, sw_tag, acc_prev);
      f32x4 acc2[2][MF];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc2[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
      rowgemm_chunk_mfma<KS, MF, T2, SW, STAGE * 2, true, 1, H16>(lds_stage[S], a_hi, a_lo, acc2);
      epilogue_store(c0, even, sw_tag);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc_prev[nf][mf] = acc2[nf][mf];
      recipe();
      // this wave's share of the next pair has landed (vmcnt retires in order: everything but the stores issued behind the
      // DMA), then all waves meet
      constexpr int N_STORES = (FIRST ? 0 : stores_of(SWP, 1)) + stores_of(SW, 0);
#ifdef OPK_TIMING
      const unsigned long long opk_w0 = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
#ifdef OPK_TIMING
      const unsigned long long opk_w1 = __builtin_readcyclecounter();
      opk_wait1 += opk_w1 - opk_w0;
#endif
      __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
      opk_wait2 += __builtin_readcyclecounter() - opk_w1;
#endif
    };
    pair_iteration(0, even, yes, yes, yes);  // (chunks 0 and 1 were requested in front of the LayerNorm and have landed)
    int stage_bit = 1;
    for (int c0 = 2; c0 < n_sw; c0 += 2, stage_bit ^= 1) {
      if (stage_bit) pair_iteration(c0, odd, no, yes, yes);
      else pair_iteration(c0, even, no, yes, yes);
    }
    if (stage_bit) pair_iteration(n_sw, odd, no, no, yes);  // first v pair; finishes the last k chunk
    else pair_iteration(n_sw, even, no, no, yes);
    stage_bit ^= 1;
    for (int c0 = n_sw + 2; c0 < p.n_chunks; c0 += 2, stage_bit ^= 1) {
      if (stage_bit) pair_iteration(c0, odd, no, no, no);
      else pair_iteration(c0, even, no, no, no);
    }
    epilogue(p.n_chunks - 1, odd, no, acc_prev);
    epilogue_store(p.n_chunks - 1, odd, no);
    OPK_STAMP(5);
    OPK_DUMP();
    return;
  }
  iteration(0, even, yes, yes, yes);
  iteration(1, odd, no, yes, yes);
  for (int c0 = 2; c0 < n_sw; c0 += 2) {
    iteration(c0, even, no, yes, yes);
    iteration(c0 + 1, odd, no, yes, yes);
  }
  if (EPI == RE_QKV) {
    iteration(n_sw, even, no, no, yes);  // first v chunk; finishes the last k chunk
    iteration(n_sw + 1, odd, no, no, no);
    for (int c0 = n_sw + 2; c0 < p.n_chunks; c0 += 2) {
      iteration(c0, even, no, no, no);
      iteration(c0 + 1, odd, no, no, no);
    }
    epilogue(p.n_chunks - 1, odd, no, acc_prev);
    epilogue_store(p.n_chunks - 1, odd, no);
  } else {
    epilogue(p.n_chunks - 1, odd, yes, acc_prev);
    epilogue_store(p.n_chunks - 1, odd, yes);
  }
  OPK_STAMP(5);
  OPK_DUMP();
#undef OPK_STAMP
#undef OPK_DUMP
}

// ----------------------------------------------------------------------------------------------
// Fragment-packed activations.  An activation matrix [rows x C] that is consumed as the MFMA operand
// of the next GEMM is stored as 1 KiB pieces  [row/16][C/32][plane][lane = 16*(k%32/8) + row%16][8 k]:
// exactly one wave-instruction of 16-byte lanes, in lane order.  Producer epilogues store whole pieces
// (one fully coalesced 1 KiB store per wave), consumers load their fragment with one fully coalesced
// 1 KiB load straight into registers -- no LDS staging, no row-strided 8-byte accesses.
// ----------------------------------------------------------------------------------------------

#ifdef OPK_PACK_KERNELS  // weight re-packing runs in op_api.hip only
// dst[ks][plane][nf][g][i][e] <- W[nf*16 + i][ks*32 + g*8 + e]   (W is [N][K]; chunk = one k-step of all N)
__global__ void pack_kstream_kernel(const float* __restrict__ src, int N, int K, int permute, u16* __restrict__ dst,
                                    int zero_lo, int* __restrict__ any_lo, int f16 = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * K) return;
  const int NF = N / 16;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t % NF);
  const int ks = (int)(t / NF);
  // permute: accumulator slot (nf, i = 4g' + r) holds output feature 32(nf>>1) + 8g' + 4(nf&1) + r, so that the
  // accumulators of fragments (2s, 2s+1) ARE the 8 k-values of lane slot g' of k-step s of the next GEMM.
  const int row = permute ? (32 * (nf >> 1) + 8 * (i >> 2) + 4 * (nf & 1) + (i & 3)) : (nf * 16 + i);
  const float v = src[(size_t)row * K + ks * 32 + g * 8 + e];
  const size_t base = ((size_t)ks * 2 * NF + nf) * 512 + (size_t)g * 128 + i * 8 + e;
  if (f16) {  // kernel set "f16": fp16 hi plane, zero lo plane
    dst[base] = f2h(v);
    dst[base + (size_t)NF * 512] = (u16)0;
    return;
  }
  const u16 h = f2bf(v);
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  dst[base] = h;
  dst[base + (size_t)NF * 512] = zero_lo ? (u16)0 : l;
}
#endif

struct KStreamParams {
  const u16* a_fp;  // fragment-packed activations [r_pad/16][n_ksteps][2 planes][512]
  const u16* wp;    // packed weights [n_ksteps][2 planes][NF][512]
  int n_ksteps;     // K / 32
  float* x;         // fp32 [r_pad][N], x += A W^T
};

// x[128 or 256 rows, N = 16*NF] += A[rows, K] W[N, K]^T with K streamed: per k-step the block DMAs one
// [N x 32] weight slab into LDS (double-buffered) while every wave pulls its own two A fragments straight
// from the fragment-packed activation (prefetched one k-step ahead) and keeps all N outputs of its 32 rows
// in accumulators (NF x 2 x 4 registers).
template <int NF, int T, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void kstream_gemm_kernel(KStreamParams p) {
  constexpr bool W_LO = (T & T_RIGHT_LO) != 0, A_LO = (T & T_LEFT_LO) != 0;
  constexpr int PLANES = W_LO ? 2 : 1;
  constexpr int STAGE = NF * PLANES * 512;        // elements per LDS stage
  constexpr int CHUNK_SRC = NF * 2 * 512;         // elements per k-step in the packed weights
  constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * (WAVES * 32) + wave * 32;
  const int nks = p.n_ksteps;

  auto stage_chunk = [&](int ks, int stage) {
    const u16* src = p.wp + (size_t)ks * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + WAVES * u;  // stage = [plane][nf] pieces; source = same order (2 planes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  // A fragments of k-step ks: piece (rb, ks, plane) of the fragment-packed activation, 16 bytes per lane
  const u16* a_base0 = p.a_fp + ((size_t)(m0 >> 4) * nks * 2) * 512 + lane * 8;
  const u16* a_base1 = a_base0 + (size_t)nks * 2 * 512;
  bf16x8 an_hi[2], an_lo[2];
  auto load_a = [&](int ks) {
    an_hi[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024);
    an_hi[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024);
    if (A_LO) {
      an_lo[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024 + 512);
      an_lo[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024 + 512);
    }
  };

  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_chunk(0, 0);
  load_a(0);
  // Retire the first fragment loads HERE (empty asm "rewrites" the registers): a load still pending at the loop
  // header makes the compiler drain everything (vmcnt(0)) right after the loop body has issued its DMA.
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    asm volatile("" : "+v"(an_hi[mf]));
    if (A_LO) asm volatile("" : "+v"(an_lo[mf]));
  }
  __syncthreads();

  for (int k0 = 0; k0 < nks; k0 += 2) {
#pragma unroll
    for (int cur = 0; cur < 2; ++cur) {
      const int ks = k0 + cur;
      if (ks >= nks) break;
      const int kn = ks + 1 < nks ? ks + 1 : ks;
      stage_chunk(kn, cur ^ 1);
      bf16x8 a_hi[2], a_lo[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        a_hi[mf] = an_hi[mf];
        a_lo[mf] = an_lo[mf];
      }
      load_a(kn);                              // prefetch the next k-step's fragments ...
      __builtin_amdgcn_sched_barrier(0);       // ... and keep the loads up here, ahead of the MFMAs
#pragma unroll
      for (int nf = 0; nf < NF; nf += 2) {  // term-major over 2 fragments x 2 row blocks (see rowgemm_kernel phase 1)
        bf16x8 wh[2], wl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
          wl[j] = W_LO ? lds_frag(&sW[cur][(NF + nf + j) * 512 + lane * 8]) : wh[j];
        }
#pragma unroll
        for (int term = 0; term < 3; ++term) {
          if ((term == 0 && !W_LO) || (term == 1 && !A_LO)) continue;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
              acc[nf + j][mf] = mfma16(term == 0 ? wl[j] : wh[j], term == 1 ? a_lo[mf] : a_hi[mf], acc[nf + j][mf]);
        }
      }
      __syncthreads();
    }
  }

  // x += acc : the weights are packed with permuted output features (pack_kstream_kernel), accumulator slot
  // (nf, g, r) is feature 32(nf>>1) + 8g + 4(nf&1) + r of token m0 + 16mf + l15
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    float* xrow = p.x + (size_t)(m0 + mf * 16 + l15) * (NF * 16) + g * 8;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
      float4 r4 = load_stream_f4(reinterpret_cast<const float*>(px));
      r4.x += acc[nf][mf][0];
      r4.y += acc[nf][mf][1];
      r4.z += acc[nf][mf][2];
      r4.w += acc[nf][mf][3];
      store_stream16(reinterpret_cast<float*>(px), r4);
    }
  }
}

}  // namespace opk
