// opk_rowgemm.hip.h -- row-stationary GEMMs (hidden <= 256) and the k-streamed output projection
// rowgemm_kernel = RowGemmBlock<...>::run() below: the instantiation's constants, the register state its phases hand to each
// other and the staging helpers are members of the block; the phases are member functions defined in their own headers:
//   opk_rowgemm_phase1.hip.h     phase1()          attention output projection, K streamed (RP_KSTREAM / RP_MLP): A1 -> acc1
//   opk_rowgemm_mlp.hip.h        mlp_phase()       whole-layer kernel: the MLP with h on chip.  Its two sections stay closures
//                                                  over the loop's register arrays, included in place: opk_rowgemm_mlp_ops.inc
//                                                  (stage DMA, GeGLU micro-operations, MFMA steps), opk_rowgemm_mlp_loop.inc
//                                                  (macro-iterations) -- as members they cost the fp16 + e4m3 instantiations
//                                                  6 - 15 %
//   opk_rowgemm_qkv_pairs.hip.h  qkv_pairs_loop()  fp16 + e4m3 kernel sets: q / k / v^T as one fragment stream per chunk pair
//   opk_rowgemm_chunks.hip.h     chunk_loop()      the chunk loop (q / k / v^T with RoPE, GeGLU) and its deferred epilogues
// and the transitions between them are functions over that state: opk_rowgemm_ln.hip.h (residual + LayerNorm, row write-back,
// layer-0 rows), opk_rowgemm_stream.hip.h (final_norm + pruning head; LDS layouts and MFMA streams).  Parameters and pack
// kernels: opk_rowgemm_pack.hip.h; the k-streamed GEMM: opk_kstream.hip.h.
#pragma once

#include "opk_rowgemm_pack.hip.h"
#include "opk_rowgemm_stream.hip.h"
#include "opk_rowgemm_ln.hip.h"

namespace opk {

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
// One block of rowgemm_kernel: the instantiation's constants, the state its phases hand to each other -- all of it registers of
// the calling wave once inlined -- and the phases as member functions (run() is the kernel body).
template <int KS, int EPI, int PRO, int T1, int T2, int OLO, int WAVES, int MF, int TW, int TM, int F8, bool H16>
struct RowGemmBlock {
  static constexpr bool WLO = F8 == 2;
  static_assert(!H16 || (F8 == 0 && T1 == 0 && T2 == 0 && OLO == 0 && TW == 0 && TM == 0 && PRO != RP_KSTREAM && EPI != RE_GEGLU),
                "fp16 operands: the single-pass whole-layer / layer-0 q/k/v kernels only");
  static constexpr int TF8 = WLO ? 3 : T_LEFT_LO;  // the term masks this instantiation stands for
  static_assert(!F8 || (PRO == RP_MLP && WAVES == 4 && MF == 2 && KS % 4 == 0 && T1 == TF8 && TW == TF8 && TM == TF8 &&
                        (EPI == RE_NONE || T2 == TF8)),
                "f16 + fp8 kernel sets: whole-layer kernel, 4 waves x 32 rows, every activation lo term (+ F8 = 2: every weight lo term)");
  static constexpr int NS8 = F8 ? KS / 4 : 1;  // K = 128 steps of the fp8 lo product
  // A block is WAVES x MF x 16 rows; the library launches 4 waves x 2 fragments = 128 rows, two blocks per CU, and
  // 4 waves x 1 fragment = 64 rows for small batches (fewer than one 128-row block per CU-slot: twice the blocks, so
  // twice the CUs work on a latency-bound request).
  // Measured alternatives on MI355X (xsmall, 256 x 512): 8 waves x 2 (256 rows, one block per CU, half the DMA
  // instructions and L2 -> LDS traffic per row) is within +-2 % on both fused kernels; 8 waves x 1 (16 rows per wave,
  // <= 128 VGPRs, 4 waves per SIMD, twice the fragment reads per MFMA) is equal on q/k/v and 10 % slower on GeGLU.
  static_assert(MF == 1 || MF == 2, "one or two 16-row fragments per wave");
  // (F8: the weight's lo part is not a second bf16 plane -- W_LO / W_LO1 describe the bf16 kernels' LDS layout only)
  static constexpr bool W_LO = !F8 && (T2 & T_RIGHT_LO) != 0, A_LO = (T2 & T_LEFT_LO) != 0;
  static constexpr bool PHASE1 = PRO == RP_KSTREAM || PRO == RP_MLP;
  static constexpr bool W_LO1 = !F8 && PHASE1 && (T1 & T_RIGHT_LO) != 0, A_LO1 = PHASE1 && (T1 & T_LEFT_LO) != 0;
  // RP_MLP: 4 waves x 32 rows, ONE wave per SIMD with the 512-register budget: the normalised rows (2 x 64 registers
  // with their lo plane) and the 256 x 32 output accumulators (128) are both resident for the whole MLP; two waves of
  // 16 rows per SIMD (256 registers each) spill.  The fragment streams prefetch by hand, so latency is covered
  // without a partner wave.
  static_assert(PRO != RP_MLP || (((WAVES == 4 && MF == 2) || (WAVES == 8 && MF == 1)) && (WLO || ((TW & T_RIGHT_LO) == 0 && (TM & T_RIGHT_LO) == 0))),
                "fused MLP: 4 waves x 32 rows or 8 waves x 16 rows, single-plane Wi and Wo");
  static constexpr int PLANES = W_LO ? 2 : 1;
  static constexpr int PLANES1 = W_LO1 ? 2 : 1;
  static constexpr int K = KS * 32;
  // F8: a chunk is [fp16 plane: KS k-steps x 2 fragments | e4m3 plane: 2 fragments x KS/4 K-steps x 2 halves] 1 KiB pieces
  // (WLO: + the e4m3 plane of the weight's lo part; the packed chunk in global memory always carries it)
  static constexpr int CHUNK_PIECES8 = F8Chunk<KS, WLO>::PIECES;
  static constexpr int CHUNK_SRC = F8 ? F8Chunk<KS, WLO>::SRC_PIECES * 512 : KS * 2 * 1024;  // elements per packed chunk in global memory
  static constexpr int STAGE = F8 ? CHUNK_PIECES8 * 512 : KS * PLANES * 1024;  // elements per LDS stage
  // two Wi chunks + one Wo slab (2 KS fragments), one plane; F8: the chunks carry their e4m3 plane; WLO: ONE chunk +
  // half a slab (KS fragments, fp16 + bf16-lo planes) per stage
  static constexpr int MLP_UNIT = PRO == RP_MLP ? (WLO ? (CHUNK_PIECES8 + 2 * KS) * 512 : F8 ? (2 * CHUNK_PIECES8 + 2 * KS) * 512 : 3 * KS * 1024) : 0;
  // phase 1 slabs: 2 KS fragments per plane; F8: two fp16 slabs + half an e4m3 K = 128 slab per stage (WLO: + the
  // same half slab of the weight's lo part)
  static constexpr int STAGE_GEMM = WLO ? 8 * KS * 512 : F8 ? 6 * KS * 512 : KS * (PLANES > PLANES1 ? PLANES : PLANES1) * 1024;
  static constexpr int STAGE_ALLOC = STAGE_GEMM > MLP_UNIT ? STAGE_GEMM : MLP_UNIT;
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  // Whole-layer kernel: its two LayerNorm weight vectors (this layer's mlp_norm, the next layer's attn_norm) live in
  // LDS.  Read from global memory inside the LayerNorm they were 32 L2 round trips per block issued just in time
  // (the compiler cannot hoist them over the asm fences), each behind an in-order vmcnt wait that also waited for the
  // write acknowledgements of the residual rows stored just before: 16 k of a block's 252 k cycles per LayerNorm.
  static constexpr bool LN_V2 = PRO == RP_MLP;
  static_assert(!F8 || LN_V2, "f16 + fp8 kernel set: LayerNorm phases of the whole-layer kernel");
  static constexpr bool FIN_HEAD = LN_V2 && EPI == RE_NONE;  // (run-time switch: p.fin_ln)
  static constexpr int SLN_SIZE = LN_V2 ? (FIN_HEAD ? 4 : 2) * KS * 32 : 4;
  static constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  // Kernel set "f16" (round 5): the q / k / v^T loop of the whole-layer kernel takes TWO chunks per LDS stage and barrier
  // (pair_iteration below); chunk `chunk` copied to element offset `elem_off` of stage `stage`
  static constexpr bool QKV2 = PRO == RP_MLP && EPI == RE_QKV && !F8 && H16 && 2 * STAGE <= STAGE_ALLOC;
  // F8 kernel sets, q / k / v^T projection: the chunks are streamed in PAIRS (qkv_pairs below), a stage holds chunks
  // 2t and 2t+1 back to back.  Instruction u of a wave copies piece u % GS of its group u / GS (GS consecutive pieces
  // through one pointer / M0 and the DMA's immediate offset); the chunk a group belongs to is a wave-uniform integer
  // select, never control flow (a DMA under a branch is drained at the join).
  static constexpr bool QKV_PAIRS = F8 != 0 && EPI == RE_QKV;
  static constexpr int FRAG_ILV = 4;  // fragment groups in flight where the reads sit between the MFMAs (frag_stream2i); two measured slower in the forward
  static constexpr int PAIR_DMA = QKV_PAIRS ? 2 * CHUNK_PIECES8 / WAVES : 1;  // DMA instructions per wave and pair
  static constexpr int PAIR_GS = (2 * CHUNK_PIECES8 / 4) % WAVES == 0 ? 4 : 2;  // pieces per group (hidden 128 / 384: 2)
  static_assert(!QKV_PAIRS || (CHUNK_PIECES8 % PAIR_GS == 0 && (2 * CHUNK_PIECES8 / PAIR_GS) % WAVES == 0 && 2 * STAGE <= STAGE_ALLOC),
                "a chunk pair is whole groups of pieces per wave and fits one LDS stage");
  // RE_QKV: RoPE rows of this lane's tokens, cos/sin [pos][8g + 4j .. +3] for half-head j.  Two-wave kernels fetch the
  // half-head of the chunk whose (deferred) epilogue runs in an iteration at the top of that iteration (the partner wave
  // covers the latency).  The one-wave-per-SIMD layer kernel (RP_MLP) has nobody to cover it -- the loads sat behind a
  // full s_waitcnt vmcnt(0) in front of each epilogue, ~1000 cycles per chunk -- so it fetches both half-heads once,
  // while the LayerNorm in front of the chunk loop runs (ROPE_PRELOAD).
  static constexpr bool ROPE_PRELOAD = EPI == RE_QKV && PRO == RP_MLP;

  const RowGemmParams& p;
  u16 (&sW)[2][STAGE_ALLOC];  // weight stages (LDS)
  float (&sLn)[SLN_SIZE];     // LayerNorm / head weight vectors (LDS; whole-layer kernel)
  int tid, lane, wave, l15, g, m0, ln_i;
  float ln_fill0 = 0.f, ln_fill1 = 0.f, ln_fill2 = 0.f, ln_fill3 = 0.f;
  bf16x8 a_hi[MF][KS], a_lo[MF][KS];
  i32x8 a_lo8[MF][NS8];  // F8: e4m3 lo plane of the in-register operand, one K = 128 fragment per 4 k-steps
  i32x8 a_h8[MF][NS8];   // WLO: e4m3 of the operand itself (multiplied against the weights' lo part)
  const float* rope_c_row[MF];
  const float* rope_s_row[MF];
  f32x4 rope_c[MF], rope_s[MF];
  f32x4 rope_cc[MF][2], rope_ss[MF][2];
  static constexpr int NF1 = 2 * KS;  // feature fragments of a row block's H outputs (phase 1, MLP)
  f32x4 acc1[NF1][MF];                 // PHASE1: the block's rows [feature fragment][row fragment], x + o Wo^T (+ h Wo^T)
  float4 xq0[(PRO == RP_MLP) ? NF1 : 1];  // RP_MLP: row fragment 0 of x, requested at the top of the kernel
  // the MLP loop's register arrays, where they live in the block (opk_rowgemm_mlp_ops.inc): the single-pass 8 x 16 kernels
  static constexpr bool MLP_REGS_IN_BLOCK = PRO == RP_MLP && WAVES == 8 && F8 == 0 && T1 == 0 && TW == 0 && TM == 0;
  struct MlpRegs {
    f32x4 acc_b[2][MF];
    uint2 hold_hi[MF];
    uint2 hold_lo[MF];
    bf16x8 h_hi[MF];
    bf16x8 h_lo[MF];
    float g_prev[MF][4];
    float g_cur[MF][4];
    float gx[MF * 4];
    float gq[MF * 4];
    uint2 pk_hi[MF];
    float pk_d[MF][4];
    f32x4 nbv[2][MF];
  } mlp_regs;
  uint32_t lds_stage[2];  // LDS byte address of this lane's 16 bytes in piece 0 of each stage (the hand-placed fragment reads add immediates)
#ifdef OPK_TIMING
  unsigned long long opk_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, opk_wait = 0, opk_wait2 = 0, opk_wait1 = 0, opk_x[4] = {0, 0, 0, 0}, opk_rt0 = 0;
#ifdef OPK_SEG_TIMING
  unsigned long long opk_seg[3] = {0, 0, 0}, opk_seg_t = 0;
#endif
#endif

  __device__ __forceinline__ RowGemmBlock(const RowGemmParams& p_, u16 (&sW_)[2][STAGE_ALLOC], float (&sLn_)[SLN_SIZE]) : p(p_), sW(sW_), sLn(sLn_) {}

  // ---- weight streaming: global -> LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave-instruction) --
  // Stage layout = [ks][plane][frag][512] = a sequence of 1 KiB pieces; wave w copies pieces w, w+4, ...
  // The copy is linear (the packing kernel already wrote fragment order), so the lane-linear LDS
  // destination the DMA imposes is exactly the layout the fragment reads want.  No staging VGPRs, and
  // the request is in flight while the MFMAs of the current chunk run.
  __device__ __forceinline__ void stage_chunk(int chunk, int stage) {
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
  }
  __device__ __forceinline__ void stage_chunk_at(int chunk, int stage, int elem_off) {
    const u16* src = p.wp + (size_t)chunk * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int elem = (wave + WAVES * u) * 512;                // position inside the chunk's single plane
      const int src_elem = (elem / 1024) * 2048 + elem % 1024;  // the source keeps both planes per k-step
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + src_elem + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][elem_off + elem]), 16, 0, 0);
    }
  }
  template <class UTag>
  __device__ __forceinline__ void stage_pair_piece(UTag u_tag, int pair, int stage) {
    constexpr int u = decltype(u_tag)::value;
    const int piece0 = PAIR_GS * (wave + WAVES * (u / PAIR_GS));  // within the pair's stage
    const int second = piece0 >= CHUNK_PIECES8 ? 1 : 0;           // group lies in chunk 2t+1
    const u16* src = p.wp + (size_t)(2 * pair + second) * CHUNK_SRC + (piece0 - second * CHUNK_PIECES8) * 512 + lane * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(&sW[0][0] + stage * STAGE_ALLOC + piece0 * 512), 16,
                                     (u % PAIR_GS) * 1024, 0);
  }
  __device__ __forceinline__ void stage_pair(int pair, int stage) {
    static_for<PAIR_DMA>([&](auto u_tag) { stage_pair_piece(u_tag, pair, stage); });
  }
  __device__ __forceinline__ void rope_rows() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int pos = p.row_pos[m0 + mf * 16 + l15];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      rope_c_row[mf] = p.rope_cos + (size_t)pos * ROPE_HALF + g * 8;
      rope_s_row[mf] = p.rope_sin + (size_t)pos * ROPE_HALF + g * 8;
      rope_c[mf] = rope_s[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void rope_preload() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rope_cc[mf][j] = *reinterpret_cast<const f32x4*>(rope_c_row[mf] + j * 4);
        rope_ss[mf][j] = *reinterpret_cast<const f32x4*>(rope_s_row[mf] + j * 4);
      }
  }

#ifdef OPK_SEG_TIMING
#define OPK_SEG_DUMP() for (int i_ = 0; i_ < 3; ++i_) p.dbg[(size_t)blockIdx.x * 16 + 11 + i_] = opk_seg[i_];
#else
#define OPK_SEG_DUMP()
#endif
#ifdef OPK_TIMING
#ifdef OPK_SEG_TIMING
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
#else
#define OPK_STAMP(i)
#define OPK_DUMP()
#endif

  // ---- the phases (defined in opk_rowgemm_phase1 / _mlp / _qkv_pairs / _chunks .hip.h) and the transitions between them ----
  __device__ __forceinline__ void phase1();          // acc1 = A1 W1^T, K1 streamed (attention output projection)
  __device__ __forceinline__ void mlp_phase();       // RP_MLP: acc1 += GeGLU(LN(acc1 + x) Wi^T) Wo^T, h on chip
  __device__ __forceinline__ void qkv_pairs_loop();  // fp16 + e4m3 sets: q / k / v^T, one fragment stream per chunk pair
  __device__ __forceinline__ void chunk_loop();      // the weight-chunk loop with its deferred epilogues
  // residual add, (store the new hidden state,) LayerNorm, split -> fragments (opk_rowgemm_ln.hip.h)
  // LOAD: acc1 += x rows from memory; STORE: write the rows back; then LayerNorm with `lnw` into a_hi / a_lo.
  template <class LoadTag, class StoreTag, class LoTag>
  __device__ __forceinline__ void residual_ln(LoadTag, StoreTag, LoTag, const float* __restrict__ lnw) {
    rowgemm_residual_ln<KS, MF, PRO, LoadTag::value, StoreTag::value, LoTag::value>(p, m0, l15, g, lnw, acc1, xq0, a_hi, a_lo);
  }
  // the same transition in the whole-layer kernel: weights from LDS (sLn), the write-back left to store_rows()
  template <class LoadTag, class LoTag>
  __device__ __forceinline__ void layer_ln(LoadTag, LoTag, int which) {
    const uint32_t sln_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)&sLn[0]);
#ifdef OPK_TIMING
    unsigned long long* const stamps = opk_x;
#else
    unsigned long long* const stamps = nullptr;
#endif
    rowgemm_layer_ln<KS, MF, F8, H16, LoadTag::value, LoTag::value>(p, m0, l15, g, sln_addr, which, acc1, xq0, a_hi, a_lo, a_lo8, a_h8, stamps);
  }
  __device__ __forceinline__ void store_rows() { rowgemm_store_rows<KS, MF>(p, m0, l15, g, acc1); }

  // the kernel body
  __device__ __forceinline__ void run() {
    tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
    l15 = lane & 15;
    g = lane >> 4;
    m0 = blockIdx.x * (WAVES * 16 * MF) + wave * (16 * MF);
    // requested before anything else, written to LDS in front of the first block barrier (index clamped, no branch:
    // a load under a branch is drained at the join)
    ln_i = tid < KS * 32 ? tid : KS * 32 - 1;
    if constexpr (LN_V2) {
      ln_fill0 = p.ln_w_mlp[ln_i];
      if (EPI != RE_NONE) ln_fill1 = p.ln_w[ln_i];
      if (FIN_HEAD && p.fin_ln != nullptr) {  // block-uniform; these few loads are the first of the kernel
        ln_fill1 = p.fin_ln[ln_i];
        ln_fill2 = p.fin_pw[ln_i];
        ln_fill3 = p.fin_pw[KS * 32 + ln_i];
      }
    }
#ifdef OPK_TIMING
    opk_rt0 = wall_clock64();  // constant 100 MHz: shader clock = cycle stamps / this
    OPK_STAMP(0);
#endif
    if constexpr (F8) set_saturating_conversions();
    if (ROPE_PRELOAD) rope_rows();  // the position index load flies during phase 1
    // LDS byte address of this lane's 16 bytes in piece 0 of each stage (the hand-placed fragment reads add immediates)
    lds_stage[0] = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0][0]) + (uint32_t)lane * 16u;
    lds_stage[1] = lds_stage[0] + (uint32_t)(STAGE_ALLOC * 2);
    if (PHASE1) {
      phase1();
      OPK_STAMP(1);
      const std::true_type yes_{};
      const std::false_type no_{};
      if constexpr (PRO == RP_KSTREAM) {
        stage_chunk(0, 0);  // first weight chunk of phase 2 flies while the LayerNorm below runs
        residual_ln(yes_, yes_, std::integral_constant<bool, A_LO>{}, p.ln_w);
      } else {
        mlp_phase();
        if constexpr (EPI == RE_NONE) {
          if (FIN_HEAD && p.fin_ln != nullptr) rowgemm_final_head<KS, MF>(p, sLn, m0, l15, g, acc1);
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

    // layer 0: rows from the embedding table / a plain split of x; lo-operand mask of a narrower policy (opk_rowgemm_ln.hip.h)
    rowgemm_prologue_rows<KS, MF, PRO, A_LO, H16>(p, m0, l15, g, a_hi, a_lo);
    if (EPI == RE_QKV && !ROPE_PRELOAD) rope_rows();
    if constexpr (LN_V2) __builtin_amdgcn_s_barrier();  // (this wave's share of chunk 0 was waited for above)
    else __syncthreads();  // chunk 0 has landed (the barrier's release waits for this wave's DMA: vmcnt(0))

    if constexpr (QKV_PAIRS) qkv_pairs_loop();  // (fp16 + e4m3 sets: one fragment stream per chunk pair)
    else chunk_loop();
  }
};

}  // namespace opk

#define OPK_RG_TPL template <int KS, int EPI, int PRO, int T1, int T2, int OLO, int WAVES, int MF, int TW, int TM, int F8, bool H16>
#define OPK_RG_BLOCK RowGemmBlock<KS, EPI, PRO, T1, T2, OLO, WAVES, MF, TW, TM, F8, H16>
#include "opk_rowgemm_phase1.hip.h"
#include "opk_rowgemm_mlp.hip.h"
#include "opk_rowgemm_qkv_pairs.hip.h"
#include "opk_rowgemm_chunks.hip.h"
#undef OPK_RG_TPL
#undef OPK_RG_BLOCK
#undef OPK_STAMP
#undef OPK_DUMP

namespace opk {

template <int KS, int EPI, int PRO, int T1, int T2, int OLO, int WAVES, int MF = 2, int TW = 0, int TM = 0, int F8 = 0, bool H16 = false>
__global__ __launch_bounds__(WAVES * 64, PRO == RP_MLP ? (WAVES == 8 ? 2 : 1) : ((MF == 1 && WAVES == 8) ? 4 : 2)) void rowgemm_kernel(RowGemmParams p) {
  using Block = RowGemmBlock<KS, EPI, PRO, T1, T2, OLO, WAVES, MF, TW, TM, F8, H16>;
  __shared__ __attribute__((aligned(16))) u16 sW[2][Block::STAGE_ALLOC];
  __shared__ __attribute__((aligned(16))) float sLn[Block::SLN_SIZE];
  Block block(p, sW, sLn);
  block.run();
}

}  // namespace opk

#include "opk_kstream.hip.h"
