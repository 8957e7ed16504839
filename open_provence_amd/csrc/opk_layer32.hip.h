// opk_layer32.hip.h -- the whole-layer kernel on v_mfma_f32_32x32x16_bf16 (hidden = 256, single-plane weights)
#pragma once

#include <type_traits>
#include <utility>

#include "opk_common.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// One launch per layer, as rowgemm_kernel<.., RP_MLP, ..> (opk_rowgemm.hip.h):
//   x += o Wo^T ; LayerNorm ; x += GeGLU(LN(x) Wi^T) Wo^T with h kept in registers ; LayerNorm ; next q / k / v^T
// but every contraction runs on the 32x32x16 shape.  Why: the loops of this kernel are bound by how many non-MFMA
// instructions one wave (one per SIMD, 512 registers) can issue beside its MFMA stream, and the budget is per MFMA
// INSTRUCTION, not per flop -- measured (microbench/mfma32_probe.hip), beside 32768 flop of matrix work:
//   two 16x16x32:  2 fma + 1 accumulator read + 0.5 LDS read   42.5 cycles      one 32x32x16:  35.0 (32 = pipe-bound)
// The 32x32 shape needs its accumulators in AGPRs (the 512-register budget puts them there) and gives a wave ONE
// 32-row fragment: lane = (row n = lane % 32, half h = lane / 32).
//   MFMA operands (A: 32 x 16, B: 16 x 32): lane (n, h) holds 8 consecutive k = 16 s + 8 h + (0..7) of row / column n.
//   Result (32 x 32, 16 registers): lane (n, h) holds column n, rows 8 (i / 4) + 4 h + i % 4.
// "Swapped" products (weights as A, token rows as B) leave a lane with 16 of a tile's 32 output features of ITS row;
// the weight rows are permuted at load time (l32_source_row) so that those are exactly the 2 x 8 consecutive k the lane
// needs as the B operand of the next contraction (registers 0..7 -> k-step 2T, 8..15 -> k-step 2T + 1), the RoPE pair
// (d, d + 32) of a q / k value, or a GeGLU input beside its gate.  Nothing crosses lanes except the two-lane row sums of
// the LayerNorms.  Inputs and outputs keep the layouts of the other kernels: o, q, k as 1 KiB pieces
// [row/16][C/32][plane][16 (k%32/8) + row%16][8], v^T as [head][row/32][plane][4][16 kg + d'][8 keys], x fp32 rows.
// ----------------------------------------------------------------------------------------------

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

enum Layer32Pack { L32_RESID = 0, L32_GEGLU = 1, L32_QKV = 2 };

// source row of slot m (0..31) of 32-row tile T of a weight matrix
__host__ __device__ inline int l32_source_row(int mode, int T, int m, int H, int I) {
  const int j = m >> 3, hh = (m >> 2) & 1, r = m & 3;
  if (mode == L32_RESID)  // accumulator register i of half hh <-> feature 32 T + 16 (i / 8) + 8 hh + i % 8
    return 32 * T + 16 * (j >> 1) + 8 * hh + 4 * (j & 1) + r;
  if (mode == L32_GEGLU) {  // tile = 16 h-columns (registers 0..7) and their gates (8..15)
    const int col = 16 * T + 8 * hh + 4 * (j & 1) + r;
    return j < 2 ? col : I + col;
  }
  const int per = H / 32;  // tiles in each of q, k, v
  if (T < 2 * per) {       // q / k: 16 d of a head (registers 0..7) and their RoPE partners d + 32 (8..15)
    const int blk = T / per, cc = T % per;
    const int d = 16 * (cc & 1) + 8 * hh + 4 * (j & 1) + r + (j >= 2 ? 32 : 0);
    return blk * H + (cc >> 1) * HEAD_DIM + d;
  }
  // v (tokens x features orientation): column m of the tile is feature slot (piece nf = m / 16, d' = m % 16) of the
  // transposed layout, in the d order of rowgemm_source_row (the attention output then is lane-contiguous)
  const int cv = T - 2 * per, nf = m >> 4, dd = m & 15;
  return 2 * H + (cv >> 1) * HEAD_DIM + 32 * (cv & 1) + 8 * (dd >> 2) + 4 * nf + (dd & 3);
}

#ifdef OPK_PACK_KERNELS
// hi plane only.  chunk-major: dst[T][s][lane][8], k-major: dst[s][T][lane][8]; lane = 32 hh + m holds k = 16 s + 8 hh + e
__global__ void pack_layer32_kernel(const float* __restrict__ src, int n_rows, int K, int mode, int kmajor, int H, int I,
                                    u16* __restrict__ dst, int f16 = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_rows * K) return;
  const int KS = K / 16, NT = n_rows / 32;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int l = (int)(t & 63); t >>= 6;
  int T, s;
  if (kmajor) {
    T = (int)(t % NT);
    s = (int)(t / NT);
  } else {
    s = (int)(t % KS);
    T = (int)(t / KS);
  }
  const int row = l32_source_row(mode, T, l & 31, H, I);
  const float v = src[(size_t)row * K + 16 * s + 8 * (l >> 5) + e];
  dst[idx] = f16 ? f2h(v) : f2bf(v);  // f16: the packs of the wave-pair kernel on fp16 operands (kernel set "f16")
}
#endif

struct Layer32Params {
#ifdef OPK_TIMING
  unsigned long long* dbg;  // [blocks][16] cycle stamps of wave 0
#endif
  const u16* o_fp;       // attention output, pieces [r_pad/16][H/32][2][512]
  float* x_io;           // residual stream fp32 [r_pad][H]
  const float* ln_mlp;   // this layer's mlp_norm weight
  const float* ln_next;  // the next layer's attn_norm weight (QKV only)
  float eps;
  const u16* wo_p;    // attention Wo, k-major   [H/16][H/32][512]
  const u16* wi_p;    // Wi, chunk-major         [I/16][H/16][512]
  const u16* wo2_p;   // MLP Wo, k-major         [I/16][H/32][512]
  const u16* wqkv_p;  // next Wqkv, chunk-major  [3H/32][H/16][512]
  int n_pairs;        // I / 32 (even)
  u16* q_fp;
  u16* k_fp;
  u16* vt_fp;
  int r_pad;
  const int32_t* row_pos;
  const float* rope_cos;
  const float* rope_sin;
  int max_pos;
};

// NT = hidden / 32.  QKV: the next layer's q / k / v^T follow (false: the last layer).  ALO: the activation-side
// operands carry their lo plane (2 passes per contraction; false = single pass).  OLO: lo planes of q (1), k (2), v (4).
template <int NT, bool QKV, bool ALO, int OLO>
__global__ __launch_bounds__(256, 1) void layer32_kernel(Layer32Params p) {
  constexpr int H = NT * 32;
  constexpr int KS = NT * 2;                // 16-wide k-steps of a K = H contraction
  constexpr int CHUNK = KS * 512;           // elements of one 32-feature weight chunk (hi plane)
  constexpr int SLAB = NT * 512;            // elements of one k-step of a k-major weight
  constexpr int STAGE = 2 * CHUNK + 2 * SLAB;  // MLP stage: [Wi chunk A | Wi chunk B | two k-steps of Wo]
  static_assert(NT == 8, "written for hidden = 256 (4 waves share the DMA of a 48 KiB stage)");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];
  __shared__ __attribute__((aligned(16))) float sLn[2 * H];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;
#ifdef OPK_TIMING
  unsigned long long opk_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, opk_wait = 0;
  const unsigned long long opk_rt0 = wall_clock64();  // constant 100 MHz: shader clock = cycle stamps / this
#define L32_STAMP(i) opk_ts[i] = __builtin_readcyclecounter()
#else
#define L32_STAMP(i)
#endif
  L32_STAMP(0);

  // LayerNorm weights -> LDS (requested first, written in front of the first barrier; see rowgemm_kernel)
  const int ln_i = tid < H ? tid : H - 1;
  const float ln_fill0 = p.ln_mlp[ln_i];
  float ln_fill1 = 0.f;
  if (QKV) ln_fill1 = p.ln_next[ln_i];

  uint32_t lds_stage[2];
  lds_stage[0] = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0][0]) + (uint32_t)lane * 16u;
  lds_stage[1] = lds_stage[0] + (uint32_t)(STAGE * 2);
  // DMA of `pieces` consecutive 1 KiB pieces, this wave's share (piece = wave + 4 u)
  auto dma_piece = [&](const u16* src_piece0, int stage, int dst_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_piece0 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(&sW[stage][dst_piece * 512]), 16, 0, 0);
  };

  // ---- phase 1: acc1 = o Wo^T (K = H streamed, 4 k-steps per LDS stage), o fragments straight from memory ----------
  bf16x8 a_hi[KS], a_lo[KS];
  {
    const u16* o_base = p.o_fp + ((size_t)((m0 >> 4) + (n >> 4)) * NT * 2) * 512 + (16 * hh + (n & 15)) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16* src = o_base + ((ks >> 1) * 2) * 512 + (ks & 1) * 256;
      a_hi[ks] = load_stream_frag(src);
      a_lo[ks] = ALO ? load_stream_frag(src + 512) : a_hi[ks];
    }
  }
  auto stage_p1 = [&](int j, int stage) {  // k-steps 4 j .. 4 j + 3: 4 NT pieces
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int piece = wave + 4 * u;
      dma_piece(p.wo_p + (size_t)(4 * j) * SLAB + piece * 512, stage, piece);
    }
  };
  stage_p1(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // residual rows: lane (n, hh) owns features 32 T + 16 jj + 8 hh + (0..7); tiles 0..3 are requested here, 4..7 when the
  // LayerNorm starts
  float* xrow = p.x_io + (size_t)(m0 + n) * H + 8 * hh;
  float4 xa[NT / 2][2][2];
#pragma unroll
  for (int T = 0; T < NT / 2; ++T)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) xa[T][jj][q4] = load_stream_f4(xrow + 32 * T + 16 * jj + 4 * q4);
  __builtin_amdgcn_sched_barrier(0);
  sLn[ln_i] = ln_fill0;
  sLn[H + ln_i] = ln_fill1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * 2) : "memory");  // all but the residual-row loads
  __builtin_amdgcn_s_barrier();

  f32x16 acc1[NT];
  {
    struct P1Off {  // step st = (k-step st / 4 of the stage, tile pair st % 4)
      static constexpr int at(int st, int j) { return ((st >> 2) * NT + 2 * (st & 3) + j) * 1024; }
    };
    static_for<KS / 4>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      constexpr int cur = j & 1;
      if constexpr (j + 1 < KS / 4) stage_p1(j + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      frag_stream2<16, 2, P1Off>(lds_stage[cur], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
        constexpr int st = decltype(step_tag)::value;
        constexpr int ks = 4 * j + (st >> 2), T = 2 * (st & 3);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ALO) {
          acc1[T] = mfma32(w0, a_lo[ks], ks == 0 ? zero : acc1[T]);
          acc1[T + 1] = mfma32(w1, a_lo[ks], ks == 0 ? zero : acc1[T + 1]);
        }
        acc1[T] = mfma32(w0, a_hi[ks], (ks == 0 && !ALO) ? zero : acc1[T]);
        acc1[T + 1] = mfma32(w1, a_hi[ks], (ks == 0 && !ALO) ? zero : acc1[T + 1]);
      });
      __syncthreads();
    });
  }
  L32_STAMP(1);

  // ---- LayerNorm of the rows in the accumulators (LOAD: acc1 += x first) -> a_hi / a_lo --------------------------
  // Vector-only phase: packed fp32 arithmetic, sums in four chains, weights from LDS (see rowgemm_kernel::layer_ln).
  auto layer_ln = [&](auto load_tag, auto lo_tag, int which) {
    constexpr bool LOAD = decltype(load_tag)::value, LO = decltype(lo_tag)::value;
    float4 xb[LOAD ? NT / 2 : 1][2][2];
    if (LOAD) {
#pragma unroll
      for (int T = NT / 2; T < NT; ++T)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) xb[LOAD ? T - NT / 2 : 0][jj][q4] = load_stream_f4(xrow + 32 * T + 16 * jj + 4 * q4);
    }
    // this lane's weights (columns 32 T + 16 jj + 8 hh .. + 7 per step) are read a step ahead, by hand (lds_read_f4)
    const uint32_t ln_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)&sLn[0]) + (uint32_t)(which * H + 8 * hh) * 4u;
    f32x4 wq[2][2];
    auto ln_read = [&](auto st_tag) {
      constexpr int st = decltype(st_tag)::value;  // step = (T, jj) = (st / 2, st % 2)
      wq[st & 1][0] = lds_read_f4<(32 * (st >> 1) + 16 * (st & 1)) * 4>(ln_addr);
      wq[st & 1][1] = lds_read_f4<(32 * (st >> 1) + 16 * (st & 1)) * 4 + 16>(ln_addr);
    };
    f32x2 v[NT][8];
    f32x2 s4[4];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      f32x16 a = acc1[T];
      if (LOAD) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            const float4 x4 = T < NT / 2 ? xa[T < NT / 2 ? T : 0][jj][q4] : xb[(LOAD && T >= NT / 2) ? T - NT / 2 : 0][jj][q4];
            a[8 * jj + 4 * q4 + 0] += x4.x;
            a[8 * jj + 4 * q4 + 1] += x4.y;
            a[8 * jj + 4 * q4 + 2] += x4.z;
            a[8 * jj + 4 * q4 + 3] += x4.w;
          }
        acc1[T] = a;
      }
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        v[T][pp] = f32x2{a[2 * pp], a[2 * pp + 1]};
        s4[pp & 3] = (T == 0 && pp < 4) ? v[T][pp] : pk_add(s4[pp & 3], v[T][pp]);
      }
    }
    const f32x2 st = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
    float sum = st.x + st.y;
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / (float)H);
    const f32x2 m2 = f32x2{mean, mean};
    f32x2 q4s[4];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        v[T][pp] = pk_sub(v[T][pp], m2);
        q4s[pp & 3] = (T == 0 && pp < 4) ? pk_mul(v[T][pp], v[T][pp]) : pk_fma(v[T][pp], v[T][pp], q4s[pp & 3]);
      }
    const f32x2 qt = pk_add(pk_add(q4s[0], q4s[1]), pk_add(q4s[2], q4s[3]));
    float qq = qt.x + qt.y;
    qq += __shfl_xor(qq, 32, 64);
    const float rstd = 1.0f / sqrtf(qq * (1.0f / (float)H) + p.eps);
    const f32x2 r2 = f32x2{rstd, rstd};
    ln_read(std::integral_constant<int, 0>{});
    ln_read(std::integral_constant<int, 1>{});
    static_for<2 * NT>([&](auto st_tag) {
      constexpr int st = decltype(st_tag)::value, T = st >> 1, jj = st & 1;
      f32x4& w0 = wq[st & 1][0];
      f32x4& w1 = wq[st & 1][1];
      lds_wait_f4<(st + 1 < 2 * NT ? 2 : 0)>(w0, w1);
      const f32x2 lw[4] = {f32x2{w0[0], w0[1]}, f32x2{w0[2], w0[3]}, f32x2{w1[0], w1[1]}, f32x2{w1[2], w1[3]}};
      uint32_t hb[4], lb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) split2_pk<LO>(pk_mul(pk_mul(v[T][4 * jj + u], r2), lw[u]), hb[u], lb[u]);
      a_hi[2 * T + jj] = as_frag(make_uint4(hb[0], hb[1], hb[2], hb[3]));
      a_lo[2 * T + jj] = as_frag(make_uint4(lb[0], lb[1], lb[2], lb[3]));
      if constexpr (LO) asm volatile("" : "+a"(a_lo[2 * T + jj]));  // the lo fragments live in AGPRs from here on
      if constexpr (st + 2 < 2 * NT) ln_read(std::integral_constant<int, st + 2>{});
    });
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4)
          store_stream16(xrow + 32 * T + 16 * jj + 4 * q4,
                         make_float4(acc1[T][8 * jj + 4 * q4], acc1[T][8 * jj + 4 * q4 + 1], acc1[T][8 * jj + 4 * q4 + 2],
                                     acc1[T][8 * jj + 4 * q4 + 3]));
  };
  const std::true_type yes_{};
  const std::false_type no_{};

  // ---- MLP ----------------------------------------------------------------------------------------------------
  // Stage t = [Wi chunk 2t | Wi chunk 2t+1 | Wo k-steps 2(t-1), 2(t-1)+1].  Macro-iteration t, one fragment stream:
  //   steps 0..KS-1   the chunk pair (two accumulators na / nb, alternating: no MFMA waits for the one before it), beside
  //                   them the GeGLU of pair t-1 in stages over all 16 values -> its two h fragments
  //   steps KS..KS+7  acc1 += h(t-1) Wo^T, beside them na / nb are copied out for the next iteration's GeGLU
  const int n_pairs = p.n_pairs;
  auto stage_piece = [&](auto u_tag, int t, int stage) {
    constexpr int u = decltype(u_tag)::value;  // 0 .. 11
    const int tc = t < n_pairs ? t : n_pairs - 1;
    const int ts = t > 0 ? t - 1 : 0;
    const int piece = wave + 4 * u;  // 48 pieces: [0, 2 KS) chunks, then 2 NT slab pieces
    const u16* src = u < (2 * KS) / 4 ? p.wi_p + (size_t)(2 * tc) * CHUNK + piece * 512
                                      : p.wo2_p + (size_t)(2 * ts) * SLAB + (piece - 2 * KS) * 512;
    dma_piece(src, stage, piece);
  };
  constexpr int UNIT_DMA = (2 * KS + 2 * NT) / 4;
  static_for<UNIT_DMA>([&](auto u) { stage_piece(u, 0, 0); });
  layer_ln(yes_, std::integral_constant<bool, ALO>{}, 0);

  f32x16 na, nb;
  float gx[16], gg[16], gq[16];  // GeGLU in flight: inputs, gates, running polynomial / result
  bf16x8 h_hi[2], h_lo[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) gx[i] = gg[i] = gq[i] = 0.f;
  h_hi[0] = h_hi[1] = h_lo[0] = h_lo[1] = as_frag(make_uint4(0u, 0u, 0u, 0u));
  // stage `stg` (0..7) of values [8 half, 8 half + 8)
  auto geglu_stage = [&](auto stg_tag, auto half_tag) {
    constexpr int stg = decltype(stg_tag)::value, half = decltype(half_tag)::value;
    static_for<8>([&](auto i_tag) {
      constexpr int i = 8 * half + decltype(i_tag)::value;
      if constexpr (stg == 0) gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
      else if constexpr (stg < 5) gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), stg);
      else if constexpr (stg == 5) gq[i] = __builtin_amdgcn_exp2f(gq[i]);
      else if constexpr (stg == 6) gq[i] = gelu_erf_finish(gq[i], gx[i]);
      else gq[i] = gq[i] * gg[i];
    });
    if constexpr (stg == 7) {  // the chunk's 8 values = this lane's k of one k-step of the Wo contraction
      uint2 h2[2], l2[2];
      split4<ALO>(&gq[8 * half], h2[0], l2[0]);
      split4<ALO>(&gq[8 * half + 4], h2[1], l2[1]);
      h_hi[half] = as_frag(make_uint4(h2[0].x, h2[0].y, h2[1].x, h2[1].y));
      h_lo[half] = as_frag(make_uint4(l2[0].x, l2[0].y, l2[1].x, l2[1].y));
    }
  };
  // slice u (0..7) of copying the finished pair out of the accumulators: inputs = registers 0..7, gates = 8..15
  auto copy_out = [&](auto u_tag) {
    constexpr int i = decltype(u_tag)::value;
    gx[i] = na[i];
    gg[i] = na[8 + i];
    gx[8 + i] = nb[i];
    gg[8 + i] = nb[8 + i];
    // materialised HERE (empty asm, "v" class): a plain copy is deferred to its use in the next iteration, and the pair
    // is then carried across the back edge in a second set of 32 AGPRs (32 accumulator moves per iteration)
    asm volatile("" : "+v"(gx[i]), "+v"(gg[i]), "+v"(gx[8 + i]), "+v"(gg[8 + i]));
  };
  struct MlpOff {
    static constexpr int at(int s, int j) {
      return s < KS ? (j * KS + s) * 1024 : (2 * KS + ((s - KS) >> 2) * NT + 2 * ((s - KS) & 3) + j) * 1024;
    }
  };
  // one MFMA : up to four vector instructions of the step's slice (the 32x32x16 shape leaves room for 4-5, see the top)
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < (ALO ? 4 : 2); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
  };
  auto macro = [&](int t, int cur, auto slab_tag) {
    constexpr bool WITH_SLAB = decltype(slab_tag)::value;  // false only for t = 0
    frag_stream2<(WITH_SLAB ? KS + 8 : KS), 2, MlpOff>(cur ? lds_stage[1] : lds_stage[0], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
      constexpr int s = decltype(step_tag)::value;
      if constexpr (s < UNIT_DMA) stage_piece(step_tag, t + 1, cur ^ 1);
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (s < KS) {
        if (ALO) {
          na = mfma32(w0, a_lo[s], s == 0 ? zero : na);
          nb = mfma32(w1, a_lo[s], s == 0 ? zero : nb);
        }
        na = mfma32(w0, a_hi[s], (s == 0 && !ALO) ? zero : na);
        nb = mfma32(w1, a_hi[s], (s == 0 && !ALO) ? zero : nb);
        if constexpr (WITH_SLAB) geglu_stage(std::integral_constant<int, (s >> 1)>{}, std::integral_constant<int, (s & 1)>{});
      } else {
        constexpr int u = s - KS, kk = u >> 2, T = 2 * (u & 3);
        if (ALO) {
          acc1[T] = mfma32(w0, h_lo[kk], acc1[T]);
          acc1[T + 1] = mfma32(w1, h_lo[kk], acc1[T + 1]);
        }
        acc1[T] = mfma32(w0, h_hi[kk], acc1[T]);
        acc1[T + 1] = mfma32(w1, h_hi[kk], acc1[T + 1]);
        copy_out(std::integral_constant<int, u>{});  // (the GeGLU of pair t-1 ended with step KS-1)
      }
      interleave();
    });
  };
  auto end_of_stage = [&]() {
#ifdef OPK_TIMING
    const unsigned long long w0_ = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
    opk_wait += __builtin_readcyclecounter() - w0_;
#endif
    // The row accumulators stay in AGPRs across the back edge (empty asm, "a" class): left to the allocator three of
    // the eight tiles spent the chunk steps in VGPRs and were copied in and out around the slab steps -- 176 accumulator
    // moves per iteration at ~8 issue cycles each.
#pragma unroll
    for (int T = 0; T < NT; ++T) asm volatile("" : "+a"(acc1[T]));
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // stage 0 has landed
  L32_STAMP(2);
  macro(0, 0, no_);
  static_for<8>([&](auto u) { copy_out(u); });
  end_of_stage();
  {  // at least one more iteration (n_pairs is even): written as do-while -- with a loop that may run zero times the
    // compiler parks 96 accumulator values in scratch for the path around it
    int t = 1;
    do {
      macro(t, t & 1, yes_);
      end_of_stage();
    } while (++t < n_pairs);
  }
  {  // tail: GeGLU of the last pair, its two Wo k-steps (stage 0: n_pairs is even)
    static_for<KS>([&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
      geglu_stage(std::integral_constant<int, (s >> 1)>{}, std::integral_constant<int, (s & 1)>{});
    });
    struct TailOff {
      static constexpr int at(int u, int j) { return (2 * KS + (u >> 2) * NT + 2 * (u & 3) + j) * 1024; }
    };
    frag_stream2<8, 2, TailOff>(lds_stage[0], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
      constexpr int u = decltype(step_tag)::value, kk = u >> 2, T = 2 * (u & 3);
      if (ALO) {
        acc1[T] = mfma32(w0, h_lo[kk], acc1[T]);
        acc1[T + 1] = mfma32(w1, h_lo[kk], acc1[T + 1]);
      }
      acc1[T] = mfma32(w0, h_hi[kk], acc1[T]);
      acc1[T + 1] = mfma32(w1, h_hi[kk], acc1[T + 1]);
    });
  }
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring
  L32_STAMP(3);

  if constexpr (!QKV) {
    store_rows();
    L32_STAMP(4);
  } else {
    // ---- next layer's q / k / v^T: two 32-feature chunks per LDS stage and iteration ---------------------------
    constexpr int N_IT = 3 * NT / 2;  // 12 chunk pairs: 4 q, 4 k (one head each), 4 v
    constexpr int N_SW = 2 * NT / 2;  // q / k pairs ("swapped": weights as the A operand)
    auto stage_pair = [&](int it, int stage) {
#pragma unroll
      for (int u = 0; u < (2 * KS) / 4; ++u) {
        const int piece = wave + 4 * u;
        dma_piece(p.wqkv_p + (size_t)(2 * it) * CHUNK + piece * 512, stage, piece);
      }
    };
    stage_pair(0, 0);
    // RoPE rows of this lane's token: cos / sin [pos][16 q' + 8 hh + (0..7)] for the two chunks q' of a head
    f32x4 rc[2][2], rs[2][2];
    {
      int pos = p.row_pos[m0 + n];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      const float* cr = p.rope_cos + (size_t)pos * ROPE_HALF + 8 * hh;
      const float* sr = p.rope_sin + (size_t)pos * ROPE_HALF + 8 * hh;
#pragma unroll
      for (int qd = 0; qd < 2; ++qd)
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          rc[qd][q4] = *reinterpret_cast<const f32x4*>(cr + 16 * qd + 4 * q4);
          rs[qd][q4] = *reinterpret_cast<const f32x4*>(sr + 16 * qd + 4 * q4);
        }
    }
    layer_ln(no_, std::integral_constant<bool, ALO>{}, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk pair 0 and the RoPE rows have landed
    store_rows();
    __builtin_amdgcn_s_barrier();
    L32_STAMP(4);

    constexpr bool Q_LO = (OLO & 1) != 0, K_LO = (OLO & 2) != 0, V_LO = (OLO & 4) != 0;
    struct QkvOff {
      static constexpr int at(int s, int j) { return (j * KS + s) * 1024; }
    };
    f32x16 qa[2], qb[2];  // [iteration parity]: chunk 2 it / 2 it + 1
    uint4 st_v[8];
    u16* st_p[2];
    float e_lo[8], e_hi[8];  // q / k: RoPE'd halves of the chunk in flight
    uint2 e_h[4], e_l[4];
    // Epilogue of pair `it` (q / k: one head, chunks q' = 0, 1; v: one head, feature halves 0, 1), cut into KS slices
    // that ride along with the steps of the NEXT pair's MFMA stream (pure register work; the stores follow the stream).
    auto epilogue_slice = [&](int it, auto sw_tag, auto s_tag, const f32x16& ca, const f32x16& cb) {
      constexpr bool SW = decltype(sw_tag)::value;
      constexpr int s = decltype(s_tag)::value;
      constexpr int ch = s >> 3, i = s & 7;  // chunk of the pair, slice within the chunk
      const f32x16& c = ch == 0 ? ca : cb;
      const size_t rb = (size_t)((m0 >> 4) + (n >> 4));
      if constexpr (SW) {
        const bool is_q = it < N_SW / 2;
        const int head = is_q ? it : it - N_SW / 2;
        u16* out = is_q ? p.q_fp : p.k_fp;
        const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
        // slice i: RoPE of register i (d = 16 ch + 8 hh + i) and its partner d + 32 (register 8 + i)
        const float cc = rc[ch][i >> 2][i & 3], ss = rs[ch][i >> 2][i & 3];
        e_lo[i] = rope_lo(c[i], c[8 + i], cc, ss) * qscale;
        e_hi[i] = rope_hi(c[i], c[8 + i], cc, ss) * qscale;
        if constexpr ((i & 3) == 3) {
          split4<(Q_LO || K_LO)>(&e_lo[i - 3], e_h[(i >> 2)], e_l[(i >> 2)]);
          split4<(Q_LO || K_LO)>(&e_hi[i - 3], e_h[2 + (i >> 2)], e_l[2 + (i >> 2)]);
        }
        if constexpr (i == 7) {
          // k-step 2 head, granule 2 ch + hh; the partners d + 32: k-step 2 head + 1
          st_p[ch] = out + ((rb * NT + 2 * head) * 2) * 512 + (16 * (2 * ch + hh) + (n & 15)) * 8;
          st_v[4 * ch + 0] = make_uint4(e_h[0].x, e_h[0].y, e_h[1].x, e_h[1].y);
          st_v[4 * ch + 1] = make_uint4(e_h[2].x, e_h[2].y, e_h[3].x, e_h[3].y);
          st_v[4 * ch + 2] = make_uint4(e_l[0].x, e_l[0].y, e_l[1].x, e_l[1].y);
          st_v[4 * ch + 3] = make_uint4(e_l[2].x, e_l[2].y, e_l[3].x, e_l[3].y);
        }
      } else {
        // v^T: lane (feature column n, hh) holds tokens 8 j + 4 hh + r; key granule kg = 2 j' + hh of the 32-token tile
        // = tokens {4 kg + r, 16 + 4 kg + r} = registers {4 j' + r, 8 + 4 j' + r}, j' = 0, 1
        if constexpr ((i & 1) == 1) {
          constexpr int grp = i >> 1, jp = grp >> 1, part = grp & 1;
          const float vv[4] = {c[8 * part + 4 * jp], c[8 * part + 4 * jp + 1], c[8 * part + 4 * jp + 2], c[8 * part + 4 * jp + 3]};
          split4<V_LO>(vv, e_h[grp], e_l[grp]);
        }
        if constexpr (i == 7) {
          const size_t tb = (size_t)(m0 >> 5);
          const size_t n4 = (size_t)(2 * ch + (n >> 4));
          st_p[ch] = p.vt_fp + ((((size_t)(it - N_SW) * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + n4) * 512 + (16 * hh + (n & 15)) * 8;
          st_v[4 * ch + 0] = make_uint4(e_h[0].x, e_h[0].y, e_h[1].x, e_h[1].y);  // j' = 0
          st_v[4 * ch + 1] = make_uint4(e_l[0].x, e_l[0].y, e_l[1].x, e_l[1].y);
          st_v[4 * ch + 2] = make_uint4(e_h[2].x, e_h[2].y, e_h[3].x, e_h[3].y);  // j' = 1
          st_v[4 * ch + 3] = make_uint4(e_l[2].x, e_l[2].y, e_l[3].x, e_l[3].y);
        }
      }
    };
    auto epilogue_store = [&](int it, auto sw_tag) {
      constexpr bool SW = decltype(sw_tag)::value;
      if constexpr (SW) {
        const bool with_lo = it < N_SW / 2 ? Q_LO : K_LO;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          store_stream16(st_p[qd], st_v[4 * qd + 0]);
          store_stream16(st_p[qd] + 1024, st_v[4 * qd + 1]);
          if ((Q_LO || K_LO) && (Q_LO == K_LO || with_lo)) {
            store_stream16(st_p[qd] + 512, st_v[4 * qd + 2]);
            store_stream16(st_p[qd] + 1536, st_v[4 * qd + 3]);
          }
        }
      } else {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            store_stream16(st_p[ch] + 32 * 8 * jp, st_v[4 * ch + 2 * jp]);
            if (V_LO) store_stream16(st_p[ch] + 32 * 8 * jp + 2048, st_v[4 * ch + 2 * jp + 1]);
          }
      }
    };
    constexpr int N_ST_SW = 2 * (2 + ((Q_LO && K_LO) ? 2 : 0));
    constexpr int N_ST_V = 4 * (1 + (V_LO ? 1 : 0));
    auto iteration = [&](int it, auto cur_tag, auto first_tag, auto sw_tag, auto swp_tag) {
      constexpr int cur = decltype(cur_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value, SW = decltype(sw_tag)::value, SWP = decltype(swp_tag)::value;
      stage_pair(it + 1 < N_IT ? it + 1 : it, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      frag_stream2<KS, 2, QkvOff>(lds_stage[cur], [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
        constexpr int s = decltype(step_tag)::value;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ALO) {
          qa[cur] = SW ? mfma32(w0, a_lo[s], s == 0 ? zero : qa[cur]) : mfma32(a_lo[s], w0, s == 0 ? zero : qa[cur]);
          qb[cur] = SW ? mfma32(w1, a_lo[s], s == 0 ? zero : qb[cur]) : mfma32(a_lo[s], w1, s == 0 ? zero : qb[cur]);
        }
        qa[cur] = SW ? mfma32(w0, a_hi[s], (s == 0 && !ALO) ? zero : qa[cur]) : mfma32(a_hi[s], w0, (s == 0 && !ALO) ? zero : qa[cur]);
        qb[cur] = SW ? mfma32(w1, a_hi[s], (s == 0 && !ALO) ? zero : qb[cur]) : mfma32(a_hi[s], w1, (s == 0 && !ALO) ? zero : qb[cur]);
        if constexpr (!FIRST) {
          epilogue_slice(it - 1, swp_tag, step_tag, qa[cur ^ 1], qb[cur ^ 1]);
          interleave();
        }
      });
      if (!FIRST) epilogue_store(it - 1, swp_tag);
      constexpr int N_STORES = FIRST ? 0 : (SWP ? N_ST_SW : N_ST_V);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
      __builtin_amdgcn_s_barrier();
    };
    const std::integral_constant<int, 0> even{};
    const std::integral_constant<int, 1> odd{};
    iteration(0, even, yes_, yes_, yes_);
    iteration(1, odd, no_, yes_, yes_);
    for (int i0 = 2; i0 < N_SW; i0 += 2) {
      iteration(i0, even, no_, yes_, yes_);
      iteration(i0 + 1, odd, no_, yes_, yes_);
    }
    iteration(N_SW, even, no_, no_, yes_);
    iteration(N_SW + 1, odd, no_, no_, no_);
    for (int i0 = N_SW + 2; i0 < N_IT; i0 += 2) {
      iteration(i0, even, no_, no_, no_);
      iteration(i0 + 1, odd, no_, no_, no_);
    }
    static_for<KS>([&](auto s_tag) { epilogue_slice(N_IT - 1, no_, s_tag, qa[1], qb[1]); });
    epilogue_store(N_IT - 1, no_);
  }
  L32_STAMP(5);
#ifdef OPK_TIMING
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 16 + i] = opk_ts[i];
    p.dbg[(size_t)blockIdx.x * 16 + 8] = opk_wait;
    p.dbg[(size_t)blockIdx.x * 16 + 15] = wall_clock64() - opk_rt0;
  }
#endif
#undef L32_STAMP
}

}  // namespace opk
