// opk_rowgemm_stream.hip.h -- LDS stage layouts and per-chunk MFMA streams of rowgemm_kernel (bf16 / fp16 planes and the
// fp16 + e4m3 format); second part of what opk_rowgemm.hip.h provides
#pragma once

#include "opk_rowgemm_pack.hip.h"

namespace opk {

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

// a reference to `a` (FIRST) or `b`: the choice is a compile-time fact, both have the same type
template <bool FIRST, class T>
__device__ __forceinline__ T& pick_ref(T& a, T& b) {
  if constexpr (FIRST) return a;
  else return b;
}

// final_norm + pruning head on the rows in the accumulators (see RowGemmParams::fin_ln).  Same arithmetic per row
// as rowgemm_kernel's layer_ln; the head's two dot products ride on the normalised values (or, fin_pre_norm, on the raw row), the
// four lanes of a row are summed, the lane of column group 0 writes the token's logits and keep-probability.
// sLn: [mlp_norm | final_norm | pruning head row 0 | row 1] weights in LDS; acc1: the block's rows (x after the MLP).
template <int KS, int MF>
__device__ __forceinline__ void rowgemm_final_head(const RowGemmParams& p, const float* sLn, int m0, int l15, int g,
                                                   const f32x4 (&acc1)[2 * KS][MF]) {
  constexpr int K = KS * 32, NF1 = 2 * KS;
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
}

}  // namespace opk
