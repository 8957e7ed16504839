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

}  // namespace opk
