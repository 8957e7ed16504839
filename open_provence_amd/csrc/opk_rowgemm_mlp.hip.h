// opk_rowgemm_mlp.hip.h -- RowGemmBlock::mlp_phase(): the whole MLP of a layer between phase 1 and the chunk loop, h kept on chip
// (whole-layer kernel, RP_MLP).  The loop's operations and register arrays are members (opk_rowgemm_mlp_members.inc).
#pragma once

namespace opk {

OPK_RG_TPL __device__ __forceinline__ void OPK_RG_BLOCK::mlp_phase() {
  const std::true_type yes_{};
  const std::false_type no_{};
  n_pairs = p.n_pairs;
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

#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    hold_hi[mf] = hold_lo[mf] = make_uint2(0u, 0u);
    h_hi[mf] = h_lo[mf] = as_frag(make_uint4(0u, 0u, 0u, 0u));
    acc_b[0][mf] = acc_b[1][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
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
  auto no_rd = [](auto, f32x4&) {};
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
}

}  // namespace opk
