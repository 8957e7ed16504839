// opk_rowgemm_qkv_pairs.hip.h -- RowGemmBlock::qkv_pairs_loop(): fp16 + e4m3 kernel sets, the q / k / v^T projection as one
// fragment stream per chunk pair
#pragma once

namespace opk {

OPK_RG_TPL __device__ __forceinline__ void OPK_RG_BLOCK::qkv_pairs_loop() {
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

}

}  // namespace opk
