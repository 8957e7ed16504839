// opk_rowgemm_chunks.hip.h -- RowGemmBlock::chunk_loop(): the weight-chunk loop with its deferred epilogues (q / k / v^T + RoPE,
// GeGLU); chunk pairs per barrier for kernel set "f16"
#pragma once

namespace opk {

OPK_RG_TPL __device__ __forceinline__ void OPK_RG_BLOCK::chunk_loop() {
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
}

}  // namespace opk
