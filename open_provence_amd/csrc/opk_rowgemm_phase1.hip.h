// opk_rowgemm_phase1.hip.h -- RowGemmBlock::phase1(): acc1 = A1 W1^T with K1 streamed (the attention output projection of the
// fused kernels); RP_MLP: also requests row fragment 0 of x (xq0) and publishes the LayerNorm weight vectors in LDS
#pragma once

namespace opk {

OPK_RG_TPL __device__ __forceinline__ void OPK_RG_BLOCK::phase1() {
  // ---- fused phase 1: x_new[32 rows, H] = x + A1[32 rows, K1] W1[H, K1]^T, K1 streamed ----------------
  // Same structure as kstream_gemm_kernel (one [H x 32] weight slab per k-step by DMA, A1 fragments straight
  // from the fragment-packed activation, prefetched one k-step ahead), all H outputs of the 32 rows in
  // accumulators.  W1's output features were permuted at load time so that accumulator fragments (2s, 2s+1)
  // are exactly lane slot g of k-step s of THIS kernel's chunk loop: residual add, LayerNorm and the hi/lo
  // split happen in registers and the hidden state makes one fp32 round trip (read + write) per block.
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
#pragma unroll
  for (int nf = 0; nf < NF1; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  // RP_MLP (one wave per SIMD, nobody to hide a load behind): everything phase 1 reads from HBM is requested up
  // front -- all K1 / 32 = KS fragment pairs of A1 (into a_hi / a_lo, which only the LayerNorm below overwrites) and
  // the residual rows x of row fragment 0 -- and the weight slabs come two k-steps per LDS stage (half the block
  // barriers, each DMA issued two k-steps ahead of its use).  Measured per 128-row block before / after: phase 1
  // 26k -> 12k cycles (8.3k are MFMAs), residual + LayerNorm 29k -> 11k (eight serialized HBM round trips gone).
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
}

}  // namespace opk
