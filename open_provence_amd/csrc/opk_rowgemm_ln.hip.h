// opk_rowgemm_ln.hip.h -- the transitions between the phases of rowgemm_kernel as functions: residual add + LayerNorm of the rows
// in the accumulators into MFMA operand fragments, and the write-back of the rows (the final head: opk_rowgemm_stream.hip.h).
// State they read and write, all in registers of the calling wave: acc1 = the block's rows [feature fragment][row fragment],
// xq0 = row fragment 0 of x requested at the top of the kernel (RP_MLP), a_hi / a_lo (/ a_lo8 / a_h8) = the fragments out.
#pragma once

#include "opk_rowgemm_stream.hip.h"

namespace opk {

// ---- transition: residual add, (store the new hidden state,) LayerNorm, split -> fragments -------------------
// LOAD: acc1 += x rows from memory; STORE: write the rows back; then LayerNorm with `lnw` into a_hi / a_lo.
template <int KS, int MF, int PRO, bool LOAD, bool STORE, bool LO, int NX>
__device__ __forceinline__ void rowgemm_residual_ln(const RowGemmParams& p, int m0, int l15, int g, const float* __restrict__ lnw,
                                                    f32x4 (&acc1)[2 * KS][MF], const float4 (&xq0)[NX], bf16x8 (&a_hi)[MF][KS],
                                                    bf16x8 (&a_lo)[MF][KS]) {
  constexpr int K = KS * 32, NF1 = 2 * KS;
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
}

// ---- the same transition in the whole-layer kernel (one wave per SIMD: a vector-only phase runs at one
// instruction per ~4.9 cycles, 8 for the conversion / accumulator-move class and for an instruction that needs
// the result of the one before it).  The row's 64 values per lane are read from the accumulators once, sums run in
// four independent chains of packed instructions, the LayerNorm weights come from LDS (sLn, above) and the
// write-back of the rows is left to store_rows(): a store in the middle puts every later counted vmcnt wait behind
// its write acknowledgement.
template <int KS, int MF, int F8, bool H16, bool LOAD, bool LO, int NS8, int NX>
__device__ __forceinline__ void rowgemm_layer_ln(const RowGemmParams& p, int m0, int l15, int g, uint32_t sln_addr, int which,
                                                 f32x4 (&acc1)[2 * KS][MF], const float4 (&xq0)[NX], bf16x8 (&a_hi)[MF][KS],
                                                 bf16x8 (&a_lo)[MF][KS], i32x8 (&a_lo8)[MF][NS8], i32x8 (&a_h8)[MF][NS8],
                                                 unsigned long long* stamps) {
  constexpr int K = KS * 32, NF1 = 2 * KS;
  constexpr bool WLO = F8 == 2;
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
  const uint32_t ln_addr = sln_addr + (uint32_t)(which * K + g * 8) * 4u;  // sln_addr: LDS byte address of the weight vectors
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
    if (stamps != nullptr && mf == 0) stamps[LOAD ? 0 : 1] = __builtin_readcyclecounter();  // (OPK_TIMING builds only)
  }
}

// The residual rows as they stand in the accumulators, in one burst behind the LayerNorm's arithmetic.  Measured
// (stamps): the arithmetic takes 5.5 k cycles, the 32 stores 5.7 k to ISSUE -- every CU of the chip writes its
// 128 KB at the same moment -- and spread between the arithmetic instructions they cost more (LayerNorm phase
// 11.2 k as a burst, 15 - 17.7 k interleaved).
template <int KS, int MF>
__device__ __forceinline__ void rowgemm_store_rows(const RowGemmParams& p, int m0, int l15, int g, const f32x4 (&acc1)[2 * KS][MF]) {
  constexpr int K = KS * 32, NF1 = 2 * KS;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    float* xrow = p.x_io + (size_t)(m0 + mf * 16 + l15) * K + g * 8;
#pragma unroll
    for (int nf = 0; nf < NF1; ++nf)
      store_stream16(xrow + 32 * (nf >> 1) + 4 * (nf & 1),
                     make_float4(acc1[nf][mf][0], acc1[nf][mf][1], acc1[nf][mf][2], acc1[nf][mf][3]));
  }
}

// ---- prologue of rowgemm_kernel: this wave's rows of the hidden state as MFMA operand fragments (a_hi / a_lo) ----------------
// RP_SPLIT (layer 0; attn_norm is Identity): the rows are built from the embedding table + embeddings.norm (p.emb_table) or read
// from x_in and split.  Every prologue but RP_MLP: the lo fragments are masked for a narrower policy (p.zero_a_lo).
template <int KS, int MF, int PRO, bool A_LO, bool H16>
__device__ __forceinline__ void rowgemm_prologue_rows(const RowGemmParams& p, int m0, int l15, int g, bf16x8 (&a_hi)[MF][KS],
                                                      bf16x8 (&a_lo)[MF][KS]) {
  constexpr int K = KS * 32;
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
}

}  // namespace opk
