// opk_layer32p.hip.h -- the whole-layer kernel as WAVE PAIRS on v_mfma_f32_32x32x16 (hidden = 256, single-pass operands)
#pragma once

#include "opk_layer32.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// One launch per layer, the arithmetic of layer32_kernel (opk_layer32.hip.h) and of rowgemm_kernel<.., RP_MLP, ..>:
//   x += o Wo^T ; LayerNorm ; x += GeGLU(LN(x) Wi^T) Wo^T with h on chip ; LayerNorm ; the next layer's q / k / v^T
// Why another form (round 6).  Every phase of this launch is bound by instruction ISSUE, not by HBM (with every HBM stream
// removed it is 6 % shorter: profiles/r06_exp_stagger_memfree.txt) and not by the matrix pipe (busy 40 % of the cycles).
// A SIMD issues about one instruction per 4.5 cycles whichever of its waves it comes from; what the forms so far pay per
// 16 cycles of matrix work:
//   8 waves x 16 rows (two waves per SIMD, 16x16x32):  one MFMA + ONE weight-fragment read (a fragment serves one MFMA)
//                                                      + 1.7 vector instructions of GeGLU + ~1.5 of moves / waits / nops;
//   4 waves x 32 rows (one wave per SIMD, 512 registers): half the fragment reads, but nothing fills the wave's own
//                                                      stalls (MFMA -> accvgpr_read, LDS latency, barriers): 65 % at best.
// Two waves per SIMD AND 32 rows per wave needs ~300 registers of the 256 -- because a wave holds all 256 output
// features of its rows (128 accumulators).  So here the two waves of a pair SHARE one 32-row tile and split the OUTPUT
// FEATURES of every contraction:
//   wave (pw, hf), pw = row tile 0..3 of the 128-row block, hf = half 0 / 1
//   attention-output projection, MLP output projection:  output tiles 4 hf .. 4 hf + 3 of 8     (64 accumulators)
//   Wi:  tile 2 t + hf of pair t (16 h-columns + their gates)                                   (16 accumulators, x 2 for the pipeline)
//   next q / k / v^T:  tile 2 it + hf of pair it
// Each weight fragment (1 KiB) is read from LDS by ONE wave and feeds one 32-cycle MFMA: half the reads per flop of the
// 16-row form, half its MFMA instructions, at two waves per SIMD.  What a wave lacks of its rows comes from its partner
// through LDS: the h fragment of the other half (1 KiB per pair step, consumed one step later so that the block's single
// barrier per step orders it), and after each LayerNorm the other half's normalised fragments (8 KiB per wave) with the
// row statistics combined from the two halves' (mean, M2) by the parallel-variance formula -- nothing else crosses waves.
// Layouts of operands and weights are layer32_kernel's (l32_source_row; packs with fp16 bits for H16).
// K order: a wave multiplies its OWN eight k-steps first, then its partner's (u = 0..15 <-> k-step (8 hf + u) mod 16): a
// feature's summation order depends on the feature only, never on where a row sits in the block.
// ----------------------------------------------------------------------------------------------

template <bool H16>
__device__ __forceinline__ f32x16 mfma32x(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <bool H16>
__device__ __forceinline__ uint32_t pack2x(float a, float b) {
  if constexpr (H16) return pack_f16x2(a, b);
  else return pack_bf16x2(a, b);
}

__device__ __forceinline__ void lds_write_frag(uint32_t lds_addr, const bf16x8& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_f2(uint32_t lds_addr, const f32x2& v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x2 lds_read_f2(uint32_t lds_addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait3(bf16x8& a, bf16x8& b, bf16x8& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}

// A stream of NSTEPS steps of NF (2 or 3) weight fragments each; fragment j of step s is at LDS byte offset Off::at(s, j)
// from address register Off::base(s, j) of `addr` (a wave's fragments sit behind up to NB wave-dependent bases); requested
// DEPTH steps ahead into DEPTH + 1 rotating register sets, waited for by count (frag_stream2 of opk_common.hip.h).
template <int NSTEPS, int NF, int DEPTH, class Off, int NB, class Body>
__device__ __forceinline__ void frag_stream_m(const uint32_t (&addr)[NB], Body&& body) {
  static_assert(NF == 2 || NF == 3, "two or three fragments per step");
  constexpr int SETS = DEPTH + 1;
  bf16x8 w[SETS][NF];
  auto read_group = [&](auto step_tag) {
    constexpr int s = decltype(step_tag)::value;
    static_for<NF>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      w[s % SETS][j] = lds_read_frag<Off::at(s, j)>(addr[Off::base(s, j)]);
    });
  };
  static_for<(DEPTH + 1 < NSTEPS ? DEPTH + 1 : NSTEPS)>([&](auto t) { read_group(t); });
  static_for<NSTEPS>([&](auto t) {
    constexpr int s = decltype(t)::value;
    constexpr int set = s % SETS;
    constexpr int ahead = (NSTEPS - 1 - s) < DEPTH ? (NSTEPS - 1 - s) : DEPTH;
    if constexpr (NF == 2) lds_wait2<2 * ahead>(w[set][0], w[set][1]);
    else lds_wait3<3 * ahead>(w[set][0], w[set][1], w[set][2]);
    body(t, w[set]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + DEPTH + 1 < NSTEPS) read_group(std::integral_constant<int, s + DEPTH + 1>{});
  });
}

// The residual stream between two launches of THIS kernel is kept "tiled" (XIN_T / XOUT_T): the 32 x 32 fp32 values of a row
// tile x feature tile as four 1 KiB pieces [i4][lane][4 floats] = registers 4 i4 .. 4 i4 + 3 of the accumulator layout, so that
// every load / store instruction of a wave moves one contiguous KiB.  Row-major rows (what the other kernels read and write:
// 16 bytes per lane in 64 different cache lines per instruction) cost this kernel 15 k cycles per block in each direction
// (profiles/r06_m32p_steps.txt); the first launch of a forward reads them, the last one writes them.
// NT = hidden / 32 (8).  QKV: the next layer's q / k / v^T follow (false: the last layer).  H16: fp16 operands (kernel set "f16").
template <int NT, bool QKV, bool H16, bool XIN_T = false, bool XOUT_T = false>
__global__ __launch_bounds__(512, 2) void layer32p_kernel(Layer32Params p) {
  static_assert(NT == 8, "written for hidden = 256");
  constexpr int H = NT * 32;
  constexpr int KS = NT * 2;                       // 16-wide k-steps of a K = H contraction
  constexpr int CHUNK = KS * 512;                  // elements of one 32-row weight tile (16 pieces)
  constexpr int SLAB = NT * 512;                   // elements of one k-step of a k-major weight (8 pieces)
  constexpr int STAGE = 2 * CHUNK + 2 * SLAB;      // MLP stage: [Wi tile 2t | Wi tile 2t+1 | two k-steps of Wo]: 48 KiB
  constexpr int STAGE_B = STAGE * 2;               // bytes
  constexpr int EXTRA = 16 * 512;                  // 16 KiB: h exchange (2 buffers x 4 pairs x 2 halves x 1 KiB)
  constexpr int XCH_B = STAGE_B;                   // LayerNorm exchange: stage 1 + EXTRA = 64 KiB = 8 waves x 8 fragments
  constexpr int HX_B = 2 * STAGE_B;                // h exchange = EXTRA
  constexpr int SW_ELEMS = (2 * STAGE + EXTRA) > NT * 2 * SLAB ? (2 * STAGE + EXTRA) : NT * 2 * SLAB;  // phase 1 keeps all of Wo: 128 KiB
  __shared__ __attribute__((aligned(16))) u16 sW[SW_ELEMS];
  __shared__ __attribute__((aligned(16))) float sLn[2 * H];
  __shared__ __attribute__((aligned(16))) float sStat[8 * 32 * 2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pw = wave & 3, hf = wave >> 2;
  const int n = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * 128 + pw * 32;
#ifdef OPK_TIMING
  unsigned long long opk_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, opk_wait = 0, opk_x[4] = {0, 0, 0, 0};
  const unsigned long long opk_rt0 = wall_clock64();
#define L32P_STAMP(i) opk_ts[i] = __builtin_readcyclecounter()
#else
#define L32P_STAMP(i)
#endif
  L32P_STAMP(0);
  if (hf) __builtin_amdgcn_s_setprio(1);  // the younger half loses every arbitration otherwise (older-first at equal priority)

  const int ln_i = tid < H ? tid : H - 1;
  const float ln_fill0 = p.ln_mlp[ln_i];
  float ln_fill1 = 0.f;
  if (QKV) ln_fill1 = p.ln_next[ln_i];

  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0]) + (uint32_t)lane * 16u;
  const uint32_t stat_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)&sStat[0]);
  auto dma_piece = [&](const u16* src_piece0, int dst_piece) {  // dst_piece counts 1 KiB pieces from the start of sW
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_piece0 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(&sW[dst_piece * 512]), 16, 0, 0);
  };
  constexpr int STAGE_PIECES = STAGE / 512;  // 48

  // ---- phase 1: acc1[i] = tile 4 hf + i of o Wo^T.  ALL of Wo (128 KiB, k-major) is requested up front into the start of sW,
  // then the o fragments, LAST the residual rows: vmcnt retires in order, so the contraction starts as soon as Wo and o are
  // there and runs while the rows of x -- needed by the LayerNorm only -- are still arriving; no barrier inside the phase ----
  float* xrow = p.x_io + (size_t)(m0 + n) * H + 8 * hh + 128 * hf;
  // tiled: piece ((row tile * NT + feature tile) * 4 + i4) of 256 floats, this lane's 4 floats at lane * 4
  float* xtile = p.x_io + ((size_t)(m0 >> 5) * NT + 4 * hf) * 1024 + lane * 4;
#pragma unroll
  for (int u = 0; u < 16; ++u) {  // 128 pieces [k-step][tile], 16 per wave
    const int piece = wave + 8 * u;
    dma_piece(p.wo_p + (size_t)piece * 512, piece);
  }
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 a[KS];  // B operands of this wave's rows: phase 1 = o, k-step order; afterwards LN(x), own-first order
  {
    const u16* o_base = p.o_fp + ((size_t)((m0 >> 4) + (n >> 4)) * NT * 2) * 512 + (16 * hh + (n & 15)) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = load_stream_frag(o_base + ((ks >> 1) * 2) * 512 + (ks & 1) * 256);
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 xa[4][2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4)
        xa[i][jj][q4] = XIN_T ? load_stream_f4(xtile + i * 1024 + (2 * jj + q4) * 256) : load_stream_f4(xrow + 32 * i + 16 * jj + 4 * q4);
  __builtin_amdgcn_sched_barrier(0);
  sLn[ln_i] = ln_fill0;
  sLn[H + ln_i] = ln_fill1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // all but the residual rows
  __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
  opk_x[0] = __builtin_readcyclecounter();
#endif

  f32x16 acc1[4];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    struct P1Off {  // step st = (k-step st / 2, own tile pair st % 2); k-steps 8..15 sit behind the second base (+ 64 KiB)
      static constexpr int at(int st, int j) { return (((st >> 1) & 7) * NT + 2 * (st & 1) + j) * 1024; }
      static constexpr int base(int st, int) { return (st >> 1) >= 8 ? 1 : 0; }
    };
    const uint32_t addr[2] = {lds0 + (uint32_t)hf * 4096u, lds0 + 65536u + (uint32_t)hf * 4096u};
    frag_stream_m<32, 2, 2, P1Off, 2>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
      constexpr int st = decltype(step_tag)::value;
      constexpr int ks = st >> 1, i0 = 2 * (st & 1);
      acc1[i0] = mfma32x<H16>(w[0], a[ks], ks == 0 ? zero16 : acc1[i0]);
      acc1[i0 + 1] = mfma32x<H16>(w[1], a[ks], ks == 0 ? zero16 : acc1[i0 + 1]);
    });
  }
  __builtin_amdgcn_s_barrier();  // every wave is done with Wo: the MLP's first stage may land on it

  // ---- MLP stage DMA: stage t = [Wi tiles 2t, 2t+1 | Wo k-steps 2(t-2), 2(t-2)+1]; pieces wave + 8 u, u = 0..5 ----
  const int n_it = p.n_pairs;  // I / 32 pair steps
  auto stage_piece = [&](auto u_tag, int t, int stage) {
    constexpr int u = decltype(u_tag)::value;
    const int tc = t < n_it ? t : n_it - 1;
    const int ts = t >= 2 ? (t - 2 < n_it ? t - 2 : n_it - 1) : 0;
    const int piece = wave + 8 * u;
    const u16* src = u < 4 ? p.wi_p + (size_t)(2 * tc) * CHUNK + piece * 512 : p.wo2_p + (size_t)(2 * ts) * SLAB + (piece - 32) * 512;
    dma_piece(src, stage * STAGE_PIECES + piece);
  };
  static_for<6>([&](auto u) { stage_piece(u, 0, 0); });  // flies during the LayerNorm
  L32P_STAMP(1);

  // ---- LayerNorm of the rows in the accumulators (LOAD: acc1 += x first) -> a[0..7] own fragments, a[8..15] the partner's ----
  auto layer_ln = [&](auto load_tag, int which) {
    constexpr bool LOAD = decltype(load_tag)::value;
    // vector-only phase: packed fp32 arithmetic (two values per instruction at a scalar FMA's issue cost here; see pk_add)
    f32x2 v[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x16 t = acc1[i];
      if constexpr (LOAD) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            const float4 x4 = xa[i][jj][q4];
            t[8 * jj + 4 * q4 + 0] += x4.x;
            t[8 * jj + 4 * q4 + 1] += x4.y;
            t[8 * jj + 4 * q4 + 2] += x4.z;
            t[8 * jj + 4 * q4 + 3] += x4.w;
          }
        acc1[i] = t;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[i][r] = f32x2{t[2 * r], t[2 * r + 1]};
    }
    // own half (128 features): mean and M2 in two passes, four packed chains each
    f32x2 s4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 8; ++r) s4[r & 3] = (i == 0 && r < 4) ? v[i][r] : pk_add(s4[r & 3], v[i][r]);
    const f32x2 st2 = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
    float sum = st2.x + st2.y;
    sum += __shfl_xor(sum, 32, 64);
    const float mean_a = sum * (1.0f / 128.0f);
    const f32x2 ma2 = f32x2{mean_a, mean_a};
    f32x2 q4s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[i][r] = pk_sub(v[i][r], ma2);
        q4s[r & 3] = (i == 0 && r < 4) ? pk_mul(v[i][r], v[i][r]) : pk_fma(v[i][r], v[i][r], q4s[r & 3]);
      }
    const f32x2 qt2 = pk_add(pk_add(q4s[0], q4s[1]), pk_add(q4s[2], q4s[3]));
    float m2_a = qt2.x + qt2.y;
    m2_a += __shfl_xor(m2_a, 32, 64);
    lds_write_f2(stat_addr + (uint32_t)((wave * 32 + n) * 8), f32x2{mean_a, m2_a});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
    if (which == 1) opk_x[1] = __builtin_readcyclecounter();
#endif
    const f32x2 other = lds_read_f2(stat_addr + (uint32_t)(((wave ^ 4) * 32 + n) * 8));
    // the two halves combined (parallel variance: n_a = n_b = 128)
    const float mean = 0.5f * (mean_a + other.x);
    const float dm = mean_a - other.x;
    const float m2 = (m2_a + other.y) + dm * dm * 64.0f;
    const float rstd = 1.0f / sqrtf(m2 * (1.0f / (float)H) + p.eps);
    const float shift = mean_a - mean;
    const f32x2 sh2 = f32x2{shift, shift}, r2 = f32x2{rstd, rstd};
    // normalise the own features, weights from LDS: columns 128 hf + 32 i + 16 jj + 8 hh + (0..7)
    const float* lw = &sLn[which * H + 128 * hf + 8 * hh];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(lw + 32 * i + 16 * jj);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(lw + 32 * i + 16 * jj + 4);
        const f32x2 wv[4] = {f32x2{w0[0], w0[1]}, f32x2{w0[2], w0[3]}, f32x2{w1[0], w1[1]}, f32x2{w1[2], w1[3]}};
        uint32_t d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 y = pk_mul(pk_mul(pk_add(v[i][4 * jj + e], sh2), r2), wv[e]);
          d[e] = pack2x<H16>(y.x, y.y);
        }
        a[2 * i + jj] = as_frag(make_uint4(d[0], d[1], d[2], d[3]));
        lds_write_frag(lds0 + (uint32_t)XCH_B + (uint32_t)((wave * 8 + 2 * i + jj) * 1024), a[2 * i + jj]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t oth = lds0 + (uint32_t)XCH_B + (uint32_t)(((wave ^ 4) * 8) * 1024);
    static_for<8>([&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      a[8 + f] = lds_read_frag<f * 1024>(oth);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4)
          store_stream16(XOUT_T ? xtile + i * 1024 + (2 * jj + q4) * 256 : xrow + 32 * i + 16 * jj + 4 * q4,
                         make_float4(acc1[i][8 * jj + 4 * q4], acc1[i][8 * jj + 4 * q4 + 1], acc1[i][8 * jj + 4 * q4 + 2], acc1[i][8 * jj + 4 * q4 + 3]));
  };
  const std::true_type yes_{};
  const std::false_type no_{};

  layer_ln(yes_, 0);
  __builtin_amdgcn_s_barrier();  // every wave has read its partner's fragments: the h exchange area (inside it) may be cleared
  const uint32_t hx_own = lds0 + (uint32_t)HX_B + (uint32_t)((pw * 2 + hf) * 1024);   // + buf * 8192
  const uint32_t hx_pair = lds0 + (uint32_t)HX_B + (uint32_t)((pw * 2) * 1024);       // + buf * 8192 + half * 1024
  {
    const bf16x8 zf = as_frag(make_uint4(0u, 0u, 0u, 0u));
    lds_write_frag(hx_own, zf);
    lds_write_frag(hx_own + 8192u, zf);
  }

  // ---- MLP: pair step t = [Wi tile 2t+hf -> na[t & 1]] + [GeGLU of na[(t-1) & 1] -> own half of h(t-1) -> LDS] + [acc1 += h(t-2) Wo^T] ----
  f32x16 na[2];
  na[0] = zero16;
  na[1] = zero16;
  float gx[8], gq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) gx[i] = gq[i] = 0.f;
  // this wave's fragment bases inside a stage (byte addresses, + stage * STAGE_B): Wi tile hf, own-first k order; Wo tiles 4 hf ..
  const uint32_t b_lo = lds0 + (uint32_t)hf * (16384u + 8192u);       // u <  8: k-step 8 hf + u      at + u * 1024
  const uint32_t b_hi = lds0 + (uint32_t)hf * (16384u - 8192u);       // u >= 8: k-step u - 8 hf      at + u * 1024
  const uint32_t b_wo = lds0 + 32768u + (uint32_t)hf * 4096u;         // Wo fragment (kk, i)          at + (8 kk + i) * 1024
  struct MlpOff {  // step g: Wi u = 2g | Wo fragment g = (kk = g / 4, i = g % 4) | Wi u = 2g + 1
    static constexpr int at(int g, int j) { return j == 1 ? ((g >> 2) * 8 + (g & 3)) * 1024 : (2 * g + (j >> 1)) * 1024; }
    static constexpr int base(int g, int j) { return j == 1 ? 2 : ((2 * g + (j >> 1)) < 8 ? 0 : 1); }
  };
  struct WoOff {  // tail: step g = Wo fragments 2g, 2g + 1
    static constexpr int at(int g, int j) { return (((2 * g + j) >> 2) * 8 + ((2 * g + j) & 3)) * 1024; }
    static constexpr int base(int, int) { return 0; }
  };
  // stage `stg` (0..7) of the GeGLU of the tile in `src` (registers 0..7 inputs, 8..15 gates); stage 7 packs the eight values
  // (= this lane's k of k-step 2(t-1)+hf of the Wo contraction) and hands the fragment to the partner
  auto geglu_stage = [&](auto stg_tag, const f32x16& src, uint32_t dst_addr) {
    constexpr int stg = decltype(stg_tag)::value;
    static_for<8>([&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value;
      if constexpr (stg == 0) {
        gx[i] = src[i];
        gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
      } else if constexpr (stg < 5) gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), stg);
      else if constexpr (stg == 5) gq[i] = __builtin_amdgcn_exp2f(gq[i]);
      else if constexpr (stg == 6) gq[i] = gelu_erf_finish(gq[i], gx[i]);
      else gq[i] = gq[i] * src[8 + i];
    });
    if constexpr (stg == 7) {
      const bf16x8 hfrag = as_frag(make_uint4(pack2x<H16>(gq[0], gq[1]), pack2x<H16>(gq[2], gq[3]), pack2x<H16>(gq[4], gq[5]), pack2x<H16>(gq[6], gq[7])));
      lds_write_frag(dst_addr, hfrag);
    }
  };
  auto interleave3 = [&]() {  // three MFMAs of a step, the step's vector slice spread between them
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
  };
  auto end_of_stage = [&]() {
#ifdef OPK_TIMING
    const unsigned long long w0_ = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef OPK_TIMING
    opk_wait += __builtin_readcyclecounter() - w0_;
#endif
  };
  auto macro = [&](int t, auto par_tag) {
    constexpr int P = decltype(par_tag)::value;  // t & 1 = LDS stage, accumulator, h buffer of h(t-2)
    bf16x8 hb[2];
    hb[0] = lds_read_frag<0>(hx_pair + (uint32_t)(P * 8192));
    hb[1] = lds_read_frag<1024>(hx_pair + (uint32_t)(P * 8192));
    const uint32_t addr[3] = {b_lo + (uint32_t)(P * STAGE_B), b_hi + (uint32_t)(P * STAGE_B), b_wo + (uint32_t)(P * STAGE_B)};
    const uint32_t h_dst = hx_own + (uint32_t)((P ^ 1) * 8192);
    frag_stream_m<8, 3, 1, MlpOff, 3>(addr, [&](auto step_tag, bf16x8(&w)[3]) {
      constexpr int g = decltype(step_tag)::value;
      if constexpr (g == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));  // (read before the stream's first group: landed with it)
      if constexpr (g < 6) stage_piece(step_tag, t + 1, P ^ 1);
      na[P] = mfma32x<H16>(w[0], a[2 * g], g == 0 ? zero16 : na[P]);
      acc1[g & 3] = mfma32x<H16>(w[1], hb[g >> 2], acc1[g & 3]);
      na[P] = mfma32x<H16>(w[2], a[2 * g + 1], na[P]);
      geglu_stage(step_tag, na[P ^ 1], h_dst);
      interleave3();
    });
    end_of_stage();
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // stage 0 has landed, the h buffers are clear
  L32P_STAMP(2);
  {
    int t = 0;
    do {  // n_it is even (checked on the host)
      macro(t, std::integral_constant<int, 0>{});
      macro(t + 1, std::integral_constant<int, 1>{});
      t += 2;
    } while (t < n_it);
  }
  {  // tail: GeGLU of the last tile (na[1]) beside acc1 += h(n_it - 2) Wo^T, then acc1 += h(n_it - 1) Wo^T
    bf16x8 hb[2];
    hb[0] = lds_read_frag<0>(hx_pair);
    hb[1] = lds_read_frag<1024>(hx_pair);
    {
      const uint32_t addr[1] = {b_wo};
      frag_stream_m<4, 2, 1, WoOff, 1>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
        constexpr int g = decltype(step_tag)::value;
        if constexpr (g == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
        if constexpr (g < 2) {  // stage n_it + 1 needs only its Wo k-steps (pieces 32..47: u = 4, 5)
          stage_piece(std::integral_constant<int, 4 + g>{}, n_it + 1, 1);
        }
        acc1[(2 * g) & 3] = mfma32x<H16>(w[0], hb[(2 * g) >> 2], acc1[(2 * g) & 3]);
        acc1[(2 * g + 1) & 3] = mfma32x<H16>(w[1], hb[(2 * g + 1) >> 2], acc1[(2 * g + 1) & 3]);
        geglu_stage(std::integral_constant<int, 2 * g>{}, na[1], hx_own + 8192u);
        geglu_stage(std::integral_constant<int, 2 * g + 1>{}, na[1], hx_own + 8192u);
      });
    }
    end_of_stage();
    hb[0] = lds_read_frag<0>(hx_pair + 8192u);
    hb[1] = lds_read_frag<1024>(hx_pair + 8192u);
    {
      const uint32_t addr[1] = {b_wo + (uint32_t)STAGE_B};
      frag_stream_m<4, 2, 1, WoOff, 1>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
        constexpr int g = decltype(step_tag)::value;
        if constexpr (g == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
        acc1[(2 * g) & 3] = mfma32x<H16>(w[0], hb[(2 * g) >> 2], acc1[(2 * g) & 3]);
        acc1[(2 * g + 1) & 3] = mfma32x<H16>(w[1], hb[(2 * g + 1) >> 2], acc1[(2 * g + 1) & 3]);
      });
    }
  }
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring
  L32P_STAMP(3);

  if constexpr (!QKV) {
    store_rows();
    L32P_STAMP(4);
  } else {
    // ---- next layer's q / k / v^T: pair step `it` = tiles 2 it (wave half 0) and 2 it + 1 (half 1), one LDS stage --------------
    constexpr int N_IT = 3 * NT / 2;  // 12 pair steps: 4 q, 4 k (one head each), 4 v
    constexpr int N_SW = 2 * NT / 2;  // q / k steps ("swapped": weights as the A operand)
    auto stage_pair = [&](int it, int stage) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int piece = wave + 8 * u;
        dma_piece(p.wqkv_p + (size_t)(2 * it) * CHUNK + piece * 512, stage * STAGE_PIECES + piece);
      }
    };
    stage_pair(0, 0);
    // RoPE rows of this lane's token: cos / sin [pos][16 hf + 8 hh + (0..7)] (the own tile = half hf of a head's rotary pairs)
    f32x4 rc[2], rs[2];
    {
      int pos = p.row_pos[m0 + n];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      const float* cr = p.rope_cos + (size_t)pos * ROPE_HALF + 16 * hf + 8 * hh;
      const float* sr = p.rope_sin + (size_t)pos * ROPE_HALF + 16 * hf + 8 * hh;
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        rc[q4] = *reinterpret_cast<const f32x4*>(cr + 4 * q4);
        rs[q4] = *reinterpret_cast<const f32x4*>(sr + 4 * q4);
      }
    }
    layer_ln(no_, 1);
#ifdef OPK_TIMING
    opk_x[2] = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pair 0 and the RoPE rows have landed
#ifdef OPK_TIMING
    opk_x[3] = __builtin_readcyclecounter();
#endif
    store_rows();
    __builtin_amdgcn_s_barrier();  // pair 0 everywhere; everyone has read its partner's fragments (stage 1 is free)
    L32P_STAMP(4);

    struct QkvOff {  // step g: k-order positions u = 2g, 2g + 1 of the own tile
      static constexpr int at(int g, int j) { return (2 * g + j) * 1024; }
      static constexpr int base(int g, int j) { return (2 * g + j) < 8 ? 0 : 1; }
    };
    f32x16 qa[2];  // [pair step parity]
    uint4 st_v[4];
    u16* st_p = nullptr;
    float e_lo[8], e_hi[8];
    uint2 e_h[4];
    const size_t rb = (size_t)((m0 >> 4) + (n >> 4));
    // Epilogue of the tile of pair step `it`, in 8 slices that ride on the NEXT step's MFMAs
    auto epilogue_slice = [&](int it, auto sw_tag, auto i_tag, const f32x16& c) {
      constexpr bool SW = decltype(sw_tag)::value;
      constexpr int i = decltype(i_tag)::value;
      if constexpr (SW) {
        const bool is_q = it < N_SW / 2;
        const int head = is_q ? it : it - N_SW / 2;
        u16* out = is_q ? p.q_fp : p.k_fp;
        const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
        // register i: d = 16 hf + 8 hh + i, register 8 + i: its RoPE partner d + 32
        const float cc = rc[i >> 2][i & 3], ss = rs[i >> 2][i & 3];
        e_lo[i] = rope_lo(c[i], c[8 + i], cc, ss) * qscale;
        e_hi[i] = rope_hi(c[i], c[8 + i], cc, ss) * qscale;
        if constexpr ((i & 3) == 3) {
          e_h[i >> 2] = make_uint2(pack2x<H16>(e_lo[i - 3], e_lo[i - 2]), pack2x<H16>(e_lo[i - 1], e_lo[i]));
          e_h[2 + (i >> 2)] = make_uint2(pack2x<H16>(e_hi[i - 3], e_hi[i - 2]), pack2x<H16>(e_hi[i - 1], e_hi[i]));
        }
        if constexpr (i == 7) {
          st_p = out + ((rb * NT + 2 * head) * 2) * 512 + (16 * (2 * hf + hh) + (n & 15)) * 8;
          st_v[0] = make_uint4(e_h[0].x, e_h[0].y, e_h[1].x, e_h[1].y);
          st_v[1] = make_uint4(e_h[2].x, e_h[2].y, e_h[3].x, e_h[3].y);
        }
      } else {
        // v^T: lane (feature column n, hh) holds tokens 8 j + 4 hh + r; key granule kg = 2 j' + hh = registers {4 j' + r, 8 + 4 j' + r}
        if constexpr ((i & 1) == 1) {
          constexpr int grp = i >> 1, jp = grp >> 1, part = grp & 1;
          e_h[grp] = make_uint2(pack2x<H16>(c[8 * part + 4 * jp], c[8 * part + 4 * jp + 1]), pack2x<H16>(c[8 * part + 4 * jp + 2], c[8 * part + 4 * jp + 3]));
        }
        if constexpr (i == 7) {
          const size_t tb = (size_t)(m0 >> 5);
          const size_t n4 = (size_t)(2 * hf + (n >> 4));
          st_p = p.vt_fp + ((((size_t)(it - N_SW) * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + n4) * 512 + (16 * hh + (n & 15)) * 8;
          st_v[0] = make_uint4(e_h[0].x, e_h[0].y, e_h[1].x, e_h[1].y);  // j' = 0
          st_v[1] = make_uint4(e_h[2].x, e_h[2].y, e_h[3].x, e_h[3].y);  // j' = 1
        }
      }
    };
    auto epilogue_store = [&](auto sw_tag) {
      constexpr bool SW = decltype(sw_tag)::value;
      if constexpr (SW) {
        store_stream16(st_p, st_v[0]);
        store_stream16(st_p + 1024, st_v[1]);
      } else {
        store_stream16(st_p, st_v[0]);
        store_stream16(st_p + 32 * 8, st_v[1]);
      }
    };
    auto interleave2 = [&]() {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    };
    auto iteration = [&](int it, auto cur_tag, auto first_tag, auto sw_tag, auto swp_tag) {
      constexpr int cur = decltype(cur_tag)::value;
      constexpr bool FIRST = decltype(first_tag)::value, SW = decltype(sw_tag)::value;
      stage_pair(it + 1 < N_IT ? it + 1 : it, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t addr[2] = {b_lo + (uint32_t)(cur * STAGE_B), b_hi + (uint32_t)(cur * STAGE_B)};
      frag_stream_m<8, 2, 2, QkvOff, 2>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
        constexpr int g = decltype(step_tag)::value;
        qa[cur] = SW ? mfma32x<H16>(w[0], a[2 * g], g == 0 ? zero16 : qa[cur]) : mfma32x<H16>(a[2 * g], w[0], g == 0 ? zero16 : qa[cur]);
        qa[cur] = SW ? mfma32x<H16>(w[1], a[2 * g + 1], qa[cur]) : mfma32x<H16>(a[2 * g + 1], w[1], qa[cur]);
        if constexpr (!FIRST) {
          epilogue_slice(it - 1, swp_tag, step_tag, qa[cur ^ 1]);
          interleave2();
        }
      });
      if constexpr (!FIRST) epilogue_store(swp_tag);
      constexpr int N_STORES = FIRST ? 0 : 2;
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
    static_for<8>([&](auto i_tag) { epilogue_slice(N_IT - 1, no_, i_tag, qa[1]); });
    epilogue_store(no_);
  }
  L32P_STAMP(5);
#ifdef OPK_TIMING
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 16 + i] = opk_ts[i];
    p.dbg[(size_t)blockIdx.x * 16 + 8] = opk_wait;
    for (int i = 0; i < 4; ++i) p.dbg[(size_t)blockIdx.x * 16 + 11 + i] = opk_x[i];
    p.dbg[(size_t)blockIdx.x * 16 + 15] = wall_clock64() - opk_rt0;
  }
#endif
#undef L32P_STAMP
}

}  // namespace opk
