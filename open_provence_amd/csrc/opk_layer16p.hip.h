// opk_layer16p.hip.h -- the whole-layer kernel as WAVE PAIRS (hidden = 256, single-pass operands), 16x16x32 MFMAs
#pragma once

#include "opk_layer32.hip.h"  // Layer32Params

namespace opk {

// ----------------------------------------------------------------------------------------------
// One launch per layer, the arithmetic of rowgemm_kernel<.., RP_MLP, ..> (opk_rowgemm.hip.h):
//   x += o Wo^T ; LayerNorm ; x += GeGLU(LN(x) Wi^T) Wo^T with h on chip ; LayerNorm ; the next layer's q / k / v^T
// Why another form (round 6).  Every phase of this launch is bound by instruction ISSUE and by the board's POWER limit, not by
// HBM (with every HBM stream removed the 8 x 16 form is 6 % shorter: profiles/r06_exp_stagger_memfree.txt) and not by the
// matrix pipe (busy 40 % of the cycles; the chip holds 1.8 - 1.95 of its 2.4 GHz under these kernels, 2.17 on all-zero data).
// What the forms so far pay per 16 cycles of matrix work:
//   8 waves x 16 rows (two waves per SIMD):  one MFMA + ONE 1 KiB weight-fragment read from LDS (a fragment serves one MFMA)
//                                            + 1.7 vector instructions of GeGLU + ~1.5 of moves / waits / nops;
//   4 waves x 32 rows (one wave per SIMD, 512 registers): half the fragment reads, but nothing fills the wave's own stalls.
// Two waves per SIMD AND 32 rows per wave needs ~300 of the 256 registers -- because a wave holds all 256 output features
// of its rows (128 accumulators).  Here the two waves of a pair SHARE one 32-row tile and split the OUTPUT FEATURES of every
// contraction:
//   wave (pw, hf), pw = row tile 0..3 of the 128-row block, hf = half 0 / 1
//   attention-output projection, MLP output projection:  output tiles 4 hf .. 4 hf + 3 of 8     (64 accumulators)
//   Wi:  tile 2 t + hf of pair step t (16 h-columns + their gates)                              (16 accumulators, x 2 for the pipeline)
//   next q / k / v^T:  tile 2 it + hf of pair step it
// A weight fragment (16 features x 32 k, 1 KiB) is read from LDS by ONE wave and multiplied against BOTH 16-row halves of the
// tile: half the LDS reads per flop of the 8 x 16 form at the same two waves per SIMD.  (The same split on 32x32x16 MFMAs
// -- half the MFMA instructions again -- was built first and measured: 12 % fewer cycles, but that shape draws so much more
// power per flop that the chip clocks 8 % lower under it: profiles/r06_pair_kernel_steps.txt.)  What a wave lacks of its rows
// comes from its partner through LDS: the other half of each h fragment (8 bytes per lane and 16 rows per pair step, consumed
// one step later so that the block's single barrier per step orders it), and after each LayerNorm the partner's normalised
// fragments (8 KiB per wave), the row statistics combined from the halves' (mean, M2) by the parallel-variance formula.
//
// Register layout (v_mfma_f32_16x16x32): lane = (n16 = lane % 16, g = lane / 16); the wave's rows are m0 + 16 mf + n16, mf = 0, 1.
//   "swapped" products (weights = X operand, token rows = Y): D[fh][mf] of a 32-feature tile: lane (n16, g) holds slots
//   16 fh + 4 g + r (r = 0..3) of row 16 mf + n16.  l16p_source_row() permutes the weight rows so that a lane's 8 slots of a
//   tile are the 8 consecutive k = 32 T + 8 g + (0..7) it needs as the Y operand of the next contraction (k-pair T), a GeGLU
//   input beside its gate (fh = 0 / 1), or a RoPE pair (d, d + 32).  Operand layouts in memory are the other kernels' own:
//   o, q, k as 1 KiB pieces [row/16][C/32][plane][16 (k%32/8) + row%16][8] (= one Y fragment), v^T as
//   [head][row/32][plane][4][16 kg + d'][8 keys]; x as fp32 rows, or TILED between two launches of this kernel (XIN_T / XOUT_T:
//   the 32 x 32 values of a row tile x feature tile as four 1 KiB pieces [2 mf + fh][lane][4 floats], so that every load / store
//   moves one contiguous KiB -- rows cost 16 bytes per lane in 16 different cache lines per instruction).
// K order: a wave multiplies its OWN four k-pairs first, then its partner's: a feature's summation order depends on the
// feature only, never on where a row sits in the block.
// ----------------------------------------------------------------------------------------------

enum Layer16pPack { L16_RESID = 0, L16_GEGLU = 1, L16_QKV = 2 };

// source row of slot m (0..31; fh = m / 16, rho = m % 16 = 4 g + r) of 32-row tile T of a weight matrix
__host__ __device__ inline int l16p_source_row(int mode, int T, int m, int H, int I) {
  const int fh = m >> 4, rho = m & 15, g = rho >> 2, r = rho & 3;
  if (mode == L16_RESID) return 32 * T + 8 * g + 4 * fh + r;
  if (mode == L16_GEGLU) {  // tile 2t + hf: h columns 32 t + 8 g + 4 hf + r (fh = 0) and their gates (fh = 1)
    const int col = 32 * (T >> 1) + 8 * g + 4 * (T & 1) + r;
    return fh ? I + col : col;
  }
  const int per = H / 32;  // tiles in each of q, k, v
  if (T < 2 * per) {       // q / k: 16 d of a head (fh = 0) and their RoPE partners d + 32 (fh = 1)
    const int blk = T / per, cc = T % per;
    return blk * H + (cc >> 1) * HEAD_DIM + 16 * (cc & 1) + 4 * g + r + 32 * fh;
  }
  // v (tokens x features orientation): column rho of fragment fh is feature slot (piece n = 2 (cv & 1) + fh, d' = rho) of the
  // transposed layout, in the d order of rowgemm_source_row (the attention output then is lane-contiguous)
  const int cv = T - 2 * per;
  return 2 * H + (cv >> 1) * HEAD_DIM + 32 * (cv & 1) + 8 * (rho >> 2) + 4 * fh + (rho & 3);
}

#ifdef OPK_PACK_KERNELS
// one plane.  chunk-major: dst[T][kp][fh][lane][8], k-major: dst[kp][T][fh][lane][8]; lane (rho, gk) holds k = 32 kp + 8 gk + e
__global__ void pack_layer16p_kernel(const float* __restrict__ src, int n_rows, int K, int mode, int kmajor, int H, int I,
                                     u16* __restrict__ dst, int f16) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_rows * K) return;
  const int KP = K / 32, NTL = n_rows / 32;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int l = (int)(t & 63); t >>= 6;
  const int fh = (int)(t & 1); t >>= 1;
  int T, kp;
  if (kmajor) {
    T = (int)(t % NTL);
    kp = (int)(t / NTL);
  } else {
    kp = (int)(t % KP);
    T = (int)(t / KP);
  }
  const int row = l16p_source_row(mode, T, 16 * fh + (l & 15), H, I);
  const float v = src[(size_t)row * K + 32 * kp + 8 * (l >> 4) + e];
  dst[idx] = f16 ? f2h(v) : f2bf(v);
}
#endif

template <bool H16>
__device__ __forceinline__ uint32_t pack2x(float a, float b) {
  if constexpr (H16) return pack_f16x2(a, b);
  else return pack_bf16x2(a, b);
}

__device__ __forceinline__ void lds_write_frag(uint32_t lds_addr, const bf16x8& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_write_u2(uint32_t lds_addr, uint32_t lo, uint32_t hi) {  // (an ext-vector operand: a 64-bit register pair)
  const u32x2 v = {lo, hi};
  asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write_f2(uint32_t lds_addr, const f32x2& v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x2 lds_read_f2(uint32_t lds_addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
// s_barrier with the compiler fenced on both sides: the bare builtin is "no memory" to the optimizer, which then moves LDS
// reads, plain loads and the LDS-DMA intrinsic across it (seen: nondeterministic q / k / v^T of one row half until a
// sequence point was added in front of the second LayerNorm -- profiles/r06_pair_kernel_steps.txt)
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait3(bf16x8& a, bf16x8& b, bf16x8& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
__device__ __forceinline__ void store_stream8(void* dst, const uint2& v) {
  typedef unsigned int u32x2_nt __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(u32x2_nt{v.x, v.y}, reinterpret_cast<u32x2_nt*>(dst));
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

// NT = hidden / 32 (8).  QKV: the next layer's q / k / v^T follow (false: the last layer).  H16: fp16 operands (kernel set "f16").
template <int NT, bool QKV, bool H16, bool XIN_T = false, bool XOUT_T = false>
__global__ __launch_bounds__(512, 2) void layer16p_kernel(Layer32Params p) {
  static_assert(NT == 8, "written for hidden = 256");
  constexpr int H = NT * 32;
  constexpr int KP = NT;                           // 32-wide k-pairs of a K = H contraction
  constexpr int CHUNK = KP * 2 * 512;              // elements of one 32-row weight tile (16 pieces: [kp][fh])
  constexpr int SLAB = NT * 2 * 512;               // elements of one k-pair of a k-major weight (16 pieces: [T][fh])
  constexpr int STAGE = 2 * CHUNK + SLAB;          // MLP stage: [Wi tile 2t | Wi tile 2t+1 | one k-pair of Wo]: 48 KiB
  constexpr int STAGE_B = STAGE * 2;               // bytes
  constexpr int EXTRA = 16 * 512;                  // 16 KiB: h exchange (2 buffers x 4 pairs x 2 row halves x 1 KiB)
  constexpr int XCH_B = STAGE_B;                   // LayerNorm exchange: stage 1 + EXTRA = 64 KiB = 8 waves x 8 fragments
  constexpr int HX_B = 2 * STAGE_B;                // h exchange = EXTRA
  constexpr int STAGE_PIECES = STAGE / 512;        // 48
  constexpr int O_PIECE0 = 80;                     // attention output of the block's rows: 64 pieces behind phase 1's second weight stage
  constexpr int SW_ELEMS = (O_PIECE0 + 64) * 512;  // 144 KiB
  static_assert(SW_ELEMS >= 2 * STAGE + EXTRA && O_PIECE0 * 512 >= STAGE + 32 * 512, "the o pieces lie behind phase 1's stages");
  __shared__ __attribute__((aligned(16))) u16 sW[SW_ELEMS];
  __shared__ __attribute__((aligned(16))) float sLn[2 * H];
  __shared__ __attribute__((aligned(16))) float sStat[8 * 2 * 16 * 2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pw = wave & 3, hf = wave >> 2;
  const int n16 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 128 + pw * 32;
#ifdef OPK_TIMING
  unsigned long long opk_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, opk_wait = 0, opk_x[4] = {0, 0, 0, 0};
  const unsigned long long opk_rt0 = wall_clock64();
#define L16P_STAMP(i) opk_ts[i] = __builtin_readcyclecounter()
#else
#define L16P_STAMP(i)
#endif
  L16P_STAMP(0);
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
  // Weight stages by buffer_load ... lds: one resource per weight tensor, the piece's byte offset as the SCALAR offset, lane * 16
  // as the (constant) vector offset -- per DMA one s_add and the M0 write, where the global_load_lds form pays a 64-bit scalar
  // add / addc on the pointer (36 of ~300 non-MFMA instructions per two MLP steps)
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto weight_rsrc = [](const u16* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(base), 0, 0x7fffffff, 0x00020000); };
  const __amdgpu_buffer_rsrc_t rs_wo = weight_rsrc(p.wo_p), rs_wi = weight_rsrc(p.wi_p), rs_wo2 = weight_rsrc(p.wo2_p), rs_qkv = weight_rsrc(p.wqkv_p);
  auto dma_weight = [&](__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, int dst_piece) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(&sW[dst_piece * 512]), 16, (int)lane16, (int)byte_off, 0, 0);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- phase 1: acc1[i] = tile 4 hf + i of o Wo^T (K = H, 2 k-pairs per LDS stage), o fragments straight from memory ----
  bf16x8 a[2][KP];  // Y operands of this wave's rows [row half][k-pair]: phase 1 = o (k order); afterwards LN(x), own-first order
  const size_t rb0 = (size_t)(m0 >> 4);
  // o pieces (row block rb0 + mf, k-pair kp) -> LDS piece O_PIECE0 + (2 pw + mf) * 8 + kp: each wave of the pair fetches the four
  // k-pairs of its half, both read all sixteen (the pair's rows are the same: fetched once instead of twice)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int mf = j >> 2, kp = 4 * hf + (j & 3);
    dma_piece(p.o_fp + (((rb0 + mf) * NT + kp) * 2) * 512, O_PIECE0 + (2 * pw + mf) * 8 + kp);
  }
  auto stage_p1 = [&](int j, int stage) {  // k-pairs 2 j, 2 j + 1: 32 pieces, 4 per wave
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int piece = wave + 8 * u;
      dma_weight(rs_wo, (uint32_t)(((2 * j) * SLAB + piece * 512) * 2), stage * STAGE_PIECES + piece);
    }
  };
  stage_p1(0, 0);
  stage_p1(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  // residual rows of the own tiles: lane (n16, g) owns features 32 T + 8 g + 4 fh + (0..3) of rows 16 mf + n16, T = 4 hf + i
  float* xrow = p.x_io + (size_t)(m0 + n16) * H + 128 * hf + 8 * g;  // + 16 mf rows
  float* xtile = p.x_io + ((size_t)(m0 >> 5) * NT + 4 * hf) * 1024 + lane * 4;  // + i * 1024 + (2 mf + fh) * 256
  float4 xa[4][2][2];  // [tile][mf][fh]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int fh = 0; fh < 2; ++fh)
        xa[i][mf][fh] = XIN_T ? load_stream_f4(xtile + i * 1024 + (2 * mf + fh) * 256) : load_stream_f4(xrow + (size_t)(16 * mf) * H + 32 * i + 4 * fh);
  __builtin_amdgcn_sched_barrier(0);
  sLn[ln_i] = ln_fill0;
  sLn[H + ln_i] = ln_fill1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // all but the residual rows: o and the first two weight stages
  block_barrier();
#ifdef OPK_TIMING
  opk_x[0] = __builtin_readcyclecounter();
#endif
  {
    const uint32_t o_addr = lds0 + (uint32_t)(O_PIECE0 * 1024) + (uint32_t)pw * 16384u;
    static_for<16>([&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      a[f >> 3][f & 7] = lds_read_frag<f * 1024>(o_addr);
    });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]), "+v"(a[0][7]),
                   "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[1][6]), "+v"(a[1][7]));
  }

  f32x4 acc1[4][2][2];  // [own tile][fh][mf]
  struct P1Off {  // step st = (k-pair st / 4 of the stage, own tile st % 4): the tile's two fragments fh = 0, 1
    static constexpr int at(int st, int j) { return ((st >> 2) * 16 + 2 * (st & 3) + j) * 1024; }
    static constexpr int base(int, int) { return 0; }
  };
  auto p1_stage = [&](auto j_tag) {
    constexpr int j = decltype(j_tag)::value;
    const uint32_t addr[1] = {lds0 + (uint32_t)((j & 1) * STAGE_B) + (uint32_t)hf * 8192u};
    frag_stream_m<8, 2, 2, P1Off, 1>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
      constexpr int st = decltype(step_tag)::value;
      constexpr int kp = 2 * j + (st >> 2), i = st & 3;
#pragma unroll
      for (int fh = 0; fh < 2; ++fh)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) acc1[i][fh][mf] = mfma16x<H16>(w[fh], a[mf][kp], kp == 0 ? zero4 : acc1[i][fh][mf]);
    });
  };
  p1_stage(std::integral_constant<int, 0>{});
  block_barrier();  // everyone has read stage 0
  stage_p1(2, 0);
  p1_stage(std::integral_constant<int, 1>{});
  block_barrier();
  stage_p1(3, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // k-pairs 4, 5 (and, in order, the residual rows) have landed
  block_barrier();
  p1_stage(std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  block_barrier();  // (stage 0 is free from here on)

  // ---- MLP stage DMA: stage t = [Wi tiles 2t, 2t+1 | Wo k-pair t-2]; pieces wave + 8 u, u = 0..5 ----
  const int n_it = p.n_pairs;  // I / 32 pair steps
  auto stage_piece = [&](auto u_tag, int t, int stage) {
    constexpr int u = decltype(u_tag)::value;
    const int tc = t < n_it ? t : n_it - 1;
    const int ts = t >= 2 ? (t - 2 < n_it ? t - 2 : n_it - 1) : 0;
    const int piece = wave + 8 * u;
    if constexpr (u < 4) dma_weight(rs_wi, (uint32_t)(((2 * tc) * CHUNK + piece * 512) * 2), stage * STAGE_PIECES + piece);
    else dma_weight(rs_wo2, (uint32_t)((ts * SLAB + (piece - 32) * 512) * 2), stage * STAGE_PIECES + piece);
  };
  static_for<6>([&](auto u) { stage_piece(u, 0, 0); });  // flies during the last phase-1 stage and the LayerNorm
  p1_stage(std::integral_constant<int, 3>{});
  block_barrier();  // stage 1 is free: the LayerNorm exchange may use it
  L16P_STAMP(1);

  // ---- LayerNorm of the rows in the accumulators (LOAD: acc1 += x first) -> a[mf][0..3] own fragments, a[mf][4..7] the partner's ----
  auto layer_ln = [&](auto load_tag, int which) {
    constexpr bool LOAD = decltype(load_tag)::value;
    // vector-only phase: packed fp32 arithmetic (two values per instruction at a scalar FMA's issue cost here; see pk_add).
    // Nothing but the accumulators themselves lives across the block barrier in the middle: each side reads them again
    // (64 registers fewer than keeping the centred values; the kernel sits at the 256-register edge here).
    // own half (128 features = this lane's 32 values x the 4 lanes g of a row): mean and M2 in two passes
    float mean_a[2], m2_a[2];
    if constexpr (!LOAD) {
      // The accumulators were last written by MFMAs and are first read HERE by packed instructions inside inline asm, which
      // the compiler's hazard recognizer cannot see: it pads "MFMA writes VGPR -> VALU reads it" (19 wait states at most) only
      // for instructions it knows.  Unpadded, the row half whose MFMAs issue last read stale registers whenever its wave was
      // the last to reach the barrier in front of this phase (nondeterministic q / k / v^T: profiles/r06_pair_kernel_steps.txt).
      asm volatile("s_nop 15\n\ts_nop 7"
                   : "+v"(acc1[0][0][0]), "+v"(acc1[0][0][1]), "+v"(acc1[0][1][0]), "+v"(acc1[0][1][1]), "+v"(acc1[1][0][0]), "+v"(acc1[1][0][1]),
                     "+v"(acc1[1][1][0]), "+v"(acc1[1][1][1]), "+v"(acc1[2][0][0]), "+v"(acc1[2][0][1]), "+v"(acc1[2][1][0]), "+v"(acc1[2][1][1]),
                     "+v"(acc1[3][0][0]), "+v"(acc1[3][0][1]), "+v"(acc1[3][1][0]), "+v"(acc1[3][1][1]));
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      f32x2 v[4][4];  // [tile][pair]: values e = 4 fh + r of the lane's 8 consecutive features of tile i
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
          f32x4 t = acc1[i][fh][mf];
          if constexpr (LOAD) {
            const float4 x4 = xa[i][mf][fh];
            t[0] += x4.x;
            t[1] += x4.y;
            t[2] += x4.z;
            t[3] += x4.w;
            acc1[i][fh][mf] = t;
          }
          v[i][2 * fh] = f32x2{t[0], t[1]};
          v[i][2 * fh + 1] = f32x2{t[2], t[3]};
        }
      f32x2 s4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] = i == 0 ? v[i][q] : pk_add(s4[q], v[i][q]);
      const f32x2 st2 = pk_add(pk_add(s4[0], s4[1]), pk_add(s4[2], s4[3]));
      float sum = st2.x + st2.y;
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      mean_a[mf] = sum * (1.0f / 128.0f);
      const f32x2 ma2 = f32x2{mean_a[mf], mean_a[mf]};
      f32x2 q4s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 c = pk_sub(v[i][q], ma2);
          q4s[q] = i == 0 ? pk_mul(c, c) : pk_fma(c, c, q4s[q]);
        }
      const f32x2 qt2 = pk_add(pk_add(q4s[0], q4s[1]), pk_add(q4s[2], q4s[3]));
      float m2 = qt2.x + qt2.y;
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      m2_a[mf] = m2;
      lds_write_f2(stat_addr + (uint32_t)(((wave * 2 + mf) * 16 + n16) * 8), f32x2{mean_a[mf], m2});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();
#ifdef OPK_TIMING
    if (which == 1) opk_x[1] = __builtin_readcyclecounter();
#endif
    // weights from LDS: columns 128 hf + 32 i + 8 g + (0..7)
    const float* lw = &sLn[which * H + 128 * hf + 8 * g];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const f32x2 other = lds_read_f2(stat_addr + (uint32_t)((((wave ^ 4) * 2 + mf) * 16 + n16) * 8));
      // the two halves combined (parallel variance: n_a = n_b = 128)
      const float mean = 0.5f * (mean_a[mf] + other.x);
      const float dm = mean_a[mf] - other.x;
      const float m2 = (m2_a[mf] + other.y) + dm * dm * 64.0f;
      const float rstd = 1.0f / sqrtf(m2 * (1.0f / (float)H) + p.eps);
      const f32x2 mn2 = f32x2{mean, mean}, r2 = f32x2{rstd, rstd};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(lw + 32 * i);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(lw + 32 * i + 4);
        const f32x4 t0 = acc1[i][0][mf], t1 = acc1[i][1][mf];
        const f32x2 y0 = pk_mul(pk_mul(pk_sub(f32x2{t0[0], t0[1]}, mn2), r2), f32x2{w0[0], w0[1]});
        const f32x2 y1 = pk_mul(pk_mul(pk_sub(f32x2{t0[2], t0[3]}, mn2), r2), f32x2{w0[2], w0[3]});
        const f32x2 y2 = pk_mul(pk_mul(pk_sub(f32x2{t1[0], t1[1]}, mn2), r2), f32x2{w1[0], w1[1]});
        const f32x2 y3 = pk_mul(pk_mul(pk_sub(f32x2{t1[2], t1[3]}, mn2), r2), f32x2{w1[2], w1[3]});
        a[mf][i] = as_frag(make_uint4(pack2x<H16>(y0.x, y0.y), pack2x<H16>(y1.x, y1.y), pack2x<H16>(y2.x, y2.y), pack2x<H16>(y3.x, y3.y)));
        lds_write_frag(lds0 + (uint32_t)XCH_B + (uint32_t)((wave * 8 + 4 * mf + i) * 1024), a[mf][i]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();
    const uint32_t oth = lds0 + (uint32_t)XCH_B + (uint32_t)(((wave ^ 4) * 8) * 1024);
    static_for<8>([&](auto f_tag) {
      constexpr int f = decltype(f_tag)::value;
      a[f >> 2][4 + (f & 3)] = lds_read_frag<f * 1024>(oth);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[0][6]), "+v"(a[0][7]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[1][6]), "+v"(a[1][7]));
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
          const f32x4 t = acc1[i][fh][mf];
          store_stream16(XOUT_T ? xtile + i * 1024 + (2 * mf + fh) * 256 : xrow + (size_t)(16 * mf) * H + 32 * i + 4 * fh, make_float4(t[0], t[1], t[2], t[3]));
        }
  };
  const std::true_type yes_{};
  const std::false_type no_{};

  layer_ln(yes_, 0);
  block_barrier();  // every wave has read its partner's fragments: the h exchange area (inside it) may be cleared
  // h exchange: buffer b, pair pw, row half mf = one Y fragment (1 KiB); this wave writes bytes 8 hf .. 8 hf + 7 of every lane's 16
  const uint32_t hx_pair = lds0 + (uint32_t)HX_B + (uint32_t)((pw * 2) * 1024);  // + buf * 8192 + mf * 1024
  const uint32_t hx_own = hx_pair + (uint32_t)hf * 8u;
  {
    lds_write_u2(hx_own, 0u, 0u);
    lds_write_u2(hx_own + 1024u, 0u, 0u);
    lds_write_u2(hx_own + 8192u, 0u, 0u);
    lds_write_u2(hx_own + 8192u + 1024u, 0u, 0u);
  }

  // ---- MLP: pair step t = [Wi tile 2t+hf -> na[t & 1]] + [GeGLU of na[(t-1) & 1] -> own half of h(t-1) -> LDS] + [acc1 += h(t-2) Wo^T] ----
  f32x4 na[2][2][2];  // [t & 1][fh: inputs | gates][mf]
#pragma unroll
  for (int q = 0; q < 8; ++q) na[q >> 2][(q >> 1) & 1][q & 1] = zero4;
  float gx[8], gq[8];  // GeGLU in flight: value 4 mf + r
#pragma unroll
  for (int i = 0; i < 8; ++i) gx[i] = gq[i] = 0.f;
  // this wave's fragment bases inside a stage (byte addresses, + stage * STAGE_B): Wi tile hf, own-first k order; Wo tiles 4 hf ..
  const uint32_t b_lo = lds0 + (uint32_t)hf * (16384u + 8192u);       // u <  4: k-pair 4 hf + u  at + (2 u + fh) * 1024
  const uint32_t b_hi = lds0 + (uint32_t)hf * (16384u - 8192u);       // u >= 4: k-pair u - 4 hf  at + (2 u + fh) * 1024
  const uint32_t b_wo = lds0 + 32768u + (uint32_t)hf * 8192u;         // Wo fragment (i, fh)      at + (2 i + fh) * 1024
  struct MlpOff {  // step s: Wi k-position u = s, fh = 0 | Wo fragment s = (tile i = s / 2, fh = s % 2) | Wi u = s, fh = 1
    static constexpr int at(int s, int j) { return j == 1 ? s * 1024 : (2 * s + (j >> 1)) * 1024; }
    static constexpr int base(int s, int j) { return j == 1 ? 2 : (s < 4 ? 0 : 1); }
  };
  struct WoOff {  // tail: step s = the two fragments of own tile s
    static constexpr int at(int s, int j) { return (2 * s + j) * 1024; }
    static constexpr int base(int, int) { return 0; }
  };
  // stage `stg` (0..7) of the GeGLU of the tile in `src` ([0] inputs, [1] gates); stage 7 packs each row half's four values
  // (= bytes 8 hf .. of the lane's k of k-pair t-1 of the Wo contraction) and hands them to the partner
  auto geglu_stage = [&](auto stg_tag, const f32x4 (&src)[2][2], uint32_t dst_addr) {
    constexpr int stg = decltype(stg_tag)::value;
    static_for<8>([&](auto i_tag) {
      constexpr int i = decltype(i_tag)::value, mf = i >> 2, r = i & 3;
      if constexpr (stg == 0) {
        gx[i] = src[0][mf][r];
        gq[i] = gelu_erf_poly(0.f, fabsf(gx[i]), 0);
      } else if constexpr (stg < 5) gq[i] = gelu_erf_poly(gq[i], fabsf(gx[i]), stg);
      else if constexpr (stg == 5) gq[i] = __builtin_amdgcn_exp2f(gq[i]);
      else if constexpr (stg == 6) gq[i] = gelu_erf_finish(gq[i], gx[i]);
      else gq[i] = gq[i] * src[1][mf][r];
    });
    if constexpr (stg == 7) {
      lds_write_u2(dst_addr, pack2x<H16>(gq[0], gq[1]), pack2x<H16>(gq[2], gq[3]));
      lds_write_u2(dst_addr + 1024u, pack2x<H16>(gq[4], gq[5]), pack2x<H16>(gq[6], gq[7]));
    }
  };
  auto interleave6 = [&]() {  // six MFMAs of a step, the step's vector slice spread between them
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
  };
  auto end_of_stage = [&]() {
#ifdef OPK_TIMING
    const unsigned long long w0_ = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    block_barrier();
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
      constexpr int s = decltype(step_tag)::value;
      if constexpr (s == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));  // (read before the stream's first group: landed with it)
      if constexpr (s < 6) stage_piece(step_tag, t + 1, P ^ 1);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) na[P][0][mf] = mfma16x<H16>(w[0], a[mf][s], s == 0 ? zero4 : na[P][0][mf]);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) acc1[s >> 1][s & 1][mf] = mfma16x<H16>(w[1], hb[mf], acc1[s >> 1][s & 1][mf]);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) na[P][1][mf] = mfma16x<H16>(w[2], a[mf][s], s == 0 ? zero4 : na[P][1][mf]);
      geglu_stage(step_tag, na[P ^ 1], h_dst);
      interleave6();
    });
    end_of_stage();
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  block_barrier();  // stage 0 has landed, the h buffers are clear
  L16P_STAMP(2);
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
        constexpr int s = decltype(step_tag)::value;
        if constexpr (s == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
        if constexpr (s < 2) stage_piece(std::integral_constant<int, 4 + s>{}, n_it + 1, 1);  // stage n_it + 1 needs only its Wo k-pair
#pragma unroll
        for (int fh = 0; fh < 2; ++fh)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf) acc1[s][fh][mf] = mfma16x<H16>(w[fh], hb[mf], acc1[s][fh][mf]);
        geglu_stage(std::integral_constant<int, 2 * s>{}, na[1], hx_own + 8192u);
        geglu_stage(std::integral_constant<int, 2 * s + 1>{}, na[1], hx_own + 8192u);
      });
    }
    end_of_stage();
    hb[0] = lds_read_frag<0>(hx_pair + 8192u);
    hb[1] = lds_read_frag<1024>(hx_pair + 8192u);
    {
      const uint32_t addr[1] = {b_wo + (uint32_t)STAGE_B};
      frag_stream_m<4, 2, 1, WoOff, 1>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
        constexpr int s = decltype(step_tag)::value;
        if constexpr (s == 0) asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
#pragma unroll
        for (int fh = 0; fh < 2; ++fh)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf) acc1[s][fh][mf] = mfma16x<H16>(w[fh], hb[mf], acc1[s][fh][mf]);
      });
    }
  }
  block_barrier();  // every wave is done with the ring
  L16P_STAMP(3);

  if constexpr (!QKV) {
    store_rows();
    L16P_STAMP(4);
  } else {
    // ---- next layer's q / k / v^T: pair step `it` = tiles 2 it (wave half 0) and 2 it + 1 (half 1), one LDS stage --------------
    constexpr int N_IT = 3 * NT / 2;  // 12 pair steps: 4 q, 4 k (one head each), 4 v
    constexpr int N_SW = 2 * NT / 2;  // q / k steps ("swapped": weights as the X operand)
    auto stage_pair = [&](int it, int stage) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int piece = wave + 8 * u;
        dma_weight(rs_qkv, (uint32_t)(((2 * it) * CHUNK + piece * 512) * 2), stage * STAGE_PIECES + piece);
      }
    };
    stage_pair(0, 0);
    // RoPE rows of this lane's two tokens: cos / sin [pos][16 hf + 4 g + (0..3)] (the own tile = half hf of a head's rotary pairs)
    f32x4 rc[2], rs[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      int pos = p.row_pos[m0 + 16 * mf + n16];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      rc[mf] = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * ROPE_HALF + 16 * hf + 4 * g);
      rs[mf] = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * ROPE_HALF + 16 * hf + 4 * g);
    }
    layer_ln(no_, 1);
#ifdef OPK_TIMING
    opk_x[2] = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pair 0 and the RoPE rows have landed
#ifdef OPK_TIMING
    opk_x[3] = __builtin_readcyclecounter();
#endif
    // (the write-back of x rides on the first eight pair steps, two KiB-stores each: issued here in one burst -- every CU of
    // the launch at once -- the sixteen stores of a wave take 7 k cycles to issue; spread, the loop is 9 k longer and the forward
    // 2 % faster: the launch's write burst (q, k, v^T and x of every CU at once) is what both forms wait for)
    block_barrier();  // pair 0 everywhere; everyone has read its partner's fragments (stage 1 is free)
    L16P_STAMP(4);

    struct QkvOff {  // step s: k-position u = s of the own tile, fragments fh = 0, 1
      static constexpr int at(int s, int j) { return (2 * s + j) * 1024; }
      static constexpr int base(int s, int) { return s < 4 ? 0 : 1; }
    };
    f32x4 qa[2][2][2];  // [pair step parity][fh][mf]
    uint2 st_v[4];
    uint4 st_w[2];
    u16* st_p[2] = {nullptr, nullptr};
    float e_lo[4], e_hi[4];
    // Epilogue of the tile of pair step `it`, in 8 slices that ride on the NEXT step's MFMAs
    auto epilogue_slice = [&](int it, auto sw_tag, auto s_tag, const f32x4 (&c)[2][2]) {
      constexpr bool SW = decltype(sw_tag)::value;
      constexpr int s = decltype(s_tag)::value, mf = s >> 2, r = s & 3;
      if constexpr (SW) {
        const bool is_q = it < N_SW / 2;
        const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
        // slot (fh = 0, r): d = 16 hf + 4 g + r, (fh = 1, r): its RoPE partner d + 32
        const float cc = rc[mf][r], ss = rs[mf][r];
        e_lo[r] = rope_lo(c[0][mf][r], c[1][mf][r], cc, ss) * qscale;
        e_hi[r] = rope_hi(c[0][mf][r], c[1][mf][r], cc, ss) * qscale;
        if constexpr (r == 3) {
          st_v[2 * mf] = make_uint2(pack2x<H16>(e_lo[0], e_lo[1]), pack2x<H16>(e_lo[2], e_lo[3]));
          st_v[2 * mf + 1] = make_uint2(pack2x<H16>(e_hi[0], e_hi[1]), pack2x<H16>(e_hi[2], e_hi[3]));
          const int head = is_q ? it : it - N_SW / 2;
          u16* out = is_q ? p.q_fp : p.k_fp;
          // piece (row block, k-step 2 head [d < 32] / 2 head + 1 [partners]); granule 2 hf + g / 2, bytes 8 (g % 2) .. of the lane's 16
          st_p[mf] = out + (((rb0 + mf) * NT + 2 * head) * 2) * 512 + (16 * (2 * hf + (g >> 1)) + n16) * 8 + 4 * (g & 1);
        }
      } else {
        // v^T pieces [head][row/32][plane][n4][lane = 16 kg + d'][8]: element e of a lane is key 4 kg + e (e < 4) / 16 + 4 kg + e - 4
        // of the 32-token block (the order of the attention kernel's P^T fragments).  This lane (feature column n16 of fragment fh,
        // g) holds tokens 16 mf + 4 g + r: all eight keys of granule kg = g, d' = n16 -- one 16-byte store per fragment, lane-linear
        if constexpr (s == 7) {
#pragma unroll
          for (int fh = 0; fh < 2; ++fh)
            st_w[fh] = make_uint4(pack2x<H16>(c[fh][0][0], c[fh][0][1]), pack2x<H16>(c[fh][0][2], c[fh][0][3]),
                                  pack2x<H16>(c[fh][1][0], c[fh][1][1]), pack2x<H16>(c[fh][1][2], c[fh][1][3]));
          const size_t tb = (size_t)(m0 >> 5);
          st_p[0] = p.vt_fp + ((((size_t)(it - N_SW) * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + 2 * hf) * 512 + lane * 8;  // piece n4 = 2 hf + fh
        }
      }
    };
    auto epilogue_store = [&](auto sw_tag) {
      constexpr bool SW = decltype(sw_tag)::value;
      if constexpr (SW) {
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          store_stream8(st_p[mf], st_v[2 * mf]);
          store_stream8(st_p[mf] + 1024, st_v[2 * mf + 1]);
        }
      } else {
        store_stream16(st_p[0], st_w[0]);
        store_stream16(st_p[0] + 512, st_w[1]);
      }
    };
    auto interleave4 = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    };
    auto iteration = [&](auto it_tag, auto first_tag, auto sw_tag, auto swp_tag) {
      constexpr int it = decltype(it_tag)::value, cur = it & 1;
      constexpr bool FIRST = decltype(first_tag)::value, SW = decltype(sw_tag)::value, SWP = decltype(swp_tag)::value;
      constexpr bool XS = it < 8;  // this step carries the x pieces of own tile it / 2, row half it % 2
      stage_pair(it + 1 < N_IT ? it + 1 : it, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t addr[2] = {b_lo + (uint32_t)(cur * STAGE_B), b_hi + (uint32_t)(cur * STAGE_B)};
      frag_stream_m<8, 2, 2, QkvOff, 2>(addr, [&](auto step_tag, bf16x8(&w)[2]) {
        constexpr int s = decltype(step_tag)::value;
#pragma unroll
        for (int fh = 0; fh < 2; ++fh)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            qa[cur][fh][mf] = SW ? mfma16x<H16>(w[fh], a[mf][s], s == 0 ? zero4 : qa[cur][fh][mf]) : mfma16x<H16>(a[mf][s], w[fh], s == 0 ? zero4 : qa[cur][fh][mf]);
        if constexpr (!FIRST) {
          epilogue_slice(it - 1, swp_tag, step_tag, qa[cur ^ 1]);
          interleave4();
        }
      });
      if constexpr (!FIRST) epilogue_store(swp_tag);
      if constexpr (XS) {
#pragma unroll
        for (int fh = 0; fh < 2; ++fh) {
          const f32x4 t = acc1[it >> 1][fh][it & 1];
          store_stream16(XOUT_T ? xtile + (it >> 1) * 1024 + (2 * (it & 1) + fh) * 256 : xrow + (size_t)(16 * (it & 1)) * H + 32 * (it >> 1) + 4 * fh,
                       make_float4(t[0], t[1], t[2], t[3]));
        }
      }
      constexpr int N_STORES = (FIRST ? 0 : (SWP ? 4 : 2)) + (XS ? 2 : 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
      block_barrier();
    };
    iteration(std::integral_constant<int, 0>{}, yes_, yes_, yes_);
    static_for<N_SW - 1>([&](auto j_tag) { iteration(std::integral_constant<int, 1 + decltype(j_tag)::value>{}, no_, yes_, yes_); });
    iteration(std::integral_constant<int, N_SW>{}, no_, no_, yes_);
    static_for<N_IT - N_SW - 1>([&](auto j_tag) { iteration(std::integral_constant<int, N_SW + 1 + decltype(j_tag)::value>{}, no_, no_, no_); });
    static_for<8>([&](auto s_tag) { epilogue_slice(N_IT - 1, no_, s_tag, qa[1]); });
    epilogue_store(no_);
  }
  L16P_STAMP(5);
#ifdef OPK_TIMING
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 16 + i] = opk_ts[i];
    p.dbg[(size_t)blockIdx.x * 16 + 8] = opk_wait;
    for (int i = 0; i < 4; ++i) p.dbg[(size_t)blockIdx.x * 16 + 11 + i] = opk_x[i];
    p.dbg[(size_t)blockIdx.x * 16 + 15] = wall_clock64() - opk_rt0;
  }
#endif
#undef L16P_STAMP
}

}  // namespace opk
