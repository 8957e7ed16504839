// opk_panel.hip.h -- panel GEMMs (hidden % 256 == 0)
#pragma once

#include <type_traits>

#include "opk_common.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// Panel GEMM (hidden > 256: base / large / en-gte): the k-streamed kernel above, tiled over N.  A block
// computes a [128 rows x 256 features] panel: per k-step it DMAs one [256 x 32] weight slab (hi + lo, 32 KiB) into
// LDS while every wave pulls its own A fragments straight from the fragment-packed activation; all 256 outputs of
// the wave's 32 rows stay in accumulators (128 VGPRs) and the epilogue writes the NEXT consumer's layout directly:
//   PE_RESIDUAL  x += acc                                   (attention / MLP output projections)
//   PE_QK        RoPE, q scale -> fragment-packed q or k    (a panel = 4 heads)
//   PE_V         -> v^T pieces (tokens as MFMA rows, so a lane ends up with 8 keys of one head dim)
//   PE_GEGLU     gelu(act) * gate -> fragment-packed h      (a panel = 128 act + the matching 128 gate features)
// The weight rows of a panel are permuted at load time (panel_source_row) so that accumulator fragments
// (2s, 2s+1) are the 8 consecutive k of lane slot g of k-step s of the consumer, with RoPE partners and GeGLU gates
// in the same lane.
// ----------------------------------------------------------------------------------------------

// source row of accumulator slot (fragment nf, row i) of panel `tile`
__device__ __forceinline__ int panel_source_row(int mode, int tile, int nf, int i, int H, int I) {
  const int within = 8 * (i >> 2) + (i & 3);  // + 4 * (fragment parity): position inside a 32-wide k-step
  if (mode == PE_RESIDUAL) return tile * 256 + 32 * (nf >> 1) + within + 4 * (nf & 1);
  if (mode == PE_QK) {  // q panels first, then k panels; 4 heads per panel; fragments (0,1): d < 32, (2,3): d >= 32
    const int per = H / 256, region = tile / per, tq = tile % per;
    const int head = tq * 4 + (nf >> 2), sub = nf & 3;
    return region * H + head * HEAD_DIM + 32 * (sub >> 1) + within + 4 * (sub & 1);
  }
  if (mode == PE_V) {  // piece n = nf & 3 of head (nf >> 2): row i is d = 32(n>>1) + 8(i>>2) + 4(n&1) + (i&3)
    const int head = tile * 4 + (nf >> 2), n = nf & 3;
    return 2 * H + head * HEAD_DIM + 32 * (n >> 1) + within + 4 * (n & 1);
  }
  // PE_GEGLU: fragments 0..7 = input columns 128 tile .. +127, fragments 8..15 = the matching gate columns
  const int nn = nf & 7;
  return (nf < 8 ? 0 : I) + tile * 128 + 32 * (nn >> 1) + within + 4 * (nn & 1);
}

#ifdef OPK_PACK_KERNELS  // weight re-packing runs in op_api.hip only
// dst[tile][ks][plane][nf 0..15][lane = 16 g + i][8] <- W[source_row(tile, nf, i)][ks*32 + g*8 + e]
// f16 = 1 (kernel set "f16"): hi plane = RNE_fp16(w), lo plane zeros, any_lo untouched
__global__ void pack_panel_kernel(const float* __restrict__ src, int n_tiles, int K, int mode, int H, int I,
                                  u16* __restrict__ dst, int zero_lo, int* __restrict__ any_lo, int f16 = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_tiles * 256 * K;
  if (idx >= total) return;
  const int KS = K / 32;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t & 15); t >>= 4;
  const int ks = (int)(t % KS);
  const int tile = (int)(t / KS);
  const int srow = panel_source_row(mode, tile, nf, i, H, I);
  const float v = src[(size_t)srow * K + ks * 32 + g * 8 + e];
  const size_t base = (((size_t)tile * KS + ks) * 2) * 8192 + (size_t)nf * 512 + (size_t)g * 128 + i * 8 + e;
  if (f16) {
    dst[base] = f2h(v);
    dst[base + 8192] = (u16)0;
    return;
  }
  const u16 h = f2bf(v);
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  dst[base] = h;
  dst[base + 8192] = zero_lo ? (u16)0 : l;
}
#endif

#ifdef OPK_PACK_KERNELS
// "f16 + fp8" kernel sets on the panel path (opk_common.hip.h for the format).  Per panel:
//   dst16[tile][ks][nf 0..15][512]            fp16(w)
//   dst8 [tile][S = ks / 4][nf][half][1 KiB]  e4m3(w), and lo_off_bytes further the same of lo(w) x 2^12
// round_bf16: the requested policy has no hi x lo(weight) term -- the weight IS its bf16 rounding.
__global__ void pack_panel_f8_kernel(const float* __restrict__ src, int n_tiles, int K, int mode, int H, int I,
                                     u16* __restrict__ dst16, u16* __restrict__ dst8, size_t lo_off_bytes, int round_bf16,
                                     int* __restrict__ not_f16, float* __restrict__ fit) {
  set_saturating_conversions();
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_tiles * 256 * K;
  if (idx >= total) return;
  const int KS = K / 32, NS8 = KS / 4;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t & 15); t >>= 4;
  const int ks = (int)(t % KS);
  const int tile = (int)(t / KS);
  const int srow = panel_source_row(mode, tile, nf, i, H, I);
  float v = src[(size_t)srow * K + ks * 32 + g * 8 + e];
  if (round_bf16) v = bf2f(f2bf(v));
  const _Float16 hv = (_Float16)v;
  note_f16_fit(v, hv, not_f16, fit);
  dst16[(((size_t)tile * KS + ks) * 16 + nf) * 512 + g * 128 + i * 8 + e] = __builtin_bit_cast(u16, hv);
  const int s8 = ks >> 2, hh = (ks & 3) >> 1, pbyte = 8 * (ks & 1) + e;
  unsigned char* d8 = reinterpret_cast<unsigned char*>(dst8 + ((((size_t)tile * NS8 + s8) * 16 + nf) * 2 + hh) * 512);
  d8[(g * 16 + i) * 16 + pbyte] = f2e4m3(v * (float)(1 << F8_W_SHIFT));
  d8[lo_off_bytes + (g * 16 + i) * 16 + pbyte] = f2e4m3((v - (float)hv) * (float)(1 << (F8_LO_SHIFT + F8_W_SHIFT)));
}
#endif

struct PanelParams {
  const u16* a_fp;   // fragment-packed activations [r_pad/16][n_ksteps][2 planes][512]
  const u16* wp;     // packed weights [n_tiles][n_ksteps][2 planes][16][512]
  int n_ksteps;      // K / 32
  int n_tiles;       // output panels (N / 256; PE_GEGLU: I / 128)
  int row_group;     // row blocks per XCD-local group (see the block map in panel_gemm_kernel)
  int r_pad;
  int hidden;        // H
  int ld_out;        // PE_RESIDUAL: H   PE_GEGLU: I
  float* x;          // PE_RESIDUAL
  u16* o0;           // PE_QK: q   PE_V: v^T   PE_GEGLU: h   (fragment-packed, hi/lo planes interleaved per piece)
  u16* o1;           // PE_QK: k
  u16* o2;           // fused q / k / v launch: v^T (panels n_qk_tiles .. n_tiles - 1)
  int n_qk_tiles;    // fused q / k / v launch: panels of q and k (= 2 H / 256)
  const int32_t* row_pos;
  const float* rope_cos;
  const float* rope_sin;
  int max_pos;
  // "f16 + fp8" kernel sets (panel_f8_block): a_fp = fp16 pieces [r_pad/16][n_ksteps][512], a_lo8 = e4m3 pieces
  // [r_pad/16][n_ksteps/2][512] (one 1 KiB half-fragment per two k-steps); wp = fp16 slabs [tile][ks][16][512], wp8 =
  // e4m3 slabs [tile][ks/4][16][2][512] and, w8_lo_off u16 elements further, the same of the weights' lo part;
  // o0 (PE_GEGLU) = fp16 pieces of h, o0_lo8 = its e4m3 pieces
  const u16* a_lo8;
  const u16* wp8;
  size_t w8_lo_off;
  u16* o0_lo8;
  // fp16 + e4m3 kernels: raised (atomicOr 1) when a value written as an fp16 operand is beyond fp16's range or not finite --
  // those kernels convert under MODE.FP16_OVFL = 1 (clamp) and their MFMAs swallow a NaN operand; rank_head_kernel turns the
  // flag into NaN outputs.  May be NULL.
  int* range_flag;
};

// T = term mask (left = the fragment-packed activation, right = the weight panel); OLO bit 0: o0 (q / v^T / h) gets a
// lo plane, bit 1: o1 (k) does.
//
// Block -> (row block, output panel), XCD-aware: the grid is one-dimensional, ceil(row blocks / 8) * 8 * n_tiles
// blocks.  Block b is dispatched to XCD b % 8 (observed placement; a speed assumption only), so with
//   row block = (b / 8 / n_tiles) * 8 + b % 8,   panel = (b / 8) % n_tiles
// the n_tiles blocks that read the same 128 activation rows follow each other on ONE XCD and share that XCD's L2:
// the activation planes leave HBM once instead of once per panel (measured on the base model, H = 768, I = 1152:
// the Wi GEMM fetched 5.4 GB per launch for 1.0 GB of operands with the row-block-major grid).
constexpr int panel_stage_elems(int T) { return 16 * (((T & T_RIGHT_LO) != 0) ? 2 : 1) * 512; }

// One block's panel: rows [row_block * 128, +128) x panel `wtile_index` of the packed weights; `tile` is the panel's index
// inside its own output tensor (q / k panels count from 0, so do the v^T panels of the fused launch).
// H16 (kernel set "f16"): the single-pass instantiation with fp16 operands (opk_common.hip.h)
template <int EPI, int T, int OLO, bool H16 = false>
__device__ __forceinline__ void panel_block(const PanelParams& p, int row_block, int wtile_index, int tile, u16* __restrict__ o0,
                                            u16 (&sW)[2][panel_stage_elems(T)]) {
  constexpr int NF = 16;
  constexpr bool W_LO = (T & T_RIGHT_LO) != 0, A_LO = (T & T_LEFT_LO) != 0;
  constexpr bool O0_LO = (OLO & 1) != 0, O1_LO = (OLO & 2) != 0;
  constexpr int PLANES = W_LO ? 2 : 1;
  constexpr int STAGE = NF * PLANES * 512;   // elements per LDS stage
  constexpr int SLAB_SRC = NF * 2 * 512;     // elements per k-step in the packed weights (both planes)
  constexpr int WAVE_PIECES = (NF * PLANES) / 4;
  constexpr bool SWAPPED = (EPI != PE_V);    // weights as the MFMA row operand, except for v^T
  static_assert(STAGE == panel_stage_elems(T), "stage size");
  static_assert(!H16 || (T == 0 && OLO == 0), "fp16 operands: single pass only");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = row_block * ROW_BM + wave * 32;
  const int nks = p.n_ksteps;
  const u16* wtile = p.wp + (size_t)wtile_index * nks * SLAB_SRC;
  uint32_t lds_stage[2];  // LDS byte address of this lane's 16 bytes in fragment 0 of each stage
  lds_stage[0] = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0][0]) + (uint32_t)lane * 16u;
  lds_stage[1] = lds_stage[0] + (uint32_t)(STAGE * 2);

  auto stage_slab = [&](int ks, int stage) {
    const u16* src = wtile + (size_t)ks * SLAB_SRC;
    // stage = [plane][nf] pieces, same order as the source (which has 2 planes).  A wave copies groups of four consecutive
    // pieces: one pointer / M0 per group, the other three through the DMA's immediate offset, which moves the global and the
    // LDS address alike (microbench/dma_offset_probe.hip) -- three scalar instructions less per piece
    static_assert(WAVE_PIECES % 4 == 0, "groups of four pieces");
    static_for<WAVE_PIECES>([&](auto u_tag) {
      constexpr int u = decltype(u_tag)::value;
      const int piece0 = 4 * (wave + 4 * (u / 4));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece0 * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece0 * 512]), 16, (u % 4) * 1024, 0);
    });
  };
  const u16* a_base = p.a_fp + ((size_t)(m0 >> 4) * nks * 2) * 512 + lane * 8;
  const size_t a_block = (size_t)nks * 2 * 512;
  bf16x8 an_hi[2], an_lo[2];
  auto load_a = [&](int ks) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      an_hi[mf] = *reinterpret_cast<const bf16x8*>(a_base + mf * a_block + (size_t)ks * 1024);
      an_lo[mf] = A_LO ? *reinterpret_cast<const bf16x8*>(a_base + mf * a_block + (size_t)ks * 1024 + 512) : an_hi[mf];
    }
  };

  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_slab(0, 0);
  load_a(0);
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {  // retire the first fragment loads in front of the loop (see kstream_gemm_kernel)
    asm volatile("" : "+v"(an_hi[mf]));
    asm volatile("" : "+v"(an_lo[mf]));
  }
  __syncthreads();

  auto step = [&](int ks, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    const int kn = ks + 1 < nks ? ks + 1 : ks;
    stage_slab(kn, cur ^ 1);
    bf16x8 a_hi[2], a_lo[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      a_hi[mf] = an_hi[mf];
      a_lo[mf] = an_lo[mf];
    }
    load_a(kn);
    __builtin_amdgcn_sched_barrier(0);
    // The slab's fragments as one hand-placed stream (frag_stream2 / 4 in opk_common.hip.h): step = fragments (nf, nf + 1)
    // [+ their lo planes], read two steps ahead with counted waits.  Left to the compiler, every pair of reads was followed
    // by s_waitcnt lgkmcnt(0) in front of its 8 MFMAs -- the LDS latency eight times per k-step.
    auto mfmas = [&](auto step_tag, const bf16x8& wh0, const bf16x8& wh1, const bf16x8& wl0, const bf16x8& wl1) {
      constexpr int nf = 2 * decltype(step_tag)::value;
#pragma unroll
      for (int term = 0; term < 3; ++term) {  // term-major over 2 fragments x 2 row blocks: no dependent MFMA pairs
        if ((term == 0 && !W_LO) || (term == 1 && !A_LO)) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf) {
            const bf16x8 w = term == 0 ? (j ? wl1 : wl0) : (j ? wh1 : wh0);
            const bf16x8 a = term == 1 ? a_lo[mf] : a_hi[mf];
            acc[nf + j][mf] = SWAPPED ? mfma16x<H16>(w, a, acc[nf + j][mf]) : mfma16x<H16>(a, w, acc[nf + j][mf]);
          }
      }
    };
    if constexpr (W_LO) {
      struct Off4 {
        static constexpr int at(int st, int j) { return (((j >> 1) ? NF : 0) + 2 * st + (j & 1)) * 1024; }
      };
      frag_stream4<NF / 2, 2, Off4>(lds_stage[cur], [&](auto step_tag, bf16x8& w0, bf16x8& w1, bf16x8& w2, bf16x8& w3) {
        mfmas(step_tag, w0, w1, w2, w3);
      });
    } else {
      struct Off2 {
        static constexpr int at(int st, int j) { return (2 * st + j) * 1024; }
      };
      frag_stream2<NF / 2, 2, Off2>(lds_stage[cur], [&](auto step_tag, bf16x8& w0, bf16x8& w1) { mfmas(step_tag, w0, w1, w0, w1); });
    }
    __syncthreads();
  };
  for (int k0 = 0; k0 < nks; k0 += 2) {  // even number of k-steps (K % 64 == 0, checked on the host)
    step(k0, std::integral_constant<int, 0>{});
    step(k0 + 1, std::integral_constant<int, 1>{});
  }

  if (EPI == PE_RESIDUAL) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      float* xrow = p.x + (size_t)(m0 + mf * 16 + l15) * p.ld_out + tile * 256 + g * 8;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
        float4 r4 = *px;  // plain on purpose: streaming (nt) access to x measured -15 % on this kernel -- the LayerNorm launch
                          // that follows finds part of x in L2
        r4.x += acc[nf][mf][0];
        r4.y += acc[nf][mf][1];
        r4.z += acc[nf][mf][2];
        r4.w += acc[nf][mf][3];
        *px = r4;
      }
    }
  } else if (EPI == PE_GEGLU) {
    const int kb_out = p.ld_out >> 5;  // k-steps per row block of h
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const size_t rb = (size_t)((m0 >> 4) + mf);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = gelu_erf(acc[2 * s][mf][r]) * acc[8 + 2 * s][mf][r];
          v[4 + r] = gelu_erf(acc[2 * s + 1][mf][r]) * acc[8 + 2 * s + 1][mf][r];
        }
        bf16x8 hi, lo;
        pack8x<O0_LO, H16>(v, hi, lo);
        u16* dst = o0 + ((rb * kb_out + (size_t)(tile * 4 + s)) * 2) * 512 + lane * 8;
        store_stream16(dst, as_u4(hi));
        if (O0_LO) store_stream16(dst + 512, as_u4(lo));
      }
    }
  } else if (EPI == PE_QK) {
    const int per = p.hidden / 256;
    const bool is_q = tile < per;
    const int tq = is_q ? tile : tile - per;
    u16* out = is_q ? o0 : p.o1;
    // head_dim^-0.5 * log2(e): the fragment-packed attention kernel exponentiates with exp2
    const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
    const int kb_out = p.hidden >> 5;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      int pos = p.row_pos[m0 + mf * 16 + l15];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      // cos / sin of d_low = 8g + 4u + r, u = 0, 1
      f32x4 c4[2], s4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        c4[u] = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * ROPE_HALF + g * 8 + u * 4);
        s4[u] = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * ROPE_HALF + g * 8 + u * 4);
      }
      const size_t rb = (size_t)((m0 >> 4) + mf);
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        float lo_half[8], hi_half[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x1 = acc[4 * hh + u][mf][r], x2 = acc[4 * hh + 2 + u][mf][r];
            lo_half[4 * u + r] = rope_lo(x1, x2, c4[u][r], s4[u][r]) * qscale;
            hi_half[4 * u + r] = rope_hi(x1, x2, c4[u][r], s4[u][r]) * qscale;
          }
        bf16x8 h0, l0, h1, l1;
        pack8x<(O0_LO || O1_LO), H16>(lo_half, h0, l0);
        pack8x<(O0_LO || O1_LO), H16>(hi_half, h1, l1);
        u16* dst = out + ((rb * kb_out + (size_t)((tq * 4 + hh) * 2)) * 2) * 512 + lane * 8;
        store_stream16(dst, as_u4(h0));
        store_stream16(dst + 1024, as_u4(h1));
        if ((O0_LO || O1_LO) && (O0_LO == O1_LO || (is_q ? O0_LO : O1_LO))) {
          store_stream16(dst + 512, as_u4(l0));
          store_stream16(dst + 1536, as_u4(l1));
        }
      }
    }
  } else {  // PE_V: accumulator rows = tokens 4g + r of block mf, column = row l15 of piece n = nf & 3
    const size_t tb = (size_t)(m0 >> 5);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const size_t head = (size_t)(tile * 4 + (nf >> 2));
      const float v[8] = {acc[nf][0][0], acc[nf][0][1], acc[nf][0][2], acc[nf][0][3],
                          acc[nf][1][0], acc[nf][1][1], acc[nf][1][2], acc[nf][1][3]};
      bf16x8 hi, lo;
      pack8x<O0_LO, H16>(v, hi, lo);
      u16* dst = o0 + (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + (size_t)(nf & 3)) * 512 + lane * 8;
      store_stream16(dst, as_u4(hi));
      if (O0_LO) store_stream16(dst + 2048, as_u4(lo));
    }
  }
}

// block -> (row block, panel) of the XCD-aware map described above; false when the block is grid padding
__device__ __forceinline__ bool panel_block_map(const PanelParams& p, int& row_block, int& tile) {
  const int xcd_slot = blockIdx.x >> 3;
  const int group_blocks = p.row_group * p.n_tiles;
  const int in_group = xcd_slot % group_blocks;
  row_block = ((xcd_slot / group_blocks) * p.row_group + in_group % p.row_group) * 8 + (blockIdx.x & 7);
  tile = in_group / p.row_group;
  return row_block * ROW_BM < p.r_pad;  // the grid is rounded up to whole groups
}

template <int EPI, int T, int OLO, bool H16 = false>
__global__ __launch_bounds__(256, 2) void panel_gemm_kernel(PanelParams p) {
  __shared__ __attribute__((aligned(16))) u16 sW[2][panel_stage_elems(T)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  panel_block<EPI, T, OLO, H16>(p, row_block, tile, tile, p.o0, sW);
}

// q, k and v^T of a layer in ONE launch: the three projections read the same normalised rows, so their panels sit side
// by side in the XCD-local group (the rows are fetched once for all 3 H / 256 panels) and a launch is saved per layer.
// The v^T panels use the un-swapped MFMA orientation: a block-uniform branch picks the instantiation.
template <int T, int OLO_QK, int OLO_V, bool H16 = false>
__global__ __launch_bounds__(256, 2) void panel_qkv_kernel(PanelParams p) {
  __shared__ __attribute__((aligned(16))) u16 sW[2][panel_stage_elems(T)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  if (tile < p.n_qk_tiles) panel_block<PE_QK, T, OLO_QK, H16>(p, row_block, tile, tile, p.o0, sW);
  else panel_block<PE_V, T, OLO_V, H16>(p, row_block, tile, tile - p.n_qk_tiles, p.o2, sW);
}

// ----------------------------------------------------------------------------------------------
// Panel GEMM in the "f16 + fp8" format (round 3; kernel sets 3 / 4 on the panel path).  Same block shape and XCD-aware
// map.  Iteration `it` multiplies the fp16 slab of k-step `it` (16 fragments x 2 row blocks on v_mfma_f32_16x16x32_f16)
// AND one quarter (4 of the 16 output fragments) of the e4m3 K = 128 product of the PREVIOUS group of four k-steps
// (lo(a) x e4m3(w); WLO: + e4m3(a) x lo(w)) -- so a stage holds one 16 KiB fp16 slab + an 8 KiB (WLO: 16 KiB) e4m3
// quarter-slab instead of the 64 + 32 KiB a whole K = 128 step would need, and the matrix pipe sees 1.5 (WLO: 2)
// units per product in EVERY panel GEMM, the MLP output projection included (its K = intermediate is streamed anyway).
// Four tail iterations finish the last group's e4m3 part.
// ----------------------------------------------------------------------------------------------
constexpr int panel_f8_stage_elems(bool wlo) { return (16 + (wlo ? 16 : 8)) * 512; }

// O16 (kernel sets 10 / 11, q / k / v^T only): the outputs are single-plane fp16 -- what attn_fp_kernel<.., H16> reads.
template <int EPI, bool WLO, int OLO, int NST, bool O16 = false>
__device__ __forceinline__ void panel_f8_block(const PanelParams& p, int row_block, int wtile_index, int tile, u16* __restrict__ o0,
                                               u16 (&sW)[NST][panel_f8_stage_elems(WLO)]) {
  static_assert(NST == 2 || NST == 3, "two or three slab stages");
  static_assert(!O16 || (OLO == 0 && (EPI == PE_QK || EPI == PE_V)), "fp16 outputs: single-plane q / k / v^T");
  constexpr int NF = 16;
  constexpr bool O0_LO = (OLO & 1) != 0, O1_LO = (OLO & 2) != 0;
  constexpr int STAGE = panel_f8_stage_elems(WLO);
  constexpr int PIECES = STAGE / 512;       // 24 or 32
  constexpr int WAVE_PIECES = PIECES / 4;
  constexpr bool SWAPPED = (EPI != PE_V);
  set_saturating_conversions();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = row_block * ROW_BM + wave * 32;
  const int nks = p.n_ksteps;  // a multiple of 4 (checked on the host)
  const int n_it = nks + 4;
  const u16* w16 = p.wp + (size_t)wtile_index * nks * (NF * 512);
  const u16* w8 = p.wp8 + (size_t)wtile_index * (nks >> 2) * (NF * 1024);
  const uint32_t lds_stage0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) u16*)&sW[0][0]) + (uint32_t)lane * 16u;

  // stage of iteration `it`: [fp16 slab of k-step it (clamped)][e4m3 quarter (it % 4) of group it / 4 - 1 (clamped)]
  // [WLO: the same quarter of lo(w)] -- clamped copies are harmless and keep the DMA count branch-free
  auto stage_it = [&](int it, int stage) {
    const int ks = it < nks ? it : nks - 1;
    const int grp = it >= 4 ? (it >> 2) - 1 : 0, q = it & 3;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + 4 * u;  // wave-uniform
      const u16* src;
      if (u < 4) src = w16 + ((size_t)ks * NF + piece) * 512;
      else if (u < 6) src = w8 + (((size_t)grp * NF + 4 * q) * 2 + (piece - 16)) * 512;
      else src = w8 + p.w8_lo_off + (((size_t)grp * NF + 4 * q) * 2 + (piece - 24)) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  const u16* a_base = p.a_fp + (size_t)(m0 >> 4) * nks * 512 + lane * 8;
  const u16* l_base = p.a_lo8 + (size_t)(m0 >> 4) * (nks >> 1) * 512 + lane * 8;
  bf16x8 an_hi[2];
  auto load_a = [&](int ks) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) an_hi[mf] = *reinterpret_cast<const bf16x8*>(a_base + ((size_t)mf * nks + ks) * 512);
  };
  // e4m3 operands: lo plane of the group being multiplied / requested for the next one; WLO: e4m3(a) of the group being
  // multiplied / being collected from the fp16 fragments as they pass
  i32x8 lo_prev[2], lo_next[2], h8_prev[2], h8_cur[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int r = 0; r < 8; ++r) lo_prev[mf][r] = lo_next[mf][r] = h8_prev[mf][r] = h8_cur[mf][r] = 0;

  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_a(0);
  stage_it(0, 0);
  if constexpr (NST == 3) {  // two slabs in flight: the request of iteration it + 2 goes out while it is multiplied
    stage_it(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAVE_PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
  } else {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) asm volatile("" : "+v"(an_hi[mf]));
    __syncthreads();
  }

  struct OffF8 {  // steps 0..7: fp16 fragment pairs; 8..11: the two halves of e4m3 fragment (step - 8); 12..15: of lo(w)
    static constexpr int at(int st, int j) { return st < 8 ? (2 * st + j) * 1024 : (16 + 2 * (st - 8) + j) * 1024; }
  };
  // live: the fp16 part (it < nks); f8_live: the e4m3 part of the previous group (it >= 4) -- compile-time, the loop is
  // cut into head (4 iterations), body and tail (4 iterations)
  int ring = 0;  // NST == 3: it % 3
  auto step = [&](int it, auto cur_tag, auto q_tag, auto live_tag, auto f8_tag) {
    const int cur = NST == 2 ? decltype(cur_tag)::value : ring;
    constexpr int q = decltype(q_tag)::value;  // it % 4
    constexpr bool live = decltype(live_tag)::value, f8_live = decltype(f8_tag)::value;
    if constexpr (NST == 2) stage_it(it + 1 < n_it ? it + 1 : it, cur ^ 1);
    bf16x8 a_hi[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) a_hi[mf] = an_hi[mf];
    auto load_lo = [&](int grp, i32x8 (&dst)[2]) {
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
        dst[mf] = f8_frag(*reinterpret_cast<const bf16x8*>(l_base + ((size_t)mf * (nks >> 1) + 2 * grp) * 512),
                          *reinterpret_cast<const bf16x8*>(l_base + ((size_t)mf * (nks >> 1) + 2 * grp + 1) * 512));
    };
    if (q == 0) {  // a new group starts: what was collected / requested becomes what is multiplied
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        h8_prev[mf] = h8_cur[mf];
        if (!WLO) lo_prev[mf] = lo_next[mf];
      }
      // WLO (16 more registers of e4m3(a) in flight): no look-ahead copy, the lo plane is requested here, eight fp16
      // steps ahead of its first use
      if (WLO && f8_live) load_lo((it >> 2) - 1, lo_prev);
    }
    if (!WLO && q == 3 && live) load_lo(it >> 2, lo_next);
    if (false) {
    }
    if (live) load_a(it + 1 < nks ? it + 1 : nks - 1);
    if constexpr (NST == 3) {  // after the register loads: the wait below must not hold back on this request
      __builtin_amdgcn_sched_barrier(0);
      stage_it(it + 2 < n_it ? it + 2 : n_it - 1, ring >= 1 ? ring - 1 : 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (WLO && live) {  // e4m3 of this k-step's activation fragment, for the product with lo(w) one group later
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        const uint4 h = as_u4(a_hi[mf]);
        constexpr int d0 = 4 * (q >> 1) + 2 * (q & 1);
        h8_cur[mf][d0] = (int)f16x4_to_e4m3(h.x, h.y);
        h8_cur[mf][d0 + 1] = (int)f16x4_to_e4m3(h.z, h.w);
      }
    }
    // (head: the 8 fp16 steps only; tail: the e4m3 steps only)
    constexpr int ST0 = live ? 0 : 8, ST1 = f8_live ? (WLO ? 16 : 12) : 8;
    frag_stream2<ST1 - ST0, (WLO ? 1 : 3), OffShift<OffF8, ST0>>(lds_stage0 + (uint32_t)cur * (uint32_t)(STAGE * 2), [&](auto step_tag, bf16x8& w0, bf16x8& w1) {
      constexpr int st = decltype(step_tag)::value + ST0;
      if constexpr (st < 8) {
        if (live) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
              const bf16x8 w = j ? w1 : w0;
              acc[2 * st + j][mf] = SWAPPED ? mfma16h(w, a_hi[mf], acc[2 * st + j][mf]) : mfma16h(a_hi[mf], w, acc[2 * st + j][mf]);
            }
        }
      } else if constexpr (st < 12) {
        if (f8_live) {
          constexpr int nf = 4 * q + (st - 8);
          const i32x8 w8f = f8_frag(w0, w1);
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            acc[nf][mf] = SWAPPED ? mfma8<true>(w8f, lo_prev[mf], acc[nf][mf]) : mfma8<false>(lo_prev[mf], w8f, acc[nf][mf]);
        }
      } else {
        if (f8_live) {
          constexpr int nf = 4 * q + (st - 12);
          const i32x8 w8f = f8_frag(w0, w1);
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
            acc[nf][mf] = SWAPPED ? mfma8w<false>(w8f, h8_prev[mf], acc[nf][mf]) : mfma8w<true>(h8_prev[mf], w8f, acc[nf][mf]);
        }
      }
    });
    if constexpr (NST == 3) {  // the slab of iteration it + 1 has landed; the request issued above may still fly
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAVE_PIECES) : "memory");
      __builtin_amdgcn_s_barrier();
      ring = ring == 2 ? 0 : ring + 1;
    } else {
      __syncthreads();
    }
  };
  const std::integral_constant<int, 0> s0{};
  const std::integral_constant<int, 1> s1{};
  const std::integral_constant<int, 2> s2{};
  const std::integral_constant<int, 3> s3{};
  const std::true_type on{};
  const std::false_type off{};
  step(0, s0, s0, on, off);
  step(1, s1, s1, on, off);
  step(2, s0, s2, on, off);
  step(3, s1, s3, on, off);
  for (int it = 4; it < nks; it += 4) {
    step(it, s0, s0, on, on);
    step(it + 1, s1, s1, on, on);
    step(it + 2, s0, s2, on, on);
    step(it + 3, s1, s3, on, on);
  }
  step(nks, s0, s0, off, on);
  step(nks + 1, s1, s1, off, on);
  step(nks + 2, s0, s2, off, on);
  step(nks + 3, s1, s3, off, on);

  if (EPI == PE_RESIDUAL) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      float* xrow = p.x + (size_t)(m0 + mf * 16 + l15) * p.ld_out + tile * 256 + g * 8;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
        float4 r4 = *px;
        r4.x += acc[nf][mf][0];
        r4.y += acc[nf][mf][1];
        r4.z += acc[nf][mf][2];
        r4.w += acc[nf][mf][3];
        *px = r4;
      }
    }
  } else if (EPI == PE_GEGLU) {
    // h in the format the MLP output projection reads: fp16 pieces [rb][I/32][512] + one e4m3 half-fragment piece per
    // two k-steps; this panel's four k-steps are two of those
    const int kb_out = p.ld_out >> 5;
    if constexpr (OLO == 2) {
      // h as (hi, lo) bf16 pieces (panel_gemm_kernel's layout): the Wi GEMM alone runs in the fp16 + e4m3 format and the
      // MLP output projection that reads h stays on the (hi, lo) bf16 kernels (OP_FLAG_PANEL_F8_WI)
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        const size_t rb = (size_t)((m0 >> 4) + mf);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = gelu_erf(acc[2 * s][mf][r]) * acc[8 + 2 * s][mf][r];
            v[4 + r] = gelu_erf(acc[2 * s + 1][mf][r]) * acc[8 + 2 * s + 1][mf][r];
          }
          bf16x8 hi, lo;
          pack8<true>(v, hi, lo);
          u16* dst = o0 + ((rb * kb_out + (size_t)(tile * 4 + s)) * 2) * 512 + lane * 8;
          store_stream16(dst, as_u4(hi));
          store_stream16(dst + 512, as_u4(lo));
        }
      }
      return;
    }
    bool out_of_range = false;  // an h beyond fp16's range (the conversion below clamps it) or not finite
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const size_t rb = (size_t)((m0 >> 4) + mf);
      uint32_t lo8[8];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float va[4], vb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          va[r] = gelu_erf(acc[2 * s][mf][r]) * acc[8 + 2 * s][mf][r];
          vb[r] = gelu_erf(acc[2 * s + 1][mf][r]) * acc[8 + 2 * s + 1][mf][r];
          out_of_range |= !(fabsf(va[r]) <= 65504.f) | !(fabsf(vb[r]) <= 65504.f);
        }
        uint2 h0, h1;
        split4_f8(va, h0, lo8[2 * s]);
        split4_f8(vb, h1, lo8[2 * s + 1]);
        store_stream16(o0 + (rb * kb_out + (size_t)(tile * 4 + s)) * 512 + lane * 8, make_uint4(h0.x, h0.y, h1.x, h1.y));
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
        store_stream16(p.o0_lo8 + (rb * (kb_out >> 1) + (size_t)(tile * 2 + pr)) * 512 + lane * 8,
                       make_uint4(lo8[4 * pr], lo8[4 * pr + 1], lo8[4 * pr + 2], lo8[4 * pr + 3]));
    }
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0 && lane == 0 && p.range_flag != nullptr) atomicOr(p.range_flag, 1);
  } else if (EPI == PE_QK) {
    // O16: q / k become fp16 operands -- beyond fp16's range they turn into Inf (and the outputs into NaN: the range guard of
    // the Python layer repeats the batch on the (hi, lo) bf16 sets) instead of a silently clamped 65504
    if constexpr (O16) {
      set_overflowing_conversions();
      // (the mode write is an asm statement and a conversion has no dependency on it: the accumulators pass through empty asm
      // statements behind it -- volatile asm keeps its order, and every conversion below depends on one of these values)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) asm volatile("" : "+v"(acc[nf][mf]));
    }
    const int per = p.hidden / 256;
    const bool is_q = tile < per;
    const int tq = is_q ? tile : tile - per;
    u16* out = is_q ? o0 : p.o1;
    const float qscale = is_q ? 0.125f * 1.44269504088896340736f : 1.0f;
    const int kb_out = p.hidden >> 5;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      int pos = p.row_pos[m0 + mf * 16 + l15];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      f32x4 c4[2], s4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        c4[u] = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * ROPE_HALF + g * 8 + u * 4);
        s4[u] = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * ROPE_HALF + g * 8 + u * 4);
      }
      const size_t rb = (size_t)((m0 >> 4) + mf);
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        float lo_half[8], hi_half[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x1 = acc[4 * hh + u][mf][r], x2 = acc[4 * hh + 2 + u][mf][r];
            lo_half[4 * u + r] = rope_lo(x1, x2, c4[u][r], s4[u][r]) * qscale;
            hi_half[4 * u + r] = rope_hi(x1, x2, c4[u][r], s4[u][r]) * qscale;
          }
        bf16x8 h0, l0, h1, l1;
        pack8x<(O0_LO || O1_LO), O16>(lo_half, h0, l0);
        pack8x<(O0_LO || O1_LO), O16>(hi_half, h1, l1);
        u16* dst = out + ((rb * kb_out + (size_t)((tq * 4 + hh) * 2)) * 2) * 512 + lane * 8;
        store_stream16(dst, as_u4(h0));
        store_stream16(dst + 1024, as_u4(h1));
        if ((O0_LO || O1_LO) && (O0_LO == O1_LO || (is_q ? O0_LO : O1_LO))) {
          store_stream16(dst + 512, as_u4(l0));
          store_stream16(dst + 1536, as_u4(l1));
        }
      }
    }
  } else {  // PE_V
    if constexpr (O16) {
      set_overflowing_conversions();
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) asm volatile("" : "+v"(acc[nf][mf]));
    }
    const size_t tb = (size_t)(m0 >> 5);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const size_t head = (size_t)(tile * 4 + (nf >> 2));
      const float v[8] = {acc[nf][0][0], acc[nf][0][1], acc[nf][0][2], acc[nf][0][3],
                          acc[nf][1][0], acc[nf][1][1], acc[nf][1][2], acc[nf][1][3]};
      bf16x8 hi, lo;
      pack8x<O0_LO, O16>(v, hi, lo);
      u16* dst = o0 + (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + (size_t)(nf & 3)) * 512 + lane * 8;
      store_stream16(dst, as_u4(hi));
      if (O0_LO) store_stream16(dst + 2048, as_u4(lo));
    }
  }
}

#ifndef OPK_PANEL_F8_STAGES
#define OPK_PANEL_F8_STAGES 2
#endif
constexpr int panel_f8_stages(bool wlo) { return wlo ? 2 : OPK_PANEL_F8_STAGES; }  // (three 32 KiB stages x 2 blocks exceed the LDS)

template <int EPI, bool WLO, int OLO>
__global__ __launch_bounds__(256, 2) void panel_f8_gemm_kernel(PanelParams p) {
  constexpr int NST = panel_f8_stages(WLO);
  __shared__ __attribute__((aligned(16))) u16 sW[NST][panel_f8_stage_elems(WLO)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  panel_f8_block<EPI, WLO, OLO, NST>(p, row_block, tile, tile, p.o0, sW);
}

template <bool WLO, int OLO_QK, int OLO_V, bool O16 = false>
__global__ __launch_bounds__(256, 2) void panel_f8_qkv_kernel(PanelParams p) {
  constexpr int NST = panel_f8_stages(WLO);
  __shared__ __attribute__((aligned(16))) u16 sW[NST][panel_f8_stage_elems(WLO)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  if (tile < p.n_qk_tiles) panel_f8_block<PE_QK, WLO, OLO_QK, NST, O16>(p, row_block, tile, tile, p.o0, sW);
  else panel_f8_block<PE_V, WLO, OLO_V, NST, O16>(p, row_block, tile, tile - p.n_qk_tiles, p.o2, sW);
}

}  // namespace opk
