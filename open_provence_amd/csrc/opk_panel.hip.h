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
__global__ void pack_panel_kernel(const float* __restrict__ src, int n_tiles, int K, int mode, int H, int I,
                                  u16* __restrict__ dst, int zero_lo, int* __restrict__ any_lo) {
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
  const u16 h = f2bf(v);
  const size_t base = (((size_t)tile * KS + ks) * 2) * 8192 + (size_t)nf * 512 + (size_t)g * 128 + i * 8 + e;
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  dst[base] = h;
  dst[base + 8192] = zero_lo ? (u16)0 : l;
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
template <int EPI, int T, int OLO>
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
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + 4 * u;  // stage = [plane][nf] pieces, same order as the source (which has 2 planes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
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
            acc[nf + j][mf] = SWAPPED ? mfma16(w, a, acc[nf + j][mf]) : mfma16(a, w, acc[nf + j][mf]);
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
        pack8<O0_LO>(v, hi, lo);
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
        pack8<(O0_LO || O1_LO)>(lo_half, h0, l0);
        pack8<(O0_LO || O1_LO)>(hi_half, h1, l1);
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
      pack8<O0_LO>(v, hi, lo);
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

template <int EPI, int T, int OLO>
__global__ __launch_bounds__(256, 2) void panel_gemm_kernel(PanelParams p) {
  __shared__ __attribute__((aligned(16))) u16 sW[2][panel_stage_elems(T)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  panel_block<EPI, T, OLO>(p, row_block, tile, tile, p.o0, sW);
}

// q, k and v^T of a layer in ONE launch: the three projections read the same normalised rows, so their panels sit side
// by side in the XCD-local group (the rows are fetched once for all 3 H / 256 panels) and a launch is saved per layer.
// The v^T panels use the un-swapped MFMA orientation: a block-uniform branch picks the instantiation.
template <int T, int OLO_QK, int OLO_V>
__global__ __launch_bounds__(256, 2) void panel_qkv_kernel(PanelParams p) {
  __shared__ __attribute__((aligned(16))) u16 sW[2][panel_stage_elems(T)];
  int row_block, tile;
  if (!panel_block_map(p, row_block, tile)) return;
  if (tile < p.n_qk_tiles) panel_block<PE_QK, T, OLO_QK>(p, row_block, tile, tile, p.o0, sW);
  else panel_block<PE_V, T, OLO_V>(p, row_block, tile, tile - p.n_qk_tiles, p.o2, sW);
}

}  // namespace opk
