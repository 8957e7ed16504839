// opk_tiled.hip.h -- generic tiled fallback: 128 x 128 x 32 GEMM tiles and attention on row-major hi/lo planes
#pragma once

#include "opk_common.hip.h"

namespace opk {

// ----------------------------------------------------------------------------------------------
// GEMM  C[m, n] = sum_k A[m, k] * W[n, k]   (A = activation planes, W = nn.Linear weight planes)
// 128 x 128 tile, 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 accumulators.
// "Swapped" orientation (X = W rows, Y = A rows) leaves every lane with 4 consecutive OUTPUT
// FEATURES of one token, so all epilogues store 8-byte (bf16 planes) / 16-byte (fp32) pieces of a
// token row; the V projection uses the plain orientation to emit V transposed ([feature][row]).
// ----------------------------------------------------------------------------------------------
enum GemmEpilogue {
  EPI_QK_ROPE = 0,  // RoPE + (q * head_dim^-0.5) -> q / k planes                HF :188-219, :271-285
  EPI_V_T = 1,      // V transposed planes [H][R_pad]
  EPI_RESIDUAL = 2, // x += C  (attention Wo, MLP Wo)                            HF :331-332
  EPI_GEGLU = 3     // gelu_erf(input) * gate -> h planes                        HF :89-91
};

struct GemmParams {
  const u16* a_hi;
  const u16* a_lo;
  const u16* w_hi;
  const u16* w_lo;
  int K;        // reduction length (multiple of 32)
  int n_tiles;  // N / 128
  int m_tiles;  // R_pad / 128
  float* x;     // EPI_RESIDUAL: [R_pad][ld_out] fp32, updated in place
  u16* o0_hi;   // QK: q planes   V_T: vt planes   GEGLU: h planes
  u16* o0_lo;
  u16* o1_hi;   // QK: k planes
  u16* o1_lo;
  int ld_out;   // row stride (elements) of the output: H (QK, RESIDUAL), I (GEGLU), R_pad (V_T)
  int hidden;   // H (QK: column where the k block starts)
  const int32_t* row_pos;
  const float* rope_cos;  // [max_pos][32]
  const float* rope_sin;
  int max_pos;
};

template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) u16 sA[2][GEMM_BM * GEMM_LDS];
  __shared__ __attribute__((aligned(16))) u16 sW[2][GEMM_BN * GEMM_LDS];

  // XCD-aware tile order: consecutive tiles (same token rows, different feature tiles) share an XCD
  // and therefore its L2 copy of the A rows.  Bijective for any grid size.
  const int nwg = gridDim.x;
  const int orig = blockIdx.x;
  const int xcd = orig & 7;
  const int qd = nwg >> 3, rem = nwg & 7;
  const int wgid = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int n_tile = wgid % p.n_tiles;
  const int m_tile = wgid / p.n_tiles;
  const int m0 = m_tile * GEMM_BM;
  const int n0 = n_tile * GEMM_BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1;   // token half of the tile
  const int wn = wave >> 1;  // feature half of the tile
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int K = p.K;

  // staging: each plane tile is 128 rows x 32 k = 512 pieces of 16 B; thread handles pieces tid, tid+256
  const int srow = tid >> 2;
  const int skc = (tid & 3) * 8;
  const u16* ga_hi = p.a_hi + (size_t)(m0 + srow) * K + skc;
  const u16* ga_lo = p.a_lo + (size_t)(m0 + srow) * K + skc;
  const u16* gw_hi = p.w_hi + (size_t)(n0 + srow) * K + skc;
  const u16* gw_lo = p.w_lo + (size_t)(n0 + srow) * K + skc;
  const size_t half_rows = (size_t)64 * K;
  const int soff0 = srow * GEMM_LDS + skc;
  const int soff1 = (srow + 64) * GEMM_LDS + skc;

  // staging registers are plain scalars (arrays captured by lambdas end up in scratch memory)
  uint4 ra_hi0, ra_hi1, rw_hi0, rw_hi1;
  uint4 ra_lo0 = make_uint4(0, 0, 0, 0), ra_lo1 = ra_lo0, rw_lo0 = ra_lo0, rw_lo1 = ra_lo0;
#define OPK_GLOAD(kt_)                                                       \
  do {                                                                       \
    const int ko_ = (kt_) * GEMM_BK;                                          \
    ra_hi0 = *reinterpret_cast<const uint4*>(ga_hi + ko_);                    \
    ra_hi1 = *reinterpret_cast<const uint4*>(ga_hi + half_rows + ko_);        \
    rw_hi0 = *reinterpret_cast<const uint4*>(gw_hi + ko_);                    \
    rw_hi1 = *reinterpret_cast<const uint4*>(gw_hi + half_rows + ko_);        \
    if (SPLIT) {                                                             \
      ra_lo0 = *reinterpret_cast<const uint4*>(ga_lo + ko_);                  \
      ra_lo1 = *reinterpret_cast<const uint4*>(ga_lo + half_rows + ko_);      \
      rw_lo0 = *reinterpret_cast<const uint4*>(gw_lo + ko_);                  \
      rw_lo1 = *reinterpret_cast<const uint4*>(gw_lo + half_rows + ko_);      \
    }                                                                        \
  } while (0)
#define OPK_LSTORE()                                              \
  do {                                                            \
    *reinterpret_cast<uint4*>(&sA[0][soff0]) = ra_hi0;            \
    *reinterpret_cast<uint4*>(&sA[0][soff1]) = ra_hi1;            \
    *reinterpret_cast<uint4*>(&sW[0][soff0]) = rw_hi0;            \
    *reinterpret_cast<uint4*>(&sW[0][soff1]) = rw_hi1;            \
    if (SPLIT) {                                                  \
      *reinterpret_cast<uint4*>(&sA[1][soff0]) = ra_lo0;          \
      *reinterpret_cast<uint4*>(&sA[1][soff1]) = ra_lo1;          \
      *reinterpret_cast<uint4*>(&sW[1][soff0]) = rw_lo0;          \
      *reinterpret_cast<uint4*>(&sW[1][soff1]) = rw_lo1;          \
    }                                                             \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / GEMM_BK;
  OPK_GLOAD(0);
  OPK_LSTORE();
  __syncthreads();

  const int a_frag = (wm * 64 + l15) * GEMM_LDS + g * 8;
  const int w_frag = (wn * 64 + l15) * GEMM_LDS + g * 8;

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) OPK_GLOAD(kt + 1);
    bf16x8 wf_hi[4], af_hi[4], wf_lo[4], af_lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wf_hi[i] = lds_frag(&sW[0][w_frag + i * 16 * GEMM_LDS]);
      af_hi[i] = lds_frag(&sA[0][a_frag + i * 16 * GEMM_LDS]);
      if (SPLIT) {
        wf_lo[i] = lds_frag(&sW[1][w_frag + i * 16 * GEMM_LDS]);
        af_lo[i] = lds_frag(&sA[1][a_frag + i * 16 * GEMM_LDS]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (EPI == EPI_V_T) {  // rows = tokens (j), cols = features (i)
          if (SPLIT) {
            acc[i][j] = mfma16(af_lo[j], wf_hi[i], acc[i][j]);
            acc[i][j] = mfma16(af_hi[j], wf_lo[i], acc[i][j]);
          }
          acc[i][j] = mfma16(af_hi[j], wf_hi[i], acc[i][j]);
        } else {  // rows = features (i), cols = tokens (j)
          if (SPLIT) {
            acc[i][j] = mfma16(wf_lo[i], af_hi[j], acc[i][j]);
            acc[i][j] = mfma16(wf_hi[i], af_lo[j], acc[i][j]);
          }
          acc[i][j] = mfma16(wf_hi[i], af_hi[j], acc[i][j]);
        }
      }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      OPK_LSTORE();
      __syncthreads();
    }
  }
#undef OPK_GLOAD
#undef OPK_LSTORE

  // ------------------------------------------------------------------------------------------
  // epilogues
  // ------------------------------------------------------------------------------------------
  if (EPI == EPI_V_T) {
    // acc[i][j][r]: token m0 + wm*64 + 16j + 4g + r, feature n0 + wn*64 + 16i + l15
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = n0 + wn * 64 + i * 16 + l15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + g * 4;
        const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        uint2 h2, l2;
        split4<SPLIT>(v, h2, l2);
        const size_t off = (size_t)f * p.ld_out + m;
        *reinterpret_cast<uint2*>(p.o0_hi + off) = h2;
        if (SPLIT) *reinterpret_cast<uint2*>(p.o0_lo + off) = l2;
      }
    }
    return;
  }

  // swapped orientation: acc[i][j][r]: feature n0 + wn*64 + 16i + 4g + r, token m0 + wm*64 + 16j + l15
  if (EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = n0 + wn * 64 + i * 16 + g * 4;
        float4* px = reinterpret_cast<float4*>(p.x + (size_t)m * p.ld_out + f);
        float4 r4 = *px;
        r4.x += acc[i][j][0];
        r4.y += acc[i][j][1];
        r4.z += acc[i][j][2];
        r4.w += acc[i][j][3];
        *px = r4;
      }
    }
    return;
  }

  if (EPI == EPI_GEGLU) {
    // weight rows were interleaved at load time: within this wave's 64 features, i = 0,1 are 32
    // "input" columns and i = 2,3 the 32 matching "gate" columns (Wi.chunk(2), HF :90).
    const int out_col0 = (n0 >> 1) + wn * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(acc[i][j][r]) * acc[i + 2][j][r];
        uint2 h2, l2;
        split4<SPLIT>(v, h2, l2);
        const size_t off = (size_t)m * p.ld_out + out_col0 + i * 16 + g * 4;
        *reinterpret_cast<uint2*>(p.o0_hi + off) = h2;
        if (SPLIT) *reinterpret_cast<uint2*>(p.o0_lo + off) = l2;
      }
    }
    return;
  }

  if (EPI == EPI_QK_ROPE) {
    // this wave's 64 features are exactly one head of q (columns < hidden) or of k.
    const int col = n0 + wn * 64;
    const bool is_q = col < p.hidden;
    u16* out_hi = is_q ? p.o0_hi : p.o1_hi;
    u16* out_lo = is_q ? p.o0_lo : p.o1_lo;
    const int out_col = is_q ? col : col - p.hidden;
    const float qscale = is_q ? 0.125f : 1.0f;  // head_dim^-0.5, exact power of two
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + l15;
      int pos = p.row_pos[m];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // d = 16i + 4g + r pairs with d + 32 (rotate_half: first half / second half, HF :188-192)
        const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + (size_t)pos * ROPE_HALF + i * 16 + g * 4);
        const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + (size_t)pos * ROPE_HALF + i * 16 + g * 4);
        const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
        const float sn[4] = {s4.x, s4.y, s4.z, s4.w};
        float lo_half[4], hi_half[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x1 = acc[i][j][r], x2 = acc[i + 2][j][r];
          lo_half[r] = rope_lo(x1, x2, cs[r], sn[r]) * qscale;
          hi_half[r] = rope_hi(x1, x2, cs[r], sn[r]) * qscale;
        }
        uint2 h2, l2;
        const size_t off = (size_t)m * p.ld_out + out_col + i * 16 + g * 4;
        split4<SPLIT>(lo_half, h2, l2);
        *reinterpret_cast<uint2*>(out_hi + off) = h2;
        if (SPLIT) *reinterpret_cast<uint2*>(out_lo + off) = l2;
        split4<SPLIT>(hi_half, h2, l2);
        *reinterpret_cast<uint2*>(out_hi + off + 32) = h2;
        if (SPLIT) *reinterpret_cast<uint2*>(out_lo + off + 32) = l2;
      }
    }
    return;
  }
}

// ----------------------------------------------------------------------------------------------
// Attention over the packed layout.  One block = 64 queries of one (sequence, head); each of the 4
// waves owns 16 queries.  Scores are computed TRANSPOSED (S^T = K Q^T) so that a lane owns ONE query
// column: the online softmax needs two cross-lane steps per tile, P never leaves registers, and
// O^T = V^T P^T consumes it directly as the MFMA Y operand.  K rows are stored in LDS in a permuted
// order chosen so that the 8 keys a lane holds after S^T are 8 CONSECUTIVE keys -> the V^T operand is
// one 16-byte LDS read.  window < 0: full attention; else keys with |q - k| <= window
// (masking_utils.py:141-150), always intersected with key < len (the padding mask).
// ----------------------------------------------------------------------------------------------
struct AttnParams {
  const u16* q_hi;
  const u16* q_lo;
  const u16* k_hi;
  const u16* k_lo;
  const u16* vt_hi;  // [H][r_pad]
  const u16* vt_lo;
  u16* o_hi;
  u16* o_lo;
  const int32_t* cu;
  int s0;
  const int32_t* roff;
  int H;
  int r_pad;
  int window;
  int zero_p_lo;  // evaluate a policy without the lo(p) x hi(v) term on this (all-terms) kernel
};

template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) u16 sK[2][ATT_BK * ATT_LDS];
  __shared__ __attribute__((aligned(16))) u16 sV[2][HEAD_DIM * ATT_LDS];

  const int s = blockIdx.z;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * ATT_BQ;
  const int seq_start = p.cu[p.s0 + s];
  const int len = p.cu[p.s0 + s + 1] - seq_start;
  if (q0 >= len) return;
  const int r0 = p.roff[s];
  const int alloc = p.roff[s + 1] - r0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int H = p.H;
  const int hcol = head * HEAD_DIM;

  const int qbase = q0 + wave * 16;
  const bool active = qbase < alloc;  // alloc is a multiple of 16: whole wave in or out
  const int qpos = qbase + l15;
  const size_t qrow = (size_t)(r0 + (active ? qpos : q0));

  bf16x8 qf_hi[2], qf_lo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf_hi[ks] = as_frag(*reinterpret_cast<const uint4*>(p.q_hi + qrow * H + hcol + ks * 32 + g * 8));
    if (SPLIT) qf_lo[ks] = as_frag(*reinterpret_cast<const uint4*>(p.q_lo + qrow * H + hcol + ks * 32 + g * 8));
  }

  int kt_lo = 0, kt_hi = (len - 1) / ATT_BK;
  if (p.window >= 0) {
    const int lo_key = q0 - p.window;
    kt_lo = lo_key > 0 ? lo_key / ATT_BK : 0;
    const int hi_key = q0 + ATT_BQ - 1 + p.window;
    const int hi_t = hi_key / ATT_BK;
    kt_hi = hi_t < kt_hi ? hi_t : kt_hi;
  }

  float m_run = -1e30f;
  float l_run = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) oacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: tile = 64 rows x 64 elements = 512 pieces of 16 B per plane; thread handles 2
  const int prow = tid >> 3;          // 0..31 (+32 for the second piece)
  const int pcol = (tid & 7) * 8;
  auto kperm = [](int key) { return (key & 32) | (((key >> 2) & 1) << 4) | (((key >> 3) & 3) << 2) | (key & 3); };

  for (int kt = kt_lo; kt <= kt_hi; ++kt) {
    const int kbase = kt * ATT_BK;
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = prow + 32 * u;
      const size_t grow = (size_t)(r0 + kbase + row) * H + hcol + pcol;
      const int kdst = kperm(row) * ATT_LDS + pcol;
      *reinterpret_cast<uint4*>(&sK[0][kdst]) = *reinterpret_cast<const uint4*>(p.k_hi + grow);
      const size_t gv = (size_t)(hcol + row) * p.r_pad + r0 + kbase + pcol;
      const int vdst = row * ATT_LDS + pcol;
      *reinterpret_cast<uint4*>(&sV[0][vdst]) = *reinterpret_cast<const uint4*>(p.vt_hi + gv);
      if (SPLIT) {
        *reinterpret_cast<uint4*>(&sK[1][kdst]) = *reinterpret_cast<const uint4*>(p.k_lo + grow);
        *reinterpret_cast<uint4*>(&sV[1][vdst]) = *reinterpret_cast<const uint4*>(p.vt_lo + gv);
      }
    }
    __syncthreads();

    // S^T tile: rows = keys (4 fragments of 16 LDS rows), column = this lane's query
    f32x4 sacc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) sacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int off = (m * 16 + l15) * ATT_LDS + ks * 32 + g * 8;
        const bf16x8 kh = lds_frag(&sK[0][off]);
        if (SPLIT) {
          const bf16x8 kl = lds_frag(&sK[1][off]);
          sacc[m] = mfma16(kl, qf_hi[ks], sacc[m]);
          sacc[m] = mfma16(kh, qf_lo[ks], sacc[m]);
        }
        sacc[m] = mfma16(kh, qf_hi[ks], sacc[m]);
      }
    }

    // element (m, r) of this lane: key = kbase + 32*(m>>1) + 8*g + 4*(m&1) + r
    // masked scores become -3e30 (below the running-max initial value -1e30): exp(masked - max) underflows to
    // exactly 0 even when a whole tile is masked for this query; branch-free (selects only).
    const int win = p.window >= 0 ? p.window : (1 << 30);
    float tile_max = -3e30f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kbase + 32 * (m >> 1) + 8 * g + 4 * (m & 1) + r;
        const int d = key - qpos;
        const bool ok = (key < len) & (d <= win) & (d >= -win);
        sacc[m][r] = ok ? sacc[m][r] : -3e30f;
        tile_max = fmaxf(tile_max, sacc[m][r]);
      }
    }
    tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 16, 64));
    tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 32, 64));
    const float m_new = fmaxf(m_run, tile_max);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    float pv[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(sacc[m][r] - m_new);
        pv[m][r] = e;
        psum += e;
      }
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      oacc[n][0] *= alpha;
      oacc[n][1] *= alpha;
      oacc[n][2] *= alpha;
      oacc[n][3] *= alpha;
    }

    // O^T += V^T P^T, two k-steps of 32 keys; lane's k-slots = keys 32t + 8g + (0..7)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v8[8] = {pv[2 * t][0],     pv[2 * t][1],     pv[2 * t][2],     pv[2 * t][3],
                           pv[2 * t + 1][0], pv[2 * t + 1][1], pv[2 * t + 1][2], pv[2 * t + 1][3]};
      uint2 h0, l0, h1, l1;
      split4<SPLIT>(v8, h0, l0);
      split4<SPLIT>(v8 + 4, h1, l1);
      const bf16x8 ph = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
      const bf16x8 pl = p.zero_p_lo ? as_frag(make_uint4(0u, 0u, 0u, 0u)) : as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int off = (n * 16 + l15) * ATT_LDS + t * 32 + g * 8;
        const bf16x8 vh = lds_frag(&sV[0][off]);
        if (SPLIT) {
          const bf16x8 vl = lds_frag(&sV[1][off]);
          oacc[n] = mfma16(vl, ph, oacc[n]);
          oacc[n] = mfma16(vh, pl, oacc[n]);
        }
        oacc[n] = mfma16(vh, ph, oacc[n]);
      }
    }
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (active) {
    // oacc[n][r]: d = 16n + 4g + r of query qpos
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float v[4] = {oacc[n][0] * inv, oacc[n][1] * inv, oacc[n][2] * inv, oacc[n][3] * inv};
      uint2 h2, l2;
      split4<SPLIT>(v, h2, l2);
      const size_t off = qrow * H + hcol + n * 16 + g * 4;
      *reinterpret_cast<uint2*>(p.o_hi + off) = h2;
      if (SPLIT) *reinterpret_cast<uint2*>(p.o_lo + off) = l2;
    }
  }
}

}  // namespace opk
