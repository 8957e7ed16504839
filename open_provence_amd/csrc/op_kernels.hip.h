// op_kernels.hip.h -- CDNA4 (gfx950) device code of the OpenProvence forward path.
//
// Packed ("unpadded") row layout: the tokens of a chunk of sequences are laid end to end, each
// sequence starting at a multiple of ROW_ALIGN rows; `row_pos[r] < 0` marks an alignment row.  All
// activations are [rows, features] row-major.  The fp32 residual stream `x` is the only fp32
// activation; every MFMA operand is stored as bf16 planes: `*_hi` = RNE(bf16(v)) and (BF16X3 mode)
// `*_lo` = RNE(bf16(v - hi)), so that  a*b ~= a_hi*b_hi + a_lo*b_hi + a_hi*b_lo  on the bf16 MFMA pipe
// with fp32 accumulation (~2^-16 relative) -- what the 1e-3 parity bar against the fp32 CPU
// reference needs (single-pass bf16 is ~1e-2, SURVEY.md headline fact 5).
//
// Arithmetic restated from (third-party) HF ModernBERT, transformers 5.15.0:
//   embeddings+LN  modeling_modernbert.py:52-71     GeGLU MLP  :74-91      RoPE :94-219
//   attention      :166-185, :222-301               layer      :304-333    heads :481-490, :569-622
// and the reference's OpenProvenceHead (open_provence/modeling_open_provence_standalone.py:434-448).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace opk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t u16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16-byte staging register (stays in VGPRs)

constexpr int ROW_ALIGN = 32;   // sequence starts are multiples of this many rows (= one wave's row group)
constexpr int GEMM_BM = 128;    // rows (tokens) per GEMM tile
constexpr int GEMM_BN = 128;    // output features per GEMM tile
constexpr int GEMM_BK = 32;     // one 16x16x32 MFMA step
constexpr int GEMM_LDS = 48;    // LDS row stride in bf16 elements (32 + 16 pad: conflict-free b128 reads)
constexpr int ATT_BQ = 64;      // queries per attention block (4 waves x 16)
constexpr int ATT_BK = 64;      // keys per tile
constexpr int ATT_LDS = 80;     // LDS row stride in bf16 elements (64 + 16 pad)
constexpr int HEAD_DIM = 64;
constexpr int ROPE_HALF = 32;

// ----------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ u16 f2bf(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs only)
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// two fp32 -> one dword of two bf16 (round to nearest even): a single v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// (hi, lo) bf16 split of a pair: hi = RNE(v), lo = RNE(v - hi); 5 VALU instructions per pair
template <bool SPLIT>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = SPLIT ? pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)) : 0u;
}

template <bool SPLIT>
__device__ __forceinline__ void split4(const float v[4], uint2& hi, uint2& lo) {
  split2<SPLIT>(v[0], v[1], hi.x, lo.x);
  split2<SPLIT>(v[2], v[3], hi.y, lo.y);
}

union FragU {
  uint4 u;
  bf16x8 v;
};
__device__ __forceinline__ bf16x8 as_frag(uint4 u) {
  FragU f;
  f.u = u;
  return f.v;
}
// Load the fragment as an ext-vector type: an LDS load typed as HIP's uint4 class makes hipcc (ROCm 7.2) treat it
// as possibly aliasing an in-flight global_load_lds DMA and drain the DMA (s_waitcnt vmcnt(0)) in front of it.
__device__ __forceinline__ bf16x8 lds_frag(const u16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// D = X * Y + C on one wave.  X fragment: row (lane & 15), k-group (lane >> 4) holds 8 consecutive k.
// Y fragment: column (lane & 15), same k-group.  D: column (lane & 15), rows 4*(lane >> 4) + r.
__device__ __forceinline__ f32x4 mfma16(bf16x8 x, bf16x8 y, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// GELU (exact-erf form) = 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |x| he(|x|),  he(a) = erfc(a / sqrt 2) / 2,
// with he(a) = 2^q(a), q a degree-5 polynomial (weighted minimax fit of log2 he on [0, 10], weight a he(a); leading
// coefficient negative, so q -> -inf and he -> 0 for large |x|).  Eight instructions: five FMAs, one v_exp_f32, one
// FMA, one med3 -- the VALU work next to the MFMAs is what these epilogues cost, instruction for instruction.  No
// cancellation on either side of zero.  Max |gelu - exact| = 6.4e-7 over [-12, 12] evaluated in fp32 (offline, against
// fp64 erf); the library erff costs ~3x the instructions.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  float q = fmaf(-4.7330930829e-04f, ax, 7.0845573209e-03f);
  q = fmaf(q, ax, -5.1827382296e-02f);
  q = fmaf(q, ax, -4.5999243855e-01f);
  q = fmaf(q, ax, -1.1507878304e+00f);
  q = fmaf(q, ax, -1.0000376701e+00f);
  const float he = __builtin_amdgcn_exp2f(q);
  // max(x, 0) as med3(x, 0, +inf): one instruction (fmaxf adds a NaN-quieting v_max x,x in front)
  return fmaf(-he, ax, __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()));
}

// ----------------------------------------------------------------------------------------------
// row map: sequence offsets (aligned) and per-row (seq, pos, token index)
// ----------------------------------------------------------------------------------------------
// roff[i] = sum_{j<i} ceil(len_j / unit) * scale  (exclusive prefix; roff[ns] = total).  unit = scale = ROW_ALIGN gives
// the aligned row offsets, unit = queries per attention block with scale = 1 the first work item of each sequence.
__global__ __launch_bounds__(1024) void seq_offsets_kernel(const int32_t* __restrict__ cu, int s0, int ns, int unit,
                                                           int scale, int32_t* __restrict__ roff) {
  __shared__ int sh[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ns; base += 1024) {
    const int i = base + tid;
    int v = 0;
    if (i < ns) {
      const int len = cu[s0 + i + 1] - cu[s0 + i];
      v = (len + unit - 1) / unit * scale;
    }
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = tid >= off ? sh[tid - off] : 0;
      __syncthreads();
      sh[tid] += t;
      __syncthreads();
    }
    const int incl = sh[tid];
    const int c = carry;
    if (i < ns) roff[i] = c + incl - v;
    __syncthreads();
    if (tid == 1023) carry = c + incl;
    __syncthreads();
  }
  if (tid == 0) roff[ns] = carry;
}

__global__ void row_map_kernel(const int32_t* __restrict__ cu, int s0, int ns, const int32_t* __restrict__ roff,
                               int r_pad, int32_t* __restrict__ row_seq, int32_t* __restrict__ row_pos,
                               int32_t* __restrict__ row_tok) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= r_pad) return;
  const int total = roff[ns];
  if (r >= total) {
    row_seq[r] = -1;
    row_pos[r] = -1;
    row_tok[r] = -1;
    return;
  }
  int lo = 0, hi = ns - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (roff[mid] <= r) lo = mid; else hi = mid - 1;
  }
  const int pos = r - roff[lo];
  const int start = cu[s0 + lo];
  const int len = cu[s0 + lo + 1] - start;
  row_seq[r] = lo;
  if (pos < len) {
    row_pos[r] = pos;
    row_tok[r] = start + pos;
  } else {
    row_pos[r] = -1;
    row_tok[r] = -1;
  }
}

// ----------------------------------------------------------------------------------------------
// LayerNorm family: one wave per row, float4 per lane-chunk, H <= 1024, H % 4 == 0
// ----------------------------------------------------------------------------------------------
constexpr int LN_MAX_CHUNKS = 4;  // float4 chunks per lane: H <= 4 * 64 * 4 = 1024

struct RowVec {
  float4 v[LN_MAX_CHUNKS];
};

__device__ __forceinline__ void row_load(const float* __restrict__ src, int H, int lane, RowVec& rv) {
  const int nchunk = H >> 2;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    rv.v[k] = (c < nchunk) ? reinterpret_cast<const float4*>(src)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// y = (x - mean) / sqrt(var + eps) * w   (biased variance, as torch.nn.LayerNorm)
__device__ __forceinline__ void row_layer_norm(RowVec& rv, const float* __restrict__ w, int H, int lane, float eps) {
  const int nchunk = H >> 2;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) s += (rv.v[k].x + rv.v[k].y) + (rv.v[k].z + rv.v[k].w);
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const float a = rv.v[k].x - mean, b = rv.v[k].y - mean, cc = rv.v[k].z - mean, d = rv.v[k].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wave_sum(q) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const float4 ww = reinterpret_cast<const float4*>(w)[c];
      rv.v[k].x = (rv.v[k].x - mean) * rstd * ww.x;
      rv.v[k].y = (rv.v[k].y - mean) * rstd * ww.y;
      rv.v[k].z = (rv.v[k].z - mean) * rstd * ww.z;
      rv.v[k].w = (rv.v[k].w - mean) * rstd * ww.w;
    }
  }
}

template <bool SPLIT>
__device__ __forceinline__ void row_store_planes(const RowVec& rv, u16* __restrict__ hi, u16* __restrict__ lo, int H,
                                                 int lane) {
  const int nchunk = H >> 2;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const float v[4] = {rv.v[k].x, rv.v[k].y, rv.v[k].z, rv.v[k].w};
      uint2 h2, l2;
      split4<SPLIT>(v, h2, l2);
      reinterpret_cast<uint2*>(hi)[c] = h2;
      if (SPLIT) reinterpret_cast<uint2*>(lo)[c] = l2;
    }
  }
}

__device__ __forceinline__ void row_store_f32(const RowVec& rv, float* __restrict__ dst, int H, int lane) {
  const int nchunk = H >> 2;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) reinterpret_cast<float4*>(dst)[c] = rv.v[k];
  }
}

// x0 = LN(E[id]) -> residual stream (fp32) and, because layer 0 has attn_norm = Identity
// (modeling_modernbert.py:309-312), directly the Wqkv operand planes.  Alignment rows get zeros.
template <bool SPLIT>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids,
                                                       const int32_t* __restrict__ row_tok,
                                                       const float* __restrict__ table, const float* __restrict__ lnw,
                                                       float eps, int H, int r_pad, int vocab, float* __restrict__ x,
                                                       u16* __restrict__ a_hi, u16* __restrict__ a_lo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= r_pad) return;
  const int tok = row_tok[row];
  RowVec rv;
  if (tok < 0) {
#pragma unroll
    for (int k = 0; k < LN_MAX_CHUNKS; ++k) rv.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    int id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    row_load(table + (size_t)id * H, H, lane, rv);
    row_layer_norm(rv, lnw, H, lane, eps);
  }
  row_store_f32(rv, x + (size_t)row * H, H, lane);
  row_store_planes<SPLIT>(rv, a_hi + (size_t)row * H, a_lo + (size_t)row * H, H, lane);
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, const float* __restrict__ lnw, float eps,
                                                 int H, int r_pad, u16* __restrict__ a_hi, u16* __restrict__ a_lo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= r_pad) return;
  RowVec rv;
  row_load(x + (size_t)row * H, H, lane, rv);
  row_layer_norm(rv, lnw, H, lane, eps);
  row_store_planes<SPLIT>(rv, a_hi + (size_t)row * H, a_lo + (size_t)row * H, H, lane);
}

// test hook: copy the residual stream rows of real tokens to the caller's packed [T, H] layout
__global__ __launch_bounds__(256) void capture_rows_kernel(const float* __restrict__ x,
                                                           const int32_t* __restrict__ row_tok, int H, int r_pad,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= r_pad) return;
  const int tok = row_tok[row];
  if (tok < 0) return;
  RowVec rv;
  row_load(x + (size_t)row * H, H, lane, rv);
  row_store_f32(rv, out + (size_t)tok * H, H, lane);
}

// final_norm + OpenProvenceHead Linear(H, 2) on every real token (standalone.py:446-448); keeps the
// normalised row for the ranking head (CLS row or, for mean pooling, every row -- written over x).
__global__ __launch_bounds__(256) void final_ln_prune_kernel(float* __restrict__ x, const float* __restrict__ lnw,
                                                             float eps, int H, int r_pad,
                                                             const int32_t* __restrict__ row_tok,
                                                             const int32_t* __restrict__ row_seq,
                                                             const int32_t* __restrict__ row_pos,
                                                             const float* __restrict__ pw, const float* __restrict__ pb,
                                                             float* __restrict__ prune_out, int keep_all_rows,
                                                             float* __restrict__ cls, float* __restrict__ capture) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= r_pad) return;
  const int tok = row_tok[row];
  if (tok < 0) return;
  RowVec rv;
  row_load(x + (size_t)row * H, H, lane, rv);
  row_layer_norm(rv, lnw, H, lane, eps);
  const int nchunk = H >> 2;
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const float4 w0 = reinterpret_cast<const float4*>(pw)[c];
      const float4 w1 = reinterpret_cast<const float4*>(pw + H)[c];
      d0 += (rv.v[k].x * w0.x + rv.v[k].y * w0.y) + (rv.v[k].z * w0.z + rv.v[k].w * w0.w);
      d1 += (rv.v[k].x * w1.x + rv.v[k].y * w1.y) + (rv.v[k].z * w1.z + rv.v[k].w * w1.w);
    }
  }
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  if (lane == 0) {
    prune_out[(size_t)tok * 2 + 0] = d0 + pb[0];
    prune_out[(size_t)tok * 2 + 1] = d1 + pb[1];
  }
  if (keep_all_rows) row_store_f32(rv, x + (size_t)row * H, H, lane);
  if (row_pos[row] == 0) row_store_f32(rv, cls + (size_t)row_seq[row] * H, H, lane);
  if (capture) row_store_f32(rv, capture + (size_t)tok * H, H, lane);
}

// ModernBertPredictionHead + classifier on the pooled row (modeling_modernbert.py:481-490, 609-622):
// logits = classifier(LN(gelu(dense(pooled)))).  One block per sequence; dense weight stored
// transposed [k][n] so that thread n reads coalesced.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void rank_head_kernel(const float* __restrict__ cls, const float* __restrict__ y,
                                                        const int32_t* __restrict__ cu, int s0,
                                                        const int32_t* __restrict__ roff, int mean_pool, int H, int nl,
                                                        const float* __restrict__ dense_t,
                                                        const float* __restrict__ head_norm, float eps,
                                                        const float* __restrict__ cls_w, const float* __restrict__ cls_b,
                                                        float* __restrict__ rank_out) {
  __shared__ float pooled[1024];
  __shared__ float z[1024];
  __shared__ float red[4];
  const int s = blockIdx.x;
  const int tid = threadIdx.x;
  const int len = cu[s0 + s + 1] - cu[s0 + s];
  if (len <= 0) {
    if (tid < nl) rank_out[(size_t)(s0 + s) * nl + tid] = 0.f;
    return;
  }
  for (int k = tid; k < H; k += 256) {
    float v;
    if (mean_pool) {
      const float* base = y + (size_t)roff[s] * H + k;
      float acc = 0.f;
      for (int p = 0; p < len; ++p) acc += base[(size_t)p * H];
      v = acc / (float)len;
    } else {
      v = cls[(size_t)s * H + k];
    }
    pooled[k] = v;
  }
  __syncthreads();
  float lsum = 0.f;
  for (int n = tid; n < H; n += 256) {
    float acc = 0.f;
    for (int k = 0; k < H; ++k) acc = fmaf(pooled[k], dense_t[(size_t)k * H + n], acc);
    const float gl = gelu_erf(acc);
    z[n] = gl;
    lsum += gl;
  }
  const float mean = block_sum_256(lsum, red) / (float)H;
  float lq = 0.f;
  for (int n = tid; n < H; n += 256) {
    const float d = z[n] - mean;
    lq += d * d;
  }
  const float var = block_sum_256(lq, red) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int c = 0; c < nl; ++c) {
    float part = 0.f;
    for (int n = tid; n < H; n += 256) part += (z[n] - mean) * rstd * head_norm[n] * cls_w[(size_t)c * H + n];
    const float tot = block_sum_256(part, red);
    if (tid == 0) rank_out[(size_t)(s0 + s) * nl + c] = tot + cls_b[c];
  }
}

// ----------------------------------------------------------------------------------------------
// weight re-packing (runs once per tensor at load time)
// ----------------------------------------------------------------------------------------------
__global__ void convert_to_f32_kernel(const void* __restrict__ src, int dtype, size_t n, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dtype == 0) {
    dst[i] = reinterpret_cast<const float*>(src)[i];
  } else if (dtype == 1) {
    dst[i] = bf2f(reinterpret_cast<const u16*>(src)[i]);
  } else {
    dst[i] = __half2float(reinterpret_cast<const __half*>(src)[i]);
  }
}

// dst planes [rows][cols]; source row for destination row r is perm(r): identity, or the GeGLU
// interleave that puts the 32 "input" rows and the 32 matching "gate" rows of Wi in one 64-row slab.
__global__ void split_planes_kernel(const float* __restrict__ src, int rows, int cols, int geglu_half,
                                    u16* __restrict__ hi, u16* __restrict__ lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  int sr = r;
  if (geglu_half > 0) {
    const int b = r >> 6, j = r & 63;
    sr = (j < 32) ? (b * 32 + j) : (geglu_half + b * 32 + (j - 32));
  }
  const float v = src[(size_t)sr * cols + c];
  const u16 h = f2bf(v);
  hi[i] = h;
  lo[i] = f2bf(v - bf2f(h));
}

__global__ void transpose_f32_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(size_t)c * rows + r] = src[i];
}

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
          lo_half[r] = (x1 * cs[r] - x2 * sn[r]) * qscale;
          hi_half[r] = (x2 * cs[r] + x1 * sn[r]) * qscale;
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
// Row-stationary GEMM for K = hidden <= 256 (the three projections whose input is the hidden state):
//   C[m, n] = sum_k A[m, k] W[n, k],  A = 128 rows per block kept IN REGISTERS as MFMA fragments
//   (each wave owns 32 rows = 2 fragments x K/32 k-steps x hi/lo), W streamed through LDS in chunks of
//   32 output features, double-buffered, one barrier per chunk (96 MFMAs per wave between barriers).
// Why: the activations are the big operand (read exactly once, never staged through LDS); the weights
// are small, L2-resident and STATIC, so they are pre-packed at load time in exactly the order the MFMA
// X/Y fragments want them ([chunk][k-step][plane][frag][k-group][row][8]) -- every LDS fragment read
// is a lane-linear, conflict-free 1 KiB ds_read_b128, every global->LDS copy a linear memcpy.
// Prologues fuse what used to be separate kernels: LayerNorm (+ hi/lo split) of the fp32 residual
// stream is computed in registers directly in fragment layout (a row lives in 4 lanes).
// ----------------------------------------------------------------------------------------------
enum RowEpilogue { RE_QKV = 0, RE_RESIDUAL = 1, RE_GEGLU = 2 };
enum RowPrologue { RP_LN = 0, RP_SPLIT = 1, RP_PLANES = 2, RP_KSTREAM = 3 };
constexpr int ROW_BM = 128;
constexpr int ROW_CHUNK = 32;  // output features per streamed chunk

struct RowGemmParams {
  const float* x_in;  // RP_LN / RP_SPLIT: fp32 [r_pad][K]
  const float* ln_w;  // RP_LN
  float eps;
  const u16* a_hi;  // RP_PLANES: planes [r_pad][K]
  const u16* a_lo;
  const u16* wp;  // packed weights, n_chunks x (K/32) x 2 planes x 2 frags x 512 elements
  int n_chunks;
  int n_swapped;  // RE_QKV: chunks [0, n_swapped) are q/k (RoPE), the rest v (transposed store)
  float* x;       // RE_RESIDUAL: fp32 [r_pad][ld_out], updated in place
  u16* o0_hi;     // RE_QKV: q   RE_GEGLU: h
  u16* o0_lo;
  u16* o1_hi;  // RE_QKV: k
  u16* o1_lo;
  u16* o2_hi;  // RE_QKV: v^T [H][r_pad]
  u16* o2_lo;
  int ld_out;  // RE_RESIDUAL: H   RE_GEGLU: I   RE_QKV: H
  int hidden;
  int r_pad;
  const int32_t* row_pos;
  const float* rope_cos;
  const float* rope_sin;
  int max_pos;
  // RP_KSTREAM (fused block): x_new = x + A1 W1^T first, A1 fragment-packed [r_pad/16][k1_steps][2][512], W1 packed
  // by pack_kstream_kernel with permuted output features; then LayerNorm(x_new) feeds the chunk loop.
  const u16* a1_fp;
  const u16* w1p;
  int k1_steps;
  float* x_io;
};

// source row of packed row `pr` (0..31) of chunk `c`
__device__ __forceinline__ int rowgemm_source_row(int mode, int c, int pr, int H, int I) {
  const int nf = pr >> 4, i = pr & 15;
  if (mode == RE_QKV) {
    const int per_block = H / ROW_CHUNK;  // chunks in each of q, k, v
    const int blk = c / per_block, cc = c % per_block;
    const int head = cc >> 1, j = cc & 1;
    // q / k: the chunk pair (j = 0, 1) of a head leaves lane slot i = 4g + r with d = 8g + 4j + r (fragment 0)
    // and its RoPE partner d + 32 (fragment 1): after the pair a lane owns 8 consecutive d of both k-steps.
    if (blk < 2) return blk * H + head * HEAD_DIM + 32 * nf + 8 * (i >> 2) + 4 * j + (i & 3);
    // v: fragment nf of chunk j becomes piece n = 2j + nf of the transposed layout, whose row i is
    // d = 32j + 8(i>>2) + 4nf + (i&3) -- the order that makes the attention output lane-contiguous.
    return 2 * H + head * HEAD_DIM + 32 * j + 8 * (i >> 2) + 4 * nf + (i & 3);
  }
  if (mode == RE_GEGLU) {  // chunk pair (2t, 2t+1): lane slot i = 4g + r -> h-column 32t + 8g + 4u + r
    const int col = 32 * (c >> 1) + 8 * (i >> 2) + 4 * (c & 1) + (i & 3);
    return nf == 0 ? col : I + col;  // input column | matching gate column
  }
  return c * ROW_CHUNK + pr;
}

// dst[chunk][ks][plane][nf][g][i][e] <- src[source_row(chunk, nf*16+i)][ks*32 + g*8 + e]
__global__ void pack_rowgemm_kernel(const float* __restrict__ src, int n_rows, int K, int mode, int H, int I,
                                    u16* __restrict__ dst) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * K;
  if (idx >= total) return;
  const int KS = K / 32;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t & 1); t >>= 1;
  const int ks = (int)(t % KS);
  const int c = (int)(t / KS);
  const int srow = rowgemm_source_row(mode, c, nf * 16 + i, H, I);
  const float v = src[(size_t)srow * K + ks * 32 + g * 8 + e];
  const u16 h = f2bf(v);
  const size_t base = (((size_t)c * KS + ks) * 2) * 1024 + (size_t)nf * 512 + (size_t)g * 128 + i * 8 + e;
  dst[base] = h;
  dst[base + 1024] = f2bf(v - bf2f(h));
}

template <bool SPLIT>
__device__ __forceinline__ void pack8(const float v[8], bf16x8& hi, bf16x8& lo) {
  uint2 h0, l0, h1, l1;
  split4<SPLIT>(v, h0, l0);
  split4<SPLIT>(v + 4, h1, l1);
  hi = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
  lo = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// One weight chunk (32 output features x K) against this wave's 32 rows: 2 x 2 accumulators, K/32 k-steps of
// 12 (split) or 4 MFMAs.  (hipcc hoists the fragment reads one k-step ahead of their MFMAs by itself; an explicit
// register double buffer only cost 16 VGPRs.)
template <int KS, int MF, bool SPLIT, bool SWAPPED>
__device__ __forceinline__ void rowgemm_chunk_mfma(const u16* stage_lane, const bf16x8 (&a_hi)[MF][KS],
                                                   const bf16x8 (&a_lo)[MF][KS], f32x4 (&acc)[2][MF]) {
  constexpr int PLANES = SPLIT ? 2 : 1;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 wh[2], wl[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      wh[nf] = lds_frag(stage_lane + (ks * PLANES) * 1024 + nf * 512);
      wl[nf] = SPLIT ? lds_frag(stage_lane + (ks * PLANES + 1) * 1024 + nf * 512) : wh[nf];
    }
    // The three product terms are issued term-major over the four accumulators: an accumulator is touched every
    // fourth MFMA, so no MFMA waits for the result of the previous one (back-to-back MFMAs on one accumulator
    // stall on the read-after-write).  (A unit-major software pipeline that prefetches the next (k-step, 16-feature)
    // fragment pair during six MFMAs measured the same: the fragment-read latency is already covered by the
    // partner wave on the SIMD.)
#pragma unroll
    for (int term = SPLIT ? 0 : 2; term < 3; ++term) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const bf16x8 w = term == 0 ? wl[nf] : wh[nf];
          const bf16x8 a = term == 1 ? a_lo[mf][ks] : a_hi[mf][ks];
          acc[nf][mf] = SWAPPED ? mfma16(w, a, acc[nf][mf]) : mfma16(a, w, acc[nf][mf]);
        }
      }
    }
  }
}

template <int KS, int EPI, int PRO, bool SPLIT, int WAVES, int MF = 2>
__global__ __launch_bounds__(WAVES * 64, (MF == 1 && WAVES == 8) ? 4 : 2) void rowgemm_kernel(RowGemmParams p) {
  // A block is WAVES x MF x 16 rows; the library launches 4 waves x 2 fragments = 128 rows, two blocks per CU, and
  // 4 waves x 1 fragment = 64 rows for small batches (fewer than one 128-row block per CU-slot: twice the blocks, so
  // twice the CUs work on a latency-bound request).
  // Measured alternatives on MI355X (xsmall, 256 x 512): 8 waves x 2 (256 rows, one block per CU, half the DMA
  // instructions and L2 -> LDS traffic per row) is within +-2 % on both fused kernels; 8 waves x 1 (16 rows per wave,
  // <= 128 VGPRs, 4 waves per SIMD, twice the fragment reads per MFMA) is equal on q/k/v and 10 % slower on GeGLU.
  static_assert(MF == 1 || MF == 2, "one or two 16-row fragments per wave");
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int K = KS * 32;
  constexpr int CHUNK_SRC = KS * 2 * 1024;        // elements per packed chunk in global memory
  constexpr int STAGE = KS * PLANES * 1024;       // elements per LDS stage
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * (WAVES * 16 * MF) + wave * (16 * MF);

  // ---- weight streaming: global -> LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave-instruction) --
  // Stage layout = [ks][plane][frag][512] = a sequence of 1 KiB pieces; wave w copies pieces w, w+4, ...
  // The copy is linear (the packing kernel already wrote fragment order), so the lane-linear LDS
  // destination the DMA imposes is exactly the layout the fragment reads want.  No staging VGPRs, and
  // the request is in flight while the MFMAs of the current chunk run.
  constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  auto stage_chunk = [&](int chunk, int stage) {
    const u16* src = p.wp + (size_t)chunk * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + WAVES * u;              // wave-uniform
      const int elem = piece * 512;                    // position inside the LDS stage
      const int ks = elem / (PLANES * 1024);
      const int rem = elem % (PLANES * 1024);
      const int src_elem = ks * 2048 + rem;            // source keeps both planes per k-step
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + src_elem + lane * 8),
          (__attribute__((address_space(3))) void*)(&sW[stage][elem]), 16, 0, 0);
    }
  };
  bf16x8 a_hi[MF][KS], a_lo[MF][KS];
  if (PRO == RP_KSTREAM) {
    // ---- fused phase 1: x_new[32 rows, H] = x + A1[32 rows, K1] W1[H, K1]^T, K1 streamed ----------------
    // Same structure as kstream_gemm_kernel (one [H x 32] weight slab per k-step by DMA, A1 fragments straight
    // from the fragment-packed activation, prefetched one k-step ahead), all H outputs of the 32 rows in
    // accumulators.  W1's output features were permuted at load time so that accumulator fragments (2s, 2s+1)
    // are exactly lane slot g of k-step s of THIS kernel's chunk loop: residual add, LayerNorm and the hi/lo
    // split happen in registers and the hidden state makes one fp32 round trip (read + write) per block.
    constexpr int NF1 = 2 * KS;
    constexpr int SLAB_SRC = NF1 * 2 * 512;
    constexpr int SLAB_PIECES = (NF1 * PLANES) / WAVES;
    static_assert((NF1 * PLANES) % WAVES == 0, "slab must split evenly over the waves");
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
        an_hi[mf] = *reinterpret_cast<const bf16x8*>(a_base0 + mf * a_block + (size_t)ks1 * 1024);
        an_lo[mf] = SPLIT ? *reinterpret_cast<const bf16x8*>(a_base0 + mf * a_block + (size_t)ks1 * 1024 + 512) : an_hi[mf];
      }
    };
    f32x4 acc1[NF1][MF];
#pragma unroll
    for (int nf = 0; nf < NF1; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc1[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
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
          wl[j] = SPLIT ? lds_frag(&sW[cur][(NF1 + nf + j) * 512 + lane * 8]) : wh[j];
        }
#pragma unroll
        for (int term = SPLIT ? 0 : 2; term < 3; ++term)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
              acc1[nf + j][mf] = mfma16(term == 0 ? wl[j] : wh[j], term == 1 ? c_lo[mf] : c_hi[mf], acc1[nf + j][mf]);
      }
      __syncthreads();
    };
    for (int k0 = 0; k0 < nks1; k0 += 2) {  // even number of k-steps (checked on the host)
      slab_step(k0, std::integral_constant<int, 0>{});
      slab_step(k0 + 1, std::integral_constant<int, 1>{});
    }
    stage_chunk(0, 0);  // first weight chunk of phase 2 flies while the LayerNorm below runs

    // ---- transition: residual add, store the new hidden state, LayerNorm, split -> fragments --------------
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      float* xrow = p.x_io + (size_t)(m0 + mf * 16 + l15) * K + g * 8;
      float sum = 0.f;
#pragma unroll
      for (int nf = 0; nf < NF1; ++nf) {
        float4* px = reinterpret_cast<float4*>(xrow + 32 * (nf >> 1) + 4 * (nf & 1));
        float4 r4 = *px;
        r4.x += acc1[nf][mf][0];
        r4.y += acc1[nf][mf][1];
        r4.z += acc1[nf][mf][2];
        r4.w += acc1[nf][mf][3];
        *px = r4;
        acc1[nf][mf] = f32x4{r4.x, r4.y, r4.z, r4.w};
        sum += (r4.x + r4.y) + (r4.z + r4.w);
        // keep the scheduler from hoisting all 16 row loads (64 more registers) on top of the accumulators
        if ((nf & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
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
        const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + ks * 32 + g * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(p.ln_w + ks * 32 + g * 8 + 4);
        const float v[8] = {(acc1[2 * ks][mf][0] - mean) * rstd * w0.x,     (acc1[2 * ks][mf][1] - mean) * rstd * w0.y,
                            (acc1[2 * ks][mf][2] - mean) * rstd * w0.z,     (acc1[2 * ks][mf][3] - mean) * rstd * w0.w,
                            (acc1[2 * ks + 1][mf][0] - mean) * rstd * w1.x, (acc1[2 * ks + 1][mf][1] - mean) * rstd * w1.y,
                            (acc1[2 * ks + 1][mf][2] - mean) * rstd * w1.z, (acc1[2 * ks + 1][mf][3] - mean) * rstd * w1.w};
        pack8<SPLIT>(v, a_hi[mf][ks], a_lo[mf][ks]);
      }
    }
  } else {
    stage_chunk(0, 0);
  }

  // ---- prologue: this wave's 32 rows as fragments ---------------------------------------------
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    if (PRO == RP_KSTREAM) break;
    const size_t row = (size_t)(m0 + mf * 16 + l15);
    if (PRO == RP_PLANES) {  // fragment-packed input: piece (row block, k-step, plane), 16 bytes per lane
      const u16* base = p.a_hi + (((size_t)((m0 >> 4) + mf) * KS) * 2) * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a_hi[mf][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)ks * 1024);
        if (SPLIT) a_lo[mf][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)ks * 1024 + 512);
      }
      // Pin the fragment loads in front of the chunk loop: an empty asm that "rewrites" each register makes
      // the compiler wait for the load HERE; otherwise it sinks the loads next to their first MFMA inside the
      // loop and then drains the weight DMA (vmcnt(0)) at the top of every iteration.
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" : "+v"(a_hi[mf][ks]));
        if (SPLIT) asm volatile("" : "+v"(a_lo[mf][ks]));
      }
    } else {
      float v[KS][8];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 f0 = *reinterpret_cast<const float4*>(p.x_in + row * K + ks * 32 + g * 8);
        const float4 f1 = *reinterpret_cast<const float4*>(p.x_in + row * K + ks * 32 + g * 8 + 4);
        v[ks][0] = f0.x; v[ks][1] = f0.y; v[ks][2] = f0.z; v[ks][3] = f0.w;
        v[ks][4] = f1.x; v[ks][5] = f1.y; v[ks][6] = f1.z; v[ks][7] = f1.w;
      }
      if (PRO == RP_LN) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) s += v[ks][e];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)K;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[ks][e] - mean;
            q += d * d;
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q / (float)K + p.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + ks * 32 + g * 8);
          const float4 w1 = *reinterpret_cast<const float4*>(p.ln_w + ks * 32 + g * 8 + 4);
          const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[ks][e] = (v[ks][e] - mean) * rstd * ww[e];
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) pack8<SPLIT>(v[ks], a_hi[mf][ks], a_lo[mf][ks]);
    }
  }

  // RE_QKV: RoPE rows of this lane's tokens: cos/sin [pos][8g + 4j .. +3] for the half-head j of the chunk whose
  // (deferred) epilogue runs in this iteration are fetched at the top of the iteration, before the DMA is issued.
  const float* rope_c_row[MF];
  const float* rope_s_row[MF];
  f32x4 rope_c[MF], rope_s[MF];
  if (EPI == RE_QKV) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int pos = p.row_pos[m0 + mf * 16 + l15];
      pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
      rope_c_row[mf] = p.rope_cos + (size_t)pos * ROPE_HALF + g * 8;
      rope_s_row[mf] = p.rope_sin + (size_t)pos * ROPE_HALF + g * 8;
      rope_c[mf] = rope_s[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();  // chunk 0 has landed (the barrier's release waits for this wave's DMA: vmcnt(0))

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
  auto epilogue = [&](int cc, auto parity_tag, auto sw_tag, const f32x4 (&av)[2][MF]) {
    constexpr int PP = decltype(parity_tag)::value;
    constexpr bool sw = decltype(sw_tag)::value;  // q/k chunk ("swapped" MFMA orientation) or v chunk
    if (EPI == RE_RESIDUAL) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const size_t row = (size_t)(m0 + mf * 16 + l15);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          float4* px = reinterpret_cast<float4*>(p.x + row * p.ld_out + cc * ROW_CHUNK + nf * 16 + g * 4);
          float4 r4 = *px;
          r4.x += av[nf][mf][0];
          r4.y += av[nf][mf][1];
          r4.z += av[nf][mf][2];
          r4.w += av[nf][mf][3];
          *px = r4;
        }
      }
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
        split4<SPLIT>(v, h2, l2);
        if (PP == 0) {
          hold_hi[mf] = h2;
          hold_lo[mf] = l2;
        } else {
          const size_t rb = (size_t)((m0 >> 4) + mf);
          const size_t off = ((rb * (size_t)(p.ld_out >> 5) + (size_t)(cc >> 1)) * 2) * 512 + lane * 8;
          *reinterpret_cast<uint4*>(p.o0_hi + off) = make_uint4(hold_hi[mf].x, hold_hi[mf].y, h2.x, h2.y);
          if (SPLIT) *reinterpret_cast<uint4*>(p.o0_hi + off + 512) = make_uint4(hold_lo[mf].x, hold_lo[mf].y, l2.x, l2.y);
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
            lo_half[r] = (x1 * c4[r] - x2 * s4[r]) * qscale;
            hi_half[r] = (x2 * c4[r] + x1 * s4[r]) * qscale;
          }
          uint2 h0, l0, h1, l1;
          split4<SPLIT>(lo_half, h0, l0);
          split4<SPLIT>(hi_half, h1, l1);
          if (PP == 0) {
            qk_hold[mf][0] = h0; qk_hold[mf][1] = l0; qk_hold[mf][2] = h1; qk_hold[mf][3] = l1;
          } else {
            const size_t rb = (size_t)((m0 >> 4) + mf);
            const size_t kb = (size_t)(cq >> 1) * 2;  // k-step of d in [0, 32); d + 32 is the next one
            const size_t off = ((rb * (size_t)(p.hidden >> 5) + kb) * 2) * 512 + lane * 8;
            *reinterpret_cast<uint4*>(out + off) = make_uint4(qk_hold[mf][0].x, qk_hold[mf][0].y, h0.x, h0.y);
            *reinterpret_cast<uint4*>(out + off + 1024) = make_uint4(qk_hold[mf][2].x, qk_hold[mf][2].y, h1.x, h1.y);
            if (SPLIT) {
              *reinterpret_cast<uint4*>(out + off + 512) = make_uint4(qk_hold[mf][1].x, qk_hold[mf][1].y, l0.x, l0.y);
              *reinterpret_cast<uint4*>(out + off + 1536) = make_uint4(qk_hold[mf][3].x, qk_hold[mf][3].y, l1.x, l1.y);
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
          const size_t off = (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + n) * 512 + lane * 8;
          const float v0[4] = {av[nf][0][0], av[nf][0][1], av[nf][0][2], av[nf][0][3]};
          uint2 h0, l0;
          split4<SPLIT>(v0, h0, l0);
          if (MF == 2) {
            const float v1[4] = {av[nf][MF - 1][0], av[nf][MF - 1][1], av[nf][MF - 1][2], av[nf][MF - 1][3]};
            uint2 h1, l1;
            split4<SPLIT>(v1, h1, l1);
            *reinterpret_cast<uint4*>(p.o2_hi + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            if (SPLIT) *reinterpret_cast<uint4*>(p.o2_hi + off + 2048) = make_uint4(l0.x, l0.y, l1.x, l1.y);
          } else {
            const int half = ((m0 >> 4) & 1) * 4;
            *reinterpret_cast<uint2*>(p.o2_hi + off + half) = h0;
            if (SPLIT) *reinterpret_cast<uint2*>(p.o2_hi + off + 2048 + half) = l0;
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
        rope_c[mf] = *reinterpret_cast<const f32x4*>(rope_c_row[mf] + (cur ^ 1) * 4);
        rope_s[mf] = *reinterpret_cast<const f32x4*>(rope_s_row[mf] + (cur ^ 1) * 4);
      }
    }
    stage_chunk(c + 1 < p.n_chunks ? c + 1 : c, cur ^ 1);
    if (EPI == RE_QKV) __builtin_amdgcn_sched_barrier(0);  // keep those loads up here, ahead of the DMA's wait
    if (!FIRST) epilogue(c - 1, std::integral_constant<int, (cur ^ 1)>{}, swp_tag, acc_prev);

    f32x4 acc[2][MF];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    rowgemm_chunk_mfma<KS, MF, SPLIT, SW>(&sW[cur][lane * 8], a_hi, a_lo, acc);
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
      constexpr int N_MFMA = KS * 2 * MF * (SPLIT ? 3 : 1);
      constexpr int N_DS = KS * 2 * PLANES;
      constexpr int DS_LATE = N_DS - 2 * PLANES;  // reads placed between the MFMAs, spread evenly
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * PLANES, 0);
#pragma unroll
      for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, SPLIT ? 2 : 5, 0);
        if (((i + 1) * DS_LATE) / N_MFMA - (i * DS_LATE) / N_MFMA == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (((i + 1) * DS_LATE) / N_MFMA - (i * DS_LATE) / N_MFMA == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    __syncthreads();
  };
  // Even chunk counts on both sides of the q/k -> v boundary (checked on the host).  The first pair is peeled so
  // that the deferred epilogue is unconditional in the steady-state loops.
  const std::integral_constant<int, 0> even{};
  const std::integral_constant<int, 1> odd{};
  const std::true_type yes{};
  const std::false_type no{};
  const int n_sw = (EPI == RE_QKV) ? p.n_swapped : p.n_chunks;
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
  } else {
    epilogue(p.n_chunks - 1, odd, yes, acc_prev);
  }
}

// ----------------------------------------------------------------------------------------------
// Fragment-packed activations.  An activation matrix [rows x C] that is consumed as the MFMA operand
// of the next GEMM is stored as 1 KiB pieces  [row/16][C/32][plane][lane = 16*(k%32/8) + row%16][8 k]:
// exactly one wave-instruction of 16-byte lanes, in lane order.  Producer epilogues store whole pieces
// (one fully coalesced 1 KiB store per wave), consumers load their fragment with one fully coalesced
// 1 KiB load straight into registers -- no LDS staging, no row-strided 8-byte accesses.
// ----------------------------------------------------------------------------------------------

// dst[ks][plane][nf][g][i][e] <- W[nf*16 + i][ks*32 + g*8 + e]   (W is [N][K]; chunk = one k-step of all N)
__global__ void pack_kstream_kernel(const float* __restrict__ src, int N, int K, int permute, u16* __restrict__ dst) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * K) return;
  const int NF = N / 16;
  size_t t = idx;
  const int e = (int)(t & 7); t >>= 3;
  const int i = (int)(t & 15); t >>= 4;
  const int g = (int)(t & 3); t >>= 2;
  const int nf = (int)(t % NF);
  const int ks = (int)(t / NF);
  // permute: accumulator slot (nf, i = 4g' + r) holds output feature 32(nf>>1) + 8g' + 4(nf&1) + r, so that the
  // accumulators of fragments (2s, 2s+1) ARE the 8 k-values of lane slot g' of k-step s of the next GEMM.
  const int row = permute ? (32 * (nf >> 1) + 8 * (i >> 2) + 4 * (nf & 1) + (i & 3)) : (nf * 16 + i);
  const float v = src[(size_t)row * K + ks * 32 + g * 8 + e];
  const u16 h = f2bf(v);
  const size_t base = ((size_t)ks * 2 * NF + nf) * 512 + (size_t)g * 128 + i * 8 + e;
  dst[base] = h;
  dst[base + (size_t)NF * 512] = f2bf(v - bf2f(h));
}

struct KStreamParams {
  const u16* a_fp;  // fragment-packed activations [r_pad/16][n_ksteps][2 planes][512]
  const u16* wp;    // packed weights [n_ksteps][2 planes][NF][512]
  int n_ksteps;     // K / 32
  float* x;         // fp32 [r_pad][N], x += A W^T
};

// x[128 or 256 rows, N = 16*NF] += A[rows, K] W[N, K]^T with K streamed: per k-step the block DMAs one
// [N x 32] weight slab into LDS (double-buffered) while every wave pulls its own two A fragments straight
// from the fragment-packed activation (prefetched one k-step ahead) and keeps all N outputs of its 32 rows
// in accumulators (NF x 2 x 4 registers).
template <int NF, bool SPLIT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void kstream_gemm_kernel(KStreamParams p) {
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int STAGE = NF * PLANES * 512;        // elements per LDS stage
  constexpr int CHUNK_SRC = NF * 2 * 512;         // elements per k-step in the packed weights
  constexpr int WAVE_PIECES = STAGE / (WAVES * 512);
  static_assert(STAGE % (WAVES * 512) == 0, "stage must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * (WAVES * 32) + wave * 32;
  const int nks = p.n_ksteps;

  auto stage_chunk = [&](int ks, int stage) {
    const u16* src = p.wp + (size_t)ks * CHUNK_SRC;
#pragma unroll
    for (int u = 0; u < WAVE_PIECES; ++u) {
      const int piece = wave + WAVES * u;  // stage = [plane][nf] pieces; source = same order (2 planes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sW[stage][piece * 512]), 16, 0, 0);
    }
  };
  // A fragments of k-step ks: piece (rb, ks, plane) of the fragment-packed activation, 16 bytes per lane
  const u16* a_base0 = p.a_fp + ((size_t)(m0 >> 4) * nks * 2) * 512 + lane * 8;
  const u16* a_base1 = a_base0 + (size_t)nks * 2 * 512;
  bf16x8 an_hi[2], an_lo[2];
  auto load_a = [&](int ks) {
    an_hi[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024);
    an_hi[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024);
    if (SPLIT) {
      an_lo[0] = *reinterpret_cast<const bf16x8*>(a_base0 + (size_t)ks * 1024 + 512);
      an_lo[1] = *reinterpret_cast<const bf16x8*>(a_base1 + (size_t)ks * 1024 + 512);
    }
  };

  f32x4 acc[NF][2];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_chunk(0, 0);
  load_a(0);
  // Retire the first fragment loads HERE (empty asm "rewrites" the registers): a load still pending at the loop
  // header makes the compiler drain everything (vmcnt(0)) right after the loop body has issued its DMA.
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    asm volatile("" : "+v"(an_hi[mf]));
    if (SPLIT) asm volatile("" : "+v"(an_lo[mf]));
  }
  __syncthreads();

  for (int k0 = 0; k0 < nks; k0 += 2) {
#pragma unroll
    for (int cur = 0; cur < 2; ++cur) {
      const int ks = k0 + cur;
      if (ks >= nks) break;
      const int kn = ks + 1 < nks ? ks + 1 : ks;
      stage_chunk(kn, cur ^ 1);
      bf16x8 a_hi[2], a_lo[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        a_hi[mf] = an_hi[mf];
        a_lo[mf] = an_lo[mf];
      }
      load_a(kn);                              // prefetch the next k-step's fragments ...
      __builtin_amdgcn_sched_barrier(0);       // ... and keep the loads up here, ahead of the MFMAs
#pragma unroll
      for (int nf = 0; nf < NF; nf += 2) {  // term-major over 2 fragments x 2 row blocks (see rowgemm_kernel phase 1)
        bf16x8 wh[2], wl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
          wl[j] = SPLIT ? lds_frag(&sW[cur][(NF + nf + j) * 512 + lane * 8]) : wh[j];
        }
#pragma unroll
        for (int term = SPLIT ? 0 : 2; term < 3; ++term)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
              acc[nf + j][mf] = mfma16(term == 0 ? wl[j] : wh[j], term == 1 ? a_lo[mf] : a_hi[mf], acc[nf + j][mf]);
      }
      __syncthreads();
    }
  }

  // x += acc : the weights are packed with permuted output features (pack_kstream_kernel), accumulator slot
  // (nf, g, r) is feature 32(nf>>1) + 8g + 4(nf&1) + r of token m0 + 16mf + l15
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    float* xrow = p.x + (size_t)(m0 + mf * 16 + l15) * (NF * 16) + g * 8;
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
}

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
enum PanelEpi { PE_RESIDUAL = 0, PE_QK = 1, PE_V = 2, PE_GEGLU = 3 };

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

// dst[tile][ks][plane][nf 0..15][lane = 16 g + i][8] <- W[source_row(tile, nf, i)][ks*32 + g*8 + e]
__global__ void pack_panel_kernel(const float* __restrict__ src, int n_tiles, int K, int mode, int H, int I,
                                  u16* __restrict__ dst) {
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
  dst[base] = h;
  dst[base + 8192] = f2bf(v - bf2f(h));
}

struct PanelParams {
  const u16* a_fp;   // fragment-packed activations [r_pad/16][n_ksteps][2 planes][512]
  const u16* wp;     // packed weights [n_tiles][n_ksteps][2 planes][16][512]
  int n_ksteps;      // K / 32
  int r_pad;
  int hidden;        // H
  int ld_out;        // PE_RESIDUAL: H   PE_GEGLU: I
  float* x;          // PE_RESIDUAL
  u16* o0;           // PE_QK: q   PE_V: v^T   PE_GEGLU: h   (fragment-packed, hi/lo planes interleaved per piece)
  u16* o1;           // PE_QK: k
  const int32_t* row_pos;
  const float* rope_cos;
  const float* rope_sin;
  int max_pos;
};

template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void panel_gemm_kernel(PanelParams p) {
  constexpr int NF = 16;
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int STAGE = NF * PLANES * 512;   // elements per LDS stage
  constexpr int SLAB_SRC = NF * 2 * 512;     // elements per k-step in the packed weights (both planes)
  constexpr int WAVE_PIECES = (NF * PLANES) / 4;
  constexpr bool SWAPPED = (EPI != PE_V);    // weights as the MFMA row operand, except for v^T
  __shared__ __attribute__((aligned(16))) u16 sW[2][STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * ROW_BM + wave * 32;
  const int tile = blockIdx.y;
  const int nks = p.n_ksteps;
  const u16* wtile = p.wp + (size_t)tile * nks * SLAB_SRC;

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
      an_lo[mf] = SPLIT ? *reinterpret_cast<const bf16x8*>(a_base + mf * a_block + (size_t)ks * 1024 + 512) : an_hi[mf];
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
#pragma unroll
    for (int nf = 0; nf < NF; nf += 2) {  // term-major over 2 fragments x 2 row blocks: no dependent MFMA pairs
      bf16x8 wh[2], wl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        wh[j] = lds_frag(&sW[cur][(nf + j) * 512 + lane * 8]);
        wl[j] = SPLIT ? lds_frag(&sW[cur][(NF + nf + j) * 512 + lane * 8]) : wh[j];
      }
#pragma unroll
      for (int term = SPLIT ? 0 : 2; term < 3; ++term)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf) {
            const bf16x8 w = term == 0 ? wl[j] : wh[j];
            const bf16x8 a = term == 1 ? a_lo[mf] : a_hi[mf];
            acc[nf + j][mf] = SWAPPED ? mfma16(w, a, acc[nf + j][mf]) : mfma16(a, w, acc[nf + j][mf]);
          }
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
        float4 r4 = *px;
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
        pack8<SPLIT>(v, hi, lo);
        u16* dst = p.o0 + ((rb * kb_out + (size_t)(tile * 4 + s)) * 2) * 512 + lane * 8;
        *reinterpret_cast<bf16x8*>(dst) = hi;
        if (SPLIT) *reinterpret_cast<bf16x8*>(dst + 512) = lo;
      }
    }
  } else if (EPI == PE_QK) {
    const int per = p.hidden / 256;
    const bool is_q = tile < per;
    const int tq = is_q ? tile : tile - per;
    u16* out = is_q ? p.o0 : p.o1;
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
            lo_half[4 * u + r] = (x1 * c4[u][r] - x2 * s4[u][r]) * qscale;
            hi_half[4 * u + r] = (x2 * c4[u][r] + x1 * s4[u][r]) * qscale;
          }
        bf16x8 h0, l0, h1, l1;
        pack8<SPLIT>(lo_half, h0, l0);
        pack8<SPLIT>(hi_half, h1, l1);
        u16* dst = out + ((rb * kb_out + (size_t)((tq * 4 + hh) * 2)) * 2) * 512 + lane * 8;
        *reinterpret_cast<bf16x8*>(dst) = h0;
        *reinterpret_cast<bf16x8*>(dst + 1024) = h1;
        if (SPLIT) {
          *reinterpret_cast<bf16x8*>(dst + 512) = l0;
          *reinterpret_cast<bf16x8*>(dst + 1536) = l1;
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
      pack8<SPLIT>(v, hi, lo);
      u16* dst = p.o0 + (((head * (size_t)(p.r_pad >> 5) + tb) * 2) * 4 + (size_t)(nf & 3)) * 512 + lane * 8;
      *reinterpret_cast<bf16x8*>(dst) = hi;
      if (SPLIT) *reinterpret_cast<bf16x8*>(dst + 2048) = lo;
    }
  }
}

// LayerNorm (or, with normalize = 0, the plain hi/lo split that layer 0 needs: its attn_norm is Identity) of the fp32
// residual stream into fragment-packed planes.  One block = one 16-row block.  Reading wants a wave per row (fully
// coalesced float4 loads, the row_layer_norm helpers above), writing wants lane = (k-group, row) of a 1 KiB piece: the
// four waves normalise four rows each, keep them in registers, and transpose through LDS one plane at a time --
// slabs of 16 lanes x 16 B per (k-step, k-group), padded by 16 B so that the 8-byte staging writes of a wave (one
// row, all k) spread over all banks; the pieces then leave as whole coalesced 1 KiB stores.
constexpr int LN_FP_SLAB = 16 * 16 + 16;  // bytes per (k-step, k-group) slab in the staging buffer

template <bool SPLIT>
__global__ __launch_bounds__(256) void ln_fp_kernel(const float* __restrict__ x, const float* __restrict__ lnw, float eps,
                                                    int H, int r_pad, int normalize, u16* __restrict__ out_fp) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[LN_MAX_CHUNKS * 8 * 4 * LN_FP_SLAB];  // H <= 1024
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rb = blockIdx.x;
  const int KS = H >> 5;
  const int nchunk = H >> 2;
  RowVec rows[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    row_load(x + (size_t)(rb * 16 + wave * 4 + j) * H, H, lane, rows[j]);
    if (normalize) row_layer_norm(rows[j], lnw, H, lane, eps);
  }
  uint2 hi[4][LN_MAX_CHUNKS], lo[4][LN_MAX_CHUNKS];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
      const float v[4] = {rows[j].v[k].x, rows[j].v[k].y, rows[j].v[k].z, rows[j].v[k].w};
      split4<SPLIT>(v, hi[j][k], lo[j][k]);
    }
#pragma unroll
  for (int plane = 0; plane < (SPLIT ? 2 : 1); ++plane) {
    if (plane) __syncthreads();  // the previous plane has been read out
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
        const int c = lane + 64 * k;  // float4 chunk = columns 4c .. 4c+3 = k-step c/8, k-group (c%8)/2, half c%2
        if (c < nchunk) {
          const int slab = (c >> 3) * 4 + ((c & 7) >> 1);
          *reinterpret_cast<uint2*>(stage + slab * LN_FP_SLAB + (wave * 4 + j) * 16 + (c & 1) * 8) = plane ? lo[j][k] : hi[j][k];
        }
      }
    __syncthreads();
    for (int ks = wave; ks < KS; ks += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(stage + (ks * 4 + (lane >> 4)) * LN_FP_SLAB + (lane & 15) * 16);
      *reinterpret_cast<uint4*>(out_fp + (((size_t)rb * KS + ks) * 2 + plane) * 512 + lane * 8) = v;
    }
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
      const bf16x8 pl = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
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


// ----------------------------------------------------------------------------------------------
// Attention on fragment-packed q / k / v^T (row-stationary path).  Same algorithm as attn_kernel above
// (transposed scores, lane-local online softmax, O^T = V^T P^T) but every byte moves as 1 KiB pieces:
//   * Q fragments: one coalesced 16-byte-per-lane load per (k-step, plane), straight into registers;
//   * K / V^T tiles: 32 pieces per 64-key tile (16 KiB hi + 16 KiB lo) copied global -> LDS by DMA
//     (global_load_lds), double-buffered, fragment reads lane-linear and conflict-free;
//   * output: the v^T piece order was chosen at QKV time so that a lane ends up with 8 consecutive head
//     dims -> two 16-byte stores per plane per lane, whole 1 KiB pieces of the fragment-packed o.
// Sequences start at multiples of 32 rows, so key tiles coincide with whole pieces.
// ----------------------------------------------------------------------------------------------
struct AttnFpParams {
  const u16* q_fp;   // [rows/16][H/32][2][512]
  const u16* k_fp;
  const u16* vt_fp;  // [heads][rows/32][2][4][512]
  u16* o_fp;         // [rows/16][H/32][2][512]
  const int32_t* cu;
  int s0;
  const int32_t* roff;
  const int32_t* qboff;  // first work item (query block) of each sequence, [ns + 1]: the grid has no empty blocks
  int ns;
  int H;
  int r_pad;
  int window;
};

// queries per block of the fragment-packed attention kernel = WAVES x 32 (4 waves: two blocks per CU; 8 waves: one
// block per CU, every K / V^T tile staged once for 256 queries -> half the DMA instructions and L2 traffic)

// Scores arrive pre-multiplied by log2(e) (folded into the q scale by the QKV epilogue), so the softmax uses
// exp2 directly: p = 2^(s - max).
template <bool SPLIT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void attn_fp_kernel(AttnFpParams p) {
  constexpr int ATT_FP_BQ = WAVES * 32;
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int K_PIECES = 8 * PLANES;              // [m 0..3][ks 0..1][plane]
  constexpr int V_PIECES = 8 * PLANES;              // [t 0..1][plane][n 0..3]
  constexpr int STAGE = (K_PIECES + V_PIECES) * 512;
  __shared__ __attribute__((aligned(16))) u16 sT[2][STAGE];

  // work item -> (sequence, query block): binary search in the per-sequence prefix of ceil(len / ATT_FP_BQ)
  int s = 0;
  {
    const int item = blockIdx.x;
    int lo_s = 0, hi_s = p.ns - 1;
    while (lo_s < hi_s) {
      const int mid = (lo_s + hi_s + 1) >> 1;
      if (p.qboff[mid] <= item) lo_s = mid; else hi_s = mid - 1;
    }
    s = lo_s;
  }
  const int head = blockIdx.y;
  const int q0 = ((int)blockIdx.x - p.qboff[s]) * ATT_FP_BQ;
  const int len = p.cu[p.s0 + s + 1] - p.cu[p.s0 + s];
  if (q0 >= len) return;
  const int r0 = p.roff[s];
  const int alloc = p.roff[s + 1] - r0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;
  const int kbn = p.H >> 5;  // k-steps per row block

  const int qbase = q0 + wave * 32;   // this wave's 32 queries = two fragments
  const bool active = qbase < alloc;  // alloc is a multiple of 32: both fragments in or out
  const size_t q_rb = (size_t)((r0 + (active ? qbase : q0)) >> 4);

  bf16x8 qf_hi[2][2], qf_lo[2][2];  // [query fragment][k-step]
#pragma unroll
  for (int qf = 0; qf < 2; ++qf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u16* src = p.q_fp + (((q_rb + qf) * kbn + head * 2 + ks) * 2) * 512 + lane * 8;
      qf_hi[qf][ks] = *reinterpret_cast<const bf16x8*>(src);
      qf_lo[qf][ks] = SPLIT ? *reinterpret_cast<const bf16x8*>(src + 512) : qf_hi[qf][ks];
    }
#pragma unroll
  for (int qf = 0; qf < 2; ++qf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // retire the loads before the tile loop (see rowgemm_kernel)
      asm volatile("" : "+v"(qf_hi[qf][ks]));
      asm volatile("" : "+v"(qf_lo[qf][ks]));
    }

  const int win = p.window >= 0 ? p.window : (1 << 30);
  int kt_lo = 0, kt_hi = (len - 1) / ATT_BK;
  if (p.window >= 0) {
    const int lo_key = q0 - p.window;
    kt_lo = lo_key > 0 ? lo_key / ATT_BK : 0;
    const int hi_t = (q0 + ATT_FP_BQ - 1 + p.window) / ATT_BK;
    kt_hi = hi_t < kt_hi ? hi_t : kt_hi;
  }

  // DMA one key tile: wave w copies pieces w, w + WAVES, ... of the stage [K: (m, ks, plane) | V^T: (t, plane, n)].
  // Everything about a piece except the tile's row block is wave-constant and computed once, in 32-bit element
  // offsets (the buffers are < 2^31 elements): per tile a piece costs one clamp, one multiply and one add.
  const int k_rb_max = (p.r_pad >> 4) - 1, v_tb_max = (p.r_pad >> 5) - 1;
  const int k_stride = kbn * 1024;  // elements per 16-row block of k
  constexpr int KPW = K_PIECES / WAVES, VPW = V_PIECES / WAVES;
  int k_m[KPW], k_off[KPW], v_t[VPW], v_off[VPW];
#pragma unroll
  for (int u = 0; u < KPW; ++u) {
    const int kp = wave + WAVES * u;
    const int rem = kp % (2 * PLANES);
    k_m[u] = kp / (2 * PLANES);
    k_off[u] = ((head * 2 + rem / PLANES) * 2 + rem % PLANES) * 512;
  }
#pragma unroll
  for (int u = 0; u < VPW; ++u) {
    const int vp = wave + WAVES * u;
    const int rem = vp % (4 * PLANES);
    v_t[u] = vp / (4 * PLANES);
    v_off[u] = head * (p.r_pad >> 5) * 4096 + (rem / 4) * 2048 + (rem % 4) * 512;
  }
  auto stage_tile = [&](int kt, int stage) {
    // A tile may reach past the last computed row (the last sequence need not fill its final 64-key tile): such
    // pieces are clamped onto the last valid one -- their keys are masked, they only have to be finite.
    const int k_rb0 = (r0 + kt * ATT_BK) >> 4;
    const int v_tb0 = (r0 + kt * ATT_BK) >> 5;
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
      int rb = k_rb0 + k_m[u];
      rb = rb < k_rb_max ? rb : k_rb_max;
      const u16* src = p.k_fp + (unsigned)(rb * k_stride + k_off[u]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 8),
                                       (__attribute__((address_space(3))) void*)(&sT[stage][(wave + WAVES * u) * 512]), 16,
                                       0, 0);
    }
#pragma unroll
    for (int u = 0; u < VPW; ++u) {
      int tb = v_tb0 + v_t[u];
      tb = tb < v_tb_max ? tb : v_tb_max;
      const u16* src = p.vt_fp + (unsigned)(tb * 4096 + v_off[u]);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + lane * 8),
          (__attribute__((address_space(3))) void*)(&sT[stage][(K_PIECES + wave + WAVES * u) * 512]), 16, 0, 0);
    }
  };

  float m_run[2] = {-1e30f, -1e30f};
  float l_run[2] = {0.f, 0.f};
  f32x4 oacc[4][2];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) oacc[n][qf] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile = [&](int kt, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    stage_tile(kt + 1 <= kt_hi ? kt + 1 : kt, cur ^ 1);  // unconditional prefetch into the idle stage
    const u16* st = &sT[cur][lane * 8];
    const int kbase = kt * ATT_BK;
    // wave-uniform tile classification: skip tiles entirely outside this wave's window (other waves of the block
    // may need them), and use the mask-free path when every (query, key) pair of the tile is visible
    const bool outside = (kbase > qbase + 31 + win) || (kbase + ATT_BK - 1 < qbase - win);
    const bool all_valid = (kbase + ATT_BK - 1 < len) && (kbase + ATT_BK - 1 - qbase <= win) && (qbase + 31 - kbase <= win);
    if (!outside) {
      f32x4 sacc[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) sacc[m][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const bf16x8 kh = lds_frag(st + ((m * 2 + ks) * PLANES) * 512);
          const bf16x8 kl = SPLIT ? lds_frag(st + ((m * 2 + ks) * PLANES + 1) * 512) : kh;
#pragma unroll
          for (int qf = 0; qf < 2; ++qf) {
            if (SPLIT) {
              sacc[m][qf] = mfma16(kl, qf_hi[qf][ks], sacc[m][qf]);
              sacc[m][qf] = mfma16(kh, qf_lo[qf][ks], sacc[m][qf]);
            }
            sacc[m][qf] = mfma16(kh, qf_hi[qf][ks], sacc[m][qf]);
          }
        }
      }

      if (!all_valid) {
        // masked scores become -3e30 (below the running-max initial value -1e30): 2^(masked - max) underflows to
        // exactly 0 even when the whole tile is masked for a query.  A query sees the keys [lo, hi] =
        // [max(q - win, 0), min(q + win, len - 1)]; element (m, r) of this lane is key kbase + 4g + (16m + r), so
        // with lo_rel = lo - kbase - 4g the test is one unsigned compare of the constant (16m + r) - lo_rel
        // against hi - lo (an empty interval is moved out of reach).
#pragma unroll
        for (int qf = 0; qf < 2; ++qf) {
          const int qpos = qbase + 16 * qf + l15;
          int lo = qpos - win, hi = qpos + win;
          lo = lo > 0 ? lo : 0;
          hi = hi < len - 1 ? hi : len - 1;
          const int span = hi - lo;
          const int lo_rel = span >= 0 ? lo - kbase - 4 * g : (1 << 29);
          const unsigned uspan = span >= 0 ? (unsigned)span : 0u;
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = (unsigned)(16 * m + r - lo_rel) <= uspan;
              sacc[m][qf][r] = ok ? sacc[m][qf][r] : -3e30f;
            }
        }
      }

      bf16x8 ph[2][2], pl[2][2];  // [k-step t][query fragment]
#pragma unroll
      for (int qf = 0; qf < 2; ++qf) {
        float tile_max = -3e30f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile_max = fmaxf(tile_max, sacc[m][qf][r]);
        tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 16, 64));
        tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 32, 64));
        // Lazy reference: the exponent reference m_run only moves when this tile's maximum exceeds it by more
        // than 2^6 (p stays <= 64, harmless in fp32 and in the hi/lo split), and the 16 accumulator rescales run
        // only in the tiles where some query of the wave moved -- a wave-uniform branch, rare after the first tile.
        const bool moved = tile_max > m_run[qf] + 6.0f;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
          const float m_new = moved ? tile_max : m_run[qf];
          const float alpha = __builtin_amdgcn_exp2f(m_run[qf] - m_new);
          m_run[qf] = m_new;
          l_run[qf] *= alpha;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            oacc[n][qf][0] *= alpha;
            oacc[n][qf][1] *= alpha;
            oacc[n][qf][2] *= alpha;
            oacc[n][qf][3] *= alpha;
          }
        }
        const float m_ref = m_run[qf];
        float psum = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // raw v_exp_f32: the argument is <= 6 and a result below 2^-126 may flush to zero
            const float e = __builtin_amdgcn_exp2f(sacc[m][qf][r] - m_ref);
            sacc[m][qf][r] = e;
            psum += e;
          }
        l_run[qf] += psum;
        // lane slot e < 4 -> key 32t + 4g + e (piece m = 2t), e >= 4 -> key 32t + 16 + 4g + (e-4) (piece 2t+1):
        // the order the QKV epilogue stored v^T in
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float v0[4] = {sacc[2 * t][qf][0], sacc[2 * t][qf][1], sacc[2 * t][qf][2], sacc[2 * t][qf][3]};
          const float v1[4] = {sacc[2 * t + 1][qf][0], sacc[2 * t + 1][qf][1], sacc[2 * t + 1][qf][2], sacc[2 * t + 1][qf][3]};
          uint2 h0, l0, h1, l1;
          split4<SPLIT>(v0, h0, l0);
          split4<SPLIT>(v1, h1, l1);
          ph[t][qf] = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
          pl[t][qf] = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
        }
      }

      // O^T += V^T P^T
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const bf16x8 vh = lds_frag(st + (K_PIECES + (t * PLANES) * 4 + n) * 512);
          const bf16x8 vl = SPLIT ? lds_frag(st + (K_PIECES + (t * PLANES + 1) * 4 + n) * 512) : vh;
#pragma unroll
          for (int qf = 0; qf < 2; ++qf) {
            if (SPLIT) {
              oacc[n][qf] = mfma16(vl, ph[t][qf], oacc[n][qf]);
              oacc[n][qf] = mfma16(vh, pl[t][qf], oacc[n][qf]);
            }
            oacc[n][qf] = mfma16(vh, ph[t][qf], oacc[n][qf]);
          }
        }
      }
    }
    __syncthreads();
  };

  stage_tile(kt_lo, 0);
  __syncthreads();
  for (int kt = kt_lo; kt <= kt_hi; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 <= kt_hi) tile(kt + 1, std::integral_constant<int, 1>{});
  }

  if (active) {
#pragma unroll
    for (int qf = 0; qf < 2; ++qf) {
      float l_tot = l_run[qf] + __shfl_xor(l_run[qf], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      // oacc[n][qf][r]: d = 32(n>>1) + 8g + 4(n&1) + r  ->  lane owns d = 8g..8g+7 of k-step (n>>1) of this head
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float v0[4] = {oacc[2 * half][qf][0] * inv, oacc[2 * half][qf][1] * inv, oacc[2 * half][qf][2] * inv,
                             oacc[2 * half][qf][3] * inv};
        const float v1[4] = {oacc[2 * half + 1][qf][0] * inv, oacc[2 * half + 1][qf][1] * inv,
                             oacc[2 * half + 1][qf][2] * inv, oacc[2 * half + 1][qf][3] * inv};
        uint2 h0, l0, h1, l1;
        split4<SPLIT>(v0, h0, l0);
        split4<SPLIT>(v1, h1, l1);
        u16* dst = p.o_fp + (((q_rb + qf) * kbn + head * 2 + half) * 2) * 512 + lane * 8;
        *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        if (SPLIT) *reinterpret_cast<uint4*>(dst + 512) = make_uint4(l0.x, l0.y, l1.x, l1.y);
      }
    }
  }
}

}  // namespace opk
