// opk_small.hip.h -- row map, LayerNorm family, heads, weight re-packing: the non-GEMM kernels (included by op_api.hip only)
#pragma once

#include "opk_common.hip.h"

namespace opk {

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
// pre_norm: the pruning head reads the row BEFORE final_norm (hidden_states[-1] of transformers 4.x, see
// op_config.prune_pre_final_norm).  keep_prob (optional) = softmax(logits)[1] = sigmoid(l1 - l0)
// (standalone.py:2918-2924 computes it on the host after the D2H copy).
__global__ __launch_bounds__(256) void final_ln_prune_kernel(float* __restrict__ x, const float* __restrict__ lnw,
                                                             float eps, int H, int r_pad,
                                                             const int32_t* __restrict__ row_tok,
                                                             const int32_t* __restrict__ row_seq,
                                                             const int32_t* __restrict__ row_pos,
                                                             const float* __restrict__ pw, const float* __restrict__ pb,
                                                             float* __restrict__ prune_out, float* __restrict__ keep_prob,
                                                             int pre_norm, int keep_all_rows, float* __restrict__ cls,
                                                             float* __restrict__ capture, const int* __restrict__ range_flag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= r_pad) return;
  const int tok = row_tok[row];
  if (tok < 0) return;
  RowVec rv;
  row_load(x + (size_t)row * H, H, lane, rv);
  const int nchunk = H >> 2;
  float d0 = 0.f, d1 = 0.f;
  auto head_dot = [&]() {
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
  };
  if (pre_norm) head_dot();
  row_layer_norm(rv, lnw, H, lane, eps);
  if (!pre_norm) head_dot();
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  if (lane == 0) {
    // range_flag: see rank_head_kernel -- the pruning logits of a flagged chunk leave as NaN too
    const float poison = (range_flag != nullptr && *range_flag != 0) ? __builtin_nanf("") : 0.f;
    const float l0 = d0 + pb[0] + poison, l1 = d1 + pb[1] + poison;
    prune_out[(size_t)tok * 2 + 0] = l0;
    prune_out[(size_t)tok * 2 + 1] = l1;
    if (keep_prob) keep_prob[tok] = 1.0f / (1.0f + expf(l0 - l1));
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
                                                        float* __restrict__ rank_out, const int* __restrict__ range_flag) {
  // range_flag (may be NULL): raised by a kernel of the fp16 + e4m3 format that met an activation beyond fp16's range (or a
  // non-finite one) where the format cannot say so itself -- under MODE.FP16_OVFL = 1 conversions clamp and the fp16 MFMA
  // takes a NaN operand for a finite number (microbench/mode_probe.hip).  The ranking logits of the chunk then leave as NaN:
  // the same signal as an Inf that travelled through the residual stream, which the callers' range guard acts on.
  const bool poisoned = range_flag != nullptr && *range_flag != 0;
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
    if (tid == 0) rank_out[(size_t)(s0 + s) * nl + c] = poisoned ? __builtin_nanf("") : tot + cls_b[c];
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
// Every weight-splitting kernel reports whether the tensor has ANY non-zero lo element (`any_lo`, set to 1; a bf16
// checkpoint has none, and then the hi x lo(weight) product term can be dropped without changing a bit), and
// `zero_lo` stores zeros in the lo plane (a precision policy without that term, run on a kernel that has it).
__global__ void split_planes_kernel(const float* __restrict__ src, int rows, int cols, int geglu_half,
                                    u16* __restrict__ hi, u16* __restrict__ lo, int zero_lo, int* __restrict__ any_lo) {
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
  const u16 l = f2bf(v - bf2f(h));
  if ((l & 0x7fffu) != 0) *any_lo = 1;
  hi[i] = h;
  lo[i] = zero_lo ? (u16)0 : l;
}

__global__ void transpose_f32_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  dst[(size_t)c * rows + r] = src[i];
}

// LayerNorm (or, with normalize = 0, the plain hi/lo split that layer 0 needs: its attn_norm is Identity) of the fp32
// residual stream into fragment-packed planes.  One block = one 16-row block.  Reading wants a wave per row (fully
// coalesced float4 loads, the row_layer_norm helpers above), writing wants lane = (k-group, row) of a 1 KiB piece: the
// four waves normalise four rows each, keep them in registers, and transpose through LDS one plane at a time --
// slabs of 16 lanes x 16 B per (k-step, k-group), padded by 16 B so that the 8-byte staging writes of a wave (one
// row, all k) spread over all banks; the pieces then leave as whole coalesced 1 KiB stores.
constexpr int LN_FP_SLAB = 16 * 16 + 16;  // bytes per (k-step, k-group) slab in the staging buffer

// H16 (kernel set "f16"): the single plane holds fp16 values
template <bool SPLIT, bool H16 = false>
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
      split4x<SPLIT, H16>(v, hi[j][k], lo[j][k]);
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

// The same for the "f16 + fp8" kernel sets (panel path): out16 = fp16 pieces [rb][ks][512], out8 = e4m3 pieces
// [rb][ks / 2][1 KiB] with a lane's 16 bytes = its 8 lo values (x 2^12) of the even k-step, then of the odd one
// (opk_common.hip.h).  H % 64 == 0.
__global__ __launch_bounds__(256) void ln_fp8_kernel(const float* __restrict__ x, const float* __restrict__ lnw, float eps, int H,
                                                     int r_pad, int normalize, u16* __restrict__ out16, u16* __restrict__ out8) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[LN_MAX_CHUNKS * 8 * 4 * LN_FP_SLAB];  // H <= 1024
  __shared__ __attribute__((aligned(16))) unsigned char stage8[LN_MAX_CHUNKS * 8 * 4 * 16 * 8];      // [ks][g][row][8 bytes]
  set_saturating_conversions();
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
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < LN_MAX_CHUNKS; ++k) {
      const int c = lane + 64 * k;  // float4 chunk = columns 4c .. 4c+3 = k-step c/8, k-group (c%8)/2, half c%2
      if (c < nchunk) {
        const float v[4] = {rows[j].v[k].x, rows[j].v[k].y, rows[j].v[k].z, rows[j].v[k].w};
        uint2 hi;
        uint32_t lo8;
        split4_f8(v, hi, lo8);
        const int slab = (c >> 3) * 4 + ((c & 7) >> 1);
        *reinterpret_cast<uint2*>(stage + slab * LN_FP_SLAB + (wave * 4 + j) * 16 + (c & 1) * 8) = hi;
        *reinterpret_cast<uint32_t*>(stage8 + (slab * 16 + (wave * 4 + j)) * 8 + (c & 1) * 4) = lo8;
      }
    }
  __syncthreads();
  for (int ks = wave; ks < KS; ks += 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(stage + (ks * 4 + (lane >> 4)) * LN_FP_SLAB + (lane & 15) * 16);
    *reinterpret_cast<uint4*>(out16 + ((size_t)rb * KS + ks) * 512 + lane * 8) = v;
  }
  for (int pk = wave; pk < (KS >> 1); pk += 4) {
    const uint2 e = *reinterpret_cast<const uint2*>(stage8 + (((2 * pk) * 4 + (lane >> 4)) * 16 + (lane & 15)) * 8);
    const uint2 o = *reinterpret_cast<const uint2*>(stage8 + (((2 * pk + 1) * 4 + (lane >> 4)) * 16 + (lane & 15)) * 8);
    *reinterpret_cast<uint4*>(out8 + ((size_t)rb * (KS >> 1) + pk) * 512 + lane * 8) = make_uint4(e.x, e.y, o.x, o.y);
  }
}

// Numerics of a narrower precision policy on a wider kernel instantiation: clear the lo plane of a fragment-packed
// tensor (every odd unit of `unit` elements) so that the wider kernel's extra product term adds exact zeros.
__global__ __launch_bounds__(256) void zero_odd_units_kernel(u16* __restrict__ base, int unit, size_t n_pairs) {
  const size_t per_unit = (size_t)unit / 8;  // 16-byte stores
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs * per_unit) return;
  const size_t pair = i / per_unit, off = i % per_unit;
  reinterpret_cast<uint4*>(base + (2 * pair + 1) * unit)[off] = make_uint4(0u, 0u, 0u, 0u);
}


// ----------------------------------------------------------------------------------------------
// Mean keep-probability of token ranges (one thread per range), bit for bit what the reference computes on the host
// with numpy: float(block_probs[start:end].mean()) (standalone.py:3075-3082) = float32 PAIRWISE sum in numpy's order
// (umath/loops_utils.h.src, @TYPE@_pairwise_sum: < 8 elements sequential; <= 128 elements eight running sums over
// blocks of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the remainder sequentially; longer ranges split
// at n/2 rounded down to a multiple of 8), divided by the count in float64 and rounded back to float32.  An empty range
// scores 1.0 (ref :3081).  process() then copies 4 bytes per FRAGMENT back instead of 4 per token and runs no numpy
// reduction per fragment.  Explicit __fadd_rn: nothing here may be contracted or reassociated.
// ----------------------------------------------------------------------------------------------
__device__ inline float np_pairwise_leaf_f32(const float* a, int n) {  // n <= 128
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = __fadd_rn(res, a[i]);
    return res;
  }
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
  }
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}
// the recursion pairwise(a, n) = pairwise(a, n2) + pairwise(a + n2, n - n2) on an explicit stack (depth <= 24 covers any int)
__device__ inline float np_pairwise_sum_f32(const float* a, int n) {
  const float* fa[24];
  int fn[24], fstage[24];
  float fleft[24];
  int sp = 0;
  fa[0] = a; fn[0] = n; fstage[0] = 0; fleft[0] = 0.f;
  float ret = 0.f;
  while (sp >= 0) {
    if (fn[sp] <= 128) {
      ret = np_pairwise_leaf_f32(fa[sp], fn[sp]);
      --sp;
      continue;
    }
    int n2 = fn[sp] / 2;
    n2 -= n2 % 8;
    if (fstage[sp] == 0) {
      fstage[sp] = 1;
      fa[sp + 1] = fa[sp]; fn[sp + 1] = n2; fstage[sp + 1] = 0;
      ++sp;
    } else if (fstage[sp] == 1) {
      fleft[sp] = ret;
      fstage[sp] = 2;
      fa[sp + 1] = fa[sp] + n2; fn[sp + 1] = fn[sp] - n2; fstage[sp + 1] = 0;
      ++sp;
    } else {
      ret = __fadd_rn(fleft[sp], ret);
      --sp;
    }
  }
  return ret;
}

// "f16 + fp8" packs, per weight tensor (note_f16_fit in opk_common.hip.h): the tensor's energy (x 2^60, as the lost part),
// and the verdict once its pack kernels have run
__global__ __launch_bounds__(256) void weight_energy_kernel(const float* __restrict__ w, size_t n, float* __restrict__ fit) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += w[i] * w[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(fit + 1, acc * 1.1529215e18f);
}
__global__ void f16_fit_close_tensor_kernel(int* __restrict__ flags, float* __restrict__ fit) {
  if (fit[0] > fit[1] * 1.4551915e-11f) flags[2] = 1;  // 2^-36
  fit[0] = 0.f;
  fit[1] = 0.f;
}

// one wave: shader cycles and 100 MHz ticks over a spin of `ticks` ticks (op_debug_clock_probe)
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* __restrict__ out) {
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    r1 = wall_clock64();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
  }
}

__global__ void segment_mean_kernel(const float* __restrict__ values, const int32_t* __restrict__ seg, int n_seg,
                                    int n_values, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seg) return;
  int start = seg[2 * i], end = seg[2 * i + 1];
  start = start < 0 ? 0 : start;
  end = end > n_values ? n_values : end;
  if (end <= start) {
    out[i] = 1.0f;
    return;
  }
  const float sum = np_pairwise_sum_f32(values + start, end - start);
  out[i] = (float)((double)sum / (double)(end - start));
}

}  // namespace opk
