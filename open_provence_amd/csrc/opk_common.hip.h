// opk_common.hip.h -- CDNA4 (gfx950) device code of the OpenProvence forward path: types, constants and helpers
// shared by every kernel family (opk_small / opk_tiled / opk_rowgemm / opk_panel / opk_attn .hip.h).
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

// Packed fp32 arithmetic (two lanes of a 64-bit register pair per instruction).  Measured with one wave per SIMD
// (microbench/mfma32_probe.hip): in a vector-only phase a v_pk_fma_f32 issues at the rate of a v_fma_f32 (4.9 cycles,
// twice the arithmetic); beside an MFMA stream it costs ~4x a scalar FMA.  So: explicit, and only in the LayerNorm
// phases of the whole-layer kernel.  Written as asm so that neither the contraction nor the vectorizer settings of an
// instantiation decide which instructions a row's arithmetic is made of.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// (hi, lo) split of a pair with the remainder computed by one packed subtraction
template <bool SPLIT>
__device__ __forceinline__ void split2_pk(f32x2 v, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(v.x, v.y);
  if (SPLIT) {
    const f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const f32x2 r = pk_sub(v, hf);
    lo = pack_bf16x2(r.x, r.y);
  } else {
    lo = 0u;
  }
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
__device__ __forceinline__ uint4 as_u4(bf16x8 v) {
  FragU f;
  f.v = v;
  return f.u;
}
// Load the fragment as an ext-vector type: an LDS load typed as HIP's uint4 class makes hipcc (ROCm 7.2) treat it
// as possibly aliasing an in-flight global_load_lds DMA and drain the DMA (s_waitcnt vmcnt(0)) in front of it.
__device__ __forceinline__ bf16x8 lds_frag(const u16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// D = X * Y + C on one wave.  X fragment: row (lane & 15), k-group (lane >> 4) holds 8 consecutive k.
// Y fragment: column (lane & 15), same k-group.  D: column (lane & 15), rows 4*(lane >> 4) + r.
__device__ __forceinline__ f32x4 mfma16(bf16x8 x, bf16x8 y, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}

// RoPE pair (HF :188-219: x cos + rotate_half(x) sin) with the fused multiply-adds written out: left to -ffp-contract
// the compiler picks which product of `x1 c - x2 s` it fuses per instantiation, and the 16-row and 32-row variants of
// one kernel then differ in the last bit of q / k -- results must not depend on the batch size that selects them.
__device__ __forceinline__ float rope_lo(float x1, float x2, float c, float s) { return __fmaf_rn(x1, c, -__fmul_rn(x2, s)); }
__device__ __forceinline__ float rope_hi(float x1, float x2, float c, float s) { return __fmaf_rn(x2, c, __fmul_rn(x1, s)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ----------------------------------------------------------------------------------------------
// Kernel set "f16 + fp8" (round 3).  The (hi, lo) bf16 pair costs two 16x16x32 MFMAs per algorithmic product.  Here
//   hi = RNE_fp16(v)                      11 significant bits, multiplied on v_mfma_f32_16x16x32_f16 against the weight
//                                         as fp16 (exact for a bf16-valued weight);
//   lo = e4m3((v - hi) * 2^12)            4 more bits, multiplied on v_mfma_scale_f32_16x16x128_f8f6f4 against the
//                                         weight as e4m3, block scale 2^-12 on the lo operand: TWICE the rate of the
//                                         16-bit shapes, accumulating into the same fp32 registers
// = 1.5 MFMA units per product instead of 2 at the same ~2^-15 operand precision (microbench/f8_probe.hip: stream
// rate on random operands 1.31 vs 1.03 PFLOP/s algorithmic; scripts/precision_emulate.py: g1_xsmall 3.6e-4 vs 1.9e-4).
// The conversions saturate (MODE.FP16_OVFL, set_saturating_conversions()): without it an e4m3 overflow is NaN.
//
// fp8 fragment of one 16-row tile and one K = 128 step S (k-steps 4S .. 4S+3 of the 16-bit shapes): lane (i, g) holds
// 32 bytes = two 16-byte halves hh = 0, 1; byte p of half hh is k = 32 (4S + 2hh + (p >> 3)) + 8g + (p & 7) -- the same
// 8 k-values per (lane, k-step) as the 16-bit fragments, so hi and lo of a value are produced by the same lane.  In
// memory a half is one 1 KiB piece [lane][16 B].  Both MFMA operands use this map (the contraction only needs them equal).
// ----------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

constexpr int F8_LO_SHIFT = 12;                   // lo planes are stored x 2^12
constexpr int F8_SCALE_LO = 127 - F8_LO_SHIFT;    // e8m0 block scale 2^-12 of an activation's lo plane
constexpr int F8_SCALE_ONE = 127;                 // e8m0 2^0 (the e4m3 copy of an activation)
// The e4m3 planes of a WEIGHT are stored x 2^6 (its lo plane x 2^18): e4m3's normal range is [2^-6, 448], and the
// weights of a trained checkpoint sit around 0.02 -- unshifted, half of them would fall on e4m3's subnormal grid
// (step 2^-9) and weights below 2^-10 would vanish from the lo product altogether.  Shifted, everything in
// [2^-12, 7] keeps its 3 mantissa bits (a larger weight saturates in the correction term only).  On reference-initialised
// weights the format's error drops 2.4x (scripts/precision_emulate.py: f16+2f8 4.3e-6 -> 1.8e-6, (hi, lo) bf16 0.9e-6);
// on O(1) weights nothing changes.
constexpr int F8_W_SHIFT = 6;
constexpr int F8_SCALE_W = 127 - F8_W_SHIFT;                     // e4m3(w x 2^6)
constexpr int F8_SCALE_W_LO = 127 - F8_LO_SHIFT - F8_W_SHIFT;    // e4m3(lo(w) x 2^18)

__device__ __forceinline__ void set_saturating_conversions() {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // MODE.FP16_OVFL: f16 / fp8 conversions clamp to max normal
}
// ... and back to IEEE overflow, for a phase that converts to fp16 only: an activation beyond fp16's range (|h| >= 65520)
// then becomes Inf and the forward's outputs NaN -- loud -- instead of a silently clamped operand (the e4m3 lo planes are
// what needs the clamp: there an overflow would be NaN for a harmless loss of the correction term)
__device__ __forceinline__ void set_overflowing_conversions() {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}

// the 16-bit product of this kernel set: operands carry fp16 bits in the registers the bf16 kernels use
__device__ __forceinline__ f32x4 mfma16h(bf16x8 x, bf16x8 y, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
}
__device__ __forceinline__ i32x8 f8_frag(bf16x8 half0, bf16x8 half1) {
  const uint4 a = __builtin_bit_cast(uint4, half0), b = __builtin_bit_cast(uint4, half1);
  return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
}
// D += X8 * Y8 (e4m3 x e4m3, K = 128): lo(activation) x e4m3(weight); LO_IS_Y: which operand is the activation's lo plane
template <bool LO_IS_Y>
__device__ __forceinline__ f32x4 mfma8(i32x8 x, i32x8 y, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, c, 0, 0, 0, LO_IS_Y ? F8_SCALE_W : F8_SCALE_LO, 0,
                                                          LO_IS_Y ? F8_SCALE_LO : F8_SCALE_W);
}
// ... e4m3(activation) x lo(weight); LO_IS_Y: which operand is the WEIGHT's lo plane
template <bool LO_IS_Y>
__device__ __forceinline__ f32x4 mfma8w(i32x8 x, i32x8 y, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, c, 0, 0, 0, LO_IS_Y ? F8_SCALE_ONE : F8_SCALE_W_LO, 0,
                                                          LO_IS_Y ? F8_SCALE_W_LO : F8_SCALE_ONE);
}

// two fp32 -> one dword of two fp16 (RNE): v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// v - float(half `HI_HALF` of h): exact, one instruction (v_fma_mix_f32 with an fp16 source)
template <int HI_HALF>
__device__ __forceinline__ float sub_f16_half(float v, uint32_t h) {
  float r;
  if (HI_HALF) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
  return r;
}
// four fp32 -> two dwords of fp16 (hi) + one dword of four e4m3 bytes (lo x 2^12): 2 + 4 + 2 = 8 VALU instructions
__device__ __forceinline__ void split4_f8(const float v[4], uint2& hi, uint32_t& lo8) {
  hi.x = pack_f16x2(v[0], v[1]);
  hi.y = pack_f16x2(v[2], v[3]);
  const float l0 = sub_f16_half<0>(v[0], hi.x), l1 = sub_f16_half<1>(v[1], hi.x);
  const float l2 = sub_f16_half<0>(v[2], hi.y), l3 = sub_f16_half<1>(v[3], hi.y);
  constexpr float inv_scale = 1.0f / (float)(1 << F8_LO_SHIFT);  // the instruction divides by its scale operand
  i16x2 w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, l0, l1, inv_scale, false);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, l2, l3, inv_scale, true);
  lo8 = __builtin_bit_cast(uint32_t, w);
}
// four fp32 -> (hi, lo) fp16 pairs (lo = RNE_fp16(v - hi)): the 16-bit form, for operands whose K is too short for fp8
__device__ __forceinline__ void split4_f16(const float v[4], uint2& hi, uint2& lo) {
  hi.x = pack_f16x2(v[0], v[1]);
  hi.y = pack_f16x2(v[2], v[3]);
  lo.x = pack_f16x2(sub_f16_half<0>(v[0], hi.x), sub_f16_half<1>(v[1], hi.x));
  lo.y = pack_f16x2(sub_f16_half<0>(v[2], hi.y), sub_f16_half<1>(v[3], hi.y));
}
// two fp16 pairs (two dwords) -> one dword of four e4m3 bytes (the value itself, unscaled): 2 VALU instructions
__device__ __forceinline__ uint32_t f16x4_to_e4m3(uint32_t h01, uint32_t h23) {
  i16x2 w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(f16x2, h01), 1.0f, false);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(w, __builtin_bit_cast(f16x2, h23), 1.0f, true);
  return __builtin_bit_cast(uint32_t, w);
}
// four fp32 -> one dword of four e4m3 bytes (unscaled)
__device__ __forceinline__ uint32_t f32x4_to_e4m3(const float v[4]) {
  i16x2 w = {0, 0};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, v[0], v[1], 1.0f, false);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, v[2], v[3], 1.0f, true);
  return __builtin_bit_cast(uint32_t, w);
}
// e4m3 of one value through the hardware conversion (weight packing; the caller has set saturating conversions)
__device__ __forceinline__ unsigned char f2e4m3(float v) {
  return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}
__device__ __forceinline__ u16 f2h(float v) { return __builtin_bit_cast(u16, (_Float16)v); }

// How well a weight fits the fp16 plane of the "f16 + fp8" sets.  flags[0]: a weight of magnitude >= 2^-14 is not an
// fp16 value (kernel set 3 needs every weight exact; set 4 carries the difference in its lo plane).  fit[0]: sum of the
// SQUARED differences of the weights below 2^-14 that miss fp16's subnormal grid (spacing 2^-24): neither plane can hold
// that difference (it is below e4m3's reach even shifted), so a tensor made of such weights -- e.g. an output projection
// scaled down by 2^-10 -- would be multiplied at fp16 single-pass accuracy (measured 2e-3 .. 4e-3 on logits), while the
// handful of tiny weights of a normally scaled tensor are harmless.  f16_fit_close_tensor_kernel (opk_small.hip.h)
// compares fit[0] with the tensor's energy fit[1] (weight_energy_kernel) and raises flags[2] when the lost part exceeds
// 2^-36 of it (the format's own relative error is ~2^-16, i.e. 2^-32 in energy); the library then keeps the (hi, lo)
// bf16 sets for the model.
__device__ __forceinline__ void note_f16_fit(float v, _Float16 hv, int* __restrict__ flags, float* __restrict__ fit) {
  const float d = v - (float)hv;
  if (d == 0.f) return;
  if (fabsf(v) >= 6.103515625e-05f) flags[0] = 1;
  else atomicAdd(fit, d * d * 1.1529215e18f);  // x 2^60: d^2 <= 2^-50 would underflow the sum's precision otherwise
}

// ---- kernel set "f16" (round 5): single-pass fp16 operands ---------------------------------------------------------
// The layouts, kernels and launch shapes of the single-pass bf16 set with every MFMA operand carried as fp16 instead
// (11 significant bits where bf16 has 8): on weights of a trained checkpoint's scale the logits err by ~1/8 of the bf16
// single pass (measured on reference-initialised xsmall, 256 x 512: 4.7e-4 -> 5e-5 against the all-terms kernels) at the
// same MFMA count.  It is never the default: op_calibrate selects it when the loaded weights allow.
// fp16's range: conversions overflow to Inf (MODE.FP16_OVFL stays 0), the outputs become NaN and the host falls back.
template <bool H16>
__device__ __forceinline__ f32x4 mfma16x(bf16x8 x, bf16x8 y, f32x4 c) {
  if constexpr (H16) return mfma16h(x, y, c);
  else return mfma16(x, y, c);
}
template <bool SPLIT, bool H16>
__device__ __forceinline__ void split2x(float a, float b, uint32_t& hi, uint32_t& lo) {
  if constexpr (H16) {
    hi = pack_f16x2(a, b);
    lo = 0u;
  } else {
    split2<SPLIT>(a, b, hi, lo);
  }
}
template <bool SPLIT, bool H16>
__device__ __forceinline__ void split4x(const float v[4], uint2& hi, uint2& lo) {
  split2x<SPLIT, H16>(v[0], v[1], hi.x, lo.x);
  split2x<SPLIT, H16>(v[2], v[3], hi.y, lo.y);
}
template <bool SPLIT, bool H16>
__device__ __forceinline__ void split2x_pk(f32x2 v, uint32_t& hi, uint32_t& lo) {
  if constexpr (H16) {
    hi = pack_f16x2(v.x, v.y);
    lo = 0u;
  } else {
    split2_pk<SPLIT>(v, hi, lo);
  }
}

// ---- hand-placed LDS fragment reads -------------------------------------------------------------------------
// While a global_load_lds DMA is in flight hipcc (ROCm 7.2) cannot count lgkmcnt: every wait it inserts in front of
// an MFMA is `s_waitcnt lgkmcnt(0)`, which also waits for the fragment reads issued just before it for the NEXT
// k-step -- and left to itself it re-uses one register set and puts most ds_read_b128 directly in front of such a
// wait (~40 % of every wave's cycles were spent there: SQ_WAIT_ANY in profiles/r02*).  So the weight-fragment reads
// and their waits are inline asm (invisible to the compiler's scoreboard), ordered by data dependencies only:
//   * volatile asm statements keep their program order among themselves (reads and waits);
//   * a wait "rewrites" the fragment registers it guards, so the MFMAs that use them cannot move above it;
//   * a read group "rewrites" one accumulator of the k-step before it, so it cannot sink below that step's MFMAs
//     (nor can they sink below it): the reads of k-step ks + 1 are in flight during all MFMAs of k-step ks.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_frag(uint32_t lds_addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
// ordered behind the producers (and ahead of the consumers) of the accumulators it "rewrites".  IN_AGPR: the
// one-wave-per-SIMD kernels (512-register budget) get their MFMA accumulators in AGPRs; a "v" constraint there makes
// the compiler copy them to VGPRs and back around every read group.
template <int OFF, bool IN_AGPR = false>
__device__ __forceinline__ bf16x8 lds_read_frag_after(uint32_t lds_addr, f32x4& p0, f32x4& p1) {
  bf16x8 v;
  if constexpr (IN_AGPR) asm volatile("ds_read_b128 %0, %3 offset:%4" : "=v"(v), "+a"(p0), "+a"(p1) : "v"(lds_addr), "n"(OFF));
  else asm volatile("ds_read_b128 %0, %3 offset:%4" : "=v"(v), "+v"(p0), "+v"(p1) : "v"(lds_addr), "n"(OFF));
  return v;
}
template <int OFF, bool IN_AGPR = false>
__device__ __forceinline__ bf16x8 lds_read_frag_after(uint32_t lds_addr, f32x4& p0, f32x4& p1, f32x4& p2, f32x4& p3) {
  bf16x8 v;
  if constexpr (IN_AGPR)
    asm volatile("ds_read_b128 %0, %5 offset:%6" : "=v"(v), "+a"(p0), "+a"(p1), "+a"(p2), "+a"(p3) : "v"(lds_addr), "n"(OFF));
  else
    asm volatile("ds_read_b128 %0, %5 offset:%6" : "=v"(v), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(lds_addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_frag_after1(uint32_t lds_addr, f32x4& p0) {  // one accumulator, in an AGPR
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %2 offset:%3" : "=v"(v), "+a"(p0) : "v"(lds_addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait2(bf16x8& a, bf16x8& b) {  // at most N fragment reads still in flight
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// The same discipline for 16-byte fp32 reads (LayerNorm weights kept in LDS): issued by hand a step ahead and waited
// for by count -- a compiler-placed read behind an in-flight DMA is always followed by s_waitcnt lgkmcnt(0).
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_f4(uint32_t lds_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait_f4(f32x4& a, f32x4& b) {  // at most N reads still in flight
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// The same discipline for an arbitrary stream of NSTEPS steps that each consume TWO weight fragments (one weight
// plane): the fragments of step s are at LDS byte offsets Off::at(s, 0 / 1) from `lds_addr` and are requested DEPTH
// steps ahead into DEPTH + 1 rotating register sets; body(step, w0, w1) holds the step's MFMAs AND the slice of
// vector work that is to run between them.  Here the order is fixed by scheduling fences instead of accumulator
// pins (the one-wave-per-SIMD kernel keeps 128 accumulators in AGPRs; "+v" pins would drag them through VGPRs):
// nothing crosses the fence after a step's body, the volatile read group and the next wait follow in program order.
// frag_stream4: the same with FOUR fragments per step (both weight planes): body(step, w0, w1, w2, w3).
template <int NSTEPS, int DEPTH, class Off, class Body>
__device__ __forceinline__ void frag_stream4(uint32_t lds_addr, Body&& body) {
  constexpr int SETS = DEPTH + 1;
  bf16x8 w[SETS][4];
  auto read_group = [&](auto step_tag) {
    constexpr int s = decltype(step_tag)::value;
    w[s % SETS][0] = lds_read_frag<Off::at(s, 0)>(lds_addr);
    w[s % SETS][1] = lds_read_frag<Off::at(s, 1)>(lds_addr);
    w[s % SETS][2] = lds_read_frag<Off::at(s, 2)>(lds_addr);
    w[s % SETS][3] = lds_read_frag<Off::at(s, 3)>(lds_addr);
  };
  static_for<(DEPTH + 1 < NSTEPS ? DEPTH + 1 : NSTEPS)>([&](auto t) { read_group(t); });
  static_for<NSTEPS>([&](auto t) {
    constexpr int s = decltype(t)::value;
    constexpr int set = s % SETS;
    constexpr int ahead = (NSTEPS - 1 - s) < DEPTH ? (NSTEPS - 1 - s) : DEPTH;
    lds_wait4<4 * ahead>(w[set][0], w[set][1], w[set][2], w[set][3]);
    body(t, w[set][0], w[set][1], w[set][2], w[set][3]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + DEPTH + 1 < NSTEPS) read_group(std::integral_constant<int, s + DEPTH + 1>{});
  });
}

// the offsets of `Off` from step S0 on (a stream over the tail of another stream's step list)
template <class Off, int S0>
struct OffShift {
  static constexpr int at(int s, int j) { return Off::at(s + S0, j); }
};

template <int NSTEPS, int DEPTH, class Off, class Body>
__device__ __forceinline__ void frag_stream2(uint32_t lds_addr, Body&& body) {
  constexpr int SETS = DEPTH + 1;
  bf16x8 w[SETS][2];
  auto read_group = [&](auto step_tag) {
    constexpr int s = decltype(step_tag)::value;
    w[s % SETS][0] = lds_read_frag<Off::at(s, 0)>(lds_addr);
    w[s % SETS][1] = lds_read_frag<Off::at(s, 1)>(lds_addr);
  };
  static_for<(DEPTH + 1 < NSTEPS ? DEPTH + 1 : NSTEPS)>([&](auto t) { read_group(t); });
  static_for<NSTEPS>([&](auto t) {
    constexpr int s = decltype(t)::value;
    constexpr int set = s % SETS;
    constexpr int ahead = (NSTEPS - 1 - s) < DEPTH ? (NSTEPS - 1 - s) : DEPTH;  // read groups that may stay in flight
    lds_wait2<2 * ahead>(w[set][0], w[set][1]);
    body(t, w[set][0], w[set][1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + DEPTH + 1 < NSTEPS) read_group(std::integral_constant<int, s + DEPTH + 1>{});
  });
}

// (a step body that may or may not be handed a read-placement callback: the callback, or the default)
template <class D>
__device__ __forceinline__ D&& rd_or(D&& d) { return static_cast<D&&>(d); }
template <class D, class R>
__device__ __forceinline__ R&& rd_or(D&&, R&& r) { return static_cast<R&&>(r); }

// frag_stream2 with the reads of the group DEPTH + 1 steps ahead issued BETWEEN the step's MFMAs instead of behind them.
// A wave that is alone on its SIMD issues one instruction per four cycles, in order: behind the last MFMA of a step only
// that MFMA's shadow (12 cycles) is there for the two ds_read_b128, the counted wait and the hazard nop in front of the
// next step's first MFMA -- measured on the bare stream of the whole-layer kernel's MLP loop (no riders, no DMA, no
// barrier): 79.5 cycles per 64-cycle step.  Here the body calls rd(0, acc) / rd(1, acc) where the reads are to go:
// each read "rewrites" the accumulator named (an AGPR), so it can neither rise above the MFMA that produces it nor sink
// below its next use.  DEPTH + 2 register sets: the set being filled is not the one being multiplied.
template <int NSTEPS, int DEPTH, class Off, class Body>
__device__ __forceinline__ void frag_stream2i(uint32_t lds_addr, Body&& body) {
  constexpr int SETS = DEPTH + 2;
  bf16x8 w[SETS][2];
  static_for<(DEPTH + 1 < NSTEPS ? DEPTH + 1 : NSTEPS)>([&](auto t) {
    constexpr int s = decltype(t)::value;
    w[s % SETS][0] = lds_read_frag<Off::at(s, 0)>(lds_addr);
    w[s % SETS][1] = lds_read_frag<Off::at(s, 1)>(lds_addr);
  });
  static_for<NSTEPS>([&](auto t) {
    constexpr int s = decltype(t)::value;
    constexpr int set = s % SETS;
    constexpr int nxt = s + DEPTH + 1;
    constexpr int ahead = (NSTEPS - 1 - s) < DEPTH ? (NSTEPS - 1 - s) : DEPTH;
    lds_wait2<2 * ahead>(w[set][0], w[set][1]);
    auto rd = [&](auto j_tag, f32x4& p0) {
      constexpr int j = decltype(j_tag)::value;
      if constexpr (nxt < NSTEPS) w[nxt % SETS][j] = lds_read_frag_after1<Off::at(nxt < NSTEPS ? nxt : 0, j)>(lds_addr, p0);
    };
    body(t, w[set][0], w[set][1], rd);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Stores of tensors that the launch writing them never reads back (q / k / v^T / o / h pieces, the residual stream's
// write-back) are non-temporal (global_store ... nt).  As ordinary stores they displace the next block's operands from
// the XCD's L2: in the whole-layer kernel the phase that starts a block (operand fetch + attention-output projection)
// took 37 k cycles behind the previous block's 393 KB of plain q / k / v^T stores, 28 k behind nt stores, 26 k with the
// residual write-back nt as well (microbench/rowgemm_ablate.hip -DOPK_TIMING; plain-store / plain-load switches: microbench/experiments/rowgemm_ablation_hooks.patch).
__device__ __forceinline__ void store_stream16(void* dst, const uint4& v) {
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(dst));
}
__device__ __forceinline__ void store_stream16(float* dst, const float4& v) {
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(dst));
}

// The matching loads for operands a launch reads exactly once (activation fragments, residual rows): global_load ... nt.
__device__ __forceinline__ bf16x8 load_stream_frag(const u16* src) {
  return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src));
}
__device__ __forceinline__ float4 load_stream_f4(const float* src) {
  typedef float f32x4_ntl __attribute__((ext_vector_type(4)));
  const f32x4_ntl v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ntl*>(src));
  return make_float4(v.x, v.y, v.z, v.w);
}

// GELU (exact-erf form) = 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |x| he(|x|),  he(a) = erfc(a / sqrt 2) / 2,
// with he(a) = 2^q(a), q a degree-5 polynomial (weighted minimax fit of log2 he on [0, 10], weight a he(a); leading
// coefficient negative, so q -> -inf and he -> 0 for large |x|).  Eight instructions: five FMAs, one v_exp_f32, one
// FMA, one med3 -- the VALU work next to the MFMAs is what these epilogues cost, instruction for instruction.  No
// cancellation on either side of zero.  Max |gelu - exact| = 6.4e-7 over [-12, 12] evaluated in fp32 (offline, against
// fp64 erf); the library erff costs ~3x the instructions.
// The evaluation in stages (the one-wave-per-SIMD layer kernel runs stage k of ALL values of a chunk in one step of
// its MFMA stream: a vector instruction that depends on the previous one costs the wave 8 issue cycles, an independent
// one 4.5 -- microbench/mfma_loop.hip): q after stage 0..4, he after 5, gelu after 6.
__device__ __forceinline__ float gelu_erf_poly(float q, float ax, int stage) {
  switch (stage) {
    case 0: return fmaf(-4.7330930829e-04f, ax, 7.0845573209e-03f);
    case 1: return fmaf(q, ax, -5.1827382296e-02f);
    case 2: return fmaf(q, ax, -4.5999243855e-01f);
    case 3: return fmaf(q, ax, -1.1507878304e+00f);
    default: return fmaf(q, ax, -1.0000376701e+00f);
  }
}
__device__ __forceinline__ float gelu_erf_finish(float he, float x) {
  // max(x, 0) in ONE instruction: fmaxf and med3(x, 0, +inf) both come out of the compiler as a NaN-quieting
  // v_max x, x followed by v_max 0, x
  float relu;
  asm("v_max_f32 %0, 0, %1" : "=v"(relu) : "v"(x));
  return fmaf(-he, fabsf(x), relu);
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  float q = gelu_erf_poly(0.f, ax, 0);
#pragma unroll
  for (int k = 1; k < 5; ++k) q = gelu_erf_poly(q, ax, k);
  return gelu_erf_finish(__builtin_amdgcn_exp2f(q), x);
}

// rowgemm_kernel epilogues: q/k/v projection (RoPE, fragment-packed q, k, v^T), GeGLU (fragment-packed h), or no chunk
// loop at all (RE_NONE: the fused layer kernel of the LAST layer only updates the residual stream); and prologues:
// plain split of x (layer 0), one fused k-streamed GEMM + residual + LayerNorm (RP_KSTREAM), or the whole
// attention-output projection + MLP of a layer with h kept on chip (RP_MLP).
enum RowEpilogue { RE_QKV = 0, RE_NONE = 1, RE_GEGLU = 2 };
enum RowPrologue { RP_SPLIT = 1, RP_KSTREAM = 3, RP_MLP = 4 };
constexpr int ROW_BM = 128;
constexpr int ROW_CHUNK = 32;  // output features per streamed chunk
enum PanelEpi { PE_RESIDUAL = 0, PE_QK = 1, PE_V = 2, PE_GEGLU = 3 };

template <bool SPLIT>
__device__ __forceinline__ void pack8(const float v[8], bf16x8& hi, bf16x8& lo) {
  uint2 h0, l0, h1, l1;
  split4<SPLIT>(v, h0, l0);
  split4<SPLIT>(v + 4, h1, l1);
  hi = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
  lo = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
}

template <bool SPLIT, bool H16>
__device__ __forceinline__ void pack8x(const float v[8], bf16x8& hi, bf16x8& lo) {
  uint2 h0, l0, h1, l1;
  split4x<SPLIT, H16>(v, h0, l0);
  split4x<SPLIT, H16>(v + 4, h1, l1);
  hi = as_frag(make_uint4(h0.x, h0.y, h1.x, h1.y));
  lo = as_frag(make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// Term mask of a contraction  left x right  (left = activation / q / p, right = weight / k / v): the hi x hi product is
// always computed; bit 0 adds lo(left) x hi(right), bit 1 adds hi(left) x lo(right).  3 = "bf16x3", 0 = single pass.
constexpr int T_LEFT_LO = 1;
constexpr int T_RIGHT_LO = 2;
__host__ __device__ constexpr int term_count(int t) { return 1 + (t & 1) + ((t >> 1) & 1); }


}  // namespace opk
