// op_internal.h -- host-side declarations shared by op_api.hip and the kernel-launch translation units
// (op_launch_*.hip).  The library is built from several translation units so that the large kernel templates
// compile in parallel; each launch unit instantiates one kernel family for every curated precision policy.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "opk_attn.hip.h"
#include "opk_common.hip.h"
#include "opk_layer32.hip.h"
#include "opk_layer16p.hip.h"
#include "opk_panel.hip.h"
#include "opk_rowgemm.hip.h"

namespace opl {

// Precision policy = one term mask per contraction family (opk::T_LEFT_LO / T_RIGHT_LO; see op_gemm_family in
// include/open_provence_hip.h for the order).  "left" is always the activation-side operand:
//   wqkv: LN(x) x Wqkv    qk: q x k    pv: p x v    attn_out: o x Wo    wi: LN(x) x Wi    mlp_out: h x Wo
struct Policy {
  int wqkv, qk, pv, attn_out, wi, mlp_out;
  int fmt = 0;  // operand format: 0 = (hi, lo) bf16 planes, 1 = fp16 hi + e4m3 lo in the whole-layer kernel (opk_common.hip.h), 2 = fp16 single plane everywhere
  constexpr bool operator==(const Policy& o) const {
    return wqkv == o.wqkv && qk == o.qk && pv == o.pv && attn_out == o.attn_out && wi == o.wi && mlp_out == o.mlp_out;
  }
};

// Curated policies: every kernel family is instantiated with exactly the product terms (and output lo planes) each
// of these needs.  Any other policy runs on the kernels of kPolicies[0] (all terms) with the unused lo operands
// cleared -- bit-identical numerics, no speed-up.
//   0  bf16x3                 every operand (hi, lo)
//   1  bf16 weights           weight lo planes are zero (bf16 checkpoint, or OP_PRECISION_BF16X2): 2 passes in the four
//                             weight GEMMs, 3 in attention (q, k, p, v are all activations)
//   2  bf16                   single pass everywhere
//   3  f16 + fp8              the terms of set 1 with the whole-layer kernel's operands as fp16 hi + e4m3 lo: 1.5 MFMA
//                             units per product instead of 2 (hidden <= 256, weights exactly representable in fp16;
//                             layer-0 q / k / v and attention run set 1's kernels, attention writes o in the new format).
//                             Never matched by terms (operator== ignores fmt): resolve_policy() upgrades 1 -> 3.
//   4  f16 + fp8, all terms   the terms of set 0 (fp32-valued weights) in the same format: the weight's lo part rides as a
//                             third e4m3 plane (bf16 for the MLP output projection): 2 MFMA units per product instead of 3,
//                             and ONE kernel per layer where set 0 needs two with h through HBM.  resolve_policy(): 0 -> 4.
//   5  f16                    (round 5) single pass with every operand as fp16: the kernels and layouts of set 2, 11 instead of
//                             8 significant bits per operand.  Never matched by terms and never a default: op_calibrate /
//                             op_select_kernel_set choose it when the loaded weights allow (reported as kernel set 7).
constexpr Policy kPolicies[] = {
    {3, 3, 3, 3, 3, 3},
    {1, 3, 3, 1, 1, 1},
    {0, 0, 0, 0, 0, 0},
    {1, 3, 3, 1, 1, 1, 1},
    {3, 3, 3, 3, 3, 3, 1},
    {0, 0, 0, 0, 0, 0, 2},
};
constexpr int PI_ALL_TERMS = 0, PI_BF16_WEIGHTS = 1, PI_BF16 = 2, PI_F16_F8 = 3, PI_F16_F8_W = 4, PI_F16 = 5;
constexpr int N_POLICIES = (int)(sizeof(kPolicies) / sizeof(kPolicies[0]));

// template arguments each kernel family derives from a policy
constexpr int qkv_olo(const Policy& p) { return (p.qk & 1) | (p.qk & 2) | ((p.pv & 2) ? 4 : 0); }  // q_lo, k_lo, v_lo
constexpr int h_olo(const Policy& p) { return p.mlp_out & 1; }
constexpr bool o_lo(const Policy& p) { return (p.attn_out & 1) != 0; }

// Launchers.  `pi` indexes kPolicies.  Each returns false when the shape has no instantiation.
bool launch_row_qkv0(hipStream_t st, const opk::RowGemmParams& p, int ks, bool small, int pi, unsigned grid);
bool launch_row_geglu_fused(hipStream_t st, const opk::RowGemmParams& p, int ks, bool small, int pi, unsigned grid);
bool launch_row_qkv_fused(hipStream_t st, const opk::RowGemmParams& p, int ks, bool small, int pi, unsigned grid);
bool launch_kstream(hipStream_t st, const opk::KStreamParams& p, int nf, int pi, unsigned grid);
// one kernel per layer: x += o Wo^T; x += GeGLU(LN(x) Wi^T) Wo^T; (with_qkv) next layer's q / k / v^T.  128-row blocks.
bool has_row_layer_fused(int pi);
bool launch_row_layer_fused(hipStream_t st, const opk::RowGemmParams& p, int ks, int pi, bool with_qkv, unsigned grid,
                            bool waves8);
// ... of the "f16 + fp8" kernel set (hidden 128 / 256)
bool launch_row_layer_f8(hipStream_t st, const opk::RowGemmParams& p, int ks, bool with_qkv, unsigned grid);
bool launch_row_layer_f8w(hipStream_t st, const opk::RowGemmParams& p, int ks, bool with_qkv, unsigned grid);  // kernel set 4
// ... of the "f16" kernel set (PI_F16): the layer-0 q / k / v projection and the whole-layer kernel with fp16 operands
bool launch_row_qkv0_h16(hipStream_t st, const opk::RowGemmParams& p, int ks, bool small, unsigned grid);
bool launch_row_layer_h16(hipStream_t st, const opk::RowGemmParams& p, int ks, bool with_qkv, unsigned grid, bool waves8);
// the same launch on the 32x32x16 shape (hidden = 256; kernel sets 1 and 2)
bool has_layer32(int pi);
bool launch_layer32(hipStream_t st, const opk::Layer32Params& p, int pi, bool with_qkv, unsigned grid);
bool launch_layer16p(hipStream_t st, const opk::Layer32Params& p, bool h16, bool with_qkv, bool xin_t, bool xout_t, unsigned grid);
// waves x kt: (8, 2) and (4, 2) full attention / long and short sequences, (4, 1) sliding window.
// zero_p_lo (pi == 0 only): the policy has no lo(p) x hi(v) term.
// f16_in_f8_out (kernel sets 10 / 11; pi = PI_F16_F8 / PI_F16_F8_W): the fp16 single-pass kernels of PI_F16 on fp16 q / k / v^T,
// o written as fp16 + e4m3 pieces for the attention output projection of the fp16 + e4m3 format.
bool launch_attn(hipStream_t st, const opk::AttnFpParams& p, int waves, int kt, int pi, bool zero_p_lo, dim3 grid,
                 bool f16_in_f8_out = false);
bool launch_panel(hipStream_t st, const opk::PanelParams& p, int epi, int pi, dim3 grid);
// q, k and v^T panels of one layer in one launch (p.n_tiles = 3 H / 256, p.n_qk_tiles = 2 H / 256, p.o2 = v^T)
bool launch_panel_qkv(hipStream_t st, const opk::PanelParams& p, int pi, dim3 grid);
// the panel GEMMs in the fp16 + e4m3 format (kernel sets 3 / 4; wlo: the weights carry their lo part = set 4)
bool launch_panel_f8(hipStream_t st, const opk::PanelParams& p, int epi, bool wlo, dim3 grid);
// o16 (kernel sets 10 / 11): q, k, v^T as single-plane fp16 for launch_attn(.., f16_in_f8_out = true)
bool launch_panel_f8_qkv(hipStream_t st, const opk::PanelParams& p, bool wlo, bool o16, dim3 grid);

}  // namespace opl
