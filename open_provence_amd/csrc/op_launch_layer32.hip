// op_launch_layer32.hip -- instantiations of the 32x32x16 whole-layer kernel (hidden = 256) for the kernel sets whose
// weights are single-plane: "bf16 weights" (activation operands hi + lo) and "bf16" (single pass).
#include "op_internal.h"

namespace opl {
using namespace opk;

namespace {
template <int PI>
bool launch_pi(hipStream_t st, const Layer32Params& p, bool with_qkv, unsigned grid) {
  constexpr Policy P = kPolicies[PI];
  constexpr bool single_plane = ((P.wqkv | P.attn_out | P.wi | P.mlp_out) & 2) == 0;
  constexpr bool uniform = (P.wqkv & 1) == (P.attn_out & 1) && (P.wi & 1) == (P.mlp_out & 1) && (P.wqkv & 1) == (P.wi & 1);
  if constexpr (!single_plane || !uniform) {
    return false;
  } else {
    constexpr bool ALO = (P.wi & 1) != 0;
    if (with_qkv) hipLaunchKernelGGL((layer32_kernel<8, true, ALO, qkv_olo(P)>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((layer32_kernel<8, false, ALO, 0>), dim3(grid), dim3(256), 0, st, p);
    return true;
  }
}
}  // namespace

bool has_layer32(int pi) {
  if (pi < 0 || pi >= N_POLICIES || kPolicies[pi].fmt != 0) return false;
  const Policy& P = kPolicies[pi];
  return ((P.wqkv | P.attn_out | P.wi | P.mlp_out) & 2) == 0 && (P.wqkv & 1) == (P.attn_out & 1) &&
         (P.wi & 1) == (P.mlp_out & 1) && (P.wqkv & 1) == (P.wi & 1);
}

bool launch_layer32(hipStream_t st, const Layer32Params& p, int pi, bool with_qkv, unsigned grid) {
  static_assert(N_POLICIES == 6, "extend the switch");
  switch (pi) {
    case 0: return launch_pi<0>(st, p, with_qkv, grid);
    case 1: return launch_pi<1>(st, p, with_qkv, grid);
    case 2: return launch_pi<2>(st, p, with_qkv, grid);
    case 3:
    case 4:
    case 5: return false;  // kernel sets 3, 4 and "f16" have their own whole-layer kernels
    default: return false;
  }
}

// the wave-pair form (opk_layer16p.hip.h): single-pass operands, bf16 or fp16; x row-major or tiled on either side
bool launch_layer16p(hipStream_t st, const Layer32Params& p, bool h16, bool with_qkv, bool xin_t, bool xout_t, unsigned grid) {
  const dim3 g(grid), b(512);
#define OPL_L16P(H16, XI, XO)                                                                   \
  do {                                                                                          \
    if (with_qkv) hipLaunchKernelGGL((layer16p_kernel<8, true, H16, XI, XO>), g, b, 0, st, p);  \
    else hipLaunchKernelGGL((layer16p_kernel<8, false, H16, XI, XO>), g, b, 0, st, p);          \
  } while (0)
  if (h16) {
    if (xin_t && xout_t) OPL_L16P(true, true, true);
    else if (xin_t) OPL_L16P(true, true, false);
    else if (xout_t) OPL_L16P(true, false, true);
    else OPL_L16P(true, false, false);
  } else {
    if (xin_t && xout_t) OPL_L16P(false, true, true);
    else if (xin_t) OPL_L16P(false, true, false);
    else if (xout_t) OPL_L16P(false, false, true);
    else OPL_L16P(false, false, false);
  }
#undef OPL_L16P
  return true;
}

}  // namespace opl
