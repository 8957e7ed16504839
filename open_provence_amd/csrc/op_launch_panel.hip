// op_launch_panel.hip -- instantiations of panel_gemm_kernel (hidden % 256 == 0: base / large / en-gte) for every
// curated precision policy.
#include "op_internal.h"

namespace opl {
using namespace opk;

namespace {

template <int PI>
bool launch_pi(hipStream_t st, const PanelParams& p, int epi, dim3 grid) {
  constexpr Policy P = kPolicies[PI];
  constexpr bool H16 = P.fmt == 2;  // kernel set "f16": fp16 operands
  const dim3 block(256);
  if (epi == PE_QK)
    hipLaunchKernelGGL((panel_gemm_kernel<PE_QK, P.wqkv, (P.qk & 3), H16>), grid, block, 0, st, p);
  else if (epi == PE_V)
    hipLaunchKernelGGL((panel_gemm_kernel<PE_V, P.wqkv, ((P.pv & 2) ? 1 : 0), H16>), grid, block, 0, st, p);
  else if (epi == PE_GEGLU)
    hipLaunchKernelGGL((panel_gemm_kernel<PE_GEGLU, P.wi, h_olo(P), H16>), grid, block, 0, st, p);
  else if (epi == 100)  // attention output projection
    hipLaunchKernelGGL((panel_gemm_kernel<PE_RESIDUAL, P.attn_out, 0, H16>), grid, block, 0, st, p);
  else if (epi == 101)  // MLP output projection
    hipLaunchKernelGGL((panel_gemm_kernel<PE_RESIDUAL, P.mlp_out, 0, H16>), grid, block, 0, st, p);
  else
    return false;
  return true;
}

template <int PI>
void launch_qkv_pi(hipStream_t st, const PanelParams& p, dim3 grid) {
  constexpr Policy P = kPolicies[PI];
  hipLaunchKernelGGL((panel_qkv_kernel<P.wqkv, (P.qk & 3), ((P.pv & 2) ? 1 : 0), P.fmt == 2>), grid, dim3(256), 0, st, p);
}

}  // namespace

bool launch_panel_qkv(hipStream_t st, const PanelParams& p, int pi, dim3 grid) {
  static_assert(N_POLICIES == 6, "extend the switch below");
  switch (pi) {
    case 0: launch_qkv_pi<0>(st, p, grid); return true;
    case 1: launch_qkv_pi<1>(st, p, grid); return true;
    case 2: launch_qkv_pi<2>(st, p, grid); return true;
    case 3: launch_qkv_pi<1>(st, p, grid); return true;
    case 4: launch_qkv_pi<0>(st, p, grid); return true;
    case 5: launch_qkv_pi<5>(st, p, grid); return true;
    default: return false;
  }
}

bool launch_panel(hipStream_t st, const PanelParams& p, int epi, int pi, dim3 grid) {
  static_assert(N_POLICIES == 6, "extend the switch below");
  switch (pi) {
    case 0: return launch_pi<0>(st, p, epi, grid);
    case 1: return launch_pi<1>(st, p, epi, grid);
    case 2: return launch_pi<2>(st, p, epi, grid);
    case 3: return launch_pi<1>(st, p, epi, grid);
    case 4: return launch_pi<0>(st, p, epi, grid);
    case 5: return launch_pi<5>(st, p, epi, grid);
    default: return false;
  }
}

// ---- "f16 + fp8" format (kernel sets 3 / 4 on the panel path): panel_f8_gemm_kernel / panel_f8_qkv_kernel ----------
namespace {
template <bool WLO>
bool launch_f8_t(hipStream_t st, const PanelParams& p, int epi, dim3 grid) {
  const dim3 block(256);
  if (epi == 103) hipLaunchKernelGGL((panel_f8_gemm_kernel<PE_GEGLU, WLO, 2>), grid, block, 0, st, p);  // h as (hi, lo) bf16 pieces
  else if (epi == PE_GEGLU) hipLaunchKernelGGL((panel_f8_gemm_kernel<PE_GEGLU, WLO, 1>), grid, block, 0, st, p);
  else if (epi == 100 || epi == 101) hipLaunchKernelGGL((panel_f8_gemm_kernel<PE_RESIDUAL, WLO, 0>), grid, block, 0, st, p);
  else return false;
  return true;
}
}  // namespace

bool launch_panel_f8(hipStream_t st, const PanelParams& p, int epi, bool wlo, dim3 grid) {
  return wlo ? launch_f8_t<true>(st, p, epi, grid) : launch_f8_t<false>(st, p, epi, grid);
}

bool launch_panel_f8_qkv(hipStream_t st, const PanelParams& p, bool wlo, bool o16, dim3 grid) {
  if (o16) {  // kernel sets 10 / 11: single-plane fp16 q, k, v^T for the fp16 attention kernels
    if (wlo) hipLaunchKernelGGL((panel_f8_qkv_kernel<true, 0, 0, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((panel_f8_qkv_kernel<false, 0, 0, true>), grid, dim3(256), 0, st, p);
    return true;
  }
  // q, k and v^T keep their (hi, lo) bf16 pieces: the attention kernels read them
  if (wlo) hipLaunchKernelGGL((panel_f8_qkv_kernel<true, 3, 1>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((panel_f8_qkv_kernel<false, 3, 1>), grid, dim3(256), 0, st, p);
  return true;
}

}  // namespace opl
