// op_launch_attn.hip -- instantiations of attn_fp_kernel for every curated precision policy.
#include "op_internal.h"

namespace opl {
using namespace opk;

namespace {

template <int PI, int WAVES, int KT>
void launch_one(hipStream_t st, const AttnFpParams& p, dim3 grid) {
  constexpr Policy P = kPolicies[PI];
  // Tile stages of the sliding-window layers: 2.  Three (requests two tiles ahead, counted end-of-tile wait; -DOPK_ATTN_
  // LOCAL_STAGES=3) measured SLOWER, 0.773 vs 0.719 ms per forward for the six launches (same-box A/B, three
  // alternations): the 48 KiB ring costs a resident block per CU, and residency hides the DMA latency better than depth.
#ifndef OPK_ATTN_LOCAL_STAGES
#define OPK_ATTN_LOCAL_STAGES 2
#endif
  constexpr int NST = KT == 1 ? OPK_ATTN_LOCAL_STAGES : 2;
  hipLaunchKernelGGL((attn_fp_kernel<P.qk, P.pv, o_lo(P), WAVES, KT, false, P.fmt == 1, NST, P.fmt == 2>), grid, dim3(WAVES * 64), 0, st, p);
}

template <int PI>
bool launch_pi(hipStream_t st, const AttnFpParams& p, int waves, int kt, dim3 grid) {
  if (waves == 8 && kt == 2) launch_one<PI, 8, 2>(st, p, grid);
  else if (waves == 4 && kt == 2) launch_one<PI, 4, 2>(st, p, grid);
  else if (waves == 4 && kt == 1) launch_one<PI, 4, 1>(st, p, grid);
  else return false;
  return true;
}

}  // namespace

bool launch_attn(hipStream_t st, const AttnFpParams& p, int waves, int kt, int pi, bool zero_p_lo, dim3 grid, bool f16_in_f8_out) {
  if (f16_in_f8_out) {
    if (zero_p_lo || (pi != PI_F16_F8 && pi != PI_F16_F8_W)) return false;
    if (waves == 8 && kt == 2) hipLaunchKernelGGL((attn_fp_kernel<0, 0, false, 8, 2, false, true, 2, true>), grid, dim3(512), 0, st, p);
    else if (waves == 4 && kt == 2) hipLaunchKernelGGL((attn_fp_kernel<0, 0, false, 4, 2, false, true, 2, true>), grid, dim3(256), 0, st, p);
    else if (waves == 4 && kt == 1)
      hipLaunchKernelGGL((attn_fp_kernel<0, 0, false, 4, 1, false, true, OPK_ATTN_LOCAL_STAGES, true>), grid, dim3(256), 0, st, p);
    else return false;
    return true;
  }
  if (zero_p_lo) {  // all-terms instantiation with lo(p) cleared
    if (pi != 0) return false;
    if (waves == 8 && kt == 2) hipLaunchKernelGGL((attn_fp_kernel<3, 3, true, 8, 2, true>), grid, dim3(512), 0, st, p);
    else if (waves == 4 && kt == 2) hipLaunchKernelGGL((attn_fp_kernel<3, 3, true, 4, 2, true>), grid, dim3(256), 0, st, p);
    else if (waves == 4 && kt == 1) hipLaunchKernelGGL((attn_fp_kernel<3, 3, true, 4, 1, true>), grid, dim3(256), 0, st, p);
    else return false;
    return true;
  }
  static_assert(N_POLICIES == 6, "extend the switch below");
  switch (pi) {
    case 5: return launch_pi<5>(st, p, waves, kt, grid);
    case 0: return launch_pi<0>(st, p, waves, kt, grid);
    case 1: return launch_pi<1>(st, p, waves, kt, grid);
    case 2: return launch_pi<2>(st, p, waves, kt, grid);
    case 3: return launch_pi<3>(st, p, waves, kt, grid);
    case 4: return launch_pi<4>(st, p, waves, kt, grid);
    default: return false;
  }
}

}  // namespace opl
