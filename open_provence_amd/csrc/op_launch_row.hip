// op_launch_row.hip -- instantiations of the row-stationary GEMM kernels (hidden 128 / 256) for every curated
// precision policy: layer-0 q/k/v projection, the two fused per-layer kernels and the last layer's k-streamed
// MLP output projection.  OPL_ROW_PART selects a subset so that the build can compile the parts in parallel.
#include "op_internal.h"

namespace opl {
using namespace opk;

namespace {

template <int KS, int EPI, int PRO, int T1, int T2, int OLO>
void launch_shape(hipStream_t st, const RowGemmParams& p, bool small, unsigned grid) {
  // 4 waves x 32 rows = 128-row blocks, two per CU.  Small batches (at most one such block per CU) use 4 waves x
  // 16 rows = 64-row blocks instead: twice the blocks, so a latency-bound request spreads over twice the CUs.
  if (small)
    hipLaunchKernelGGL((rowgemm_kernel<KS, EPI, PRO, T1, T2, OLO, 4, 1>), dim3(grid), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((rowgemm_kernel<KS, EPI, PRO, T1, T2, OLO, 4, 2>), dim3(grid), dim3(256), 0, st, p);
}

template <int EPI, int PRO, int T1, int T2, int OLO>
bool launch_ks(hipStream_t st, const RowGemmParams& p, int ks, bool small, unsigned grid) {
  if (ks == 8) launch_shape<8, EPI, PRO, T1, T2, OLO>(st, p, small, grid);
  else if (ks == 4) launch_shape<4, EPI, PRO, T1, T2, OLO>(st, p, small, grid);
  else return false;
  return true;
}

}  // namespace

// (kernel set 3 = the operand terms of set 1; only the whole-layer kernel has its own instantiation: OPL_ROW_PART 4)
// (kernel set "f16" = PI_F16 has its own launchers: OPL_ROW_PART 6; the two-kernels-per-layer forms do not exist for it)
#define OPL_SWITCH(CALL)                              \
  static_assert(N_POLICIES == 6, "extend the switch"); \
  switch (pi) {                                       \
    case 0: return CALL(0);                           \
    case 1: return CALL(1);                           \
    case 2: return CALL(2);                           \
    case 3: return CALL(1);                           \
    case 4: return CALL(0);                           \
    default: return false;                            \
  }

#if OPL_ROW_PART == 0
bool launch_row_geglu_fused(hipStream_t st, const RowGemmParams& p, int ks, bool small, int pi, unsigned grid) {
#define OPL_CALL(PI) (launch_ks<RE_GEGLU, RP_KSTREAM, kPolicies[PI].attn_out, kPolicies[PI].wi, h_olo(kPolicies[PI])>(st, p, ks, small, grid))
  OPL_SWITCH(OPL_CALL)
#undef OPL_CALL
}
#endif

#if OPL_ROW_PART == 1
bool launch_row_qkv_fused(hipStream_t st, const RowGemmParams& p, int ks, bool small, int pi, unsigned grid) {
#define OPL_CALL(PI) (launch_ks<RE_QKV, RP_KSTREAM, kPolicies[PI].mlp_out, kPolicies[PI].wqkv, qkv_olo(kPolicies[PI])>(st, p, ks, small, grid))
  OPL_SWITCH(OPL_CALL)
#undef OPL_CALL
}
#endif

#if OPL_ROW_PART == 2
bool launch_row_qkv0(hipStream_t st, const RowGemmParams& p, int ks, bool small, int pi, unsigned grid) {
  if (pi == PI_F16) return launch_row_qkv0_h16(st, p, ks, small, grid);
#define OPL_CALL(PI) (launch_ks<RE_QKV, RP_SPLIT, 0, kPolicies[PI].wqkv, qkv_olo(kPolicies[PI])>(st, p, ks, small, grid))
  OPL_SWITCH(OPL_CALL)
#undef OPL_CALL
}

namespace {
template <int T>
bool launch_kstream_t(hipStream_t st, const KStreamParams& p, int nf, unsigned grid) {
  if (nf == 16) hipLaunchKernelGGL((kstream_gemm_kernel<16, T, 4>), dim3(grid), dim3(256), 0, st, p);
  else if (nf == 8) hipLaunchKernelGGL((kstream_gemm_kernel<8, T, 4>), dim3(grid), dim3(256), 0, st, p);
  else return false;
  return true;
}
}  // namespace

bool launch_kstream(hipStream_t st, const KStreamParams& p, int nf, int pi, unsigned grid) {
#define OPL_CALL(PI) (launch_kstream_t<kPolicies[PI].mlp_out>(st, p, nf, grid))
  OPL_SWITCH(OPL_CALL)
#undef OPL_CALL
}
#endif

#if OPL_ROW_PART == 3
// Whole-layer kernel (attention output projection + MLP + next layer's q/k/v projection, h kept on chip): kernel sets
// whose weights are single-plane only (bf16 checkpoints / bf16x2 / bf16), 4 waves x 32 rows per block, one block per CU.
namespace {
template <int PI, int KS>
bool launch_layer_ks(hipStream_t st, const RowGemmParams& p, bool with_qkv, unsigned grid, bool waves8) {
  constexpr Policy P = kPolicies[PI];
  if constexpr ((P.wi & 2) != 0 || (P.mlp_out & 2) != 0) {
    return false;
  } else {
    if (waves8 && with_qkv)
      hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, P.attn_out, P.wqkv, qkv_olo(P), 8, 1, P.wi, P.mlp_out>), dim3(grid),
                         dim3(512), 0, st, p);
    else if (waves8)
      hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, P.attn_out, 0, 0, 8, 1, P.wi, P.mlp_out>), dim3(grid), dim3(512), 0,
                         st, p);
    else if (with_qkv)
      hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, P.attn_out, P.wqkv, qkv_olo(P), 4, 2, P.wi, P.mlp_out>), dim3(grid),
                         dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, P.attn_out, 0, 0, 4, 2, P.wi, P.mlp_out>), dim3(grid), dim3(256), 0,
                         st, p);
    return true;
  }
}
template <int PI>
bool launch_layer_pi(hipStream_t st, const RowGemmParams& p, int ks, bool with_qkv, unsigned grid, bool waves8) {
  if (ks == 8) return launch_layer_ks<PI, 8>(st, p, with_qkv, grid, waves8);
  if (ks == 4) return launch_layer_ks<PI, 4>(st, p, with_qkv, grid, waves8);
  return false;
}
}  // namespace

bool has_row_layer_fused(int pi) {
  return pi == PI_F16_F8_W || pi == PI_F16 || (pi >= 0 && pi < N_POLICIES && (kPolicies[pi].wi & 2) == 0 && (kPolicies[pi].mlp_out & 2) == 0);
}

bool launch_row_layer_fused(hipStream_t st, const RowGemmParams& p, int ks, int pi, bool with_qkv, unsigned grid, bool waves8) {
  if (pi == PI_F16_F8) return !waves8 && launch_row_layer_f8(st, p, ks, with_qkv, grid);
  if (pi == PI_F16_F8_W) return !waves8 && launch_row_layer_f8w(st, p, ks, with_qkv, grid);
  if (pi == PI_F16) return launch_row_layer_h16(st, p, ks, with_qkv, grid, waves8);
#define OPL_CALL(PI) (launch_layer_pi<PI>(st, p, ks, with_qkv, grid, waves8))
  OPL_SWITCH(OPL_CALL)
#undef OPL_CALL
}
#endif

#if OPL_ROW_PART == 4
// Whole-layer kernel of the "f16 + fp8" kernel set (4 waves x 32 rows, one block per CU, 132 KiB of LDS).
namespace {
template <int KS>
void launch_layer_f8_ks(hipStream_t st, const RowGemmParams& p, bool with_qkv, unsigned grid) {
  constexpr Policy P = kPolicies[PI_F16_F8];
  if (with_qkv)
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, P.attn_out, P.wqkv, qkv_olo(P), 4, 2, P.wi, P.mlp_out, 1>), dim3(grid),
                       dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, P.attn_out, 0, 0, 4, 2, P.wi, P.mlp_out, 1>), dim3(grid), dim3(256), 0,
                       st, p);
}
}  // namespace

bool launch_row_layer_f8(hipStream_t st, const RowGemmParams& p, int ks, bool with_qkv, unsigned grid) {
  if (ks == 8) launch_layer_f8_ks<8>(st, p, with_qkv, grid);
  else if (ks == 4) launch_layer_f8_ks<4>(st, p, with_qkv, grid);
  else return false;
  return true;
}
#endif

#if OPL_ROW_PART == 5
// Whole-layer kernel of kernel set 4: fp32-valued weights, fp16 + e4m3 operands with the weights' lo part (F8 = 2).
namespace {
template <int KS>
void launch_layer_f8w_ks(hipStream_t st, const RowGemmParams& p, bool with_qkv, unsigned grid) {
  constexpr Policy P = kPolicies[PI_F16_F8_W];
  if (with_qkv)
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, P.attn_out, P.wqkv, qkv_olo(P), 4, 2, P.wi, P.mlp_out, 2>), dim3(grid),
                       dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, P.attn_out, 0, 0, 4, 2, P.wi, P.mlp_out, 2>), dim3(grid), dim3(256), 0,
                       st, p);
}
}  // namespace

bool launch_row_layer_f8w(hipStream_t st, const RowGemmParams& p, int ks, bool with_qkv, unsigned grid) {
  if (ks == 8) launch_layer_f8w_ks<8>(st, p, with_qkv, grid);
  else if (ks == 4) launch_layer_f8w_ks<4>(st, p, with_qkv, grid);
  else return false;
  return true;
}
#endif

#if OPL_ROW_PART == 6
// Kernel set "f16" (PI_F16): the single-pass instantiations with fp16 operands (rowgemm_kernel<..., H16 = true>).
namespace {
template <int KS>
void launch_qkv0_h16_ks(hipStream_t st, const RowGemmParams& p, bool small, unsigned grid) {
  if (small) hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_SPLIT, 0, 0, 0, 4, 1, 0, 0, 0, true>), dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_SPLIT, 0, 0, 0, 4, 2, 0, 0, 0, true>), dim3(grid), dim3(256), 0, st, p);
}
template <int KS>
void launch_layer_h16_ks(hipStream_t st, const RowGemmParams& p, bool with_qkv, unsigned grid, bool waves8) {
  if (waves8 && with_qkv)
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, 0, 0, 0, 8, 1, 0, 0, 0, true>), dim3(grid), dim3(512), 0, st, p);
  else if (waves8)
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, 0, 0, 0, 8, 1, 0, 0, 0, true>), dim3(grid), dim3(512), 0, st, p);
  else if (with_qkv)
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_QKV, RP_MLP, 0, 0, 0, 4, 2, 0, 0, 0, true>), dim3(grid), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((rowgemm_kernel<KS, RE_NONE, RP_MLP, 0, 0, 0, 4, 2, 0, 0, 0, true>), dim3(grid), dim3(256), 0, st, p);
}
}  // namespace

bool launch_row_qkv0_h16(hipStream_t st, const RowGemmParams& p, int ks, bool small, unsigned grid) {
  if (ks == 8) launch_qkv0_h16_ks<8>(st, p, small, grid);
  else if (ks == 4) launch_qkv0_h16_ks<4>(st, p, small, grid);
  else return false;
  return true;
}

bool launch_row_layer_h16(hipStream_t st, const RowGemmParams& p, int ks, bool with_qkv, unsigned grid, bool waves8) {
  if (ks == 8) launch_layer_h16_ks<8>(st, p, with_qkv, grid, waves8);
  else if (ks == 4) launch_layer_h16_ks<4>(st, p, with_qkv, grid, waves8);
  else return false;
  return true;
}
#endif

}  // namespace opl
