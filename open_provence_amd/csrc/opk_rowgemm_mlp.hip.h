// opk_rowgemm_mlp.hip.h -- RowGemmBlock::mlp_phase(): the whole MLP of a layer between phase 1 and the chunk loop, h kept on chip
// (whole-layer kernel, RP_MLP).  Its two sections -- the loop's operations (stage DMA, GeGLU micro-operations, MFMA steps) and
// the macro-iterations -- are opk_rowgemm_mlp_ops.inc / opk_rowgemm_mlp_loop.inc, included in place: the operations are
// closures over the loop's register arrays.
#pragma once

namespace opk {

OPK_RG_TPL __device__ __forceinline__ void OPK_RG_BLOCK::mlp_phase() {
  const std::true_type yes_{};
  const std::false_type no_{};
  {
    {
#include "opk_rowgemm_mlp_ops.inc"
#include "opk_rowgemm_mlp_loop.inc"
    }
  }
}

}  // namespace opk
