// k_b3t.hip -- output_proj + MSE of a training step on the bf16 matrix pipe at fp32 accuracy (gemm_b3t.h; precision = 1).
#include "kernels.h"
#include "gemm_b3t.h"

namespace osd {

// hipErrorInvalidValue: shape or alignment outside this kernel -- the caller runs the fp32 launch
hipError_t launch_mse_b3t(hipStream_t s, const GemmArgs& g, const EpiMse::Args& a) {
  if (use_big_tile(g.F, g.P)) return launch_gemm_b3t<TileBig, EpiMse, 16>(s, g, a);
  return launch_gemm_b3t<Tile64, EpiMse, 32>(s, g, a);
}

}  // namespace osd
