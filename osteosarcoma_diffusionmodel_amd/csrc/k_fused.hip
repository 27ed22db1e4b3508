// k_fused.hip -- input_proj (+t_emb +c_proj), output_proj (+posterior), output_proj (+MSE).
#include "kernels.h"
#include "launch.h"

namespace osd {

hipError_t launch_input(hipStream_t s, const GemmArgs& g, const EpiInput::Args& a, bool a_zero_padded) {
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, EpiInput>(s, g, a, a_zero_padded);
  if (use_tile64(g.F, g.P)) return launch_gemm<Tile64, true, true, EpiInput>(s, g, a, a_zero_padded);
  return launch_gemm<TileSmall, true, true, EpiInput>(s, g, a, a_zero_padded);
}
hipError_t launch_posterior(hipStream_t s, const GemmArgs& g, const EpiPosterior::Args& a) {
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, EpiPosterior>(s, g, a);
  return launch_gemm<TileSmall, true, true, EpiPosterior>(s, g, a);
}
hipError_t launch_mse(hipStream_t s, const GemmArgs& g, const EpiMse::Args& a) {
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, EpiMse>(s, g, a);
  return launch_gemm<TileSmall, true, true, EpiMse>(s, g, a);
}

}  // namespace osd
