// k_linear.hip -- plain Linear / generic GEMM instantiations (bias, optional SiLU, optional accumulate).
#include "kernels.h"
#include "launch.h"

namespace osd {

template <bool AKC, bool BKC, bool SILU, bool ACC>
static hipError_t go(hipStream_t s, const GemmArgs& g, const float* bias, float* out, int ldo, long long slice_stride = 0) {
  typename EpiBias<SILU, ACC>::Args ea{bias, out, ldo, slice_stride};
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, AKC, BKC, EpiBias<SILU, ACC>>(s, g, ea);
  const long small_tiles = (long)((g.F + 63) / 64) * ((g.P + 127) / 128) * (g.kchunk > 0 ? (g.K + g.kchunk - 1) / g.kchunk : 1);
  if constexpr (!(AKC && BKC)) {
    // backward layouts: 64x64 tiles when the 64x128 grid would leave most CUs idle
    if (small_tiles < 512) return launch_gemm<Tile64, AKC, BKC, EpiBias<SILU, ACC>>(s, g, ea);
  }
  return launch_gemm<TileSmall, AKC, BKC, EpiBias<SILU, ACC>>(s, g, ea);
}

// wgrad with the batch reduction split over blockIdx.y: slice y writes slab + y*slice_stride
hipError_t launch_wgrad_splitk(hipStream_t s, const GemmArgs& g, float* slabs, int ldo, long long slice_stride) {
  return go<false, false, false, false>(s, g, nullptr, slabs, ldo, slice_stride);
}

hipError_t launch_linear(hipStream_t s, const GemmArgs& g, bool a_kc, bool b_kc, const float* bias,
                         float* out, int ldo, bool silu, bool accumulate) {
  if (a_kc && b_kc) {            // forward
    if (accumulate) return hipErrorInvalidValue;
    return silu ? go<true, true, true, false>(s, g, bias, out, ldo) : go<true, true, false, false>(s, g, bias, out, ldo);
  }
  if (silu) return hipErrorInvalidValue;
  if (!a_kc && b_kc)             // dgrad
    return accumulate ? go<false, true, false, true>(s, g, bias, out, ldo) : go<false, true, false, false>(s, g, bias, out, ldo);
  if (!a_kc && !b_kc)            // wgrad
    return accumulate ? go<false, false, false, true>(s, g, bias, out, ldo) : go<false, false, false, false>(s, g, bias, out, ldo);
  return accumulate ? go<true, false, false, true>(s, g, bias, out, ldo) : go<true, false, false, false>(s, g, bias, out, ldo);
}

}  // namespace osd
