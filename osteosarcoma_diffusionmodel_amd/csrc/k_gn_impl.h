// k_gn_impl.h -- shared body of k_gn.hip (DROP=false) and k_gn_drop.hip (DROP=true).
#include "kernels.h"
#include "launch.h"

namespace osd {

// Tile choice.  The 128 x 128 tile of the LDS-DMA kernel holds 64 accumulators per lane; with the dropout code or with many
// small groups (GW <= 16: 8+ statistics pairs per row block) on top, the epilogue does not fit the 256-VGPR budget of two
// workgroups per CU and spills, so those pairs take the 64 x 128 tile (32 accumulators) -- and the two that do not fit
// there either (groups of 128: a wave must span 128 features; groups of 8 with dropout) stay on the register-staged kernel.
template <int GW, bool DROP>
static hipError_t gn_go(hipStream_t s, const GemmArgs& g, const GnArgs& a) {
  typedef EpiGnSilu<GW, DROP> E;
  typename E::Args ea{a.bias, a.gamma, a.beta, a.out, a.ldo, a.z_out, a.ldz, a.stats, a.drop_mode, a.mask, a.ldm,
                      a.keep_scale, a.p_drop, a.seed, a.row_offset, a.step, a.tag, a.step_dev};
  if constexpr (GW > 64) {
    return launch_gemm<TileWide, true, true, E, false>(s, g, ea);
  } else {
    constexpr bool BIG_OK = !DROP && GW >= 32;
    if constexpr (BIG_OK) {
      if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, E>(s, g, ea);
    }
    if constexpr (GW <= 32) {
      if (use_tile64(g.F, g.P)) return launch_gemm<Tile64, true, true, E>(s, g, ea);      // 32-feature waves own whole groups
    }
    return launch_gemm<TileSmall, true, true, E, !(DROP && GW <= 8)>(s, g, ea);
  }
}

template <bool DROP>
static hipError_t gn_dispatch(hipStream_t s, const GemmArgs& g, int gw, const GnArgs& a) {
  switch (gw) {
    case 4: return gn_go<4, DROP>(s, g, a);
    case 8: return gn_go<8, DROP>(s, g, a);
    case 16: return gn_go<16, DROP>(s, g, a);
    case 32: return gn_go<32, DROP>(s, g, a);
    case 64: return gn_go<64, DROP>(s, g, a);
    case 128: return gn_go<128, DROP>(s, g, a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace osd
