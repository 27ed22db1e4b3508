// k_gn_drop.hip -- Linear+GroupNorm(8)+SiLU+Dropout (train mode, first half of every block).
#include "k_gn_impl.h"
namespace osd {
hipError_t launch_gn_silu_drop(hipStream_t s, const GemmArgs& g, int gw, const GnArgs& a) { return gn_dispatch<true>(s, g, gw, a); }
}  // namespace osd
