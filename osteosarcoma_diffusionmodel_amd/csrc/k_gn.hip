// k_gn.hip -- Linear+GroupNorm(8)+SiLU, no dropout (eval mode, second half of every block).
#include "k_gn_impl.h"
namespace osd {
bool gn_width_supported(int gw) { return gw == 4 || gw == 8 || gw == 16 || gw == 32 || gw == 64 || gw == 128; }
hipError_t launch_gn_silu(hipStream_t s, const GemmArgs& g, int gw, const GnArgs& a) { return gn_dispatch<false>(s, g, gw, a); }
}  // namespace osd
