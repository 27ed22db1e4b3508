// launch.h -- host-side launcher for gemm_kernel instantiations.
#pragma once
#include <vector>
#include "gemm.h"

namespace osd {

typedef Tile<128, 128, 64, 64> TileBig;     // 2x2 waves of 64f x 64p; sampling-sized batches
typedef Tile<64, 128, 64, 32> TileSmall;    // 1x4 waves of 64f x 32p; training-sized batches
typedef Tile<128, 128, 128, 32> TileWide;   // 1x4 waves of 128f x 32p; GroupNorm groups of 128

inline int gemm_grid(int F, int P, int BF, int BP) {
  const int nft = (F + BF - 1) / BF, npt = (P + BP - 1) / BP;
  return ((npt + 7) / 8) * 8 * nft;
}

// true when the big tile already gives every CU at least one workgroup
inline bool use_big_tile(int F, int P) {
  const long tiles = (long)((F + 127) / 128) * ((P + 127) / 128);
  return tiles >= 256;
}

// Every instantiation registers itself at load time; prepare_kernels() (called from
// osd_create, never inside a stream capture) raises each kernel's dynamic-LDS limit.
struct KernelReg { const void* fn; int lds_bytes; };
std::vector<KernelReg>& kernel_registry();
hipError_t prepare_kernels();

template <class T, bool AKC, bool BKC, class Epi>
struct GemmRegistrar {
  GemmRegistrar() { kernel_registry().push_back({reinterpret_cast<const void*>(gemm_kernel<T, AKC, BKC, Epi>), T::LDS_BYTES}); }
  static GemmRegistrar instance;
};
template <class T, bool AKC, bool BKC, class Epi>
GemmRegistrar<T, AKC, BKC, Epi> GemmRegistrar<T, AKC, BKC, Epi>::instance;

template <class T, bool AKC, bool BKC, class Epi>
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g, const typename Epi::Args& ea) {
  (void)&GemmRegistrar<T, AKC, BKC, Epi>::instance;   // odr-use: forces the registration
  auto kern = gemm_kernel<T, AKC, BKC, Epi>;
  if (g.F <= 0 || g.P <= 0) return hipSuccess;
  const int grid = gemm_grid(g.F, g.P, T::BF, T::BP);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), T::LDS_BYTES, s, g, ea);
  return hipGetLastError();
}

}  // namespace osd
