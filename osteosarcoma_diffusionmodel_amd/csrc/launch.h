// launch.h -- host-side launcher for gemm_kernel instantiations.
#pragma once
#include <vector>
#include "gemm.h"
#include "gemm_glds.h"
#include <type_traits>

namespace osd {

typedef Tile<128, 128, 64, 64> TileBig;     // 2x2 waves of 64f x 64p; sampling-sized batches
typedef Tile<64, 128, 64, 32> TileSmall;    // 1x4 waves of 64f x 32p; training-sized batches
typedef Tile<128, 128, 128, 32> TileWide;   // 1x4 waves of 128f x 32p; GroupNorm groups of 128
typedef Tile<64, 64, 32, 32> Tile64;        // 2x2 waves of 32f x 32p; backward GEMMs at training batch sizes

inline int gemm_grid(int F, int P, int BF, int BP) {
  const int nft = (F + BF - 1) / BF, npt = (P + BP - 1) / BP;
  return ((npt + 7) / 8) * 8 * nft;
}

// true when the big tile already gives every CU at least one workgroup
inline bool use_big_tile(int F, int P) {
  const long tiles = (long)((F + 127) / 128) * ((P + 127) / 128);
  return tiles >= 256;
}
// training-sized batches: true when even the 64x128 tile leaves CUs idle and 64x64 tiles should be used
inline bool use_tile64(int F, int P) {
  const long tiles = (long)((F + 63) / 64) * ((P + 127) / 128);
  return tiles < 256;
}

// Every instantiation registers itself at load time; prepare_kernels() (called from
// osd_create, never inside a stream capture) raises each kernel's dynamic-LDS limit.
struct KernelReg { const void* fn; int lds_bytes; };
std::vector<KernelReg>& kernel_registry();
hipError_t prepare_kernels();

template <class T, bool AKC, bool BKC, class Epi, bool FAST, int NG = 1>
struct GemmRegistrar {
  GemmRegistrar() { kernel_registry().push_back({reinterpret_cast<const void*>(gemm_kernel<T, AKC, BKC, Epi, FAST, NG>), NG * T::LDS_BYTES}); }
  static GemmRegistrar instance;
};
template <class T, bool AKC, bool BKC, class Epi, bool FAST, int NG>
GemmRegistrar<T, AKC, BKC, Epi, FAST, NG> GemmRegistrar<T, AKC, BKC, Epi, FAST, NG>::instance;

// which (tile, epilogue) pairs get the two-wave-group instantiations: the epilogue opts in (static constexpr bool KSPLIT2)
template <class E, class = void> struct epi_ksplit2 : std::false_type {};
template <class E> struct epi_ksplit2<E, std::void_t<decltype(E::KSPLIT2)>> : std::integral_constant<bool, E::KSPLIT2> {};

inline bool ptr_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Preconditions of the branch-free (FAST) operand loads, see gemm.h.
inline bool gemm_fast_ok(const GemmArgs& g, bool a_kc, bool b_kc) {
  if (!ptr_al16(g.A) || !ptr_al16(g.B0) || g.lda % 4 || g.ldb0 % 4 || g.K < 4) return false;
  const bool two = b_kc && g.K0 < g.K;
  if (a_kc) { if (g.K % 4) return false; } else { if (g.F % 4 || g.F < 4) return false; }
  if (b_kc) {
    if (g.K % 4) return false;
    if (two && (g.K0 % BK || (g.K - g.K0) % 4 || g.K - g.K0 < 4 || !ptr_al16(g.B1) || g.ldb1 % 4)) return false;
  } else {
    if (g.P % 4 || g.P < 4) return false;
  }
  return true;
}

template <class T, bool AKC, bool BKC, class Epi, bool FAST>
hipError_t launch_gemm_v(hipStream_t s, const GemmArgs& g, const typename Epi::Args& ea) {
  (void)&GemmRegistrar<T, AKC, BKC, Epi, FAST>::instance;   // odr-use: forces the registration
  auto kern = gemm_kernel<T, AKC, BKC, Epi, FAST>;
  const int grid = gemm_grid(g.F, g.P, T::BF, T::BP);
  const int slices = g.kchunk > 0 ? (g.K + g.kchunk - 1) / g.kchunk : 1;
  if constexpr (epi_ksplit2<Epi>::value && T::BF * T::BP <= 64 * 64 && FAST) {
    // two wave groups (gemm.h): about one tile per CU and a long K loop -- the first dgrad of a training step (K = D = 2000: 63 K tiles)
    if (g.ksplit && slices == 1 && grid <= 320 && g.K >= 32 * BK) {
      (void)&GemmRegistrar<T, AKC, BKC, Epi, FAST, 2>::instance;
      hipLaunchKernelGGL((gemm_kernel<T, AKC, BKC, Epi, FAST, 2>), dim3(grid), dim3(2 * NTHREADS), 2 * T::LDS_BYTES, s, g, ea);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid, slices), dim3(NTHREADS), T::LDS_BYTES, s, g, ea);
  return hipGetLastError();
}

// Preconditions of the direct-to-LDS forward kernel (gemm_glds.h).  a_zero_padded: the caller
// guarantees A is readable and zero for k in [K, roundup(K, BK)).
inline bool glds_ok(const GemmArgs& g, bool a_zero_padded) {
  if (!ptr_al16(g.A) || !ptr_al16(g.B0) || g.lda % 4 || g.ldb0 % 4 || g.K < 4 || g.K % 4) return false;
  if (g.K % BK && !a_zero_padded && g.a_kmax <= 0) return false;
  if (g.a_kmax > 0 && (g.a_kmax % 4 || g.a_kmax < 4 || g.a_kmax > g.K || g.K % BK)) return false;
  if (g.K0 < g.K && (g.K0 % BK || (g.K - g.K0) % 4 || g.K - g.K0 < 4 || !ptr_al16(g.B1) || g.ldb1 % 4)) return false;
  return true;
}

template <class T, class Epi, int NG = 1>
struct GldsRegistrar {
  GldsRegistrar() { kernel_registry().push_back({reinterpret_cast<const void*>(gemm_glds_kernel<T, Epi, NG>), NG * GldsTile<T>::LDS_BYTES}); }
  static GldsRegistrar instance;
};
template <class T, class Epi, int NG>
GldsRegistrar<T, Epi, NG> GldsRegistrar<T, Epi, NG>::instance;

int persist_mode();   // OSD_PERSIST env: 0 (default) one tile per workgroup; 1 / 2: persistent tile walk on 512 / 256 workgroups

template <class T, class Epi>
hipError_t launch_gemm_glds(hipStream_t s, const GemmArgs& g0, const typename Epi::Args& ea) {
  (void)&GldsRegistrar<T, Epi>::instance;
  GemmArgs g = g0;
  int grid = gemm_grid(g.F, g.P, T::BF, T::BP);
  if constexpr (epi_ksplit2<Epi>::value && T::BF * T::BP <= 64 * 128) {
    // two wave groups: only worth it when the launch has about one tile per CU and a LONG K loop to split -- measured at the
    // training batch: input_proj (63 K steps) 58 -> 49 us, the 1024-deep decoder layer 32 -> 28 us, but 8-16 K steps +-0 (the
    // second group's prologue, the hand-over through LDS and its barrier cost what the shorter loop saves)
    if (g.ksplit && grid <= 320 && g.K >= 32 * BK) {
      (void)&GldsRegistrar<T, Epi, 2>::instance;
      g.persist = 0;
      hipLaunchKernelGGL((gemm_glds_kernel<T, Epi, 2>), dim3(grid), dim3(2 * NTHREADS), 2 * GldsTile<T>::LDS_BYTES, s, g, ea);
      return hipGetLastError();
    }
  }
  // persistent patient-tile walk: 512 workgroups (2 per CU), each keeping one feature tile
  const int nft = (g.F + T::BF - 1) / T::BF;
  g.persist = 0;
  const int pm = persist_mode();          // 1: 512 workgroups, 2: 256 (leaves room for a second stream's kernel)
  const int pgrid = pm == 2 ? 256 : 512;
  if (pm && !g.tri && grid > pgrid && nft <= pgrid / 8 && (pgrid / 8) % nft == 0) { g.persist = 1; grid = pgrid; }
  hipLaunchKernelGGL((gemm_glds_kernel<T, Epi>), dim3(grid), dim3(NTHREADS), GldsTile<T>::LDS_BYTES, s, g, ea);
  return hipGetLastError();
}

// a_zero_padded only matters for the forward (KC x KC) layout.  ALLOW_GLDS = false keeps a (tile, epilogue) pair off the
// LDS-DMA kernel: its instantiation is then not even compiled (used for pairs that would not fit 256 VGPRs there).
template <class T, bool AKC, bool BKC, class Epi, bool ALLOW_GLDS = true>
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g, const typename Epi::Args& ea, bool a_zero_padded = false) {
  if (g.F <= 0 || g.P <= 0) return hipSuccess;
  if constexpr (AKC && BKC && ALLOW_GLDS) {
    if (glds_ok(g, a_zero_padded) && Epi::fast_ok(ea, g.F)) {
      return launch_gemm_glds<T, Epi>(s, g, ea);
    }
  }
  if (g.a_kmax > 0) return hipErrorInvalidValue;     // a clamped A operand exists in the LDS-DMA kernel only: the caller must not get here
  if (gemm_fast_ok(g, AKC, BKC) && Epi::fast_ok(ea, g.F)) return launch_gemm_v<T, AKC, BKC, Epi, true>(s, g, ea);
  return launch_gemm_v<T, AKC, BKC, Epi, false>(s, g, ea);
}

}  // namespace osd
