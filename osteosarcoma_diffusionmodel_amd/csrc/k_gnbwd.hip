// k_gnbwd.hip -- dgrad GEMMs whose epilogue is the GroupNorm+SiLU(+dropout) backward of the layer they feed (EpiGnBwd).
#include "kernels.h"
#include "kernels_train.h"
#include "launch.h"

namespace osd {

template <int GW, bool DROP>
static hipError_t gnbwd_go(hipStream_t s, const GemmArgs& g, const GnBwdEpi& a) {
  typedef EpiGnBwd<GW, DROP> E;
  typename E::Args ea{a.z, a.ldz, a.stats, a.gamma, a.beta, a.gz, a.ldg, a.gy, a.ldy, a.accumulate, a.drop_mode, a.mask, a.ldm,
                      a.keep_scale, a.p_drop, a.seed, a.row_offset, a.step, a.tag};
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, false, true, E>(s, g, ea);
  if constexpr (GW <= 32) {
    const long small_tiles = (long)((g.F + 63) / 64) * ((g.P + 127) / 128);
    if (small_tiles < 512) return launch_gemm<Tile64, false, true, E>(s, g, ea);      // 32-feature waves own whole groups
  }
  return launch_gemm<TileSmall, false, true, E>(s, g, ea);
}

bool dgrad_gnbwd_supported(int gw) { return gw == 32 || gw == 64; }

// out-features of the GEMM = channels of the layer whose GroupNorm backward runs in the epilogue; gw = its group width
hipError_t launch_dgrad_gnbwd(hipStream_t s, const GemmArgs& g, int gw, const GnBwdEpi& a) {
  const bool drop = a.drop_mode != 0;
  if (gw == 32) return drop ? gnbwd_go<32, true>(s, g, a) : gnbwd_go<32, false>(s, g, a);
  if (gw == 64) return drop ? gnbwd_go<64, true>(s, g, a) : gnbwd_go<64, false>(s, g, a);
  return hipErrorInvalidValue;
}

// ---- two dgrad GEMMs that read the same upstream gradient in ONE launch -----------------------------------------------------
// A decoder block's first Linear has two inputs (models/diffusion.py:250: cat[h, skip]), so its backward is two dgrads over the
// same gz: the main one (epilogue = GroupNorm backward of the layer below, on the critical path) and the skip connection's share
// (plain store, needed much later).  Launched one after the other each is a 256-tile launch of ~15 us on a 512-slot machine;
// here problem 2's tiles follow problem 1's in one grid, so the second slot of every CU works on the skip share meanwhile.
template <class T1, class E1, class T2, class E2>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_dual_kernel(GemmArgs g1, typename E1::Args e1, int grid1, GemmArgs g2, typename E2::Args e2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < grid1) {
    const int nft = (g1.F + T1::BF - 1) / T1::BF, npt = (g1.P + T1::BP - 1) / T1::BP;
    const int b = blockIdx.x, idx = b >> 3;
    const int ft = idx % nft, pt = (idx / nft) * 8 + (b & 7);
    if (pt >= npt) return;
    gemm_tile<T1, false, true, E1, true>(g1, e1, ft * T1::BF, pt * T1::BP, smem);
  } else {
    const int nft = (g2.F + T2::BF - 1) / T2::BF, npt = (g2.P + T2::BP - 1) / T2::BP;
    const int b = blockIdx.x - grid1, idx = b >> 3;
    const int ft = idx % nft, pt = (idx / nft) * 8 + (b & 7);
    if (pt >= npt) return;
    gemm_tile<T2, false, true, E2, true>(g2, e2, ft * T2::BF, pt * T2::BP, smem);
  }
}

template <class T1, int GW>
static hipError_t dual_go(hipStream_t s, const GemmArgs& g1, const GnBwdEpi& a, const GemmArgs& g2, float* out2, int ldo2) {
  typedef EpiGnBwd<GW, false> E1;
  typedef EpiBias<false, false> E2;
  typedef Tile64 T2;
  const typename E1::Args e1{a.z, a.ldz, a.stats, a.gamma, a.beta, a.gz, a.ldg, a.gy, a.ldy, a.accumulate, 0, nullptr, 0,
                             a.keep_scale, a.p_drop, a.seed, a.row_offset, a.step, a.tag};
  const E2::Args e2{nullptr, out2, ldo2, 0};
  if (!gemm_fast_ok(g1, false, true) || !E1::fast_ok(e1, g1.F) || !gemm_fast_ok(g2, false, true) || !E2::fast_ok(e2, g2.F)) return hipErrorInvalidValue;
  constexpr int lds = T1::LDS_BYTES > T2::LDS_BYTES ? T1::LDS_BYTES : T2::LDS_BYTES;
  auto kern = gemm_dual_kernel<T1, E1, T2, E2>;
  static bool attr_set[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const int grid1 = gemm_grid(g1.F, g1.P, T1::BF, T1::BP), grid2 = gemm_grid(g2.F, g2.P, T2::BF, T2::BP);
  hipLaunchKernelGGL(kern, dim3(grid1 + grid2), dim3(NTHREADS), lds, s, g1, e1, grid1, g2, e2);
  return hipGetLastError();
}

// main dgrad (GroupNorm backward epilogue of width gw, no dropout behind that layer) + plain dgrad into out2, one launch.
// hipErrorInvalidValue: shapes / alignment outside this kernel's tile code -- the caller launches the two separately.
hipError_t launch_dgrad_gnbwd_dual(hipStream_t s, const GemmArgs& g1, int gw, const GnBwdEpi& a, const GemmArgs& g2, float* out2, int ldo2) {
  if (a.drop_mode != 0 || use_big_tile(g1.F, g1.P) || use_big_tile(g2.F, g2.P)) return hipErrorInvalidValue;
  if (gw == 32) return dual_go<Tile64, 32>(s, g1, a, g2, out2, ldo2);
  if (gw == 64) return dual_go<TileSmall, 64>(s, g1, a, g2, out2, ldo2);
  return hipErrorInvalidValue;
}

// d gamma[c] += sum_r gy[r][c] * zhat[r][c], d beta[c] += sum_r gy[r][c] for every listed layer, one launch (targets zeroed by
// the caller; float atomics over the 64-row blocks).  (Tried: 16-byte loads, eight rows in flight, the row lanes of a block reduced in
// LDS before the atomics -- the kernel itself 41 -> 37 us, the STEP 0.915 -> 0.93 ms in two of three same-box runs: it runs on the side
// stream beside the dgrad tail, and its fewer, fatter blocks get in that chain's way.  One atomic per thread instead: 98 us.)
__global__ void k_gn_colsums(const GnColItem* __restrict__ items) {
  const GnColItem it = items[blockIdx.z];
  const int64_t r0 = (int64_t)blockIdx.y * 64;
  if (r0 >= it.rows) return;
  const int64_t r1 = r0 + 64 < it.rows ? r0 + 64 : it.rows;
  const int ngrp = it.C / it.gw;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < it.C; c += gridDim.x * blockDim.x) {
    const int grp = c / it.gw;
    float sb = 0.f, sg = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float gy = it.gy[r * it.ldy + c];
      const float2 st = *reinterpret_cast<const float2*>(it.stats + (r * ngrp + grp) * 2);
      const float zh = (it.z[r * it.ldz + c] - st.x) * st.y;
      sb += gy;
      sg += gy * zh;
    }
    atomicAdd(it.dbeta + c, sb);
    atomicAdd(it.dgamma + c, sg);
  }
}
hipError_t launch_gn_colsums(hipStream_t s, const GnColItem* d_items, int n_items, int64_t max_rows) {
  if (n_items <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_gn_colsums, dim3(2, (unsigned)((max_rows + 63) / 64), (unsigned)n_items), dim3(256), 0, s, d_items);
  return hipGetLastError();
}

}  // namespace osd
