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

// d gamma[c] += sum_r gy[r][c] * zhat[r][c], d beta[c] += sum_r gy[r][c] for every listed layer, one launch (targets zeroed by
// the caller; float atomics over the 64-row blocks)
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
