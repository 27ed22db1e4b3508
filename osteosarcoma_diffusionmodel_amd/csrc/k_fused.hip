// k_fused.hip -- input_proj (+t_emb +c_proj), output_proj (+posterior), output_proj (+MSE).
#include "kernels.h"
#include "launch.h"

namespace osd {

hipError_t launch_input(hipStream_t s, const GemmArgs& g, const EpiInput::Args& a, bool a_zero_padded) {
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, EpiInput>(s, g, a, a_zero_padded);
  if (use_tile64(g.F, g.P)) return launch_gemm<Tile64, true, true, EpiInput>(s, g, a, a_zero_padded);
  return launch_gemm<TileSmall, true, true, EpiInput>(s, g, a, a_zero_padded);
}
// ---- input_proj for small batches: split-K over workgroups ------------------------------------------------------------
// At sampling batches of ~1000 rows input_proj (K = D = 2000 ... 5142) is 16-64 output tiles of 63-161 sequential K steps: one
// launch of 113 us at the reference's default generation workload (1000 patients, D = 5142) on a machine of 256 CUs.  Here K is
// cut into `slices` ranges: slice y of tile (f, p) accumulates its range and stores the partial tile to slab y; k_input_reduce
// then sums the slabs in slice order and applies input_proj's epilogue  h = ((sum + b) + t_emb[t]) + c_proj  (EpiInput).  A
// different fp32 summation order than the single-pass kernels: results agree to ~1e-6 relative, not bitwise.
__global__ void k_input_reduce(const float* __restrict__ slabs, int slices, long long stride, EpiInput::Args a, int P, int F) {
  const int c4n = F >> 2;
  const long long total = (long long)P * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i / c4n);
    const int f = 4 * (int)(i - (long long)p * c4n);
    const float* sp = slabs + (size_t)p * F + f;
    float4 acc = *reinterpret_cast<const float4*>(sp);
    for (int k = 1; k < slices; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)k * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int t = a.t_index ? a.t_index[p] : (a.t_dev ? *a.t_dev : a.t_imm);
    const float4 b = *reinterpret_cast<const float4*>(a.bias + f);
    const float4 te = *reinterpret_cast<const float4*>(a.temb + (size_t)t * a.ldt + f);
    const float4 cp = *reinterpret_cast<const float4*>(a.cproj + (size_t)p * a.ldc + f);
    float4 o;
    o.x = ((acc.x + b.x) + te.x) + cp.x; o.y = ((acc.y + b.y) + te.y) + cp.y;
    o.z = ((acc.z + b.z) + te.z) + cp.z; o.w = ((acc.w + b.w) + te.w) + cp.w;
    *reinterpret_cast<float4*>(a.out + (size_t)p * a.ldo + f) = o;
  }
}

// slabs: slices x P x F floats.  hipErrorInvalidValue when the operands do not meet the preconditions (caller falls back)
hipError_t launch_input_splitk(hipStream_t s, const GemmArgs& g, const EpiInput::Args& a, float* slabs, int slices) {
  GemmArgs gs = g;
  gs.kchunk = ((g.K + slices - 1) / slices + BK - 1) / BK * BK;
  const int ns = (g.K + gs.kchunk - 1) / gs.kchunk;
  const long long stride = (long long)g.P * g.F;
  typedef EpiBias<false, false> E;
  const E::Args ea{nullptr, slabs, g.F, stride};
  if (g.K0 < g.K || !gemm_fast_ok(gs, true, true) || !E::fast_ok(ea, g.F) || !EpiInput::fast_ok(a, g.F) || a.ldt % 4) return hipErrorInvalidValue;
  hipError_t e = launch_gemm_v<Tile64, true, true, E, true>(s, gs, ea);
  if (e != hipSuccess) return e;
  const long long total = (long long)g.P * (g.F >> 2);
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_input_reduce, dim3(grid), dim3(256), 0, s, slabs, ns, stride, a, g.P, g.F);
  return hipGetLastError();
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
