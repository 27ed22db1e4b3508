// k_fused.hip -- input_proj (+t_emb +c_proj), output_proj (+posterior), output_proj (+MSE).
#include <stdlib.h>
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

// ---- Linear + GroupNorm(8) + SiLU for small batches: split-K over workgroups + one reduce kernel --------------------------------
// At the reference's generation sizes (333 - 1000 patients per scenario) a 512-deep Linear+GroupNorm layer is 64 output tiles of 16
// sequential K steps on a machine with 512 workgroup slots: 20 us of which the matrix pipes are busy for a fraction.  Here every
// K panel is cut into slices (blockIdx.y of gemm_kernel): slice y of a tile stores its partial sums to slab y; k_gn_reduce then
// adds the slabs in slice order, the bias, and applies GroupNorm(8) + SiLU (models/diffusion.py:200-204; EpiGnSilu's arithmetic on
// another fp32 summation order -- an option, like the input_proj split, never the default).  One wave per row: a lane owns C / 64
// consecutive channels, so each of the 8 groups is 8 adjacent lanes and its statistics are three xor-shuffles.
template <int CPL>       // channels per lane: C = 64 * CPL
__global__ __launch_bounds__(256) void k_gn_reduce(const float* __restrict__ slabs, int slices, long long stride, const float* __restrict__ bias,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
                                                   int ldo, int P) {
  constexpr int C = 64 * CPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  float v[CPL];
  const float* sp = slabs + (size_t)row * C + lane * CPL;
#pragma unroll
  for (int j = 0; j < CPL; j += 4) {
    const float4 a = *reinterpret_cast<const float4*>(sp + j);
    v[j] = a.x; v[j + 1] = a.y; v[j + 2] = a.z; v[j + 3] = a.w;
  }
  for (int k = 1; k < slices; ++k) {
#pragma unroll
    for (int j = 0; j < CPL; j += 4) {
      const float4 a = *reinterpret_cast<const float4*>(sp + (size_t)k * stride + j);
      v[j] += a.x; v[j + 1] += a.y; v[j + 2] += a.z; v[j + 3] += a.w;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) { v[j] += bias[lane * CPL + j]; s += v[j]; }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o);
  const float m = s * (1.0f / (8 * CPL));
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) { const float d = v[j] - m; q = fmaf(d, d, q); }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) q += __shfl_xor(q, o);
  const float r = 1.0f / sqrtf(q * (1.0f / (8 * CPL)) + GN_EPS);
  float* op = out + (size_t)row * ldo + lane * CPL;
#pragma unroll
  for (int j = 0; j < CPL; j += 4) {
    float4 y;
    y.x = silu_f(fmaf((v[j] - m) * r, gamma[lane * CPL + j], beta[lane * CPL + j]));
    y.y = silu_f(fmaf((v[j + 1] - m) * r, gamma[lane * CPL + j + 1], beta[lane * CPL + j + 1]));
    y.z = silu_f(fmaf((v[j + 2] - m) * r, gamma[lane * CPL + j + 2], beta[lane * CPL + j + 2]));
    y.w = silu_f(fmaf((v[j + 3] - m) * r, gamma[lane * CPL + j + 3], beta[lane * CPL + j + 3]));
    *reinterpret_cast<float4*>(op + j) = y;
  }
}

// slabs: room for `slices` x P x F floats per panel (two panels: 2 x slices).  hipErrorInvalidValue when the shape is outside the
// path (the caller then takes the single-pass kernel): F in {256, 512, 1024}, eval mode, aligned operands.
hipError_t launch_gn_silu_splitk(hipStream_t s, const GemmArgs& g, const GnArgs& a, float* slabs, int slices) {
  if (a.z_out || a.drop_mode != 0 || (g.F != 256 && g.F != 512 && g.F != 1024) || a.ldo % 4 || !al16(a.out) || slices < 2) return hipErrorInvalidValue;
  typedef EpiBias<false, false> E;
  const long long stride = (long long)g.P * g.F;
  int n_slabs = 0;
  // one split GEMM per K panel (gemm_kernel's split-K walks a single panel)
  const int panels = g.K0 < g.K ? 2 : 1;
  for (int pnl = 0; pnl < panels; ++pnl) {
    GemmArgs gs = g;
    const int kp = panels == 1 ? g.K : (pnl == 0 ? g.K0 : g.K - g.K0);
    if (pnl == 1) { gs.A = g.A + g.K0; gs.B0 = g.B1; gs.ldb0 = g.ldb1; }
    gs.B1 = nullptr; gs.ldb1 = 0; gs.K = kp; gs.K0 = kp; gs.ksplit = 0;
    const int want = panels == 1 ? slices : (slices + 1) / 2;
    gs.kchunk = ((kp + want - 1) / want + BK - 1) / BK * BK;
    const int ns = (kp + gs.kchunk - 1) / gs.kchunk;
    const E::Args ea{nullptr, slabs + (long long)n_slabs * stride, g.F, stride};
    if (!gemm_fast_ok(gs, true, true) || !E::fast_ok(ea, g.F)) return hipErrorInvalidValue;
    hipError_t e = launch_gemm_v<Tile64, true, true, E, true>(s, gs, ea);
    if (e != hipSuccess) return e;
    n_slabs += ns;
  }
  const int grid = (g.P + 3) / 4;
  if (g.F == 256) hipLaunchKernelGGL(k_gn_reduce<4>, dim3(grid), dim3(256), 0, s, slabs, n_slabs, stride, a.bias, a.gamma, a.beta, a.out, a.ldo, g.P);
  else if (g.F == 512) hipLaunchKernelGGL(k_gn_reduce<8>, dim3(grid), dim3(256), 0, s, slabs, n_slabs, stride, a.bias, a.gamma, a.beta, a.out, a.ldo, g.P);
  else hipLaunchKernelGGL(k_gn_reduce<16>, dim3(grid), dim3(256), 0, s, slabs, n_slabs, stride, a.bias, a.gamma, a.beta, a.out, a.ldo, g.P);
  return hipGetLastError();
}

hipError_t launch_posterior(hipStream_t s, const GemmArgs& g, const EpiPosterior::Args& a) {
  // below one full round of 128 x 128 tiles (512 workgroup slots) the 64 x 128 tile spreads the same work over twice the workgroups
  static const long big_from = [] { const char* e = getenv("OSD_POST_BIG_FROM"); return e ? atol(e) : 512L; }();
  const long big_tiles = (long)((g.F + 127) / 128) * ((g.P + 127) / 128);
  if (big_tiles >= big_from) return launch_gemm<TileBig, true, true, EpiPosterior>(s, g, a);
  return launch_gemm<TileSmall, true, true, EpiPosterior>(s, g, a);
}
hipError_t launch_mse(hipStream_t s, const GemmArgs& g, const EpiMse::Args& a) {
  if (use_big_tile(g.F, g.P)) return launch_gemm<TileBig, true, true, EpiMse>(s, g, a);
  return launch_gemm<TileSmall, true, true, EpiMse>(s, g, a);
}

}  // namespace osd
