// Probe: the fp32-accuracy K loop on the bf16 matrix pipe (csrc/gemm_bf3.h) against the fp32 MFMA tile kernel
// (csrc/gemm_glds.h) on single layers of the denoiser at sampling batch.
//   1. numerics: both kernels against an fp64 host reference (max |delta| / max |ref|), K = 256 / 512 / 2000
//   2. rate: per layer shape, microseconds per launch and effective TFLOP/s (2 K N rows FLOP) of both kernels, full chip,
//      with the real epilogues (GroupNorm + SiLU + plane split / posterior + Philox) and with an empty epilogue (K loop alone)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../osteosarcoma_diffusionmodel_amd/csrc -o split_probe split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "gemm_bf3.h"
#include "b3_pack.h"

using namespace osd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// empty epilogue: the K loop alone (one conditional store keeps the accumulators alive)
struct EpiB3Null : EpiB3Base {
  struct Args { float* sink; };
  static __device__ __forceinline__ const float* prm_ptr(const Args&, int) { return nullptr; }
  static __device__ __forceinline__ void apply(f32x16 (&acc)[2][2], const Args& a, const B3Ctx& c, B3NoState&) {
    const int lane = c.lane;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e-30f) a.sink[lane] = s;
  }
};
struct EpiNull32 {
  static constexpr bool COUNTED_STORES = false;
  static constexpr bool XBUF = false;
  struct Args { float* sink; };
  template <int NFB> struct Pre {};
  template <int NFB, bool FAST> static __device__ __forceinline__ Pre<NFB> prefetch(const Args&, int, int, int) { return {}; }
  template <int NFB, int NPB, bool FAST>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>&, int, int, int lane, int, int) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
      for (int j = 0; j < NPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e-30f) a.sink[lane] = s;
  }
};

typedef Tile<128, 128, 64, 64> TileBig;

static int grid_for(int F, int P) {
  const int nft = (F + 127) / 128, npt = (P + 127) / 128;
  return ((npt + 7) / 8) * 8 * nft;
}

template <class K, class... A>
static float time_kernel(K kern, int grid, int block, int lds, int reps, A... args) {
  hipEvent_t e0, e1;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, args...);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, args...);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

static uint4* pack(const float* d_src, int ld, long long R, int K) {
  const int nkb = b3_nkb(K);
  const long long units = b3_units(R, K);
  uint4* d = nullptr;
  CK(hipMalloc(&d, (size_t)units * 16));
  const long long total = b3_tiles(R) * nkb * 256;
  hipLaunchKernelGGL(k_b3_pack, dim3((unsigned)std::min<long long>((total + 255) / 256, 65535)), dim3(256), 0, 0, d_src, ld, R, K, d, nkb, total);
  CK(hipGetLastError());
  return d;
}

static void numerics(int F, int P, int K, bool positive_x, int ld) {
  std::mt19937 rng(1234 + K);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> W((size_t)F * K), X((size_t)P * K), bias(F);
  const float ws = 1.0f / std::sqrt((float)K);
  for (auto& w : W) w = nd(rng) * ws;
  for (auto& x : X) { x = nd(rng); if (positive_x) x = x / (1.f + std::exp(-x)); }
  for (auto& b : bias) b = nd(rng) * 0.1f;
  float *dW, *dX, *dB, *dO3, *dO32;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dB, F * 4));
  CK(hipMalloc(&dO3, (size_t)P * F * 4)); CK(hipMalloc(&dO32, (size_t)P * F * 4));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, bias.data(), F * 4, hipMemcpyHostToDevice));
  uint4* pW = pack(dW, K, F, K);
  uint4* pX = pack(dX, K, P, K);
  Bf3Args g{pW, b3_nkb(K), pX, b3_nkb(K), nullptr, 0, F, P, nullptr};
  EpiB3Bias::Args ea{dB, dO3, F};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<EpiB3Bias, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<EpiB3Bias, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES));
  if (ld == 0) hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Bias, 0>), dim3(grid_for(F, P)), dim3(256), B3_LDS_BYTES, 0, g, ea);
  else hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Bias, 2>), dim3(grid_for(F, P)), dim3(384), B3_LDS_BYTES, 0, g, ea);
  CK(hipGetLastError());
  // fp32 MFMA kernel on the same operands (K % 32 == 0 or zero-padded copy)
  const int Kp = (K + 31) / 32 * 32;
  float *dWp, *dXp;
  CK(hipMalloc(&dWp, (size_t)F * Kp * 4)); CK(hipMalloc(&dXp, (size_t)P * Kp * 4));
  CK(hipMemset(dWp, 0, (size_t)F * Kp * 4)); CK(hipMemset(dXp, 0, (size_t)P * Kp * 4));
  CK(hipMemcpy2D(dWp, Kp * 4, dW, K * 4, K * 4, F, hipMemcpyDeviceToDevice));
  CK(hipMemcpy2D(dXp, Kp * 4, dX, K * 4, K * 4, P, hipMemcpyDeviceToDevice));
  GemmArgs g32{};
  g32.A = dWp; g32.lda = Kp; g32.B0 = dXp; g32.ldb0 = Kp; g32.K0 = Kp; g32.F = F; g32.P = P; g32.K = Kp;
  typedef EpiBias<false, false> E32;
  E32::Args e32{dB, dO32, F, 0};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<TileBig, E32>), hipFuncAttributeMaxDynamicSharedMemorySize, GldsTile<TileBig>::LDS_BYTES));
  hipLaunchKernelGGL((gemm_glds_kernel<TileBig, E32>), dim3(grid_for(F, P)), dim3(256), GldsTile<TileBig>::LDS_BYTES, 0, g32, e32);
  CK(hipGetLastError());
  std::vector<float> O3((size_t)P * F), O32((size_t)P * F);
  CK(hipMemcpy(O3.data(), dO3, O3.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(O32.data(), dO32, O32.size() * 4, hipMemcpyDeviceToHost));
  double mref = 0, e3 = 0, e32m = 0, rms3 = 0, rms32 = 0;
  for (int p = 0; p < P; ++p)
    for (int f = 0; f < F; ++f) {
      double s = bias[f];
      const float* w = &W[(size_t)f * K];
      const float* x = &X[(size_t)p * K];
      for (int k = 0; k < K; ++k) s += (double)w[k] * (double)x[k];
      mref = std::max(mref, std::fabs(s));
      const double d3 = std::fabs((double)O3[(size_t)p * F + f] - s), d32 = std::fabs((double)O32[(size_t)p * F + f] - s);
      e3 = std::max(e3, d3); e32m = std::max(e32m, d32);
      rms3 += d3 * d3; rms32 += d32 * d32;
    }
  const double n = (double)P * F;
  printf("numerics LD=%d F=%d P=%d K=%d %s: max|ref| %.3f | bf16x3 max %.3e rms %.3e | fp32 MFMA max %.3e rms %.3e   (relative to max|ref|)\n", ld, F, P, K,
         positive_x ? "x=silu(N(0,1))" : "x=N(0,1)", mref, e3 / mref, std::sqrt(rms3 / n) / mref, e32m / mref, std::sqrt(rms32 / n) / mref);
  for (void* q : {(void*)dW, (void*)dX, (void*)dB, (void*)dO3, (void*)dO32, (void*)pW, (void*)pX, (void*)dWp, (void*)dXp}) CK(hipFree(q));
}

// one layer shape: K -> N over P rows; kind 0 = K loop alone, 1 = GroupNorm + SiLU (N = 256 / 512), 2 = output_proj + posterior (Philox)
template <class Epi, int NKB = 0>
static void time_b3(int grid, int reps, const Bf3Args& g, const typename Epi::Args& ea, float* us) {
  us[0] = time_kernel(gemm_bf3_kernel<Epi, 0, NKB>, grid, 256, B3_LDS_BYTES, reps, g, ea);
  us[1] = B3_RING == 3 ? time_kernel(gemm_bf3_kernel<Epi, 2, NKB>, grid, 384, B3_LDS_BYTES, reps, g, ea) : us[0];      // the loader variant is written for three slots
}
static void rate(int K, int N, int P, int kind, int reps) {
  const int Kp = (K + 31) / 32 * 32;
  float *dW, *dX, *dPar, *dOut, *dSink, *dCoef, *dXs;
  unsigned long long* dSt;
  CK(hipMalloc(&dW, (size_t)N * Kp * 4)); CK(hipMalloc(&dX, (size_t)P * Kp * 4));
  CK(hipMalloc(&dPar, (size_t)3 * 4096 * 4)); CK(hipMalloc(&dOut, (size_t)P * N * 4)); CK(hipMalloc(&dSink, 4096)); CK(hipMalloc(&dCoef, 4096 * 16));
  CK(hipMalloc(&dXs, (size_t)P * N * 4));
  const int grid = grid_for(N, P);
  CK(hipMalloc(&dSt, (size_t)grid * 32));
  {
    std::vector<float> h((size_t)N * Kp);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = nd(rng) / std::sqrt((float)K);
    CK(hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)4096 * Kp);
    for (auto& v : hx) v = nd(rng);
    for (long long r = 0; r < P; r += 4096) CK(hipMemcpy(dX + (size_t)r * Kp, hx.data(), (size_t)std::min<long long>(4096, P - r) * Kp * 4, hipMemcpyHostToDevice));
    std::vector<float> par(3 * 4096, 0.5f), coef(4096 * 4, 0.3f);
    CK(hipMemcpy(dPar, par.data(), par.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dCoef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dXs, 0, (size_t)P * N * 4));
  }
  uint4* pW = pack(dW, Kp, N, K);
  uint4* pX = pack(dX, Kp, P, K);
  uint4* pO = nullptr;
  CK(hipMalloc(&pO, (size_t)b3_units(P, N) * 16));
  CK(hipDeviceSynchronize());
  Bf3Args g{pW, b3_nkb(K), pX, b3_nkb(K), nullptr, 0, N, P, nullptr};
  GemmArgs g32{};
  g32.A = dW; g32.lda = Kp; g32.B0 = dX; g32.ldb0 = Kp; g32.K0 = Kp; g32.F = N; g32.P = P; g32.K = Kp;
  float us3[2] = {0, 0}, us32 = 0;
  double ghz[2] = {0, 0};
  const char* name = "";
  const int l32 = GldsTile<TileBig>::LDS_BYTES;
  if (kind == 0) {
    name = "K loop alone";
    time_b3<EpiB3Null>(grid, reps, g, EpiB3Null::Args{dSink}, us3);
    us32 = time_kernel(gemm_glds_kernel<TileBig, EpiNull32>, grid, 256, l32, reps, g32, EpiNull32::Args{dSink});
    // shader clock during the K loop: s_memtime cycles per s_memrealtime tick (100 MHz), median workgroup
    for (int ld = 0; ld < (B3_RING == 3 ? 2 : 1); ++ld) {
      Bf3Args gs = g; gs.stamps = dSt;
      if (ld == 0) hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Null, 0>), dim3(grid), dim3(256), B3_LDS_BYTES, 0, gs, EpiB3Null::Args{dSink});
      else hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Null, 2>), dim3(grid), dim3(384), B3_LDS_BYTES, 0, gs, EpiB3Null::Args{dSink});
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> st((size_t)grid * 4);
      CK(hipMemcpy(st.data(), dSt, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> f;
      for (int b = 0; b < grid; ++b) if (st[4 * b + 1] > 0) f.push_back((double)st[4 * b] / (double)st[4 * b + 1] * 0.1);
      std::sort(f.begin(), f.end());
      ghz[ld] = f.empty() ? 0 : f[f.size() / 2];
    }
  } else if (kind == 1) {
    name = "GroupNorm+SiLU";
    B3Out o{pO, b3_nkb(N)};
    if (N / 8 == 64) {
      time_b3<EpiB3Gn<64>>(grid, reps, g, EpiB3Gn<64>::Args{dPar, dPar + 4096, dPar + 8192, o}, us3);
      EpiGnSilu<64, false>::Args e{}; e.bias = dPar; e.gamma = dPar + 4096; e.beta = dPar + 8192; e.out = dOut; e.ldo = N;
      us32 = time_kernel(gemm_glds_kernel<TileBig, EpiGnSilu<64, false>>, grid, 256, l32, reps, g32, e);
    } else {
      time_b3<EpiB3Gn<32>>(grid, reps, g, EpiB3Gn<32>::Args{dPar, dPar + 4096, dPar + 8192, o}, us3);
      EpiGnSilu<32, false>::Args e{}; e.bias = dPar; e.gamma = dPar + 4096; e.beta = dPar + 8192; e.out = dOut; e.ldo = N;
      us32 = time_kernel(gemm_glds_kernel<TileBig, EpiGnSilu<32, false>>, grid, 256, l32, reps, g32, e);
    }
  } else {
    name = "posterior+Philox";
    EpiB3Post::Args e{}; e.bias = dPar; e.x = dXs; e.ldx = N; e.coef = dCoef; e.t_imm = 500; e.seed = 1; e.o = B3Out{pO, b3_nkb(N)};
    e.x_tile = (N % 4 == 0 && B3_RING == 3) ? 1 : 0;
    if (K == 256) time_b3<EpiB3Post, 16>(grid, reps, g, e, us3);
    else time_b3<EpiB3Post, 0>(grid, reps, g, e, us3);
    {   // the same launch with the direct x accesses and the generator behind the loop (the first version of this epilogue)
      float u2[2];
      e.x_tile = 0;
      time_b3<EpiB3Post, 0>(grid, reps, g, e, u2);
      printf("     (posterior with direct x accesses and the generator in the epilogue: LD0 %.1f us, LD2 %.1f us)\n", u2[0], u2[1]);
    }
    EpiPosterior::Args e2{}; e2.bias = dPar; e2.xin = dXs; e2.ldx = N; e2.xout = dXs; e2.ldo = N; e2.coef = dCoef; e2.t_imm = 500; e2.ldzz = N; e2.seed = 1; e2.t_first = 500;
    us32 = time_kernel(gemm_glds_kernel<TileBig, EpiPosterior>, grid, 256, l32, reps, g32, e2);
  }
  if (kind != 0) {
    Bf3Args gs = g; gs.stamps = dSt;
    CK(hipMemset(dSt, 0, (size_t)grid * 32));
    if (kind == 1) {
      B3Out o{pO, b3_nkb(N)};
      if (N / 8 == 64) hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Gn<64>, 0, 0>), dim3(grid), dim3(256), B3_LDS_BYTES, 0, gs, EpiB3Gn<64>::Args{dPar, dPar + 4096, dPar + 8192, o});
      else hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Gn<32>, 0, 0>), dim3(grid), dim3(256), B3_LDS_BYTES, 0, gs, EpiB3Gn<32>::Args{dPar, dPar + 4096, dPar + 8192, o});
    } else {
      EpiB3Post::Args e{}; e.bias = dPar; e.x = dXs; e.ldx = N; e.coef = dCoef; e.t_imm = 500; e.seed = 1; e.o = B3Out{pO, b3_nkb(N)}; e.x_tile = B3_RING == 3 ? 1 : 0;
      if (K == 256) hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Post, 0, 16>), dim3(grid), dim3(256), B3_LDS_BYTES, 0, gs, e);
      else hipLaunchKernelGGL((gemm_bf3_kernel<EpiB3Post, 0, 0>), dim3(grid), dim3(256), B3_LDS_BYTES, 0, gs, e);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)grid * 4);
    CK(hipMemcpy(st.data(), dSt, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> kc, ec;
    for (int b = 0; b < grid; ++b) if (st[4 * b + 1] > 0) { kc.push_back((double)st[4 * b]); ec.push_back((double)st[4 * b + 2]); }
    std::sort(kc.begin(), kc.end()); std::sort(ec.begin(), ec.end());
    if (!kc.empty()) printf("     stamps (LD0, median workgroup): prologue + K loop %.0f cycles, epilogue + drain %.0f cycles (MFMA time of the tile alone: %d)\n",
                            kc[kc.size() / 2], ec[ec.size() / 2], b3_nkb(K) * 768);
  }
  const double flop = 2.0 * K * N * (double)P;
  printf("rate K=%4d N=%4d P=%d %-16s: bf16x3 LD0 %7.1f us %6.1f TF | LD2 %7.1f us %6.1f TF (%.3f of 417) | fp32 %7.1f us %6.1f TF (%.3f of 157.3) | speedup %.2fx / %.2fx", K, N, P,
         name, us3[0], flop / us3[0] * 1e-6, us3[1], flop / us3[1] * 1e-6, flop / us3[1] * 1e-6 / 417.0, us32, flop / us32 * 1e-6, flop / us32 * 1e-6 / 157.3,
         us32 / us3[0], us32 / us3[1]);
  if (kind == 0) printf(" | clock %.2f / %.2f GHz", ghz[0], ghz[1]);
  printf("\n");
  for (void* q : {(void*)dW, (void*)dX, (void*)dPar, (void*)dOut, (void*)dSink, (void*)dCoef, (void*)dXs, (void*)pW, (void*)pX, (void*)pO, (void*)dSt}) CK(hipFree(q));
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 65536;
  if (argc > 2 && argv[2][0] == 'p') {      // the posterior launch only
    rate(256, 2000, P, 2, 20);
    return 0;
  }
  if (argc > 2) {           // K loops only (the -DB3_EXP=n timing variants)
    rate(512, 512, P, 0, 20);
    rate(256, 256, P, 0, 20);
    rate(2000, 256, P, 0, 20);
    rate(256, 2000, P, 0, 20);
    return 0;
  }
  for (int ld = 0; ld <= (B3_RING == 3 ? 2 : 0); ld += 2) {
    numerics(512, 256, 256, true, ld);
    numerics(512, 256, 512, true, ld);
    numerics(256, 200, 2000, false, ld);
    numerics(200, 130, 40, false, ld);
  }
  const int reps = 20;
  rate(512, 512, P, 0, reps);
  rate(512, 512, P, 1, reps);
  rate(256, 256, P, 0, reps);
  rate(256, 256, P, 1, reps);
  rate(256, 512, P, 1, reps);
  rate(1024, 256, P, 1, reps);
  rate(2000, 256, P, 0, reps);
  rate(256, 2000, P, 0, reps);
  rate(256, 2000, P, 2, reps);
  return 0;
}
