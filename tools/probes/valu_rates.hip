// Probe: issue cost (cycles per wave-instruction, one wave per SIMD, independent chains) of the VALU operations the epilogues use.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <stdint.h>
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  float f[8]; uint32_t u[8]; uint64_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { f[j] = 1.0f + 0.01f * (lane + j); u[j] = 0x9E3779B9u * (lane + j + 1); w[j] = u[j]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) f[j] = __builtin_fmaf(f[j], 0.999f, 0.001f);
      if (OP == 1) f[j] = __builtin_amdgcn_exp2f(f[j]) * 0.25f;                        // + 1 mul
      if (OP == 2) f[j] = __builtin_amdgcn_rcpf(f[j]) + 0.5f;                           // + 1 add
      if (OP == 3) f[j] = __builtin_amdgcn_sqrtf(f[j]) + 1.0f;
      if (OP == 4) f[j] = __builtin_amdgcn_logf(f[j]) + 2.0f;
      if (OP == 5) f[j] = __builtin_amdgcn_sinf(f[j]) + 1.5f;
      if (OP == 6) u[j] = __umulhi(u[j], 0xD2511F53u) + 1u;
      if (OP == 7) u[j] = u[j] * 0xD2511F53u + 1u;
      if (OP == 8) { const uint64_t p = (uint64_t)u[j] * 0xD2511F53u; u[j] = (uint32_t)(p >> 32) ^ (uint32_t)p; }   // mad_u64_u32 + xor
      if (OP == 9) u[j] = (u[j] ^ 0x12345u) + (u[j] >> 3);                              // xor, shift, add
      if (OP == 10) { float2 a = make_float2(f[j], f[j ^ 1]); a.x = a.x * 0.999f + 0.001f; a.y = a.y * 0.999f + 0.001f; f[j] = a.x; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) r += f[j] + (float)u[j] + (float)w[j];
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}
template <int OP> void run(const char* name, int extra, unsigned long long* out, float* sink) {
  const int grid = 256, iters = 4000;
  std::vector<unsigned long long> h(grid * 4);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, iters, out, sink); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h.data(), out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double a = 0; for (auto v : h) a += v;
  a /= grid * 4.0 * iters * 8;
  printf("%-44s %6.1f cycles per iteration element (%d full-rate op(s) included)\n", name, a, extra);
}
int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 4 * 8); (void)hipMalloc(&sink, 64);
  run<0>("v_fma_f32", 0, out, sink); run<1>("v_exp_f32 + v_mul", 1, out, sink); run<2>("v_rcp_f32 + v_add", 1, out, sink);
  run<3>("v_sqrt_f32 + v_add", 1, out, sink); run<4>("v_log_f32 + v_add", 1, out, sink); run<5>("v_sin_f32 + v_add", 1, out, sink);
  run<6>("v_mul_hi_u32 + v_add", 1, out, sink); run<7>("v_mul_lo_u32 + v_add (or v_mad_u32_u24?)", 1, out, sink);
  run<8>("v_mad_u64_u32 + v_xor", 1, out, sink); run<9>("xor + shift + add", 3, out, sink);
  return 0;
}
