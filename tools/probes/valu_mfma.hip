// Probe: does VALU work of one wave slow the fp32 MFMAs of another wave on the same SIMD (and the reverse)?
//   hipcc --offload-arch=gfx950 -O3 -o valu_mfma valu_mfma.hip && ./valu_mfma
// One 8-wave workgroup per CU (LDS request forces it): waves 0-3 issue independent v_mfma_f32_32x32x2_f32, waves 4-7
// (the second wave of each SIMD) issue one of: independent v_fma_f32 chains, v_exp_f32, ds_write/ds_read_b128, global stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void probe(int do_mfma, int other, int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float res = 0.f;
  if (wave < 4) {
    if (do_mfma) {
      f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
      const float x = 1.0f + lane * 1e-3f, y = 0.5f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // do_mfma: 1 back to back; 2 an s_nop 0 after every MFMA; 3 an s_sleep 0; 4 s_sleep 1 after every fourth
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          if (do_mfma == 2) asm volatile("s_nop 0"); else if (do_mfma == 3) __builtin_amdgcn_s_sleep(0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
          if (do_mfma == 2) asm volatile("s_nop 0"); else if (do_mfma == 3) __builtin_amdgcn_s_sleep(0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
          if (do_mfma == 2) asm volatile("s_nop 0"); else if (do_mfma == 3) __builtin_amdgcn_s_sleep(0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
          if (do_mfma == 2) asm volatile("s_nop 0"); else if (do_mfma == 3) __builtin_amdgcn_s_sleep(0); else if (do_mfma == 4) __builtin_amdgcn_s_sleep(1);
        }
      }
      res = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (other >= 10) { __builtin_amdgcn_s_setprio(3); other -= 10; }
    if (other == 1) {            // 16 VALU fma per iteration, 8 independent chains
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = lane + j;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) res += v[j];
    } else if (other == 2) {     // 16 v_exp_f32 per iteration
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (lane + j) * 1e-3f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.25f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) res += v[j];
    } else if (other == 3) {     // 8 ds_write_b128 + 8 ds_read_b128 per iteration (conflict-free linear)
      float4* p = reinterpret_cast<float4*>(smem) + (wave - 4) * 1024 + lane;
      float4 v = make_float4(lane, 1, 2, 3);
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p[64 * j] = v;
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float4 r = p[64 * j]; s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w; }
        v = s;
      }
      res = v.x + v.y;
    } else if (other == 4 || other == 5) {     // 8 global float4 stores per iteration: 4 = full 128-byte row segments, 5 = 32 rows x 32 bytes
      float* base = sink + ((size_t)blockIdx.x * 4 + (wave - 4)) * 64 * 512 + 4096;
      const float4 v = make_float4(lane, 1, 2, 3);
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float* q = (other == 4) ? base + (size_t)(8 * j + (lane >> 3)) * 512 + 4 * (lane & 7)
                                  : base + (size_t)(lane & 31) * 512 + 8 * j + 4 * (lane >> 5) + ((j >> 2) ? 2048 * 8 : 0);
          *reinterpret_cast<float4*>(q) = v;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (res == 12345.678f) sink[0] = res;
}

// same wave: each MFMA followed by KV independent v_fma_f32
template <int KV>
__global__ __launch_bounds__(256) void probe_same(int iters, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  const float x = 1.0f + lane * 1e-3f, y = 0.5f;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = lane + j;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < KV; ++j) v[j % 16] = __builtin_fmaf(v[j % 16], 0.999f, 0.001f);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < KV; ++j) v[(j + 4) % 16] = __builtin_fmaf(v[(j + 4) % 16], 0.999f, 0.001f);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < KV; ++j) v[(j + 8) % 16] = __builtin_fmaf(v[(j + 8) % 16], 0.999f, 0.001f);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < KV; ++j) v[(j + 12) % 16] = __builtin_fmaf(v[(j + 12) % 16], 0.999f, 0.001f);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float res = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
  for (int j = 0; j < 16; ++j) res += v[j];
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (res == 12345.678f) sink[0] = res;
}
template <int KV>
static void run_same(unsigned long long* out, float* sink, std::vector<unsigned long long>& h) {
  const int grid = 256, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe_same<KV>, dim3(grid), dim3(256), 0, 0, iters, out, sink); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
  double a = 0;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < 4; ++w) a += h[g * 8 + w];
  printf("same wave, %2d v_fma_f32 after every MFMA: %6.1f cycles per MFMA\n", KV, a / (grid * 4.0 * iters * 16));
}

int main() {
  const int grid = 256, iters = 4000;
  unsigned long long* out; float* sink;
  hipMalloc(&out, grid * 8 * 8); hipMalloc(&sink, (size_t)grid * 4 * 64 * 512 * 4 + (1 << 22));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  std::vector<unsigned long long> h(grid * 8);
  const char* names[] = {"-", "16 v_fma_f32", "16 v_exp_f32", "8 ds_write_b128 + 8 ds_read_b128", "8 full-line global stores", "8 quarter-line global stores"};
  for (int other : {0, 1, 11, 2, 3, 13, 4, 5})
    for (int do_mfma = 0; do_mfma <= 1; ++do_mfma) {
      if (!other && !do_mfma) continue;
      if (other >= 10 && !do_mfma) continue;
      if (other >= 10) printf("(other waves at s_setprio 3) ");
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 100 * 1024, 0, do_mfma, other, iters, out, sink);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
      double a = 0, b = 0;
      for (int g = 0; g < grid; ++g) { for (int w = 0; w < 4; ++w) a += h[g * 8 + w]; for (int w = 4; w < 8; ++w) b += h[g * 8 + w]; }
      a /= grid * 4.0 * iters; b /= grid * 4.0 * iters;
      printf("mfma %d  other %-34s : MFMA waves %8.1f cycles / 16 MFMAs (%.1f per MFMA)   other waves %8.1f cycles / iteration\n", do_mfma, names[other % 10], a, a / 16, b);
    }
  printf("-- MFMA wave yielding (2: s_nop 0 after every MFMA, 3: s_sleep 0, 4: s_sleep 1 after every fourth), other wave 16 v_fma_f32 / 8+8 LDS ops:\n");
  for (int mode = 2; mode <= 4; ++mode)
    for (int other : {1, 3}) {
      for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 100 * 1024, 0, mode, other, iters, out, sink); (void)hipDeviceSynchronize(); }
      (void)hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
      double a = 0, b = 0;
      for (int g = 0; g < grid; ++g) { for (int w = 0; w < 4; ++w) a += h[g * 8 + w]; for (int w = 4; w < 8; ++w) b += h[g * 8 + w]; }
      a /= grid * 4.0 * iters; b /= grid * 4.0 * iters;
      printf("yield mode %d  other %-34s : MFMA waves %8.1f cycles / 16 MFMAs (%.1f per MFMA)   other waves %8.1f cycles / iteration\n", mode, names[other], a, a / 16, b);
    }
  run_same<0>(out, sink, h); run_same<1>(out, sink, h); run_same<2>(out, sink, h); run_same<4>(out, sink, h);
  run_same<8>(out, sink, h); run_same<12>(out, sink, h); run_same<16>(out, sink, h);
  return 0;
}
