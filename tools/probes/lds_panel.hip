// Probe: the K loop of an LDS-resident reverse chain.  One 4-wave workgroup per CU owns 64 patients whose activations sit in LDS
// as [64][K + 4]; a wave computes 32 * NFB features x 64 patients with v_mfma_f32_32x32x2_f32, weights streamed straight from
// global memory into registers (fragment-ordered copy: one fully contiguous 1 KiB wave load per (feature block, 8 k)), DEPTH
// loads ahead, no barrier inside a layer.  Question: how close to the matrix peak does a LONE wave per SIMD get when nothing but
// its own prefetch distance hides the L2 / Infinity-Cache latency, with all 256 CUs streaming the 10 MB of weights?
//   hipcc --offload-arch=gfx950 -O3 -o lds_panel lds_panel.hip && ./lds_panel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gv4;

// n8 (8-k blocks of the segment) is a positive multiple of DEPTH.  Branch-free groups: hipcc's s_waitcnt insertion gives up on
// counted waits (vmcnt(N), N > 0) as soon as the loads sit in conditional blocks, and a vmcnt(0) per group is no prefetch at all.
template <int NFB, int DEPTH, int NPB = 2>
__device__ __forceinline__ void kseg(f32x16 (&acc)[NFB][NPB], gv4 wl, int fbs, int n8, const float* bl, int ldb) {
  v4f aq[DEPTH][NFB];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = wl[(size_t)fb * fbs + d * 64];
  v4f bq[2][NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) bq[0][pb] = *reinterpret_cast<const v4f*>(bl + pb * 32 * ldb);
  auto group = [&](int i0, bool refill) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int i = i0 + d;
      const int in = i + 1 < n8 ? i + 1 : i;          // scalar clamp: the last block re-reads itself
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) bq[(d + 1) & 1][pb] = *reinterpret_cast<const v4f*>(bl + pb * 32 * ldb + 8 * in);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d][fb][e], bq[d & 1][pb][e], acc[fb][pb], 0, 0, 0);
      if (refill) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = wl[(size_t)fb * fbs + (size_t)(i + DEPTH) * 64];
      }
    }
  };
  int i0 = 0;
  for (; i0 < n8 - DEPTH; i0 += DEPTH) group(i0, true);
  group(i0, false);
}

// layers: F = 128 * NFB features (4 waves x 32 NFB), K as given; layer l reads weights at float4 offset (l * F * K / 4) % wrap
template <int NFB, int DEPTH, int NW = 4, int NPB = 2>
__global__ __launch_bounds__(64 * NW, NPB == 1 ? 2 : 1) void probe(const float* __restrict__ w, long long wrap4, int K, int layers, int skew, float* sink,
                                                unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float panel[];
  const int ldb = K + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 32 * NPB * ldb; i += 64 * NW) panel[i] = 1e-3f * (float)((i * 7 + blockIdx.x) & 63);
  __syncthreads();
  const int F = 32 * NW * NFB;
  const int n8 = K / 8;
  const int fbs = n8 * 64;
  const long long per_layer4 = (long long)F * K / 4;
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // skew: workgroups start at different layers (what a steady state with random phases looks like to the L2)
  int l0 = skew ? (int)(blockIdx.x * 7 % 12) : 0;
  // skew 2: every workgroup at its own position of the stream (no two CUs in lockstep: no L2 sharing between them)
  const long long sub4 = skew == 2 ? (long long)blockIdx.x * (wrap4 / 256) / 64 * 64 : 0;
  for (int l = 0; l < layers; ++l) {
    f32x16 acc[NFB][NPB];
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
      for (int j = 0; j < NPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const long long base4 = ((long long)(l + l0) * per_layer4 + sub4) % wrap4;
    gv4 wl = (gv4)(w) + base4 + (long long)(wave * NFB) * fbs + lane;
    kseg<NFB, DEPTH, NPB>(acc, wl, fbs, n8, panel + l31 * ldb + 4 * h, ldb);
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
      for (int j = 0; j < NPB; ++j) s += acc[i][j][0] + acc[i][j][7];
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (s == 123.456f) sink[tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NFB, int DEPTH, int NW = 4, int NPB = 2>
static void run(const char* name, const float* w, long long wrap4, int K, int layers, int skew, float* sink, unsigned long long* cyc, int grid) {
  const int lds = 32 * NPB * (K + 4) * 4;
  if (NPB == 1) grid *= 2;          // two workgroups per CU, 32 patients each
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NFB, DEPTH, NW, NPB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NFB, DEPTH, NW, NPB>), dim3(grid), dim3(64 * NW), lds, 0, w, wrap4, K, layers, skew, sink, cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 1) {
      const double flop = 2.0 * 32 * NPB * (32.0 * NW * NFB) * K * layers * grid;
      printf("%-28s K=%4d F=%3d depth=%d skew=%d grid=%d: %8.3f ms  %7.2f TFLOP/s  (%.3f of 157.3)\n", name, K, 32 * NW * NFB, DEPTH, skew, grid, ms,
             flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
}

int main() {
  const long long floats = 16ll << 20;          // 64 MB of "weights": beyond one XCD's L2, inside the Infinity Cache
  float* w; float* sink; unsigned long long* cyc;
  hipMalloc(&w, floats * 4); hipMalloc(&sink, 4096); hipMalloc(&cyc, 8 * 1024);
  std::vector<float> hw(floats);
  for (long long i = 0; i < floats; ++i) hw[i] = 1e-3f * (float)(i % 97);
  hipMemcpy(w, hw.data(), floats * 4, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount;
  const long long small4 = (2ll << 20) / 16;                // 2 MB window: L2 resident
  const long long real4 = 10660000ll / 16;                 // ~10.66 MB: the model's weights
  const long long big4 = floats / 4 - (512ll * 512 / 4);    // a layer read from the last start stays inside the buffer
  const int layers = 240;
  run<4, 4, 4, 1>("2 WG x 32 pat F512 apart", w, real4, 512, layers, 2, sink, cyc, grid);
  run<4, 2, 4, 1>("2 WG x 32 pat F512 d2", w, real4, 512, layers, 2, sink, cyc, grid);
  run<2, 4, 4, 1>("2 WG x 32 pat F256 apart", w, real4, 256, layers * 4, 2, sink, cyc, grid);
  run<2, 8, 4, 1>("2 WG x 32 pat F256 d8", w, real4, 256, layers * 4, 2, sink, cyc, grid);
  run<2, 4, 8>("8 waves F512 all apart", w, real4, 512, layers, 2, sink, cyc, grid);
  run<2, 8, 8>("8 waves F512 all apart", w, real4, 512, layers, 2, sink, cyc, grid);
  run<1, 8, 8>("8 waves F256 all apart", w, real4, 256, layers * 4, 2, sink, cyc, grid);
  run<1, 8, 8>("8 waves F256 K512", w, real4, 512, layers * 2, 2, sink, cyc, grid);
  run<2, 4, 8>("8 waves F512 K256", w, real4, 256, layers * 2, 2, sink, cyc, grid);
  run<4, 4>("F512 L2-resident", w, small4, 512, layers, 0, sink, cyc, grid);
  run<4, 4>("F512 10.7MB lockstep", w, real4, 512, layers, 0, sink, cyc, grid);
  run<4, 4>("F512 10.7MB skewed", w, real4, 512, layers, 1, sink, cyc, grid);
  run<4, 4>("F512 10.7MB all apart", w, real4, 512, layers, 2, sink, cyc, grid);
  run<4, 8>("F512 10.7MB all apart", w, real4, 512, layers, 2, sink, cyc, grid);
  run<2, 8>("F256 10.7MB all apart", w, real4, 256, layers * 4, 2, sink, cyc, grid);
  run<2, 8>("F256 K512 all apart", w, real4, 512, layers * 2, 2, sink, cyc, grid);
  run<4, 8>("F512 10.7MB skewed", w, real4, 512, layers, 1, sink, cyc, grid);
  run<4, 8>("F512 64MB skewed", w, big4, 512, layers, 1, sink, cyc, grid);
  run<2, 4>("F256 L2-resident", w, small4, 256, layers * 4, 0, sink, cyc, grid);
  run<2, 4>("F256 10.7MB skewed", w, real4, 256, layers * 4, 1, sink, cyc, grid);
  run<2, 8>("F256 10.7MB skewed", w, real4, 256, layers * 4, 1, sink, cyc, grid);
  run<2, 8>("F256 K512 10.7MB skewed", w, real4, 512, layers * 2, 1, sink, cyc, grid);
  run<4, 8>("F512 K256 10.7MB skewed", w, real4, 256, layers * 2, 1, sink, cyc, grid);
  return 0;
}
