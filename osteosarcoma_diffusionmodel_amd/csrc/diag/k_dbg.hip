// k_dbg.hip -- micro-benchmark of the GEMM inner loop (diagnostic entry point, not in osdiff.h):
// isolates what the matrix pipe sustains with / without LDS fragment reads, barriers and DMA.
#include "handle.h"
#include "gemm_glds.h"

namespace osd {

// MODE 0..4: loop variants; STORE 0: one float per thread, 1: the epilogue's [patient][feature] float4-per-quad
// pattern (each wave-instruction touches 32 rows), 2: the same bytes as whole 512-byte rows per half wave
template <int MODE, int STORE = 0>
__global__ __launch_bounds__(256, 2) void k_mfma_rate(const float* src, float* dst, int nk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wf = (wave >> 1) * 64, wp = (wave & 1) * 64;
  for (int i = tid; i < 16384; i += 256) smem[i] = (float)((i * 7) % 13) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float* As0 = smem; float* As1 = smem + 4096; float* Bs0 = smem + 8192; float* Bs1 = smem + 12288;
  int a_rd[2], a_sw[2], b_rd[2], b_sw[2];
  for (int fb = 0; fb < 2; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * 32; a_sw[fb] = h ^ ((R >> 1) & 7); }
  for (int pb = 0; pb < 2; ++pb) { const int R = wp + 32 * pb + l31; b_rd[pb] = R * 32; b_sw[pb] = h ^ ((R >> 1) & 7); }
  float ra[2][4], rb[2][4];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) { ra[i][e] = 0.001f * (lane + e + i); rb[i][e] = 0.002f * (lane - e + i); }
  const float* gbase = src + (size_t)(blockIdx.x % 64) * 4096 + (wave * 64 + lane) * 4;
  // MODE 4: operand B as in input_proj -- this block's own 128 rows of a [rows][2000] fp32 matrix
  // (HBM / Infinity-Cache resident), 8 rows x 128 B per wave-instruction; A from a small shared panel
  const float* xrow[4];
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    xrow[j] = src + 65536 + ((size_t)blockIdx.x * 128 + row) * 2000 + 4 * (lane & 7);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const float* Ac = (kt & 1) ? As1 : As0;
    const float* Bc = (kt & 1) ? Bs1 : Bs0;
    if (MODE == 4) {
      const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr((kt & 1) ? As0 : As1) + (unsigned)wave * 1024u);
      const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr((kt & 1) ? Bs0 : Bs1) + (unsigned)wave * 1024u);
      const int k0 = (kt % 62) * 32;
#pragma unroll
      for (int j = 0; j < 4; ++j) { glds16(gbase + j * 1024 + (kt & 7) * 4096, la + j * 4096u); glds16(xrow[j] + k0, lb + j * 4096u); }
    } else if (MODE >= 3) {
      const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr((kt & 1) ? As0 : As1) + (unsigned)wave * 1024u);
      const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr((kt & 1) ? Bs0 : Bs1) + (unsigned)wave * 1024u);
#pragma unroll
      for (int j = 0; j < 4; ++j) { glds16(gbase + j * 1024, la + j * 4096u); glds16(gbase + 8192 + j * 1024, lb + j * 4096u); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a[2][4], bb[2][4];
      if (MODE >= 1) {
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) { const float4 t = *reinterpret_cast<const float4*>(&Ac[a_rd[fb] + 4 * (a_sw[fb] ^ (2 * i))]); a[fb][0] = t.x; a[fb][1] = t.y; a[fb][2] = t.z; a[fb][3] = t.w; }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) { const float4 t = *reinterpret_cast<const float4*>(&Bc[b_rd[pb] + 4 * (b_sw[pb] ^ (2 * i))]); bb[pb][0] = t.x; bb[pb][1] = t.y; bb[pb][2] = t.z; bb[pb][3] = t.w; }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) for (int e = 0; e < 4; ++e) { a[q][e] = ra[q][e]; bb[q][e] = rb[q][e]; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[pb][e], acc[fb][pb], 0, 0, 0);
    }
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) __syncthreads();
  }
  if (STORE == 0) {
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    dst[blockIdx.x * 256 + tid] = s;
  } else {
    // output tile: 128 patients x 128 features of a [rows][512] matrix, this block's rows
    float* tile = dst + (size_t)blockIdx.x * 128 * 512;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[fb][pb][4 * q], acc[fb][pb][4 * q + 1], acc[fb][pb][4 * q + 2], acc[fb][pb][4 * q + 3]);
          if (STORE == 1) {
            const int f = wf + 32 * fb + 8 * q + 4 * h, p = wp + 32 * pb + l31;
            *reinterpret_cast<float4*>(tile + (size_t)p * 512 + f) = v;
          } else {
            const int j = (fb * 2 + pb) * 4 + q;                 // 16 instructions, each 2 rows x 512 B
            const int p = wave * 32 + j * 2 + h;
            *reinterpret_cast<float4*>(tile + (size_t)p * 512 + 4 * l31) = v;
          }
        }
  }
}

}  // namespace osd

using namespace osd;

extern "C" int osd_dbg_mfma_rate(osd_handle* h, int mode, int nk, int grid, const float* src, float* dst, float* ms_out) {
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipEvent_t e0, e1;
  OSD_HIP(hipEventCreate(&e0));
  OSD_HIP(hipEventCreate(&e1));
  auto run = [&]() {
    switch (mode) {
      case 0: hipLaunchKernelGGL((k_mfma_rate<0>), grid, 256, 65536, h->stream, src, dst, nk); break;
      case 1: hipLaunchKernelGGL((k_mfma_rate<1>), grid, 256, 65536, h->stream, src, dst, nk); break;
      case 2: hipLaunchKernelGGL((k_mfma_rate<2>), grid, 256, 65536, h->stream, src, dst, nk); break;
      case 3: hipLaunchKernelGGL((k_mfma_rate<3>), grid, 256, 65536, h->stream, src, dst, nk); break;
      case 5: hipLaunchKernelGGL((k_mfma_rate<3, 1>), grid, 256, 65536, h->stream, src, dst, nk); break;
      case 6: hipLaunchKernelGGL((k_mfma_rate<3, 2>), grid, 256, 65536, h->stream, src, dst, nk); break;
      default: hipLaunchKernelGGL((k_mfma_rate<4>), grid, 256, 65536, h->stream, src, dst, nk); break;
    }
  };
  static bool attr = false;
  if (!attr) {
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    OSD_HIP(hipFuncSetAttribute((const void*)k_mfma_rate<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr = true;
  }
  run();
  OSD_HIP(hipEventRecord(e0, h->stream));
  for (int i = 0; i < 5; ++i) run();
  OSD_HIP(hipEventRecord(e1, h->stream));
  OSD_HIP(hipEventSynchronize(e1));
  float ms = 0;
  OSD_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / 5;
  OSD_HIP(hipEventDestroy(e0));
  OSD_HIP(hipEventDestroy(e1));
  return OSD_OK;
}

// census: which hardware slots do the workgroups of a 2-per-CU launch land on?
namespace osd {
__global__ __launch_bounds__(256, 2) void k_census(unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  smem[threadIdx.x] = 1.f;       // touch LDS so the dynamic size limits residency to 2 per CU
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(0xF804);      // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg(0xF814);     // HW_REG_XCC_ID
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  // stay resident long enough for the whole grid to be placed
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);
}
}  // namespace osd
extern "C" int osd_dbg_census(osd_handle* h, int grid, unsigned* out) {
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_HIP(hipFuncSetAttribute((const void*)osd::k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(osd::k_census, grid, 256, 65536, h->stream, out);
  OSD_HIP(hipGetLastError());
  OSD_HIP(hipStreamSynchronize(h->stream));
  return OSD_OK;
}

// ---- variant: BK = 16, three LDS stages, DMA two tiles ahead with a counted vmcnt ------------------
namespace osd {
template <int STAGES>
__global__ __launch_bounds__(256, 2) void k_mfma_rate16(const float* src, float* dst, int nk16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wf = (wave >> 1) * 64, wp = (wave & 1) * 64;
  constexpr int TILE = 128 * 16;            // floats per operand tile
  for (int i = tid; i < STAGES * 2 * TILE; i += 256) smem[i] = (float)((i * 7) % 13) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_rd[2], a_sw[2], b_rd[2], b_sw[2];
  for (int fb = 0; fb < 2; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * 16; a_sw[fb] = h ^ ((R >> 2) & 3); }
  for (int pb = 0; pb < 2; ++pb) { const int R = wp + 32 * pb + l31; b_rd[pb] = R * 16; b_sw[pb] = h ^ ((R >> 2) & 3); }
  const float* xrow[2];
  for (int j = 0; j < 2; ++j) {
    const int row = (j * 4 + wave) * 16 + (lane >> 2);
    xrow[j] = src + 65536 + ((size_t)blockIdx.x * 128 + row) * 2000 + 4 * ((lane & 3) ^ ((row >> 2) & 3));
  }
  const float* gbase = src + (size_t)(blockIdx.x % 64) * 4096 + (wave * 64 + lane) * 4;
  auto stage = [&](int kt) {
    const int st = kt % STAGES;
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(smem + st * 2 * TILE) + (unsigned)wave * 1024u);
    const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(smem + st * 2 * TILE + TILE) + (unsigned)wave * 1024u);
    const int k0 = (kt % 124) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) { glds16(gbase + j * 1024 + (kt & 7) * 2048, la + j * 4096u); glds16(xrow[j] + k0, lb + j * 4096u); }
  };
  for (int p = 0; p < STAGES - 1; ++p) stage(p);
  if (STAGES == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk16; ++kt) {
    const float* Ac = smem + (kt % STAGES) * 2 * TILE;
    const float* Bc = Ac + TILE;
    stage(kt + STAGES - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float a[2][4], bb[2][4];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) { const float4 t = *reinterpret_cast<const float4*>(&Ac[a_rd[fb] + 4 * (a_sw[fb] ^ (2 * i))]); a[fb][0] = t.x; a[fb][1] = t.y; a[fb][2] = t.z; a[fb][3] = t.w; }
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) { const float4 t = *reinterpret_cast<const float4*>(&Bc[b_rd[pb] + 4 * (b_sw[pb] ^ (2 * i))]); bb[pb][0] = t.x; bb[pb][1] = t.y; bb[pb][2] = t.z; bb[pb][3] = t.w; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[pb][e], acc[fb][pb], 0, 0, 0);
    }
    // tile kt+1 must have landed; the 4 DMAs of tile kt+2 (just issued) may stay in flight
    if (STAGES == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  dst[blockIdx.x * 256 + tid] = s;
}
}  // namespace osd
extern "C" int osd_dbg_mfma_rate16(osd_handle* h, int stages, int nk16, int grid, const float* src, float* dst, float* ms_out) {
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipEvent_t e0, e1;
  OSD_HIP(hipEventCreate(&e0));
  OSD_HIP(hipEventCreate(&e1));
  const int lds = stages * 2 * 128 * 16 * 4;
  auto run = [&]() {
    if (stages == 3) hipLaunchKernelGGL((osd::k_mfma_rate16<3>), grid, 256, lds, h->stream, src, dst, nk16);
    else hipLaunchKernelGGL((osd::k_mfma_rate16<2>), grid, 256, lds, h->stream, src, dst, nk16);
  };
  run();
  OSD_HIP(hipEventRecord(e0, h->stream));
  for (int i = 0; i < 5; ++i) run();
  OSD_HIP(hipEventRecord(e1, h->stream));
  OSD_HIP(hipEventSynchronize(e1));
  float ms = 0;
  OSD_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / 5;
  OSD_HIP(hipEventDestroy(e0));
  OSD_HIP(hipEventDestroy(e1));
  return OSD_OK;
}

// diagnostic: one Linear+GroupNorm+SiLU launch with per-wave phase stamps (prologue / K loop / epilogue)
#include "kernels.h"
#include "launch.h"
extern "C" int osd_dbg_stamp_gn(osd_handle* h, const float* x, int K, const float* w, const float* b, const float* gamma, const float* beta,
                                int64_t n, int N, float* y, unsigned long long* stamps) {
  OSD_HIP(hipSetDevice(h->cfg.device));
  GemmArgs g{};
  g.A = w; g.lda = K; g.B0 = x; g.ldb0 = K; g.K0 = K; g.F = N; g.P = (int)n; g.K = K; g.stamps = stamps;
  GnArgs ga{};
  ga.bias = b; ga.gamma = gamma; ga.beta = beta; ga.out = y; ga.ldo = N;
  OSD_HIP(launch_gn_silu(h->stream, g, N / 8, ga));
  return OSD_OK;
}

// diagnostic: where a workgroup of the persistent chain kernel spends its cycles.  `buf` (device, 8 x u64 per workgroup, >= 512
// workgroups) receives, for the chain runs that follow, dependency wait / tile prologue / K loop / epilogue+drain / kernel
// total / units; pass null to switch the stamps off again.
extern "C" int osd_dbg_chain_stamps(osd_handle* h, unsigned long long* buf) {
  if (!h) return OSD_EINVAL;
  h->chain_stamps = buf;
  return OSD_OK;
}

// diagnostic: read the chain kernel's sync words ([status, queue, -, -], then progress[0..]) while a launch may still be
// running (own stream, so it does not wait for the handle's stream)
extern "C" int osd_dbg_chain_peek(osd_handle* h, unsigned* out8) {
  if (!h || !h->chain_sync) return OSD_ESTATE;
  hipStream_t ps;
  OSD_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  OSD_HIP(hipMemcpyAsync(out8, h->chain_sync, 16, hipMemcpyDeviceToHost, ps));
  OSD_HIP(hipMemcpyAsync(out8 + 4, h->chain_sync + 4 + 2048, 16, hipMemcpyDeviceToHost, ps));
  OSD_HIP(hipStreamSynchronize(ps));
  OSD_HIP(hipStreamDestroy(ps));
  return OSD_OK;
}
