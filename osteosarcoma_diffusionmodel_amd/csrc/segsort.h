// segsort.h -- segmented LSD radix sort of fp32 keys, equal-length segments (the per-feature samples of the two-sample KS
// statistic, utils/validation.py:238-249).  Four passes of 8 bits over order-preserving uint32 images of the floats.
//
// A WAVE owns SEG_CHUNK consecutive keys of one segment per pass and walks them in order, 64 at a time:
//   count   : per-wave 256-bin histogram in LDS (ds_add), written digit-major: cnt[seg][digit][wave]
//   scan    : one workgroup per segment turns the counts into exclusive offsets in (digit, wave) order
//   scatter : the wave re-reads its keys; lanes with the same digit find each other with eight ballots (one per digit bit),
//             rank = popcount(lower lanes of the match mask); the wave's running bucket cursors live in LDS.  Walking in
//             order and ranking by lane keeps every pass stable, which LSD needs.
// No float atomics, no dependence on launch geometry: the result is the unique sorted sequence.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osd {

constexpr int SEG_CHUNK = 4096;           // keys per wave and pass
constexpr int SEG_WAVES = 4;              // waves per workgroup

__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// in: floats (FIRST) or keys; one wave per chunk
template <bool FIRST>
__global__ __launch_bounds__(64 * SEG_WAVES) void k_seg_count(const uint32_t* __restrict__ in, long long n, int wps, int shift, int* __restrict__ cnt) {
  __shared__ int hist[SEG_WAVES][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int seg = blockIdx.y;
  const int w = blockIdx.x * SEG_WAVES + wv;                 // chunk index inside the segment
  for (int i = lane; i < 256; i += 64) hist[wv][i] = 0;
  __syncthreads();
  if (w < wps) {
    const uint32_t* src = in + (long long)seg * n;
    const long long k0 = (long long)w * SEG_CHUNK;
    const long long k1 = k0 + SEG_CHUNK < n ? k0 + SEG_CHUNK : n;
    for (long long k = k0 + lane; k < k1; k += 64) {
      const uint32_t key = FIRST ? f2key(__uint_as_float(src[k])) : src[k];
      atomicAdd(&hist[wv][(key >> shift) & 255u], 1);
    }
  }
  __syncthreads();
  if (w < wps)
    for (int d = lane; d < 256; d += 64) cnt[((long long)seg * 256 + d) * wps + w] = hist[wv][d];
}

// cnt[seg][digit][wave] -> exclusive offsets in (digit, wave) order; one workgroup of 256 threads per segment
__global__ __launch_bounds__(256) void k_seg_scan(int* __restrict__ cnt, int wps) {
  __shared__ int tot[256];
  const int d = threadIdx.x;
  int* c = cnt + ((long long)blockIdx.x * 256 + d) * wps;
  int s = 0;
  for (int w = 0; w < wps; ++w) s += c[w];
  tot[d] = s;
  __syncthreads();
  // exclusive scan over the 256 digit totals (Hillis-Steele on the inclusive values)
  int v = s;
  for (int o = 1; o < 256; o <<= 1) {
    const int t = d >= o ? tot[d - o] : 0;
    __syncthreads();
    v += t;
    tot[d] = v;
    __syncthreads();
  }
  int run = v - s;
  for (int w = 0; w < wps; ++w) { const int x = c[w]; c[w] = run; run += x; }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(64 * SEG_WAVES) void k_seg_scatter(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long long n, int wps, int shift,
                                                                const int* __restrict__ offs) {
  __shared__ int cur[SEG_WAVES][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int seg = blockIdx.y;
  const int w = blockIdx.x * SEG_WAVES + wv;
  if (w >= wps) return;                                       // whole waves leave: no barrier below
  for (int d = lane; d < 256; d += 64) cur[wv][d] = offs[((long long)seg * 256 + d) * wps + w];
  const uint32_t* src = in + (long long)seg * n;
  uint32_t* dst = out + (long long)seg * n;
  const long long k0 = (long long)w * SEG_CHUNK;
  const long long k1 = k0 + SEG_CHUNK < n ? k0 + SEG_CHUNK : n;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (long long kb = k0; kb < k1; kb += 64) {
    const long long k = kb + lane;
    const bool ok = k < k1;
    uint32_t key = 0;
    if (ok) key = FIRST ? f2key(__uint_as_float(src[k])) : src[k];
    const uint32_t dg = (key >> shift) & 255u;
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(ok && ((dg >> b) & 1u));
      m &= ((dg >> b) & 1u) ? bal : ~bal;
    }
    if (ok) {
      const int rank = __popcll(m & lt);
      const int base = cur[wv][dg];                           // read by every lane of the group before its leader moves the cursor
      const uint32_t o = LAST ? __float_as_uint(key2f(key)) : key;
      dst[base + rank] = o;
      if ((m >> lane) <= 1ull) cur[wv][dg] = base + __popcll(m);      // highest lane of the group
    }
  }
}

}  // namespace osd
