// b3_pack.h -- fp32 rows -> bf16x3 planes buffer (gemm_bf3.h).  Included by ONE translation unit of the library (split.hip) and by the probe.
#pragma once
#include "gemm_bf3.h"

namespace osd {

// ---- fp32 rows -> planes (weights at load time; x_T at the start of a chain) --------------------------------------------------------
// src[R][ld] fp32, valid K columns; dst = planes buffer [ceil(R/128)][nkb][768]; rows beyond R and k beyond K become zeros.
// One thread per (tile, k block, row block, lane): 8 floats in, three 16-byte units out.
__global__ void k_b3_pack(const float* __restrict__ src, int ld, long long R, int K, uint4* __restrict__ dst, int nkb, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int rb = (int)((i >> 6) & 3);
    const long long tk = i >> 8;                 // tile * nkb + kb
    const int kb = (int)(tk % nkb);
    const long long tile = tk / nkb;
    const long long row = tile * B3_ROWS + rb * 32 + (lane & 31);
    const int k0 = kb * B3_KB + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (row < R && k0 + e < K) ? src[(size_t)row * ld + k0 + e] : 0.f;
    const Split4 lo = split4(make_float4(v[0], v[1], v[2], v[3])), hi = split4(make_float4(v[4], v[5], v[6], v[7]));
    uint4* o = dst + (size_t)tk * B3_STAGE_U4 + rb * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o[pl * 256] = make_uint4(lo.p[pl].x, lo.p[pl].y, hi.p[pl].x, hi.p[pl].y);
  }
}

}  // namespace osd
