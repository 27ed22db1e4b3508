// xpose.h -- per-wave LDS row transposer for the epilogues (chain kernel and the per-layer kernels' wide epilogues).
#pragma once
#include "gemm_glds.h"

namespace osd {

// The MFMA fragment layout (gemm.h) gives a lane ONE row and 4 consecutive features per register quad, so a float4 store
// straight from the accumulators touches 32 rows x 32 bytes: a quarter of a 128-byte line per row, 32 L2 write transactions per
// wave-instruction, and the other three quarters of each line arrive with later instructions.  In-kernel stamps put 22 000 of a
// GroupNorm tile's 31 000 epilogue cycles into issuing those 16 stores (the statistics take 600).  The epilogues therefore
// turn every 32-feature block through LDS -- the K loop's second operand buffer is idle during an epilogue, 4-8 KB per wave --
// and store (and load side inputs such as x_t / cond_proj / the noise target) as full 128-byte row segments, 8 rows per
// wave-instruction.
//
// WaveXpose: [32 * NPB rows][32 floats], 16-byte chunks XOR-swizzled by (row ^ (row >> 3)) & 7.  A plain row & 7 serves the
// 8-lane groups of ds_write_b128 and the row-side reads, but ds_read_b128 is banked over the lane groups
// {0-3, 12-15, 20-27}, ..., where rows 0 and 24 (or 2 and 26, ...) share row & 7 and parity: 2-way conflicts (PMC: 0.14 % of
// the kernel's cycles).  Folding row >> 3 in makes the fragment-side read, the row-side read and both writes conflict-free.
// LDS instructions of one wave execute in order, so a block's reads follow its writes (and the next block's writes follow these
// reads) without any barrier; the compiler keeps the order because the accesses may alias.
// Global memory is reached through ldg4 / stg4 (gemm_glds.h): explicitly global accesses even where hipcc only sees a generic
// pointer.

__device__ __forceinline__ int xsw(int row) { return (row ^ (row >> 3)) & 7; }
template <int NPB>
struct WaveXpose {
  float* buf;
  // fragment side: lane (l31, h) owns row 32 pb + l31, chunk 2 q + h of the block
  // (xsw(32 pb + l31) = xsw(l31) ^ 4 pb: one lane constant, the rest folds into the unrolled pb / q)
  __device__ __forceinline__ void put(int pb, int q, int l31, int h, float4 v) const {
    const int sw = h ^ xsw(l31);
    *reinterpret_cast<float4*>(buf + l31 * 32 + pb * 1024 + 4 * (sw ^ ((2 * q) ^ (4 * pb & 7)))) = v;
  }
  __device__ __forceinline__ float4 get(int pb, int q, int l31, int h) const {
    const int sw = h ^ xsw(l31);
    return *reinterpret_cast<const float4*>(buf + l31 * 32 + pb * 1024 + 4 * (sw ^ ((2 * q) ^ (4 * pb & 7))));
  }
  // row side: instruction i moves rows 8 i .. 8 i + 7, eight lanes per row (128 contiguous bytes)
  template <bool GUARD>
  __device__ __forceinline__ void store_rows(float* __restrict__ g, int ld, int lane, int rows, int cols) const {
    const int c = lane & 7, r = lane >> 3, cr = c ^ r;      // xsw(8 i + r) = r ^ (i & 7)
    // four rows-of-eight at a time: the LDS reads of a group are all in flight before the first store waits for one
#pragma unroll
    for (int i0 = 0; i0 < 4 * NPB; i0 += 4) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(buf + r * 32 + (i0 + j) * 256 + 4 * (cr ^ ((i0 + j) & 7)));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 8 * (i0 + j) + r;
        if (!GUARD || (row < rows && 4 * c < cols)) stg4(g + (size_t)row * ld + 4 * c, v[j]);
      }
    }
  }
  // rows beyond `rows` / chunks beyond `cols` re-read the last valid ones (finite values that are never stored)
  template <bool GUARD>
  __device__ __forceinline__ void load_rows(const float* __restrict__ g, int ld, int lane, int rows, int cols) const {
    const int c = lane & 7, r = lane >> 3, cr = c ^ r;
    float4 v[4 * NPB];
#pragma unroll
    for (int i = 0; i < 4 * NPB; ++i) {
      int row = 8 * i + r;
      int cc = 4 * c;
      if (GUARD) { row = row < rows ? row : rows - 1; cc = cc < cols - 4 ? cc : cols - 4; }
      v[i] = ldg4(g + (size_t)row * ld + cc);
    }
#pragma unroll
    for (int i = 0; i < 4 * NPB; ++i) *reinterpret_cast<float4*>(buf + r * 32 + i * 256 + 4 * (cr ^ (i & 7))) = v[i];
  }
  // load_rows in two halves, for a caller that has other work between the loads and their first use (chain_panel.h: a block's
  // x_t rows are requested while the previous block is being computed)
  template <bool GUARD>
  __device__ __forceinline__ void issue_rows(float4 (&v)[4 * NPB], const float* __restrict__ g, int ld, int lane, int rows, int cols) const {
    const int c = lane & 7, r = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4 * NPB; ++i) {
      int row = 8 * i + r;
      int cc = 4 * c;
      if (GUARD) { row = row < rows ? row : rows - 1; cc = cc < cols - 4 ? cc : cols - 4; }
      v[i] = ldg4(g + (size_t)row * ld + cc);
    }
  }
  __device__ __forceinline__ void commit_rows(const float4 (&v)[4 * NPB], int lane) const {
    const int c = lane & 7, r = lane >> 3, cr = c ^ r;
#pragma unroll
    for (int i = 0; i < 4 * NPB; ++i) *reinterpret_cast<float4*>(buf + r * 32 + i * 256 + 4 * (cr ^ (i & 7))) = v[i];
  }
};

}  // namespace osd
