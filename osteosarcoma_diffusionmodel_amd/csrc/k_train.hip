// k_train.hip -- HBM-bound kernels of the backward pass and the optimizer:
// GroupNorm+SiLU(+dropout) backward with per-row wavefront reductions, column sums,
// the time-embedding scatter, clip_grad_norm_ + AdamW over a flat buffer.
#include "kernels_train.h"
#include "gemm.h"
#include "rng.h"

namespace osd {

static inline int ew_grid(int64_t work, int per_block = 256) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- zero several small buffers in one launch --------------------------------------
__global__ void k_zero_many(ZeroList zl) {
  const int i = blockIdx.y;
  float* p = zl.ptr[i];
  const int64_t n = zl.count[i];
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) p[j] = 0.f;
}
hipError_t launch_zero_many(hipStream_t s, const ZeroList& zl) {
  if (zl.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_zero_many, dim3(64, zl.n), 256, 0, s, zl);
  return hipGetLastError();
}

// ---- SiLU forward / backward on a dense [n][c] buffer ---------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void k_silu_fwd(const float* u, float* y, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = u[i];
    y[i] = x / (1.0f + expf(-x));
  }
}
hipError_t launch_silu_fwd(hipStream_t s, const float* u, float* y, int64_t total) {
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_silu_fwd, ew_grid(total), 256, 0, s, u, y, total);
  return hipGetLastError();
}
// ---- ConditionalEmbedding, training forward (models/diffusion.py:101-105): u0 = c W0^T + b0, ce1 = SiLU(u0), ce2 = ce1 W2^T + b2
// in ONE launch (the three outputs are all kept for backward).  Both layers are 64 wide (the reference's literal) and the
// whole job is 17 MFLOP at batch 4096: as two tile GEMMs and an elementwise pass it cost three launches of 5-9 us each.
// A workgroup = 4 rows x 64 features; W2 sits in LDS ([f][k], pitch 65: conflict-free), sums run k = 0, 1, ... with one FMA
// each and the bias is added last, as the GEMM epilogue does.
__global__ __launch_bounds__(256) void k_cond_mlp_fwd(const float* __restrict__ cond, int cd, const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, int64_t n,
                                                      float* __restrict__ u0, float* __restrict__ ce1, float* __restrict__ ce2) {
  __shared__ float w2s[64 * 65];
  __shared__ float ys[4][64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) w2s[(i >> 6) * 65 + (i & 63)] = w2[i];
  const int f = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const float bias0 = b0[f], bias2 = b2[f];
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n; r0 += (int64_t)gridDim.x * 4) {
    const int64_t r = r0 + rl;
    float u = 0.f;
    if (r < n) {
      for (int k = 0; k < cd; ++k) u = fmaf(w0[f * cd + k], cond[r * cd + k], u);
      u += bias0;
      const float y = u / (1.0f + expf(-u));
      u0[r * 64 + f] = u;
      ce1[r * 64 + f] = y;
      ys[rl][f] = y;
    }
    __syncthreads();                       // W2 staged (first trip) and this trip's activations visible
    if (r < n) {
      float v = 0.f;
#pragma unroll 16
      for (int k = 0; k < 64; ++k) v = fmaf(w2s[f * 65 + k], ys[rl][k], v);
      ce2[r * 64 + f] = v + bias2;
    }
    __syncthreads();                       // ys is rewritten by the next trip
  }
}
hipError_t launch_cond_mlp_fwd(hipStream_t s, const float* cond, int cd, const float* w0, const float* b0, const float* w2, const float* b2, int64_t n,
                               float* u0, float* ce1, float* ce2) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_cond_mlp_fwd, dim3((unsigned)blocks), dim3(256), 0, s, cond, cd, w0, b0, w2, b2, n, u0, ce1, ce2);
  return hipGetLastError();
}

__global__ void k_silu_bwd(const float* u, const float* g, float* gu, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = u[i];
    const float sg = sigmoid_f(x);
    gu[i] = g[i] * (sg * (1.0f + x * (1.0f - sg)));
  }
}
hipError_t launch_silu_bwd(hipStream_t s, const float* u, const float* g, float* gu, int64_t total) {
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_silu_bwd, ew_grid(total), 256, 0, s, u, g, gu, total);
  return hipGetLastError();
}

// ---- the conditioning branch's backward below h0, one launch instead of five (scatter_rows, two 64-wide dgrads, silu_bwd, small_wgrad) ----
//   g_temb[t[r]][:] += g_h0[r][:]                                  (time embedding table; zeroed by the caller)
//   g_ce2[r][k] = sum_n g_h0[r][n] W_cp[n][k]       n < H0, k < 64   (cond_proj: h0 = ... + ce2 W_cp^T)
//   g_u[r][k]   = (sum_n g_ce2[r][n] W_ce2[n][k]) * silu'(u0[r][k]) (ConditionalEmbedding's second Linear and its SiLU)
//   dW0[k][j] += sum_r g_u[r][k] cond[r][j],  db0[k] += sum_r g_u[r][k]      (its first Linear, cd <= 4 inputs; optional)
// A workgroup owns 16 rows; wave w owns output columns 16 w .. + 15 of both GEMMs (v_mfma_f32_16x16x4_f32: M = 16 columns, N = the 16
// rows, K in steps of 4).  The A operands -- the wave's column slice of W_cp (H0 / 4 registers) and of W_ce2 (16) -- are loaded straight
// into registers, all at once, in the prologue; the B operand is the rows' g_h0 tile in LDS (row stride H0 + 4: lanes (row, k) read 64
// different banks), which the scatter reads too.  0.17 GFLOP at batch 4 096: the five launches it replaces were launch- and latency-bound
// (6 + 9 + 5 + 5 + 7 us alone, 50-60 us beside the column sums of the side stream), not FLOP-bound.
// Built and dropped on the way, each parity-green:
//   * VALU version, thread = (column, 4-8 rows), weights streamed from L2 in the K loop: 53 us -- one L2 round trip per four k;
//   * the same with both weight matrices in LDS (101 KB, one workgroup per CU): 36-50 us, of which 28 the first GEMM's K loop: its
//     tile reads were wave-uniform ds_read_b128 (every lane the same address), which this LDS serves at ~1/25 of the rate of a
//     conflict-free read -- not a broadcast;
//   * scatter atomics issued from the float4 registers of the tile copy (16 bytes between lanes: four times the cache lines per
//     instruction), a t[row] load inside the scatter loop (a vmcnt(0) wait, which the atomics count on too, in front of every atomic),
//     loads whose only use was a store under a branch (hipcc sinks the load into the branch and drains vmcnt there: 24 serial L2 round
//     trips in the prologue), __threadfence() before the ticket (an agent-scope release writes back the XCD's whole L2: 97 us, and the
//     column sums beside it 100 instead of 43).
typedef float cb_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_cond_bwd(const float* __restrict__ g_h0, int H0, const int* __restrict__ t, float* g_temb,
                                                  const float* __restrict__ w_cp, const float* __restrict__ w_ce2,
                                                  const float* __restrict__ u0, int64_t rows, float* __restrict__ g_ce2, float* __restrict__ g_u,
                                                  const float* __restrict__ cond, int cd, float* part, float* dw0, float* db0) {
  constexpr int R = 16;
  extern __shared__ float cb_lds[];
  const int ldt = H0 + 4;
  float* tile = cb_lds;                 // [16][H0 + 4]; later the [64 (cd + 1)] exchange of the small weight gradient
  float* g2 = tile + R * ldt;           // [16][68]
  __shared__ int cb_t[R];
  __shared__ float cb_c[R * 4];         // the rows' conditions, [16][4] with zeros behind cd and behind the last row
  __shared__ int cb_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4, col0 = wave * 16;
  const int64_t row0 = (int64_t)blockIdx.x * R, last_row = rows - 1;
  const int q = H0 >> 2;
  // ---- prologue: every global load of the kernel, none under a branch, nothing stored before all are issued ----
  float a1[64], a2[16];                 // A[m][k]: lane (m, kg) holds W[4 ks + kg][col0 + m]
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) { const int kr = 4 * ks + kg; a1[ks] = w_cp[(kr < H0 ? kr : H0 - 1) * 64 + col0 + m]; }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) a2[ks] = w_ce2[(4 * ks + kg) * 64 + col0 + m];
  // this lane's outputs (the accumulator fragment): row m of the tile, columns col0 + 4 kg .. + 3
  const int64_t orow = row0 + m;
  const float4 ux = *reinterpret_cast<const float4*>(u0 + (orow < rows ? orow : last_row) * 64 + col0 + 4 * kg);
  {
    const int64_t row = row0 + (tid & (R - 1));                      // threads 16.. write the same values again
    const int tv = t[row < rows ? row : last_row];
    const int ci = tid & (4 * R - 1), cr = ci >> 2, cj = ci & 3;
    const int64_t crow = row0 + cr;
    const float cv = (cond ? cond : u0)[(crow < rows ? crow : last_row) * (cond ? cd : 0) + (cj < cd ? cj : 0)];
    float4 v[4];
    for (int base = tid; base < R * q; base += 256 * 4) {      // 1 024 float4 at H0 = 256: one round of four loads
      const float* src[4];
      float* dst[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // behind the end: the last element once more, to its own place
        const int i = base + 256 * u, ic = i < R * q ? i : R * q - 1;
        const int r = ic / q, c4 = ic - r * q;
        const int64_t grow = row0 + r;
        live[u] = grow < rows;
        src[u] = g_h0 + (live[u] ? grow : last_row) * H0 + 4 * c4;
        dst[u] = tile + r * ldt + 4 * c4;
      }
      __builtin_amdgcn_sched_barrier(0);      // addresses first, then the four loads back to back (hipcc had each load wait behind the previous one's use)
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(dst[u]) = live[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    cb_t[tid & (R - 1)] = (g_temb && row < rows) ? tv : -1;
    cb_c[ci] = (cond && cj < cd && crow < rows) ? cv : 0.f;
  }
  __syncthreads();
  // ---- g_ce2 = g_h0 W_cp: B[k][n] = tile[n][k], lane (n = m, kg) reads tile[m][4 ks + kg] ----
  cb_f4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* brow = tile + m * ldt + kg;
#pragma unroll
  for (int g = 0; g < 8; ++g)
    if (8 * g < q) {                     // H0 is a multiple of 32: whole groups of eight K steps
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[8 * g + e], brow[4 * (8 * g + e)], acc, 0, 0, 0);
    }
  // the scatter (behind the first GEMM's LDS reads, so that its atomics drain under the rest of the kernel): lanes = consecutive
  // columns of one row, one atomic instruction covers 256 contiguous bytes; the rows' t come from LDS
  if (g_temb)
    for (int i = tid; i < R * H0; i += 256) {
      const int r = i / H0, c = i - r * H0;
      const int tr = cb_t[r];
      if (tr >= 0) atomicAdd(g_temb + (int64_t)tr * H0 + c, tile[r * ldt + c]);
    }
  // D[mm][n]: lane (n = m, kg) holds columns col0 + 4 kg .. + 3 of row m
  if (orow < rows) *reinterpret_cast<float4*>(g_ce2 + orow * 64 + col0 + 4 * kg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(g2 + m * 68 + col0 + 4 * kg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  // ---- g_ce1 = g_ce2 W_ce2, then the SiLU backward ----
  cb_f4 acc2 = {0.f, 0.f, 0.f, 0.f};
  const float* b2 = g2 + m * 68 + kg;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[ks], b2[4 * ks], acc2, 0, 0, 0);
  float gu[4];
  {
    const float xs[4] = {ux.x, ux.y, ux.z, ux.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float sg = sigmoid_f(xs[v]);
      gu[v] = acc2[v] * (sg * (1.0f + xs[v] * (1.0f - sg)));      // rows behind the last: acc2 = 0 (their tile rows are zero)
    }
  }
  if (orow < rows) *reinterpret_cast<float4*>(g_u + orow * 64 + col0 + 4 * kg) = make_float4(gu[0], gu[1], gu[2], gu[3]);
  if (!cond) return;
  // ---- ConditionalEmbedding's first Linear: sums over the 16 rows = over the 16 lanes that share kg ----
  // Every workgroup adding into the same 64 (cd + 1) addresses serialises in one L2 channel (k_small_wgrad below: 7 us alone with 16
  // blocks, 18 beside the column sums), so the workgroups add into 16 copies 4 KB apart and the last one to finish folds the copies
  // into the gradient (and leaves copies and counter zero again).
  const float4 cr4 = *reinterpret_cast<const float4*>(cb_c + m * 4);
  const float cs[4] = {cr4.x, cr4.y, cr4.z, cr4.w};
  float ps[5][4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    ps[0][v] = gu[v];
#pragma unroll
    for (int j = 0; j < 4; ++j) ps[1 + j][v] = gu[v] * cs[j];
  }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float x = ps[j][v];
      x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
      ps[j][v] = x;
    }
  float* red = tile;                    // the scatter above has read the tile: wait for every wave before overwriting it
  __syncthreads();
  if (m == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int n = col0 + 4 * kg + v;
      red[64 * cd + n] = ps[0][v];
#pragma unroll
      for (int j = 0; j < 4; ++j) if (j < cd) red[n * cd + j] = ps[1 + j][v];
    }
  }
  __syncthreads();
  const int nv = 64 * (cd + 1);
  float* mine = part + (blockIdx.x & 15) * 1024;
  for (int i = tid; i < nv; i += 256) atomicAdd(mine + i, red[i]);
  // every atomic of this workgroup acknowledged by L2 before its ticket is drawn: a counter wait, not __threadfence() (see above)
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  unsigned* counter = reinterpret_cast<unsigned*>(part + 16 * 1024);
  if (tid == 0) cb_last = atomicAdd(counter, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!cb_last) return;
  for (int i = tid; i < nv; i += 256) {
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) sum += __hip_atomic_load(part + c * 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int c = 0; c < 16; ++c) part[c * 1024 + i] = 0.f;
    if (i < 64 * cd) dw0[i] += sum; else db0[i - 64 * cd] += sum;
  }
  if (tid == 0) *counter = 0u;
}
// shapes and alignments the one-launch kernel takes (H0 / 4 <= 64 weight registers per lane, whole groups of eight K steps, float4 accesses)
bool cond_bwd_ok(int H0, const float* g_h0, const float* u0, const float* g_ce2, const float* g_u) {
  if (H0 < 32 || H0 % 32 != 0 || H0 > 256) return false;
  return (((uintptr_t)g_h0 | (uintptr_t)u0 | (uintptr_t)g_ce2 | (uintptr_t)g_u) & 15) == 0;
}
hipError_t launch_cond_bwd(hipStream_t s, const float* g_h0, int H0, const int* t, float* g_temb, const float* w_cp, const float* w_ce2,
                           const float* u0, int64_t rows, float* g_ce2, float* g_u, const float* cond, int cd, float* part, float* dw0, float* db0) {
  if (rows <= 0) return hipSuccess;
  if (!cond_bwd_ok(H0, g_h0, u0, g_ce2, g_u)) return hipErrorInvalidValue;
  if (cond && (cd < 1 || cd > 4 || !part || !dw0 || !db0)) return hipErrorInvalidValue;
  const size_t lds = (size_t)(16 * (H0 + 4) + 16 * 68) * sizeof(float);      // <= 21 KB
  hipLaunchKernelGGL(k_cond_bwd, (unsigned)((rows + 15) / 16), 256, lds, s, g_h0, H0, t, g_temb, w_cp, w_ce2, u0, rows, g_ce2, g_u, cond, cd, part, dw0, db0);
  return hipGetLastError();
}

// ---- column sums: out[c] += sum_r in[r][c]   (out zeroed by the caller) --------------------
__global__ void k_colsum(const float* in, int ld, int64_t rows, int cols, int rows_per_block, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  if (c >= cols) return;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += in[r * ld + c];
  atomicAdd(out + c, s);
}
hipError_t launch_colsum(hipStream_t s, const float* in, int ld, int64_t rows, int cols, float* out) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const int rpb = 64;
  dim3 grid((cols + 255) / 256, (unsigned)((rows + rpb - 1) / rpb));
  hipLaunchKernelGGL(k_colsum, grid, 256, 0, s, in, ld, rows, cols, rpb, out);
  return hipGetLastError();
}

// ---- time embedding: g_table[t[r]][c] += g[r][c]   (table zeroed by the caller) -------------
__global__ void k_scatter_rows(const float* g, const int* t, int64_t rows, int cols, float* table) {
  const int64_t total = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    atomicAdd(table + (int64_t)t[r] * cols + c, g[i]);
  }
}
hipError_t launch_scatter_rows(hipStream_t s, const float* g, const int* t, int64_t rows, int cols, float* table) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scatter_rows, ew_grid(rows * cols), 256, 0, s, g, t, rows, cols, table);
  return hipGetLastError();
}

// ---- GroupNorm(8) + SiLU (+ dropout) backward ------------------------------------------------
// One wave per row at a time; lane l owns columns 4l + 256j.. (float4), so the GW/4 lanes of a
// group are adjacent and the two per-group means are a __shfl_xor butterfly.  Per-column sums
// (d gamma, d beta, d bias) accumulate in registers over the wave's rows, then one atomic each.
//   y = zhat*gamma + beta, a = silu(y), out = a * keep
//   g_y = g*keep*silu'(y);  g_zhat = g_y*gamma
//   g_z = rstd * (g_zhat - mean_g(g_zhat) - zhat*mean_g(g_zhat*zhat))
constexpr int GN_BWD_WAVES = 16;     // waves per block: at a training batch of 4096 rows every wave owns one row (4 waves per SIMD)
template <int GW, int NJ>
__global__ __launch_bounds__(NJ <= 2 ? 64 * GN_BWD_WAVES : 32 * GN_BWD_WAVES) void k_gn_silu_bwd(GnBwdArgs a) {      // wide rows (C > 512): 8 waves, 256 VGPRs each
  extern __shared__ __attribute__((aligned(16))) float red[];      // [3][waves of the block][C]
  const int NW = blockDim.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int C = a.C;
  float acc_g[NJ][4], acc_b[NJ][4], acc_z[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc_g[j][e] = acc_b[j][e] = acc_z[j][e] = 0.f;
  float4 gam[NJ], bet[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4 * lane + 256 * j;
    gam[j] = ld4g(a.gamma, c, C);
    bet[j] = ld4g(a.beta, c, C);
  }
  for (int64_t r = wave; r < a.rows; r += nwaves) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4 * lane + 256 * j;
      const bool in = c < C;
      const float4 g4 = in ? ld4g(a.g + r * C, c, C) : make_float4(0, 0, 0, 0);
      const float4 z4 = in ? ld4g(a.z + r * C, c, C) : make_float4(0, 0, 0, 0);
      float mean = 0.f, rstd = 0.f;
      if (in) { const float* sp = a.stats + (r * (C / GW) + c / GW) * 2; mean = sp[0]; rstd = sp[1]; }
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.drop_mode == 1 && in) {
        const float4 m4 = ld4g(a.mask + r * C, c, C);
        keep[0] = m4.x * a.keep_scale; keep[1] = m4.y * a.keep_scale; keep[2] = m4.z * a.keep_scale; keep[3] = m4.w * a.keep_scale;
      } else if (a.drop_mode == 2 && in) {
        const uint4 rr = philox_at(a.seed, a.row_offset + (uint32_t)r, (uint32_t)(c >> 2), a.step, a.tag);
        keep[0] = (u01(rr.x) >= a.p_drop) ? a.keep_scale : 0.f;
        keep[1] = (u01(rr.y) >= a.p_drop) ? a.keep_scale : 0.f;
        keep[2] = (u01(rr.z) >= a.p_drop) ? a.keep_scale : 0.f;
        keep[3] = (u01(rr.w) >= a.p_drop) ? a.keep_scale : 0.f;
      }
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
      const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
      const float gm[4] = {gam[j].x, gam[j].y, gam[j].z, gam[j].w};
      const float bt[4] = {bet[j].x, bet[j].y, bet[j].z, bet[j].w};
      float zh[4], gzh[4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        zh[e] = (zv[e] - mean) * rstd;
        const float y = zh[e] * gm[e] + bt[e];
        const float sg = sigmoid_f(y);
        const float gy = gv[e] * keep[e] * (sg * (1.0f + y * (1.0f - sg)));
        acc_g[j][e] += gy * zh[e];
        acc_b[j][e] += gy;
        gzh[e] = gy * gm[e];
        s1 += gzh[e];
        s2 += gzh[e] * zh[e];
      }
#pragma unroll
      for (int o = 1; o < GW / 4; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      const float m1 = s1 * (1.0f / GW), m2 = s2 * (1.0f / GW);
      float gz[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { gz[e] = rstd * (gzh[e] - m1 - zh[e] * m2); acc_z[j][e] += gz[e]; }
      if (in) st4g(a.gz + r * C, c, C, make_float4(gz[0], gz[1], gz[2], gz[3]));
    }
  }
  // column sums: waves of the block through LDS, then one partial row per block (summed by
  // k_partial_reduce in a fixed order: deterministic, no atomics)
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * lane + 256 * j + e;
      if (c < C) {
        red[(0 * NW + w) * C + c] = acc_g[j][e];
        red[(1 * NW + w) * C + c] = acc_b[j][e];
        red[(2 * NW + w) * C + c] = acc_z[j][e];
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
    const int q = i / C, c = i - q * C;
    float v = 0.f;
    for (int ww = 0; ww < NW; ww += 4)                    // fixed order within the block
      v += (red[(q * NW + ww) * C + c] + red[(q * NW + ww + 1) * C + c]) + (red[(q * NW + ww + 2) * C + c] + red[(q * NW + ww + 3) * C + c]);
    if (a.atomic_cols) atomicAdd((q == 0 ? a.dgamma : q == 1 ? a.dbeta : a.dbias) + c, v);     // targets zeroed by the caller
    else a.partials[(size_t)blockIdx.x * 3 * C + i] = v;
  }
}

// out_q[c] = sum_b partials[b][q][c], q = 0..2 -> (dgamma, dbeta, dbias).  64 columns per block, the
// blocks-of-partials axis split over 4 thread rows (independent loads in flight), fixed summation order.
__global__ __launch_bounds__(256) void k_partial_reduce(const float* partials, int nb, int C, float* o0, float* o1, float* o2) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, bp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (i < 3 * C) {
#pragma unroll 8
    for (int b = bp; b < nb; b += 4) s += partials[(size_t)b * 3 * C + i];
  }
  part[bp][cx] = s;
  __syncthreads();
  if (bp == 0 && i < 3 * C) {
    const float v = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
    const int q = i / C, c = i - q * C;
    (q == 0 ? o0 : q == 1 ? o1 : o2)[c] = v;
  }
}

static int gn_bwd_waves(int C) { return (size_t)3 * GN_BWD_WAVES * C * 4 <= 160 * 1024 ? GN_BWD_WAVES : GN_BWD_WAVES / 2; }
int gn_bwd_blocks(int64_t rows) {
  int blocks = (int)((rows + GN_BWD_WAVES - 1) / GN_BWD_WAVES);      // one row per wave until the block cap
  if (blocks > GN_BWD_MAX_BLOCKS) blocks = GN_BWD_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return blocks;
}

template <int GW>
static hipError_t gn_bwd_go(hipStream_t s, const GnBwdArgs& a) {
  const int nj = (a.C + 255) / 256;
  const int blocks = gn_bwd_blocks(a.rows);
  const int nw = nj <= 2 ? gn_bwd_waves(a.C) : GN_BWD_WAVES / 2;
  const size_t lds = (size_t)3 * nw * a.C * sizeof(float);               // 96 KB at C = 512
  static bool raised[5] = {false, false, false, false, false};           // per (GW, nj) instantiation of this function template
  auto go = [&](auto kern) -> hipError_t {
    if (!raised[nj] && lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      raised[nj] = true;
    }
    hipLaunchKernelGGL(kern, blocks, 64 * nw, lds, s, a);
    return hipSuccess;
  };
  hipError_t e = hipSuccess;
  switch (nj) {
    case 1: e = go(k_gn_silu_bwd<GW, 1>); break;
    case 2: e = go(k_gn_silu_bwd<GW, 2>); break;
    case 3: e = go(k_gn_silu_bwd<GW, 3>); break;
    case 4: e = go(k_gn_silu_bwd<GW, 4>); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  if (!a.atomic_cols) hipLaunchKernelGGL(k_partial_reduce, (3 * a.C + 63) / 64, 256, 0, s, a.partials, blocks, a.C, a.dgamma, a.dbeta, a.dbias);
  return hipGetLastError();
}
hipError_t launch_gn_silu_bwd(hipStream_t s, int gw, const GnBwdArgs& a) {
  if (a.rows <= 0) return hipSuccess;
  switch (gw) {
    case 4: return gn_bwd_go<4>(s, a);
    case 8: return gn_bwd_go<8>(s, a);
    case 16: return gn_bwd_go<16>(s, a);
    case 32: return gn_bwd_go<32>(s, a);
    case 64: return gn_bwd_go<64>(s, a);
    case 128: return gn_bwd_go<128>(s, a);
    default: return hipErrorInvalidValue;
  }
}

// ---- split-K wgrad: out[r][c] = sum_s slabs[s][r][c] (fixed order), out has leading dimension ldo ----
__global__ void k_slab_reduce(const float* slabs, int ns, int rows, int cols, int64_t stride, float* out, int ldo) {
  const int c4n = cols >> 2;
  const int64_t total = (int64_t)rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4n);
    const int c = 4 * (int)(i - (int64_t)r * c4n);
    const float* p = slabs + (size_t)r * cols + c;
    float4 acc = *reinterpret_cast<const float4*>(p);
    for (int sidx = 1; sidx < ns; ++sidx) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sidx * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)r * ldo + c) = acc;
  }
}
hipError_t launch_slab_reduce(hipStream_t s, const float* slabs, int ns, int rows, int cols, int64_t stride, float* out, int ldo) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_slab_reduce, ew_grid((int64_t)rows * cols / 4), 256, 0, s, slabs, ns, rows, cols, stride, out, ldo);
  return hipGetLastError();
}

// ---- wgrad of a Linear with very few inputs (the condition MLP, cond_dim ~ 3):
// dW[n][k] = sum_m gz[m][n] * x[m][k], one thread per (n,k), rows split over blocks, atomics at the end
// dw and dbias (optional: the threads of column k == 0 also carry sum_m gz[m][n]) are ADDED to: zeroed by the caller
__global__ __launch_bounds__(1024) void k_small_wgrad(const float* x, int kin, const float* gz, int ldg, int nout, int64_t rows, int rows_per_block, float* dw, float* dbias) {
  // thread = (output i = n * kin + k, row group rg of 4): a block covers rows_per_block rows, each group a quarter of them; the
  // groups meet in LDS so that a block issues ONE atomic per output (the atomics on 192 addresses serialise in L2: with one
  // block per 64 rows they, not the loads, set the kernel's 20 us)
  __shared__ float part[3][256], partb[3][256];
  const int i = threadIdx.x & 255, rg = threadIdx.x >> 8;
  const bool live = i < nout * kin;
  const int n = live ? i / kin : 0, k = live ? i - n * kin : 0;
  const int per = rows_per_block / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block + (int64_t)rg * per;
  int64_t r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float s = 0.f, sb = 0.f;
  if (live) {
#pragma unroll 8
    for (int64_t r = r0; r < r1; ++r) { const float g = gz[r * ldg + n]; s += g * x[r * kin + k]; sb += g; }
  }
  if (rg > 0) { part[rg - 1][i] = s; partb[rg - 1][i] = sb; }
  __syncthreads();
  if (rg == 0 && live) {
    s += part[0][i] + part[1][i] + part[2][i];
    sb += partb[0][i] + partb[1][i] + partb[2][i];
    atomicAdd(dw + i, s);
    if (dbias && k == 0) atomicAdd(dbias + n, sb);
  }
}
hipError_t launch_small_wgrad(hipStream_t s, const float* x, int kin, const float* gz, int ldg, int nout, int64_t rows, float* dw, float* dbias) {
  if (rows <= 0) return hipSuccess;
  if (nout * kin > 256) return hipErrorInvalidValue;
  // one atomic per output and block: the atomics on <= 256 addresses of one L2 channel serialise, so big batches take fewer, longer blocks
  const int rpb = rows >= 4096 ? 256 : 64;
  hipLaunchKernelGGL(k_small_wgrad, (unsigned)((rows + rpb - 1) / rpb), 1024, 0, s, x, kin, gz, ldg, nout, rows, rpb, dw, dbias);
  return hipGetLastError();
}

// ---- clip_grad_norm_ + AdamW over flat buffers -----------------------------------------------
// Global L2 norm without atomics and without state: k_sumsq leaves one double per workgroup (at most 256, one workgroup per
// CU, each thread streaming float4s) and every workgroup of k_adamw adds those partials up again in a fixed order (2 KB from
// L2).  Deterministic, no accumulator to zero, nothing that depends on the previous call (r2: two atomically accumulated sums
// alternating with the step's parity -- a repeated or skipped step found a dirty accumulator).
constexpr int NORM_PARTS = 256;
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, int64_t n, double* parts) {
  double s = 0.0;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(g) & 15) == 0 ? n >> 2 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
  }
  for (int64_t i = 4 * n4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = g[i];
    s += (double)v * (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double part[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__device__ __forceinline__ void adamw_one(float& pi, float& gi, float& mi, float& vi, const AdamArgs& a, float coef) {
  if (a.max_norm > 0.f) gi = __fmul_rn(gi, coef);                       // clip_grad_norm_ scales the grads in place
  pi = __fmul_rn(pi, a.decay);                                          // p.mul_(1 - lr*wd)
  mi = __fadd_rn(mi, __fmul_rn(a.one_minus_b1, __fsub_rn(gi, mi)));     // exp_avg.lerp_(g, 1-b1)
  vi = __fmul_rn(vi, a.b2);
  vi = __fadd_rn(vi, __fmul_rn(__fmul_rn(a.one_minus_b2, gi), gi));     // mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(vi), a.bc2_sqrt), a.eps);
  pi = __fadd_rn(pi, __fmul_rn(a.neg_step_size, __fdiv_rn(mi, denom)));   // addcdiv_(m, denom, -lr/bc1)
}
// HBM-bound: 7 streams of n floats (read p, g, m, v; write p, m, v; g is written back only when clipping is on).  16-byte
// accesses when the four arrays are 16-byte aligned (flat parameter buffers are), scalar tail / fallback otherwise.
__global__ __launch_bounds__(256) void k_adamw(float* p, float* g, float* m, float* v, int64_t n, AdamArgs a, const double* parts, int nparts,
                                               float* norm_out) {
  __shared__ double sh[4];
  {
    double s = (int)threadIdx.x < nparts ? parts[threadIdx.x] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
  }
  float coef = 1.0f;
  const float norm = (float)sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
  if (a.max_norm > 0.f) {
    coef = a.max_norm / (norm + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
  }
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int64_t n4 = al ? n >> 2 : 0;
  float4* p4 = reinterpret_cast<float4*>(p); float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pi = p4[i], gi = g4[i], mi = m4[i], vi = v4[i];
    adamw_one(pi.x, gi.x, mi.x, vi.x, a, coef);
    adamw_one(pi.y, gi.y, mi.y, vi.y, a, coef);
    adamw_one(pi.z, gi.z, mi.z, vi.z, a, coef);
    adamw_one(pi.w, gi.w, mi.w, vi.w, a, coef);
    if (a.max_norm > 0.f) g4[i] = gi;
    p4[i] = pi; m4[i] = mi; v4[i] = vi;
  }
  for (int64_t i = 4 * n4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    adamw_one(pi, gi, mi, vi, a, coef);
    if (a.max_norm > 0.f) g[i] = gi;
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// norm_ws: device scratch of NORM_PARTS (256) doubles, no initial state required
hipError_t launch_clip_adamw(hipStream_t s, float* p, float* g, float* m, float* v, int64_t n, const AdamArgs& a, double* norm_ws, float* norm_out) {
  if (n <= 0) return hipSuccess;
  int grid = ew_grid(n, 256 * 8);
  if (grid > NORM_PARTS) grid = NORM_PARTS;
  hipLaunchKernelGGL(k_sumsq, grid, 256, 0, s, g, n, norm_ws);
  hipLaunchKernelGGL(k_adamw, ew_grid(n, 256 * 4), 256, 0, s, p, g, m, v, n, a, norm_ws, grid, norm_out);      // one float4 per thread
  return hipGetLastError();
}

}  // namespace osd
