// k_elem.hip -- HBM-bound elementwise kernels: Philox fills, q_sample, mixup, copies.
#include "kernels.h"

namespace osd {

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_add_int(int* p, int d) { *p += d; }
hipError_t launch_set_int(hipStream_t s, int* p, int v) { hipLaunchKernelGGL(k_set_int, 1, 1, 0, s, p, v); return hipGetLastError(); }
hipError_t launch_add_int(hipStream_t s, int* p, int d) { hipLaunchKernelGGL(k_add_int, 1, 1, 0, s, p, d); return hipGetLastError(); }

static inline int ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// one thread per (row, 4 columns)
__global__ void k_fill_randn(float* out, int ld, int64_t rows, int cols, uint64_t seed, uint32_t row_offset, uint32_t step, uint32_t tag) {
  const int c4n = (cols + 3) >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c4 = (int)(i - r * c4n);
    const float4 n = randn4(seed, row_offset + (uint32_t)r, (uint32_t)c4, step, tag);
    st4g(out + r * ld, 4 * c4, cols, n);
  }
}
hipError_t launch_fill_randn(hipStream_t s, float* out, int ld, int64_t rows, int cols, uint64_t seed, uint32_t row_offset, uint32_t step, uint32_t tag) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fill_randn, ew_grid(rows * ((cols + 3) / 4)), 256, 0, s, out, ld, rows, cols, seed, row_offset, step, tag);
  return hipGetLastError();
}

__global__ void k_copy2d(const float* src, int lds, float* dst, int ldd, int64_t rows, int cols) {
  const int c4n = (cols + 3) >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = 4 * (int)(i - r * c4n);
    st4g(dst + r * ldd, c, cols, ld4g(src + r * lds, c, cols));
  }
}
hipError_t launch_copy2d(hipStream_t s, const float* src, int lds, float* dst, int ldd, int64_t rows, int cols) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy2d, ew_grid(rows * ((cols + 3) / 4)), 256, 0, s, src, lds, dst, ldd, rows, cols);
  return hipGetLastError();
}

// q_sample (models/diffusion.py:337-340): x_t = sqrt_ac[t]*x0 + sqrt_1m[t]*eps, two roundings of the
// products then one add, as torch evaluates it.
// t == null: the row's timestep is drawn here -- torch.randint(0, T, (B,)) stand-in of models/diffusion.py:361, the same Philox
// word k_randint uses -- and written to t_out by the row's first thread.
// x_t rows have a stride of ldxt >= cols floats; the pad columns [cols, ldxt) are written as zeros (the training forward's
// input_proj reads them against clamped weights, gemm.h: a_kmax).  zl: buffers of the training call that must be zero before
// anything accumulates into them -- zeroed here, by the same grid, instead of by a launch of their own.
// Only the first ZERO_BLOCKS workgroups walk the list: every entry costs a wave two dependent scalar loads from the argument block
// (~40 entries: 10-15 us of latency in front of the wave's real work when all 4096 workgroups did it -- measured).
constexpr int ZERO_BLOCKS = 32;
__device__ __forceinline__ void zero_list(const ZeroList& zl) {
  if (blockIdx.x >= ZERO_BLOCKS) return;
  const int nb = gridDim.x < ZERO_BLOCKS ? gridDim.x : ZERO_BLOCKS;
  for (int e = 0; e < zl.n; ++e) {
    float* p = zl.ptr[e];
    const int64_t n = zl.count[e];
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)nb * blockDim.x) p[j] = 0.f;
  }
}
__device__ __forceinline__ void zero_pad(float* row, int cols, int ldxt, int c) {
  if (c + 4 >= cols)
    for (int k = cols; k < ldxt; ++k) row[k] = 0.f;
}

__global__ void k_q_sample(const float* x0, const int* t, const float* sqrt_ac, const float* sqrt_1m, const float* noise_in,
                           int64_t rows, int cols, uint64_t seed, uint32_t row_offset, float* x_t, int ldxt, float* noise_out, int* t_out, int T,
                           ZeroList zl) {
  zero_list(zl);
  const int c4n = (cols + 3) >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = 4 * (int)(i - r * c4n);
    int tt;
    if (t) tt = t[r];
    else {
      const uint4 rr = philox_at(seed, row_offset + (uint32_t)r, 0u, 0u, TAG_TSTEP);
      tt = (int)(((uint64_t)rr.x * (uint64_t)T) >> 32);
      if (c == 0) t_out[r] = tt;
    }
    const float a = sqrt_ac[tt], b = sqrt_1m[tt];
    const float4 x = ld4g(x0 + r * cols, c, cols);
    float4 n;
    if (noise_in) n = ld4g(noise_in + r * cols, c, cols);
    else n = randn4(seed, row_offset + (uint32_t)r, (uint32_t)(c >> 2), 0u, TAG_QNOISE);
    float4 o;
    o.x = __fadd_rn(__fmul_rn(a, x.x), __fmul_rn(b, n.x));
    o.y = __fadd_rn(__fmul_rn(a, x.y), __fmul_rn(b, n.y));
    o.z = __fadd_rn(__fmul_rn(a, x.z), __fmul_rn(b, n.z));
    o.w = __fadd_rn(__fmul_rn(a, x.w), __fmul_rn(b, n.w));
    st4g(x_t + r * ldxt, c, cols, o);
    zero_pad(x_t + r * ldxt, cols, ldxt, c);
    if (noise_out && noise_out != noise_in) st4g(noise_out + r * cols, c, cols, n);
  }
}
hipError_t launch_q_sample(hipStream_t s, const float* x0, const int* t, const float* sqrt_ac, const float* sqrt_1m,
                           const float* noise_in, int64_t rows, int cols, uint64_t seed, uint32_t row_offset,
                           float* x_t, float* noise_out, int* t_out, int T, int ldxt, const ZeroList* zl) {
  if (rows <= 0) return hipSuccess;
  if (!t && (!t_out || T < 1)) return hipErrorInvalidValue;
  ZeroList z{};
  if (zl) z = *zl;
  hipLaunchKernelGGL(k_q_sample, ew_grid(rows * ((cols + 3) / 4)), 256, 0, s, x0, t, sqrt_ac, sqrt_1m, noise_in, rows, cols, seed,
                     row_offset, x_t, ldxt > 0 ? ldxt : cols, noise_out, t_out, T, z);
  return hipGetLastError();
}

// The training step's prologue in one pass (SURVEY a5 + a11): batch row r is taken from a device-resident dataset -- row idx_a[r],
// mixed with row idx_b[r] as MixupAugmentation does (utils/train.py:117-119: lam * v + (1 - lam) * v[perm], products rounded
// separately, so the result has the bits of the reference's two-step evaluation) -- and goes straight through q_sample; the mixed
// row itself is only written when somebody needs it (x0_out: the constraint losses).  The row's first thread also mixes the
// condition row into cond_out.  Replaces gather + k_mixup3 + k_q_sample (three passes over the batch) by one.
// One workgroup = QS_ROWS batch rows; a thread walks the row's 16-byte quads t, t + 256, ... (no 64-bit division per element, the
// row's scalars -- t, the two schedule values, the two dataset rows -- are wave-uniform and loaded once), QS_ROWS x 2 independent
// quads per thread in flight at D = 2000.  First version (one quad per thread over a flat index): 37 us = 3.6 TB/s; row by row with
// the zero list folded in: 40 us; all QS_ROWS rows' loads ahead of the stores: 35.4 us = 3.8 TB/s (QS_ROWS 2: 35.9, 8: 46.8).
constexpr int QS_ROWS = 4;
__global__ __launch_bounds__(256) void k_q_sample_src(BatchSrc b, const int* t, const float* sqrt_ac, const float* sqrt_1m, const float* noise_in,
                                                      int64_t rows, int cols, int cd, uint64_t seed, uint32_t row_offset, float* x_t, int ldxt,
                                                      float* noise_out, int* t_out, int T, float* cond_out, float* x0_out, ZeroList zl) {
  zero_list(zl);
  const int c4n = (cols + 3) >> 2;
  for (int64_t r0 = (int64_t)blockIdx.x * QS_ROWS; r0 < rows; r0 += (int64_t)gridDim.x * QS_ROWS) {
    // the rows' scalars first (wave-uniform loads), then per 256-quad column block ALL dataset loads of the QS_ROWS rows before any
    // store: the stores may alias the loads as far as hipcc knows, so a load placed behind a store waits for it -- one row at a time
    // the kernel ran at 3.4 TB/s with two loads in flight per thread
    float a[QS_ROWS], bb[QS_ROWS];
    const float* xa[QS_ROWS]; const float* xb[QS_ROWS];
    int nrow = 0;
#pragma unroll
    for (int rr = 0; rr < QS_ROWS; ++rr) {
      const int64_t r = r0 + rr;
      a[rr] = bb[rr] = 0.f; xa[rr] = xb[rr] = b.data;
      if (r >= rows) continue;                       // uniform over the workgroup
      nrow = rr + 1;
      int tt;
      if (t) tt = t[r];
      else {
        const uint4 rnd = philox_at(seed, row_offset + (uint32_t)r, 0u, 0u, TAG_TSTEP);
        tt = (int)(((uint64_t)rnd.x * (uint64_t)T) >> 32);
        if (threadIdx.x == 0) t_out[r] = tt;
      }
      a[rr] = sqrt_ac[tt]; bb[rr] = sqrt_1m[tt];
      const int64_t ia = b.idx_a ? b.idx_a[r] : r;
      const int64_t ib = b.idx_b ? b.idx_b[r] : 0;
      xa[rr] = b.data + ia * b.ldd;
      xb[rr] = b.data + ib * b.ldd;
      if ((int)threadIdx.x < cd && cond_out) {
        float v = b.cond[ia * b.ldc + threadIdx.x];
        if (b.idx_b) v = __fadd_rn(__fmul_rn(b.lam, v), __fmul_rn(b.oml, b.cond[ib * b.ldc + threadIdx.x]));
        cond_out[r * cd + threadIdx.x] = v;
      }
    }
    for (int q0 = 0; q0 < c4n; q0 += 256) {
      const int q = q0 + (int)threadIdx.x;
      if (q >= c4n) continue;
      const int c = 4 * q;
      float4 x[QS_ROWS], y[QS_ROWS], nz[QS_ROWS];
#pragma unroll
      for (int rr = 0; rr < QS_ROWS; ++rr) {
        if (rr >= nrow) break;
        x[rr] = ld4g(xa[rr], c, cols);
        if (b.idx_b) y[rr] = ld4g(xb[rr], c, cols);
        if (noise_in) nz[rr] = ld4g(noise_in + (r0 + rr) * cols, c, cols);
      }
#pragma unroll
      for (int rr = 0; rr < QS_ROWS; ++rr) {
        if (rr >= nrow) break;
        const int64_t r = r0 + rr;
        float4 xv = x[rr];
        if (b.idx_b) {
          xv.x = __fadd_rn(__fmul_rn(b.lam, xv.x), __fmul_rn(b.oml, y[rr].x));
          xv.y = __fadd_rn(__fmul_rn(b.lam, xv.y), __fmul_rn(b.oml, y[rr].y));
          xv.z = __fadd_rn(__fmul_rn(b.lam, xv.z), __fmul_rn(b.oml, y[rr].z));
          xv.w = __fadd_rn(__fmul_rn(b.lam, xv.w), __fmul_rn(b.oml, y[rr].w));
        }
        float4 n;
        if (noise_in) n = nz[rr];
        else n = randn4(seed, row_offset + (uint32_t)r, (uint32_t)q, 0u, TAG_QNOISE);
        float4 o;
        o.x = __fadd_rn(__fmul_rn(a[rr], xv.x), __fmul_rn(bb[rr], n.x));
        o.y = __fadd_rn(__fmul_rn(a[rr], xv.y), __fmul_rn(bb[rr], n.y));
        o.z = __fadd_rn(__fmul_rn(a[rr], xv.z), __fmul_rn(bb[rr], n.z));
        o.w = __fadd_rn(__fmul_rn(a[rr], xv.w), __fmul_rn(bb[rr], n.w));
        st4g(x_t + r * ldxt, c, cols, o);
        zero_pad(x_t + r * ldxt, cols, ldxt, c);
        if (noise_out && noise_out != noise_in) st4g(noise_out + r * cols, c, cols, n);
        if (x0_out) st4g(x0_out + r * cols, c, cols, xv);
      }
    }
  }
}
hipError_t launch_q_sample_src(hipStream_t s, const BatchSrc& b, const int* t, const float* sqrt_ac, const float* sqrt_1m, const float* noise_in,
                               int64_t rows, int cols, int cd, uint64_t seed, uint32_t row_offset, float* x_t, float* noise_out, int* t_out, int T,
                               float* cond_out, float* x0_out, int ldxt, const ZeroList* zl) {
  if (rows <= 0) return hipSuccess;
  if (!t && (!t_out || T < 1)) return hipErrorInvalidValue;
  ZeroList z{};
  if (zl) z = *zl;
  if (cd > 256) return hipErrorInvalidValue;
  int64_t blocks = (rows + QS_ROWS - 1) / QS_ROWS;
  if (blocks > 4096) blocks = 4096;
  if (blocks < (zl ? 32 : 1)) blocks = zl ? 32 : 1;      // the zero list is walked by the first 32 workgroups
  hipLaunchKernelGGL(k_q_sample_src, dim3((unsigned)blocks), 256, 0, s, b, t, sqrt_ac, sqrt_1m, noise_in, rows, cols, cd, seed,
                     row_offset, x_t, ldxt > 0 ? ldxt : cols, noise_out, t_out, T, cond_out, x0_out, z);
  return hipGetLastError();
}

// caller-supplied timestep indices clamped into [lo, hi]: an index outside [0, T) is a caller error (the Python
// shim raises IndexError as the reference's buffer gather would); the clamp only keeps the table gathers in bounds
__global__ void k_clamp_int(const int* in, int64_t n, int lo, int hi, int* out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = in[i];
    out[i] = v < lo ? lo : (v > hi ? hi : v);
  }
}
hipError_t launch_clamp_int(hipStream_t s, const int* in, int64_t n, int lo, int hi, int* out) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_clamp_int, ew_grid(n), 256, 0, s, in, n, lo, hi, out);
  return hipGetLastError();
}

// torch.randint(0, T, (B,)) stand-in (models/diffusion.py:361): uniform ints from Philox.
__global__ void k_randint(int* out, int64_t n, int hi, uint64_t seed, uint32_t row_offset) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 r = philox_at(seed, row_offset + (uint32_t)i, 0u, 0u, TAG_TSTEP);
    out[i] = (int)(((uint64_t)r.x * (uint64_t)hi) >> 32);
  }
}
hipError_t launch_randint(hipStream_t s, int* out, int64_t n, int hi, uint64_t seed, uint32_t row_offset) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_randint, ew_grid(n), 256, 0, s, out, n, hi, seed, row_offset);
  return hipGetLastError();
}

// mixup (utils/train.py:117-119): lam*v + (1-lam)*v[perm], products rounded separately.
__global__ void k_mixup(const float* v, const int64_t* perm, float lam, float oml, int64_t rows, int cols, float* out) {
  const int c4n = (cols + 3) >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = 4 * (int)(i - r * c4n);
    const float4 a = ld4g(v + r * cols, c, cols);
    const float4 b = ld4g(v + perm[r] * cols, c, cols);
    float4 o;
    o.x = __fadd_rn(__fmul_rn(lam, a.x), __fmul_rn(oml, b.x));
    o.y = __fadd_rn(__fmul_rn(lam, a.y), __fmul_rn(oml, b.y));
    o.z = __fadd_rn(__fmul_rn(lam, a.z), __fmul_rn(oml, b.z));
    o.w = __fadd_rn(__fmul_rn(lam, a.w), __fmul_rn(oml, b.w));
    st4g(out + r * cols, c, cols, o);
  }
}
hipError_t launch_mixup(hipStream_t s, const float* v, const int64_t* perm, double lam, int64_t rows, int cols, float* out) {
  if (rows <= 0) return hipSuccess;
  // python: lam and (1 - lam) are float64 scalars; torch multiplies an fp32 tensor by each as fp32
  const float oml = (float)(1.0 - lam);
  hipLaunchKernelGGL(k_mixup, ew_grid(rows * ((cols + 3) / 4)), 256, 0, s, v, perm, (float)lam, oml, rows, cols, out);
  return hipGetLastError();
}

// the three tensors of MixupAugmentation.__call__ (data [n][D], conditions [n][cd], survival [n]) in one launch
__global__ void k_mixup3(const float* d, const float* c, const float* sv, const int64_t* perm, float lam, float oml, int64_t rows, int D, int cd,
                         float* od, float* oc, float* os) {
  const int d4 = (D + 3) >> 2, c4 = (cd + 3) >> 2;
  const int per_row = d4 + c4 + 1;
  const int64_t total = rows * per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    int j = (int)(i - r * per_row);
    const int64_t pr = perm[r];
    const float* src; float* dst; int cols, col;
    if (j < d4) { src = d; dst = od; cols = D; col = 4 * j; }
    else if (j < d4 + c4) { src = c; dst = oc; cols = cd; col = 4 * (j - d4); }
    else { src = sv; dst = os; cols = 1; col = 0; }
    if (!src || !dst) continue;
    const float4 a = ld4g(src + r * cols, col, cols);
    const float4 b = ld4g(src + pr * cols, col, cols);
    float4 o;
    o.x = __fadd_rn(__fmul_rn(lam, a.x), __fmul_rn(oml, b.x));
    o.y = __fadd_rn(__fmul_rn(lam, a.y), __fmul_rn(oml, b.y));
    o.z = __fadd_rn(__fmul_rn(lam, a.z), __fmul_rn(oml, b.z));
    o.w = __fadd_rn(__fmul_rn(lam, a.w), __fmul_rn(oml, b.w));
    st4g(dst + r * cols, col, cols, o);
  }
}
hipError_t launch_mixup3(hipStream_t s, const float* d, const float* c, const float* sv, const int64_t* perm, double lam, int64_t rows, int D,
                         int cd, float* od, float* oc, float* os) {
  if (rows <= 0) return hipSuccess;
  const float oml = (float)(1.0 - lam);
  hipLaunchKernelGGL(k_mixup3, ew_grid(rows * ((D + 3) / 4 + (cd + 3) / 4 + 1)), 256, 0, s, d, c, sv, perm, (float)lam, oml, rows, D, cd, od, oc, os);
  return hipGetLastError();
}

__global__ void k_threshold(const float* x, int ldx, int64_t rows, int cols, float thr, float* out) {
  const int64_t total = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    out[i] = (x[r * ldx + c] > thr) ? 1.0f : 0.0f;
  }
}
hipError_t launch_threshold(hipStream_t s, const float* x, int ldx, int64_t rows, int cols, float thr, float* out) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_threshold, ew_grid(rows * cols), 256, 0, s, x, ldx, rows, cols, thr, out);
  return hipGetLastError();
}

}  // namespace osd
