// validate.hip -- on-GPU validation metrics (utils/validation.py): RBF-MMD as a
// blocked Gram GEMM with an exp+reduce epilogue, per-feature two-sample Kolmogorov-Smirnov extremes over
// a hand-written segmented radix sort (segsort.h), within-pathway mean correlation and Pearson correlation as wavefront reductions.
// Entry points are stream/device based (no model handle) and synchronous: they return host scalars.
#include <limits.h>
#include <vector>
#include "handle.h"
#include "kernels.h"
#include "launch.h"
#include "segsort.h"

namespace osd {

__global__ void k_rowsumsq(const float* x, int64_t rows, int cols, float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nw) {
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[r * cols + c]; s = fmaf(v, v, s); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = s;
  }
}

// dst[f][r] = src[r][f] for f < nf
__global__ void k_gather_cols(const float* src, int ld, int64_t rows, int nf, float* dst) {
  const int64_t total = rows * nf;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nf;
    const int f = (int)(i - r * nf);
    dst[(int64_t)f * rows + r] = src[r * ld + f];
  }
}

__device__ __forceinline__ long long upper_bound_cnt(const float* a, long long n, float v) {   // #{a_i <= v}, a sorted
  long long lo = 0, hi = n;
  while (lo < hi) { const long long mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
  return lo;
}

// extremes over all sample points v of cnt(a<=v)*n2 - cnt(b<=v)*n1 (scipy.stats.ks_2samp's cddiffs, exact integers)
__global__ void k_ks_extremes(const float* a, const float* b, long long n1, long long n2, long long* omax, long long* omin) {
  const int f = blockIdx.y;
  const float* af = a + (long long)f * n1;
  const float* bf = b + (long long)f * n2;
  long long mx = LLONG_MIN, mn = LLONG_MAX;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * blockDim.x) {
    const float v = i < n1 ? af[i] : bf[i - n1];
    const long long d = upper_bound_cnt(af, n1, v) * n2 - upper_bound_cnt(bf, n2, v) * n1;
    mx = d > mx ? d : mx;
    mn = d < mn ? d : mn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long x = __shfl_xor(mx, o), y = __shfl_xor(mn, o);
    mx = x > mx ? x : mx;
    mn = y < mn ? y : mn;
  }
  if ((threadIdx.x & 63) == 0) { atomicMax(omax + f, mx); atomicMin(omin + f, mn); }
}

// per selected column: sum and sum of squares (double) -- lane j of a wave owns columns j, j+64, ...
template <int MAXJ>
__global__ void k_col_moments(const float* x, int ld, int64_t rows, const int* cols, int g, double* sum, double* sumsq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  double s[MAXJ], q[MAXJ];
  int cj[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) { s[j] = q[j] = 0.0; const int gi = lane + 64 * j; cj[j] = gi < g ? cols[gi] : -1; }
  for (int64_t r = wave; r < rows; r += nw)
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (cj[j] >= 0) { const double v = x[r * ld + cj[j]]; s[j] += v; q[j] += v * v; }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
    if (cj[j] >= 0) { atomicAdd(sum + lane + 64 * j, s[j]); atomicAdd(sumsq + lane + 64 * j, q[j]); }
}

// S = sum_rows (sum_g (x - mu_g) * inv_sd_g)^2
template <int MAXJ>
__global__ void k_rowz_sq(const float* x, int ld, int64_t rows, const int* cols, int g, const double* mu, const double* isd, double* out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int cj[MAXJ];
  double m[MAXJ], w[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int gi = lane + 64 * j;
    cj[j] = gi < g ? cols[gi] : -1;
    m[j] = gi < g ? mu[gi] : 0.0;
    w[j] = gi < g ? isd[gi] : 0.0;
  }
  double acc = 0.0;
  for (int64_t r = wave; r < rows; r += nw) {
    double z = 0.0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (cj[j] >= 0) z += ((double)x[r * ld + cj[j]] - m[j]) * w[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    acc += z * z;
  }
  if (lane == 0) atomicAdd(out, acc);
}

// five sums for Pearson: n is implicit
__global__ void k_pearson_sums(const float* a, int lda, const float* b, int ldb, int64_t rows, double* out5) {
  double sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const double x = a[r * lda], y = b[r * ldb];
    sa += x; sb += y; saa += x * x; sbb += y * y; sab += x * y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); saa += __shfl_xor(saa, o); sbb += __shfl_xor(sbb, o); sab += __shfl_xor(sab, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out5, sa); atomicAdd(out5 + 1, sb); atomicAdd(out5 + 2, saa); atomicAdd(out5 + 3, sbb); atomicAdd(out5 + 4, sab);
  }
}

static hipError_t rbf_sum(hipStream_t s, const float* X, int64_t n, const float* sqx, const float* Y, int64_t m, const float* sqy, int D,
                          float gamma, double* dsum) {
  GemmArgs g{};
  g.A = X; g.lda = D; g.B0 = Y; g.ldb0 = D; g.K0 = D; g.F = (int)n; g.P = (int)m; g.K = D;
  // K(X, X) is symmetric: the tiles on or above the diagonal, the off-diagonal ones weighted 2 -- half the MFMA work of the rectangle
  static_assert(TileBig::BF == 128 && TileBig::BP == 128, "the triangular schedule compares 128-row tile indices");
  g.tri = (X == Y && n == m && sqx == sqy) ? 1 : 0;
  EpiRbfSum::Args ea{sqx, sqy, gamma, dsum, g.tri};
  return launch_gemm<TileBig, true, true, EpiRbfSum>(s, g, ea);
}

struct DevBuf {
  void* p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  ~DevBuf() { if (p) { hipError_t e = hipFree(p); (void)e; } }
};

}  // namespace osd

using namespace osd;

extern "C" {

int osd_val_mmd(void* stream, int device, const float* X, int64_t n, const float* Y, int64_t m, int D, double gamma, double* mmd_out) {
  if (!X || !Y || !mmd_out || n <= 0 || m <= 0 || D <= 0 || n > INT_MAX / 2 || m > INT_MAX / 2) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  OSD_HIP(prepare_kernels());
  hipStream_t s = (hipStream_t)stream;
  DevBuf sq, sums;
  OSD_HIP(sq.alloc((size_t)(n + m) * 4));
  constexpr int NS = EpiRbfSum::RBF_SLOTS;
  OSD_HIP(sums.alloc(3 * NS * sizeof(double)));
  float* sqx = (float*)sq.p;
  float* sqy = sqx + n;
  double* d = (double*)sums.p;
  OSD_HIP(hipMemsetAsync(d, 0, 3 * NS * sizeof(double), s));
  hipLaunchKernelGGL(k_rowsumsq, 1024, 256, 0, s, X, n, D, sqx);
  hipLaunchKernelGGL(k_rowsumsq, 1024, 256, 0, s, Y, m, D, sqy);
  if (gamma <= 0) gamma = 1.0 / D;                                  // utils/validation.py:283-284
  OSD_HIP(rbf_sum(s, X, n, sqx, X, n, sqx, D, (float)gamma, d));
  OSD_HIP(rbf_sum(s, Y, m, sqy, Y, m, sqy, D, (float)gamma, d + NS));
  OSD_HIP(rbf_sum(s, X, n, sqx, Y, m, sqy, D, (float)gamma, d + 2 * NS));
  std::vector<double> slots(3 * NS);
  OSD_HIP(hipMemcpyAsync(slots.data(), d, slots.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  double h[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < NS; ++i) h[k] += slots[(size_t)k * NS + i];
  const double v = h[0] / ((double)n * n) + h[1] / ((double)m * m) - 2.0 * h[2] / ((double)n * m);
  *mmd_out = sqrt(v > 0 ? v : 0.0);
  return OSD_OK;
}

int osd_val_rbf_sum(void* stream, int device, const float* A, int64_t n, const float* B, int64_t m, int D, double gamma, double* sum_out) {
  if (!A || !B || !sum_out || n <= 0 || m <= 0 || D <= 0 || n > INT_MAX / 2 || m > INT_MAX / 2 || gamma <= 0) { set_error("bad argument (gamma must be > 0)"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  OSD_HIP(prepare_kernels());
  hipStream_t s = (hipStream_t)stream;
  DevBuf sq, sums;
  OSD_HIP(sq.alloc((size_t)(n + m) * 4));
  constexpr int NS = EpiRbfSum::RBF_SLOTS;
  OSD_HIP(sums.alloc(NS * sizeof(double)));
  float* sqa = (float*)sq.p;
  float* sqb = sqa + n;
  OSD_HIP(hipMemsetAsync(sums.p, 0, NS * sizeof(double), s));
  hipLaunchKernelGGL(k_rowsumsq, 1024, 256, 0, s, A, n, D, sqa);
  if (A == B && n == m) sqb = sqa;                     // one operand: the symmetric (triangular) schedule of rbf_sum
  else hipLaunchKernelGGL(k_rowsumsq, 1024, 256, 0, s, B, m, D, sqb);
  OSD_HIP(rbf_sum(s, A, n, sqa, B, m, sqb, D, (float)gamma, (double*)sums.p));
  std::vector<double> slots(NS);
  OSD_HIP(hipMemcpyAsync(slots.data(), sums.p, NS * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  double t = 0.0;
  for (double v : slots) t += v;
  *sum_out = t;
  return OSD_OK;
}

int osd_val_ks_extremes(void* stream, int device, const float* real, int64_t n1, const float* synth, int64_t n2, int ld, int nf,
                        int64_t* dmax_out, int64_t* dmin_out) {
  if (!real || !synth || !dmax_out || !dmin_out || n1 <= 0 || n2 <= 0 || nf <= 0 || nf > ld) { set_error("bad argument"); return OSD_EINVAL; }
  if ((double)n1 * nf > 2.0e9 || (double)n2 * nf > 2.0e9) { set_error("too many items for one segmented sort"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf cols, sorted, ext, tmp;
  const size_t na = (size_t)n1 * nf, nb = (size_t)n2 * nf;
  OSD_HIP(cols.alloc((na + nb) * 4));
  OSD_HIP(sorted.alloc((na + nb) * 4));
  OSD_HIP(ext.alloc((size_t)2 * nf * sizeof(long long)));
  float* ca = (float*)cols.p; float* cb = ca + na;
  float* sa = (float*)sorted.p; float* sb = sa + na;
  hipLaunchKernelGGL(k_gather_cols, 2048, 256, 0, s, real, ld, n1, nf, ca);
  hipLaunchKernelGGL(k_gather_cols, 2048, 256, 0, s, synth, ld, n2, nf, cb);
  // four stable 8-bit passes, ping-pong between the two buffers: the sorted floats end up back in `cols`
  {
    const int wps_a = (int)((n1 + SEG_CHUNK - 1) / SEG_CHUNK), wps_b = (int)((n2 + SEG_CHUNK - 1) / SEG_CHUNK);
    OSD_HIP(tmp.alloc((size_t)nf * 256 * (size_t)(wps_a > wps_b ? wps_a : wps_b) * sizeof(int)));
    int* cnt = (int*)tmp.p;
    auto sort_segments = [&](float* buf0, float* buf1, long long n, int wps) -> hipError_t {
      uint32_t* a = (uint32_t*)buf0; uint32_t* b = (uint32_t*)buf1;
      const dim3 grid((unsigned)((wps + SEG_WAVES - 1) / SEG_WAVES), (unsigned)nf), block(64 * SEG_WAVES);
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        if (pass == 0) hipLaunchKernelGGL((k_seg_count<true>), grid, block, 0, s, a, n, wps, shift, cnt);
        else hipLaunchKernelGGL((k_seg_count<false>), grid, block, 0, s, a, n, wps, shift, cnt);
        hipLaunchKernelGGL(k_seg_scan, dim3((unsigned)nf), dim3(256), 0, s, cnt, wps);
        if (pass == 0) hipLaunchKernelGGL((k_seg_scatter<true, false>), grid, block, 0, s, a, b, n, wps, shift, cnt);
        else if (pass == 3) hipLaunchKernelGGL((k_seg_scatter<false, true>), grid, block, 0, s, a, b, n, wps, shift, cnt);
        else hipLaunchKernelGGL((k_seg_scatter<false, false>), grid, block, 0, s, a, b, n, wps, shift, cnt);
        uint32_t* t = a; a = b; b = t;
      }
      return hipGetLastError();
    };
    OSD_HIP(sort_segments(ca, sa, (long long)n1, wps_a));
    OSD_HIP(sort_segments(cb, sb, (long long)n2, wps_b));
    sa = ca; sb = cb;                     // an even number of passes: back in the first buffer
  }
  std::vector<long long> init((size_t)2 * nf);
  for (int f = 0; f < nf; ++f) { init[f] = LLONG_MIN; init[nf + f] = LLONG_MAX; }
  long long* emax = (long long*)ext.p; long long* emin = emax + nf;
  OSD_HIP(hipMemcpyAsync(emax, init.data(), init.size() * sizeof(long long), hipMemcpyHostToDevice, s));
  int bx = (int)((n1 + n2 + 255) / 256);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_ks_extremes, dim3(bx, nf), 256, 0, s, sa, sb, (long long)n1, (long long)n2, emax, emin);
  OSD_HIP(hipGetLastError());
  std::vector<long long> res((size_t)2 * nf);
  OSD_HIP(hipMemcpyAsync(res.data(), emax, res.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < nf; ++f) { dmax_out[f] = res[f]; dmin_out[f] = res[nf + f]; }
  return OSD_OK;
}

static int check_cols(const int32_t* cols_host, int g, int ld) {
  if (!cols_host || g < 1 || g > 512) { set_error("bad column list (1 <= columns <= 512)"); return OSD_EINVAL; }
  for (int i = 0; i < g; ++i)
    if (cols_host[i] < 0 || cols_host[i] >= ld) { set_error("column index out of range"); return OSD_EINVAL; }
  return OSD_OK;
}

int osd_val_col_moments(void* stream, int device, const float* data, int64_t rows, int ld, const int32_t* cols_host, int g,
                        double* sum_host, double* sumsq_host) {
  if (!data || !sum_host || !sumsq_host || rows < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_TRY(check_cols(cols_host, g, ld));
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf dc, dm;
  OSD_HIP(dc.alloc((size_t)g * sizeof(int)));
  OSD_HIP(dm.alloc((size_t)2 * g * sizeof(double)));
  OSD_HIP(hipMemcpyAsync(dc.p, cols_host, (size_t)g * sizeof(int), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemsetAsync(dm.p, 0, (size_t)2 * g * sizeof(double), s));
  int blocks = (int)((rows + 15) / 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((k_col_moments<8>), blocks, 256, 0, s, data, ld, rows, (const int*)dc.p, g, (double*)dm.p, (double*)dm.p + g);
  std::vector<double> hm((size_t)2 * g);
  OSD_HIP(hipMemcpyAsync(hm.data(), dm.p, (size_t)2 * g * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < g; ++i) { sum_host[i] = hm[i]; sumsq_host[i] = hm[g + i]; }
  return OSD_OK;
}

int osd_val_rowz_sq(void* stream, int device, const float* data, int64_t rows, int ld, const int32_t* cols_host, int g, const double* mu_host,
                    const double* isd_host, double* S_host) {
  if (!data || !mu_host || !isd_host || !S_host || rows < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_TRY(check_cols(cols_host, g, ld));
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf dc, dm;
  OSD_HIP(dc.alloc((size_t)g * sizeof(int)));
  OSD_HIP(dm.alloc((size_t)(2 * g + 1) * sizeof(double)));
  double* mu = (double*)dm.p; double* isd = mu + g; double* S = isd + g;
  OSD_HIP(hipMemcpyAsync(dc.p, cols_host, (size_t)g * sizeof(int), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemcpyAsync(mu, mu_host, (size_t)g * sizeof(double), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemcpyAsync(isd, isd_host, (size_t)g * sizeof(double), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemsetAsync(S, 0, sizeof(double), s));
  int blocks = (int)((rows + 15) / 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL((k_rowz_sq<8>), blocks, 256, 0, s, data, ld, rows, (const int*)dc.p, g, mu, isd, S);
  OSD_HIP(hipMemcpyAsync(S_host, S, sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_val_mean_offdiag_corr(void* stream, int device, const float* data, int64_t rows, int ld, const int32_t* cols_host, int g,
                              double* out) {
  if (!data || !cols_host || !out || rows < 2 || g < 2 || g > 512) { set_error("bad argument (2 <= genes <= 512, rows >= 2)"); return OSD_EINVAL; }
  std::vector<double> sum((size_t)g), sumsq((size_t)g), mu((size_t)g), isd((size_t)g);
  OSD_TRY(osd_val_col_moments(stream, device, data, rows, ld, cols_host, g, sum.data(), sumsq.data()));
  for (int i = 0; i < g; ++i) {
    const double m = sum[i] / rows;
    const double var = (sumsq[i] - rows * m * m) / (rows - 1);          // ddof = 1, as pandas .corr()
    mu[i] = m;
    isd[i] = var > 0 ? 1.0 / sqrt(var) : NAN;                           // constant column -> NaN, as pandas
  }
  double hs = 0;
  OSD_TRY(osd_val_rowz_sq(stream, device, data, rows, ld, cols_host, g, mu.data(), isd.data(), &hs));
  *out = (hs / (rows - 1) - g) / ((double)g * (g - 1));
  return OSD_OK;
}

int osd_val_column_sums(void* stream, int device, const float* x, int64_t rows, int ld, int cols, double* sums_host) {
  if (!x || !sums_host || rows < 1 || cols < 1 || ld < cols) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf d;
  OSD_HIP(d.alloc((size_t)2 * cols * sizeof(double)));
  OSD_HIP(hipMemsetAsync(d.p, 0, (size_t)2 * cols * sizeof(double), s));
  OSD_HIP(cons_column_sums(s, x, ld, rows, cols, (double*)d.p));
  OSD_HIP(hipMemcpyAsync(sums_host, d.p, (size_t)cols * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_val_gram(void* stream, int device, const float* x, int64_t rows, int ld, const int32_t* cols_host, int g, double* gram_host) {
  if (!x || !cols_host || !gram_host || rows < 1 || g < 1 || g > CONS_MAX_SET) { set_error("bad argument (1 <= columns <= %d)", CONS_MAX_SET); return OSD_EINVAL; }
  for (int i = 0; i < g; ++i)
    if (cols_host[i] < 0 || cols_host[i] >= ld) { set_error("column index out of range"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf dc, dm, dC;
  OSD_HIP(dc.alloc((size_t)g * sizeof(int)));
  OSD_HIP(dm.alloc((size_t)ld * sizeof(float2)));
  OSD_HIP(dC.alloc((size_t)CONS_MAX_SET * CONS_MAX_SET * sizeof(double)));
  std::vector<float2> ident((size_t)ld, make_float2(0.f, 1.f));            // no standardisation: raw products
  OSD_HIP(hipMemcpyAsync(dc.p, cols_host, (size_t)g * sizeof(int), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemcpyAsync(dm.p, ident.data(), (size_t)ld * sizeof(float2), hipMemcpyHostToDevice, s));
  OSD_HIP(hipMemsetAsync(dC.p, 0, (size_t)CONS_MAX_SET * CONS_MAX_SET * sizeof(double), s));
  OSD_HIP(cons_gram(s, x, ld, rows, (const int*)dc.p, g, (const int*)dc.p, g, (const float2*)dm.p, (double*)dC.p));
  std::vector<double> full((size_t)CONS_MAX_SET * CONS_MAX_SET);
  OSD_HIP(hipMemcpyAsync(full.data(), dC.p, full.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < g; ++i)
    for (int j = 0; j < g; ++j) gram_host[(size_t)i * g + j] = full[(size_t)i * CONS_MAX_SET + j];
  return OSD_OK;
}

int osd_val_pearson_sums(void* stream, int device, const float* a, int lda, const float* b, int ldb, int64_t rows, double* out5_host) {
  if (!a || !b || !out5_host || rows < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  DevBuf d;
  OSD_HIP(d.alloc(5 * sizeof(double)));
  OSD_HIP(hipMemsetAsync(d.p, 0, 5 * sizeof(double), s));
  int blocks = (int)((rows + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_pearson_sums, blocks, 256, 0, s, a, lda, b, ldb, rows, (double*)d.p);
  OSD_HIP(hipMemcpyAsync(out5_host, d.p, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_val_pearson(void* stream, int device, const float* a, int lda, const float* b, int ldb, int64_t rows, double* out) {
  if (!out || rows < 2) { set_error("bad argument"); return OSD_EINVAL; }
  double h[5];
  OSD_TRY(osd_val_pearson_sums(stream, device, a, lda, b, ldb, rows, h));
  const double n = (double)rows;
  const double cov = h[4] - h[0] * h[1] / n, va = h[2] - h[0] * h[0] / n, vb = h[3] - h[1] * h[1] / n;
  *out = cov / sqrt(va * vb);
  return OSD_OK;
}

}  // extern "C"
