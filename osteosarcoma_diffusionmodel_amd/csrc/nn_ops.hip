// nn_ops.hip -- layer-level ops behind the cVAE mirror (models/cvae.py): concat-free Linear
// forward/backward on the MFMA GEMM kernels, BatchNorm1d + ReLU + Dropout forward/backward (batch statistics as
// double column sums), reparameterisation and the VAE loss.  Stream/device based, asynchronous, no model handle.
#include <math.h>
#include "handle.h"
#include "kernels.h"
#include "kernels_train.h"
#include "rng.h"

namespace osd {

static int ew_blocks(int64_t total) { int64_t b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

__device__ __forceinline__ float keep_of(const float* mask, int64_t i, int64_t r, int c, float p, float scale, uint64_t seed, uint32_t tag) {
  if (mask) return mask[i] * scale;
  const uint4 rr = philox_at(seed, (uint32_t)r, (uint32_t)(c >> 2), 0u, tag);
  const uint32_t w = (c & 3) == 0 ? rr.x : (c & 3) == 1 ? rr.y : (c & 3) == 2 ? rr.z : rr.w;
  return (u01(w) >= p) ? scale : 0.f;
}

// column sums of v and v*v (v = z) or of g_act and g_act*zhat (backward): block = 64 columns x 4 row phases
template <bool BWD>
__global__ void k_bn_colsums(const float* z, const float* gy, int64_t rows, int C, int rows_per_block, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, int use_bn, float p, float scale, const float* mask, uint64_t seed,
                             uint32_t tag, int drop, double* s0, double* s1) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  double a = 0.0, b = 0.0;
  if (c < C) {
    float m = 0.f, is = 1.f, ga = 1.f, be = 0.f;
    if (BWD && use_bn) { m = mean[c]; is = invstd[c]; ga = gamma[c]; be = beta[c]; }
    for (int64_t r = r0 + ty; r < r1; r += 4) {
      const int64_t i = r * C + c;
      if (!BWD) { const double v = z[i]; a += v; b += v * v; }
      else {
        const float zh = (z[i] - m) * is;
        const float pre = zh * ga + be;
        float g = pre > 0.f ? gy[i] : 0.f;
        if (drop) g *= keep_of(mask, i, r, c, p, scale, seed, tag);
        a += (double)g; b += (double)g * (double)zh;
      }
    }
  }
  __shared__ double sh[2][4][64];
  sh[0][ty][tx] = a; sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
    atomicAdd(s0 + c, (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]));
    atomicAdd(s1 + c, (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]));
  }
}

// batch mean / biased variance -> (mean, invstd); running statistics as nn.BatchNorm1d updates them
__global__ void k_bn_finish(const double* s0, const double* s1, int64_t rows, int C, double momentum, double eps, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = s0[c] / (double)rows;
  double var = s1[c] / (double)rows - m * m;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)m;
  save_invstd[c] = (float)(1.0 / sqrt(var + eps));
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
  if (running_var) running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * var * (double)rows / (double)(rows - 1));
}
__global__ void k_bn_eval_stats(const float* running_mean, const float* running_var, int C, double eps, float* save_mean, float* save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = running_mean[c];
  save_invstd[c] = (float)(1.0 / sqrt((double)running_var[c] + eps));
}

__global__ void k_bn_act_fwd(const float* z, int64_t rows, int C, const float* gamma, const float* beta, const float* mean, const float* invstd,
                             int use_bn, int drop, float p, float scale, const float* mask, uint64_t seed, uint32_t tag, float* y) {
  const int64_t total = rows * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    float v = z[i];
    if (use_bn) v = (v - mean[c]) * invstd[c] * gamma[c] + beta[c];
    v = v > 0.f ? v : 0.f;
    if (drop) v *= keep_of(mask, i, r, c, p, scale, seed, tag);
    y[i] = v;
  }
}

// dz = gamma*invstd*(g_act - sum(g_act)/n - zhat*sum(g_act*zhat)/n) in training mode; without the batch terms in eval mode
__global__ void k_bn_act_bwd(const float* gy, const float* z, int64_t rows, int C, const float* gamma, const float* beta, const float* mean,
                             const float* invstd, int use_bn, int training, int drop, float p, float scale, const float* mask, uint64_t seed,
                             uint32_t tag, const double* s0, const double* s1, float* dz, float* dgamma, float* dbeta) {
  const int64_t total = rows * C;
  const double inv_n = 1.0 / (double)rows;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    float m = 0.f, is = 1.f, ga = 1.f, be = 0.f;
    if (use_bn) { m = mean[c]; is = invstd[c]; ga = gamma[c]; be = beta[c]; }
    const float zh = (z[i] - m) * is;
    const float pre = zh * ga + be;
    float g = pre > 0.f ? gy[i] : 0.f;
    if (drop) g *= keep_of(mask, i, r, c, p, scale, seed, tag);
    if (use_bn) {
      float t = g;
      if (training) t = (float)((double)g - s0[c] * inv_n - (double)zh * s1[c] * inv_n);
      dz[i] = ga * is * t;
    } else {
      dz[i] = g;
    }
    if (use_bn && r == 0) { dbeta[c] = (float)s0[c]; dgamma[c] = (float)s1[c]; }
  }
}

__global__ void k_reparam(const float* mu, const float* logvar, const float* eps_in, uint64_t seed, int64_t rows, int Lz, float* z, float* eps_out) {
  const int64_t total = rows * Lz;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float e;
    if (eps_in) e = eps_in[i];
    else {
      const int64_t r = i / Lz;
      const int c = (int)(i - r * Lz);
      const float4 n4 = randn4(seed, (uint32_t)r, (uint32_t)(c >> 2), 0u, TAG_USER + 0x7eu);
      e = (c & 3) == 0 ? n4.x : (c & 3) == 1 ? n4.y : (c & 3) == 2 ? n4.z : n4.w;
    }
    if (eps_out) eps_out[i] = e;
    z[i] = mu[i] + e * expf(0.5f * logvar[i]);
  }
}

__global__ void k_reparam_bwd(const float* gz, const float* mu, const float* z, int64_t count, float* d_lv) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    d_lv[i] = 0.5f * gz[i] * (z[i] - mu[i]);
}

// parts[1] += sum (xr - x)^2 / n ; parts[2] += -0.5 sum(1 + lv - mu^2 - exp(lv)) / n ; parts[0] = their sum
__global__ void k_vae_loss(const float* xr, const float* x, int64_t nd, const float* mu, const float* lv, int64_t nl, double inv_n, float* parts,
                           float* d_recon, float* d_mu, float* d_lv) {
  double rec = 0.0, kl = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nd; i += stride) {
    const float d = xr[i] - x[i];
    rec += (double)d * (double)d;
    if (d_recon) d_recon[i] = (float)(2.0 * inv_n) * d;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nl; i += stride) {
    const float m = mu[i], l = lv[i], e = expf(l);
    kl += (double)(1.f + l - m * m - e);
    if (d_mu) d_mu[i] = (float)inv_n * m;
    if (d_lv) d_lv[i] = (float)(0.5 * inv_n) * (e - 1.f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { rec += __shfl_xor(rec, o); kl += __shfl_xor(kl, o); }
  __shared__ double sh[2][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wv] = rec; sh[1][wv] = kl; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = (float)(((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3])) * inv_n);
    const float k = (float)(-0.5 * ((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3])) * inv_n);
    atomicAdd(parts + 1, r); atomicAdd(parts + 2, k); atomicAdd(parts, r + k);
  }
}

// loss += mean (a - b)^2 ; da = 2 (a - b) / count
__global__ void k_mse_mean(const float* a, const float* b, int64_t count, float* loss, float* da) {
  double acc = 0.0;
  const double inv = 1.0 / (double)count;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    acc += (double)d * (double)d;
    if (da) da[i] = (float)(2.0 * inv) * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, (float)(acc * inv));
}

}  // namespace osd

using namespace osd;

extern "C" {

int osd_nn_linear(void* stream, int device, const float* x1, int K1, const float* x2, int K2, const float* w, const float* b, int64_t n, int N,
                  float* y) {
  if (!x1 || !w || !y || K1 < 1 || K2 < 0 || N < 1 || (K2 > 0 && !x2)) { set_error("bad argument"); return OSD_EINVAL; }
  if (n < 1 || n > 0x7fffffffLL) { set_error("row count out of range"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g{};
  g.A = w; g.lda = K1 + K2; g.B0 = x1; g.ldb0 = K1; g.K0 = K1; g.F = N; g.P = (int)n; g.K = K1;
  if (K2 > 0 && K1 % 4 == 0) {            // two K panels in one launch (concat-free)
    g.B1 = x2; g.ldb1 = K2; g.K = K1 + K2;
    OSD_HIP(launch_linear(s, g, true, true, b, y, N, false, false));
    return OSD_OK;
  }
  if (K2 == 0) {
    OSD_HIP(launch_linear(s, g, true, true, b, y, N, false, false));
    return OSD_OK;
  }
  // K1 not a multiple of 4: materialise the concatenation (torch.cat, models/cvae.py:54) and run one panel
  const int Kt = K1 + K2;
  float* cat = nullptr;
  OSD_HIP(hipMallocAsync((void**)&cat, (size_t)n * Kt * 4, s));
  OSD_HIP(launch_copy2d(s, x1, K1, cat, Kt, n, K1));
  OSD_HIP(launch_copy2d(s, x2, K2, cat + K1, Kt, n, K2));
  g.B0 = cat; g.ldb0 = Kt; g.K0 = Kt; g.K = Kt;
  OSD_HIP(launch_linear(s, g, true, true, b, y, N, false, false));
  OSD_HIP(hipFreeAsync(cat, s));
  return OSD_OK;
}

int osd_nn_linear_bwd(void* stream, int device, const float* x1, int K1, const float* x2, int K2, const float* w, const float* gy, int64_t n,
                      int N, float* dx1, float* dw, float* db) {
  if (!x1 || !w || !gy || !dw || K1 < 1 || K2 < 0 || N < 1 || (K2 > 0 && !x2)) { set_error("bad argument"); return OSD_EINVAL; }
  if (n < 1 || n > 0x7fffffffLL) { set_error("row count out of range"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int Kt = K1 + K2;
  {  // dw[N][:K1] = gy^T x1  (out[p = n_out][f = k_in] = sum_m A(f, m) B(p, m), both operands stored [m][.])
    GemmArgs g{};
    g.A = x1; g.lda = K1; g.B0 = gy; g.ldb0 = N; g.K0 = (int)n; g.F = K1; g.P = N; g.K = (int)n;
    OSD_HIP(launch_linear(s, g, false, false, nullptr, dw, Kt, false, false));
    if (K2 > 0) {
      g.A = x2; g.lda = K2; g.F = K2;
      OSD_HIP(launch_linear(s, g, false, false, nullptr, dw + K1, Kt, false, false));
    }
  }
  if (db) {
    OSD_HIP(hipMemsetAsync(db, 0, (size_t)N * 4, s));
    OSD_HIP(launch_colsum(s, gy, N, n, N, db));
  }
  if (dx1) {  // dx1[m][k] = sum_n gy[m][n] w[n][k]
    GemmArgs g{};
    g.A = w; g.lda = Kt; g.B0 = gy; g.ldb0 = N; g.K0 = N; g.F = K1; g.P = (int)n; g.K = N;
    OSD_HIP(launch_linear(s, g, false, true, nullptr, dx1, K1, false, false));
  }
  return OSD_OK;
}

int osd_nn_bn_relu_dropout(void* stream, int device, const float* z, int64_t n, int C, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, double momentum, double eps, int training, int use_bn, double p_drop,
                           const float* mask, uint64_t seed, uint32_t tag, float* y, float* save_mean, float* save_invstd) {
  if (!z || !y || n < 1 || C < 1) { set_error("bad argument"); return OSD_EINVAL; }
  if (use_bn && (!gamma || !beta || !save_mean || !save_invstd)) { set_error("BatchNorm needs gamma, beta and the two save buffers"); return OSD_EINVAL; }
  if (use_bn && !training && (!running_mean || !running_var)) { set_error("eval-mode BatchNorm needs the running statistics"); return OSD_EINVAL; }
  if (use_bn && training && n < 2) { set_error("Expected more than 1 value per channel when training, got input size [%lld, %d]", (long long)n, C); return OSD_EINVAL; }
  if (p_drop < 0.0 || p_drop >= 1.0) { set_error("dropout probability has to be in [0, 1), got %g", p_drop); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int drop = (training && p_drop > 0.0) ? 1 : 0;
  const float scale = (float)(1.0 / (1.0 - p_drop));
  if (use_bn) {
    if (training) {
      double* sums = nullptr;
      OSD_HIP(hipMallocAsync((void**)&sums, (size_t)2 * C * 8, s));
      OSD_HIP(hipMemsetAsync(sums, 0, (size_t)2 * C * 8, s));
      const int rpb = 256;
      dim3 grid((C + 63) / 64, (unsigned)((n + rpb - 1) / rpb));
      hipLaunchKernelGGL((k_bn_colsums<false>), grid, 256, 0, s, z, nullptr, n, C, rpb, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 1.f, nullptr,
                         0ull, 0u, 0, sums, sums + C);
      hipLaunchKernelGGL(k_bn_finish, (C + 255) / 256, 256, 0, s, sums, sums + C, n, C, momentum, eps, running_mean, running_var, save_mean,
                         save_invstd);
      OSD_HIP(hipFreeAsync(sums, s));
    } else {
      hipLaunchKernelGGL(k_bn_eval_stats, (C + 255) / 256, 256, 0, s, running_mean, running_var, C, eps, save_mean, save_invstd);
    }
  }
  hipLaunchKernelGGL(k_bn_act_fwd, ew_blocks(n * C), 256, 0, s, z, n, C, gamma, beta, save_mean, save_invstd, use_bn, drop, (float)p_drop, scale,
                     mask, seed, (uint32_t)TAG_DROPOUT + (tag & 0xffu), y);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

int osd_nn_bn_relu_dropout_bwd(void* stream, int device, const float* gy, const float* z, int64_t n, int C, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd, int training, int use_bn, double p_drop,
                               const float* mask, uint64_t seed, uint32_t tag, float* dz, float* dgamma, float* dbeta) {
  if (!gy || !z || !dz || n < 1 || C < 1) { set_error("bad argument"); return OSD_EINVAL; }
  if (use_bn && (!gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta)) { set_error("BatchNorm backward needs gamma, beta, the saved statistics and both gradient outputs"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  const int drop = (training && p_drop > 0.0) ? 1 : 0;
  const float scale = (float)(1.0 / (1.0 - p_drop));
  const uint32_t t = (uint32_t)TAG_DROPOUT + (tag & 0xffu);
  double* sums = nullptr;
  if (use_bn) {
    OSD_HIP(hipMallocAsync((void**)&sums, (size_t)2 * C * 8, s));
    OSD_HIP(hipMemsetAsync(sums, 0, (size_t)2 * C * 8, s));
    const int rpb = 256;
    dim3 grid((C + 63) / 64, (unsigned)((n + rpb - 1) / rpb));
    hipLaunchKernelGGL((k_bn_colsums<true>), grid, 256, 0, s, z, gy, n, C, rpb, gamma, beta, save_mean, save_invstd, use_bn, (float)p_drop, scale,
                       mask, seed, t, drop, sums, sums + C);
  }
  hipLaunchKernelGGL(k_bn_act_bwd, ew_blocks(n * C), 256, 0, s, gy, z, n, C, gamma, beta, save_mean, save_invstd, use_bn, training, drop,
                     (float)p_drop, scale, mask, seed, t, sums, sums ? sums + C : nullptr, dz, dgamma, dbeta);
  if (sums) OSD_HIP(hipFreeAsync(sums, s));
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

int osd_nn_reparameterize(void* stream, int device, const float* mu, const float* logvar, const float* eps_in, uint64_t seed, int64_t n,
                          int Lz, float* z, float* eps_out) {
  if (!mu || !logvar || !z || n < 1 || Lz < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipLaunchKernelGGL(k_reparam, ew_blocks(n * Lz), 256, 0, (hipStream_t)stream, mu, logvar, eps_in, seed, n, Lz, z, eps_out);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

int osd_nn_reparameterize_bwd(void* stream, int device, const float* gz, const float* mu, const float* z, int64_t count, float* d_logvar) {
  if (!gz || !mu || !z || !d_logvar || count < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipLaunchKernelGGL(k_reparam_bwd, ew_blocks(count), 256, 0, (hipStream_t)stream, gz, mu, z, count, d_logvar);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

int osd_nn_vae_loss(void* stream, int device, const float* x_recon, const float* x, const float* mu, const float* logvar, int64_t n, int D,
                    int Lz, float* parts3, float* d_recon, float* d_mu, float* d_logvar) {
  if (!x_recon || !x || !mu || !logvar || !parts3 || n < 1 || D < 1 || Lz < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  OSD_HIP(hipMemsetAsync(parts3, 0, 12, s));
  int blocks = ew_blocks(n * D);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_vae_loss, blocks, 256, 0, s, x_recon, x, n * D, mu, logvar, n * (int64_t)Lz, 1.0 / (double)n, parts3, d_recon, d_mu, d_logvar);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

int osd_nn_mixup(void* stream, int device, const float* v, const int64_t* perm, double lam, int64_t rows, int cols, float* out) {
  if (!v || !perm || !out || rows < 1 || cols < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  OSD_HIP(launch_mixup((hipStream_t)stream, v, perm, lam, rows, cols, out));
  return OSD_OK;
}

int osd_nn_mixup3(void* stream, int device, const float* data, const float* cond, const float* surv, const int64_t* perm, double lam, int64_t rows,
                  int data_cols, int cond_cols, float* data_out, float* cond_out, float* surv_out) {
  if (!perm || rows < 1 || data_cols < 1 || cond_cols < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  OSD_HIP(launch_mixup3((hipStream_t)stream, data_out ? data : nullptr, cond_out ? cond : nullptr, surv_out ? surv : nullptr, perm, lam, rows,
                        data_cols, cond_cols, data_out, cond_out, surv_out));
  return OSD_OK;
}

int osd_nn_mse(void* stream, int device, const float* a, const float* b, int64_t count, float* loss_out, float* da) {
  if (!a || !b || !loss_out || count < 1) { set_error("bad argument"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  OSD_HIP(hipMemsetAsync(loss_out, 0, 4, s));
  int blocks = ew_blocks(count);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_mse_mean, blocks, 256, 0, s, a, b, count, loss_out, da);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

}  // extern "C"
