// constraints.hip -- pathway-coherence and mutation-expression constraint losses with their gradients
// (definitions in constraints.h) as wavefront-reduce kernels: one wave owns a row, lanes stride over the member
// columns, sums go through xor-shuffles; batch moments accumulate in double.
#include <math.h>
#include <vector>
#include "handle.h"
#include "constraints.h"

namespace osd {

static int64_t up8(int64_t v) { return (v + 7) / 8 * 8; }

int64_t cons_acc_doubles(const ConsPlan& p, int cols) {
  return up8(4 * (int64_t)cols) + up8(2 * (int64_t)p.n_pathways) + up8(p.nnz) + 2 * CONS_MAX_SET * CONS_MAX_SET;
}
int64_t cons_carve(const ConsPlan& p, int64_t rows, int cols, char* base, ConsWs* w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* q = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return q; };
  w->acc_doubles = cons_acc_doubles(p, cols);
  w->acc = (double*)take(w->acc_doubles * 8);
  w->mi_r = (float2*)take((int64_t)cols * 8);
  w->mi_t = (float2*)take((int64_t)cols * 8);
  w->s = (float*)take(rows * (int64_t)(p.n_pathways > 0 ? p.n_pathways : 1) * 4);
  w->E = (float*)take((CONS_MAX_SET * (CONS_MAX_SET + 1) + 2 * CONS_MAX_SET) * 4);
  w->coef = (float*)take(up8(p.n_pathways > 0 ? p.n_pathways : 1) * 4);
  return off;
}

// ---- batch moments of every column: block = 64 columns x 4 row phases --------------------------------------
__global__ void k_cons_moments(const float* x, int ld, int64_t rows, int cols, int rows_per_block, double* sum, double* sumsq) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  double s = 0.0, q = 0.0;
  if (c < cols)
    for (int64_t r = r0 + ty; r < r1; r += 4) { const double v = x[r * ld + c]; s += v; q += v * v; }
  __shared__ double sh[2][4][64];
  sh[0][ty][tx] = s; sh[1][ty][tx] = q;
  __syncthreads();
  if (ty == 0 && c < cols) {
    s = (sh[0][0][tx] + sh[0][1][tx]) + (sh[0][2][tx] + sh[0][3][tx]);
    q = (sh[1][0][tx] + sh[1][1][tx]) + (sh[1][2][tx] + sh[1][3][tx]);
    atomicAdd(sum + c, s); atomicAdd(sumsq + c, q);
  }
}
// (mean, 1/std) with ddof = 1; a constant column gets 1/std = 0 (it then drops out of every correlation)
__global__ void k_cons_finish(const double* sum, const double* sumsq, int64_t rows, int cols, float2* mi) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const double m = sum[c] / (double)rows;
  const double var = (sumsq[c] - (double)rows * m * m) / (double)(rows - 1);
  const double scale = sumsq[c] / (double)rows;
  const bool alive = var > 1e-12 * scale && var > 0.0;
  mi[c] = make_float2((float)m, alive ? (float)(1.0 / sqrt(var)) : 0.f);
}
hipError_t cons_column_sums(hipStream_t s, const float* x, int ld, int64_t rows, int cols, double* sum2) {
  const int rpb = 256;
  dim3 grid((cols + 63) / 64, (unsigned)((rows + rpb - 1) / rpb));
  hipLaunchKernelGGL(k_cons_moments, grid, 256, 0, s, x, ld, rows, cols, rpb, sum2, sum2 + cols);
  return hipGetLastError();
}
hipError_t cons_moments(hipStream_t s, const float* x, int ld, int64_t rows, int cols, double* sum2, float2* mi) {
  const int rpb = 256;
  dim3 grid((cols + 63) / 64, (unsigned)((rows + rpb - 1) / rpb));
  hipLaunchKernelGGL(k_cons_moments, grid, 256, 0, s, x, ld, rows, cols, rpb, sum2, sum2 + cols);
  hipLaunchKernelGGL(k_cons_finish, (cols + 255) / 256, 256, 0, s, sum2, sum2 + cols, rows, cols, mi);
  return hipGetLastError();
}

// ---- pathway coherence ----------------------------------------------------------------------------------------
// s[r][P] = sum over the members g of P of z[r][g]
__global__ void k_pw_rowsum(const float* x, int ld, int64_t rows, const int* off, const int* mem, int P, const float2* mi, float* s) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nw) {
    const float* xr = x + r * ld;
    for (int p = 0; p < P; ++p) {
      float acc = 0.f;
      for (int e = off[p] + lane; e < off[p + 1]; e += 64) { const int g = mem[e]; const float2 m = mi[g]; acc += (xr[g] - m.x) * m.y; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) s[r * P + p] = acc;
    }
  }
}
// q[e] = sum_r s[r][P_e] * z[r][g_e]
__global__ void k_pw_q(const float* x, int ld, int64_t rows, int rows_per_block, const int* mem, const int* of, int nnz, int P,
                       const float2* mi, const float* s, double* q) {
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += gridDim.x * blockDim.x) {
    const int g = mem[e], p = of[e];
    const float2 m = mi[g];
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc = fmaf(s[r * P + p], (x[r * ld + g] - m.x) * m.y, acc);
    atomicAdd(q + e, (double)acc);
  }
}
// per pathway: c_P, the loss and dL/dS_P
__global__ void k_pw_coef(const double* S, const int* off, const int* mem, int P, const float2* mi, int64_t rows, float w_loss, float w_grad,
                          float* coef, float* loss_out, float* part_out) {
  __shared__ double sh[256];
  int valid = 0;
  for (int p = 0; p < P; ++p) valid += (off[p + 1] - off[p] >= 2) ? 1 : 0;
  double local = 0.0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int G = off[p + 1] - off[p];
    if (G < 2 || valid == 0) { coef[p] = 0.f; continue; }
    int alive = 0;
    for (int e = off[p]; e < off[p + 1]; ++e) alive += mi[mem[e]].y > 0.f ? 1 : 0;
    const double gg = (double)G * (double)(G - 1);
    const double c = (S[p] / (double)(rows - 1) - (double)alive) / gg;
    local += (1.0 - c) / (double)valid;
    coef[p] = (float)(-(double)w_grad / ((double)valid * gg * (double)(rows - 1)));
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    atomicAdd(loss_out, (float)(sh[0] * (double)w_loss));
    if (part_out) atomicAdd(part_out, (float)sh[0]);
  }
}
// dx[r][g] += coef_P * (2/sd_g) * (s[r][P] - z[r][g] * q_e / (N-1));  a wave's atomics to one address issue in pathway order
__global__ void k_pw_grad(const float* x, int ld, int64_t rows, const int* off, const int* mem, int P, const float2* mi, const float* s,
                          const double* q, const float* coef, float inv_nm1, float* dx) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nw) {
    const float* xr = x + r * ld;
    float* dr = dx + r * ld;
    for (int p = 0; p < P; ++p) {
      const float cp = coef[p];
      if (cp == 0.f) continue;
      const float sp = s[r * P + p];
      for (int e = off[p] + lane; e < off[p + 1]; e += 64) {
        const int g = mem[e];
        const float2 m = mi[g];
        const float z = (xr[g] - m.x) * m.y;
        atomicAdd(dr + g, cp * 2.f * m.y * (sp - z * ((float)q[e] * inv_nm1)));
      }
    }
  }
}

hipError_t cons_pathway(hipStream_t st, const ConsPlan& p, const ConsWs& w, const float* x, int ld, int64_t rows, int cols, float w_loss,
                        float w_grad, float* loss_out, float* part_out, float* dx) {
  const int P = p.n_pathways;
  double* Ssum = w.acc + up8(4 * (int64_t)cols);
  double* q = Ssum + up8(2 * (int64_t)P);
  int blocks = (int)((rows + 3) / 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pw_rowsum, blocks, 256, 0, st, x, ld, rows, p.pw_off, p.pw_mem, P, w.mi_r, w.s);
  {  // S_P = sum_r s[r][P]^2 through the moments kernel (its sumsq output)
    const int rpb = 256;
    dim3 grid((P + 63) / 64, (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(k_cons_moments, grid, 256, 0, st, w.s, P, rows, P, rpb, Ssum, Ssum + P);
  }
  hipLaunchKernelGGL(k_pw_coef, 1, 256, 0, st, Ssum + P, p.pw_off, p.pw_mem, P, w.mi_r, rows, w_loss, w_grad, w.coef, loss_out, part_out);
  if (dx) {
    const int rpb = 128;
    dim3 grid((p.nnz + 255) / 256, (unsigned)((rows + rpb - 1) / rpb));
    if (grid.x > 64) grid.x = 64;
    hipLaunchKernelGGL(k_pw_q, grid, 256, 0, st, x, ld, rows, rpb, p.pw_mem, p.pw_of, p.nnz, P, w.mi_r, w.s, q);
    hipLaunchKernelGGL(k_pw_grad, blocks, 256, 0, st, x, ld, rows, p.pw_off, p.pw_mem, P, w.mi_r, w.s, q, w.coef, 1.f / (float)(rows - 1), dx);
  }
  return hipGetLastError();
}

// ---- mutation-expression correlation block -----------------------------------------------------------------
// C[i][j] += sum_r zA[r][i] * zB[r][j]: lane j owns column j of C, zA is broadcast lane by lane
__global__ void k_me_corr(const float* x, int ld, int64_t rows, int rows_per_block, const int* ca, int na, const int* cb, int nb, const float2* mi,
                          double* C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ga = lane < na ? ca[lane] : -1, gb = lane < nb ? cb[lane] : -1;
  const float2 ma = ga >= 0 ? mi[ga] : make_float2(0.f, 0.f), mb = gb >= 0 ? mi[gb] : make_float2(0.f, 0.f);
  float acc[CONS_MAX_SET];
#pragma unroll
  for (int i = 0; i < CONS_MAX_SET; ++i) acc[i] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t r = r0 + wv; r < r1; r += 4) {
    const float za = ga >= 0 ? (x[r * ld + ga] - ma.x) * ma.y : 0.f;
    const float zb = gb >= 0 ? (x[r * ld + gb] - mb.x) * mb.y : 0.f;
#pragma unroll
    for (int i = 0; i < CONS_MAX_SET; ++i)
      if (i < na) acc[i] = fmaf(__shfl(za, i), zb, acc[i]);
  }
  __shared__ float sh[CONS_MAX_SET][CONS_MAX_SET + 1];
  for (int v = 0; v < 4; ++v) {            // the four waves add their partials in a fixed order
    if (wv == v) {
#pragma unroll
      for (int i = 0; i < CONS_MAX_SET; ++i) sh[i][lane] = v == 0 ? acc[i] : sh[i][lane] + acc[i];
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < na * CONS_MAX_SET; idx += blockDim.x) {
    const int i = idx >> 6, j = idx & 63;
    if (j < nb) atomicAdd(C + i * CONS_MAX_SET + j, (double)sh[i][j]);
  }
}
// E = dL/dC_recon, the loss, and the two contraction terms of the standardisation backward
__global__ void k_me_E(const double* Cr, const double* Ct, int na, int nb, int64_t rows, float w_loss, float w_grad, float* E, float* loss_out,
                       float* part_out) {
  __shared__ double sh[256];
  float* rowterm = E + CONS_MAX_SET * (CONS_MAX_SET + 1);
  float* colterm = rowterm + CONS_MAX_SET;
  const double inv = 1.0 / (double)(rows - 1), cnt = (double)na * (double)nb;
  double local = 0.0;
  for (int idx = threadIdx.x; idx < CONS_MAX_SET * CONS_MAX_SET; idx += blockDim.x) {
    const int i = idx >> 6, j = idx & 63;
    double e = 0.0;
    if (i < na && j < nb) {
      const double d = (Cr[idx] - Ct[idx]) * inv;
      local += d * d / cnt;
      e = 2.0 * (double)w_grad * d / cnt;
    }
    E[i * (CONS_MAX_SET + 1) + j] = (float)e;
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    atomicAdd(loss_out, (float)(sh[0] * (double)w_loss));
    if (part_out) atomicAdd(part_out, (float)sh[0]);
  }
  __syncthreads();
  if (threadIdx.x < CONS_MAX_SET) {
    const int i = threadIdx.x;
    double rt = 0.0, ct = 0.0;
    for (int j = 0; j < CONS_MAX_SET; ++j) {
      rt += (double)E[i * (CONS_MAX_SET + 1) + j] * Cr[i * CONS_MAX_SET + j] * inv;
      ct += (double)E[j * (CONS_MAX_SET + 1) + i] * Cr[j * CONS_MAX_SET + i] * inv;
    }
    rowterm[i] = (float)rt; colterm[i] = (float)ct;
  }
}
// dxA[r][i] += (1/sd_i)/(N-1) * (sum_j E[i][j] zB[r][j] - zA[r][i] * rowterm[i]), and the mirror for B
__global__ void k_me_grad(const float* x, int ld, int64_t rows, const int* ca, int na, const int* cb, int nb, const float2* mi, const float* Eg,
                          float inv_nm1, float* dx) {
  __shared__ float E[CONS_MAX_SET * (CONS_MAX_SET + 1) + 2 * CONS_MAX_SET];
  for (int i = threadIdx.x; i < CONS_MAX_SET * (CONS_MAX_SET + 1) + 2 * CONS_MAX_SET; i += blockDim.x) E[i] = Eg[i];
  __syncthreads();
  const float* rowterm = E + CONS_MAX_SET * (CONS_MAX_SET + 1);
  const float* colterm = rowterm + CONS_MAX_SET;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int ga = lane < na ? ca[lane] : -1, gb = lane < nb ? cb[lane] : -1;
  const float2 ma = ga >= 0 ? mi[ga] : make_float2(0.f, 0.f), mb = gb >= 0 ? mi[gb] : make_float2(0.f, 0.f);
  for (int64_t r = wave; r < rows; r += nw) {
    const float za = ga >= 0 ? (x[r * ld + ga] - ma.x) * ma.y : 0.f;
    const float zb = gb >= 0 ? (x[r * ld + gb] - mb.x) * mb.y : 0.f;
    float aa = 0.f, ab = 0.f;
    for (int j = 0; j < nb; ++j) aa = fmaf(E[lane * (CONS_MAX_SET + 1) + j], __shfl(zb, j), aa);
    for (int i = 0; i < na; ++i) ab = fmaf(E[i * (CONS_MAX_SET + 1) + lane], __shfl(za, i), ab);
    if (ga >= 0) atomicAdd(dx + r * ld + ga, ma.y * inv_nm1 * (aa - za * rowterm[lane]));
    if (gb >= 0) atomicAdd(dx + r * ld + gb, mb.y * inv_nm1 * (ab - zb * colterm[lane]));
  }
}

hipError_t cons_gram(hipStream_t s, const float* x, int ld, int64_t rows, const int* ca, int na, const int* cb, int nb, const float2* mi, double* C) {
  const int rpb = 128;
  hipLaunchKernelGGL(k_me_corr, (int)((rows + rpb - 1) / rpb), 256, 0, s, x, ld, rows, rpb, ca, na, cb, nb, mi, C);
  return hipGetLastError();
}

hipError_t cons_mutexpr(hipStream_t st, const ConsPlan& p, const ConsWs& w, const float* x_recon, const float* x_true, int ld, int64_t rows,
                        int cols, float w_loss, float w_grad, float* loss_out, float* part_out, float* dx) {
  double* Cr = w.acc + up8(4 * (int64_t)cols) + up8(2 * (int64_t)p.n_pathways) + up8(p.nnz);
  double* Ct = Cr + CONS_MAX_SET * CONS_MAX_SET;
  const int rpb = 128;
  const int blocks = (int)((rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(k_me_corr, blocks, 256, 0, st, x_recon, ld, rows, rpb, p.cols_a, p.n_a, p.cols_b, p.n_b, w.mi_r, Cr);
  hipLaunchKernelGGL(k_me_corr, blocks, 256, 0, st, x_true, ld, rows, rpb, p.cols_a, p.n_a, p.cols_b, p.n_b, w.mi_t, Ct);
  hipLaunchKernelGGL(k_me_E, 1, 256, 0, st, Cr, Ct, p.n_a, p.n_b, rows, w_loss, w_grad, w.E, loss_out, part_out);
  if (dx) {
    int gb = (int)((rows + 3) / 4);
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(k_me_grad, gb, 256, 0, st, x_recon, ld, rows, p.cols_a, p.n_a, p.cols_b, p.n_b, w.mi_r, w.E, 1.f / (float)(rows - 1), dx);
  }
  return hipGetLastError();
}

// ---- x0_hat from the eps prediction and its backward ---------------------------------------------------------
__global__ void k_x0hat(const float* x_t, const int* t, const float* sac, const float* s1m, int64_t rows, int D, float* eps) {
  const int64_t total = rows * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int ti = t[r];
    eps[i] = (x_t[i] - s1m[ti] * eps[i]) / sac[ti];
  }
}
__global__ void k_x0hat_bwd(const float* g, const int* t, const float* sac, const float* s1m, int64_t rows, int D, float* d_eps) {
  const int64_t total = rows * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int ti = t[r];
    d_eps[i] += g[i] * (-(s1m[ti] / sac[ti]));
  }
}
static int ew_blocks(int64_t total) { int64_t b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : b); }
hipError_t launch_x0hat(hipStream_t s, const float* x_t, const int* t_idx, const float* sqrt_ac, const float* sqrt_1m, int64_t rows, int D,
                        float* eps_inout) {
  hipLaunchKernelGGL(k_x0hat, ew_blocks(rows * D), 256, 0, s, x_t, t_idx, sqrt_ac, sqrt_1m, rows, D, eps_inout);
  return hipGetLastError();
}
hipError_t launch_x0hat_bwd(hipStream_t s, const float* g_x0, const int* t_idx, const float* sqrt_ac, const float* sqrt_1m, int64_t rows, int D,
                            float* d_eps) {
  hipLaunchKernelGGL(k_x0hat_bwd, ew_blocks(rows * D), 256, 0, s, g_x0, t_idx, sqrt_ac, sqrt_1m, rows, D, d_eps);
  return hipGetLastError();
}

// ---- host side: plans ------------------------------------------------------------------------------------------
void cons_free_plan(ConsPlan* p) {
  int* bufs[] = {p->pw_off, p->pw_mem, p->pw_of, p->cols_a, p->cols_b};
  for (int* b : bufs) if (b) (void)hipFree(b);
  *p = ConsPlan{};
}

// Validates and uploads; on failure the plan is left empty.
int cons_build_plan(const int32_t* off, const int32_t* mem, int n_pathways, const int32_t* ca, int na, const int32_t* cb, int nb, int cols,
                    ConsPlan* out) {
  *out = ConsPlan{};
  if (n_pathways < 0 || na < 0 || nb < 0) { set_error("negative count"); return OSD_EINVAL; }
  if ((na > 0) != (nb > 0)) { set_error("the mutation-expression block needs both column sets"); return OSD_EINVAL; }
  if (na > CONS_MAX_SET || nb > CONS_MAX_SET) { set_error("at most %d columns per side of the mutation-expression block", CONS_MAX_SET); return OSD_EUNSUPPORTED; }
  if (n_pathways > 0 && (!off || !mem)) { set_error("null pathway arrays"); return OSD_EINVAL; }
  if (na > 0 && (!ca || !cb)) { set_error("null column sets"); return OSD_EINVAL; }
  int nnz = 0, maxc = -1;
  if (n_pathways > 0) {
    if (off[0] != 0) { set_error("pathway offsets must start at 0"); return OSD_EINVAL; }
    for (int p = 0; p < n_pathways; ++p)
      if (off[p + 1] < off[p]) { set_error("pathway offsets must be non-decreasing"); return OSD_EINVAL; }
    nnz = off[n_pathways];
    for (int e = 0; e < nnz; ++e) {
      if (mem[e] < 0 || mem[e] >= cols) { set_error("pathway member column %d out of range [0, %d)", mem[e], cols); return OSD_EINVAL; }
      maxc = mem[e] > maxc ? mem[e] : maxc;
    }
  }
  for (int i = 0; i < na; ++i) { if (ca[i] < 0 || ca[i] >= cols) { set_error("column %d out of range", ca[i]); return OSD_EINVAL; } maxc = ca[i] > maxc ? ca[i] : maxc; }
  for (int i = 0; i < nb; ++i) { if (cb[i] < 0 || cb[i] >= cols) { set_error("column %d out of range", cb[i]); return OSD_EINVAL; } maxc = cb[i] > maxc ? cb[i] : maxc; }
  auto up = [&](const int32_t* src, int n, int** dst) -> int {
    if (n <= 0) return OSD_OK;
    OSD_HIP(hipMalloc((void**)dst, (size_t)n * 4));
    OSD_HIP(hipMemcpy(*dst, src, (size_t)n * 4, hipMemcpyHostToDevice));
    return OSD_OK;
  };
  int rc = OSD_OK;
  if (n_pathways > 0) {
    std::vector<int32_t> of((size_t)(nnz > 0 ? nnz : 1));
    for (int p = 0; p < n_pathways; ++p) for (int e = off[p]; e < off[p + 1]; ++e) of[e] = p;
    if ((rc = up(off, n_pathways + 1, &out->pw_off)) != OSD_OK || (rc = up(mem, nnz, &out->pw_mem)) != OSD_OK ||
        (rc = up(of.data(), nnz, &out->pw_of)) != OSD_OK) { cons_free_plan(out); return rc; }
  }
  if ((rc = up(ca, na, &out->cols_a)) != OSD_OK || (rc = up(cb, nb, &out->cols_b)) != OSD_OK) { cons_free_plan(out); return rc; }
  out->n_pathways = n_pathways; out->nnz = nnz; out->n_a = na; out->n_b = nb; out->max_col = maxc;
  return OSD_OK;
}

}  // namespace osd

using namespace osd;

namespace {
struct Scratch {
  char* p = nullptr;
  ~Scratch() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" {

int osd_loss_pathway_coherence(void* stream, int device, const float* x, int64_t rows, int ld, int cols, const int32_t* offsets_host,
                               const int32_t* members_host, int n_pathways, double weight, float* loss_out, float* dx) {
  if (!x || !loss_out || rows < 2 || cols < 1 || ld < cols || n_pathways < 1) { set_error("bad argument (rows >= 2, 1 <= cols <= ld, n_pathways >= 1)"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  ConsPlan plan;
  OSD_TRY(cons_build_plan(offsets_host, members_host, n_pathways, nullptr, 0, nullptr, 0, cols, &plan));
  Scratch sc;
  ConsWs w;
  const int64_t bytes = cons_carve(plan, rows, cols, nullptr, &w);
  if (hipMalloc((void**)&sc.p, (size_t)bytes) != hipSuccess) { cons_free_plan(&plan); set_error("hipMalloc of %lld bytes failed", (long long)bytes); return OSD_ENOMEM; }
  cons_carve(plan, rows, cols, sc.p, &w);
  hipError_t e = hipMemsetAsync(w.acc, 0, (size_t)w.acc_doubles * 8, s);
  if (e == hipSuccess) e = cons_moments(s, x, ld, rows, cols, w.acc, w.mi_r);
  if (e == hipSuccess) e = cons_pathway(s, plan, w, x, ld, rows, cols, (float)weight, (float)weight, loss_out, nullptr, dx);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cons_free_plan(&plan);
  OSD_HIP(e);
  return OSD_OK;
}

int osd_loss_mutation_expression(void* stream, int device, const float* x_recon, const float* x_true, int64_t rows, int ld, int cols,
                                 const int32_t* cols_a_host, int n_a, const int32_t* cols_b_host, int n_b, double weight, float* loss_out,
                                 float* dx) {
  if (!x_recon || !x_true || !loss_out || rows < 2 || cols < 1 || ld < cols || n_a < 1 || n_b < 1) { set_error("bad argument (rows >= 2, 1 <= cols <= ld, non-empty column sets)"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  hipStream_t s = (hipStream_t)stream;
  ConsPlan plan;
  OSD_TRY(cons_build_plan(nullptr, nullptr, 0, cols_a_host, n_a, cols_b_host, n_b, cols, &plan));
  Scratch sc;
  ConsWs w;
  const int64_t bytes = cons_carve(plan, rows, cols, nullptr, &w);
  if (hipMalloc((void**)&sc.p, (size_t)bytes) != hipSuccess) { cons_free_plan(&plan); set_error("hipMalloc of %lld bytes failed", (long long)bytes); return OSD_ENOMEM; }
  cons_carve(plan, rows, cols, sc.p, &w);
  hipError_t e = hipMemsetAsync(w.acc, 0, (size_t)w.acc_doubles * 8, s);
  if (e == hipSuccess) e = cons_moments(s, x_recon, ld, rows, cols, w.acc, w.mi_r);
  if (e == hipSuccess) e = cons_moments(s, x_true, ld, rows, cols, w.acc + 2 * (int64_t)cols, w.mi_t);
  if (e == hipSuccess) e = cons_mutexpr(s, plan, w, x_recon, x_true, ld, rows, cols, (float)weight, (float)weight, loss_out, nullptr, dx);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cons_free_plan(&plan);
  OSD_HIP(e);
  return OSD_OK;
}

}  // extern "C"
