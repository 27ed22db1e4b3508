// kernels_train.h -- wrappers of the backward / optimizer kernels (k_train.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch_src.h"

namespace osd {

hipError_t launch_zero_many(hipStream_t s, const ZeroList& zl);

hipError_t launch_silu_fwd(hipStream_t s, const float* u, float* y, int64_t total);
hipError_t launch_cond_mlp_fwd(hipStream_t s, const float* cond, int cd, const float* w0, const float* b0, const float* w2, const float* b2, int64_t n,
                               float* u0, float* ce1, float* ce2);
hipError_t launch_silu_bwd(hipStream_t s, const float* u, const float* g, float* gu, int64_t total);
hipError_t launch_colsum(hipStream_t s, const float* in, int ld, int64_t rows, int cols, float* out);
hipError_t launch_scatter_rows(hipStream_t s, const float* g, const int* t, int64_t rows, int cols, float* table);
// scatter_rows + the dgrads of cond_proj and ConditionalEmbedding's second Linear + its SiLU backward (+ its first Linear's weight gradient) in one launch (k_train.hip)
bool cond_bwd_ok(int H0, const float* g_h0, const float* u0, const float* g_ce2, const float* g_u);
hipError_t launch_cond_bwd(hipStream_t s, const float* g_h0, int H0, const int* t, float* g_temb, const float* w_cp, const float* w_ce2,
                           const float* u0, int64_t rows, float* g_ce2, float* g_u,
                           // optional (cond != null, cd <= 4): ConditionalEmbedding's first Linear's weight and bias gradients ADDED into dw0 [64][cd] / db0 [64];
                           // part: COND_BWD_PART_FLOATS zeroed floats (16 partial copies + a counter; the kernel leaves them zero)
                           const float* cond, int cd, float* part, float* dw0, float* db0);
constexpr int COND_BWD_PART_FLOATS = 16 * 1024 + 64;

struct GnBwdArgs {
  const float* g;        // dL/d(output of the half block) [rows][C]
  const float* z;        // pre-norm activations           [rows][C]
  const float* stats;    // (mean, rstd)                   [rows][8][2]
  const float* gamma; const float* beta;
  float* gz;             // dL/dz                          [rows][C]
  float* dgamma; float* dbeta; float* dbias;   // [C], overwritten
  float* partials;                             // workspace [GN_BWD_MAX_BLOCKS][3][C]
  int atomic_cols;                             // 1: per-block column sums go straight into (pre-zeroed) dgamma / dbeta / dbias with float
                                               // atomics (one launch less per layer); 0: partial rows + k_partial_reduce, deterministic
  int64_t rows; int C;
  int drop_mode; const float* mask; float keep_scale; float p_drop;
  uint64_t seed; uint32_t row_offset; uint32_t step; uint32_t tag;
};
constexpr int GN_BWD_MAX_BLOCKS = 256;
hipError_t launch_gn_silu_bwd(hipStream_t s, int gw, const GnBwdArgs& a);
hipError_t launch_slab_reduce(hipStream_t s, const float* slabs, int ns, int rows, int cols, int64_t stride, float* out, int ldo);
hipError_t launch_small_wgrad(hipStream_t s, const float* x, int kin, const float* gz, int ldg, int nout, int64_t rows, float* dw, float* dbias);

// ---- grouped weight gradients (wgrad_group.h / wgrad_group.hip) -----------------------------------------------
// dw[nout][kin] (leading dimension lddw) = sum over rows of gz[row][nout] * x[row][kin]
struct WgPending { const float* x; int ldx; int kin; const float* gz; int ldg; int nout; int64_t rows; float* dw; int lddw; float* bias[3]; };
bool wgrad_group_ok(const WgPending& w);

// ---- dgrad fused with the GroupNorm backward of the layer it feeds (k_gnbwd.hip, EpiGnBwd in epilogues.h) -------------
struct GnBwdEpi {
  const float* z; int ldz; const float* stats; const float* gamma; const float* beta;
  float* gz; int ldg; float* gy; int ldy; int accumulate;
  int drop_mode; const float* mask; int ldm; float keep_scale; float p_drop;
  uint64_t seed; uint32_t row_offset; uint32_t step; uint32_t tag;
};
struct GemmArgs;
bool dgrad_gnbwd_supported(int gw);
hipError_t launch_dgrad_gnbwd(hipStream_t s, const GemmArgs& g, int gw, const GnBwdEpi& a);
hipError_t launch_dgrad_gnbwd_dual(hipStream_t s, const GemmArgs& g1, int gw, const GnBwdEpi& a, const GemmArgs& g2, float* out2, int ldo2);
struct GnColItem { const float* gy; int ldy; const float* z; int ldz; const float* stats; int C, gw; int64_t rows; float* dgamma; float* dbeta; };
hipError_t launch_gn_colsums(hipStream_t s, const GnColItem* d_items, int n_items, int64_t max_rows);

struct AdamArgs {
  float decay;          // 1 - lr*wd
  float one_minus_b1, b2, one_minus_b2;
  float bc2_sqrt, eps, neg_step_size;
  float max_norm;
};
hipError_t launch_clip_adamw(hipStream_t s, float* p, float* g, float* m, float* v, int64_t n, const AdamArgs& a, double* norm_ws, float* norm_out);

}  // namespace osd
