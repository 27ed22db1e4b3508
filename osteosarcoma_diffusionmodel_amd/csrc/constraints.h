// constraints.h -- biological constraint losses (north_star): the reference declares them at
// models/cvae.py:262-302 as stubs that return 0.0, so the definitions below are this library's; they are off
// unless configured and then add to the eps-MSE loss of osd_train_loss_fwd_bwd.
//
//   pathway coherence     L_pc = mean_P (1 - c_P),  c_P = mean off-diagonal Pearson correlation (over the batch rows)
//                         of the member columns of pathway P  (the quantity utils/validation.py:144-173 reports)
//   mutation-expression   L_me = mean_{i in A, j in B} (corr_recon(i, j) - corr_true(i, j))^2
//                         ("MSE on correlation matrices", models/cvae.py:296-297)
//
// All kernels are stream-ordered (no host synchronisation); batch statistics accumulate in double.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osd {

constexpr int CONS_MAX_SET = 64;   // columns per side of the mutation-expression block

// Device-resident description of the constraint sets.
struct ConsPlan {
  int* pw_off = nullptr;      // [n_pathways + 1]
  int* pw_mem = nullptr;      // [nnz] member column of entry e
  int* pw_of = nullptr;       // [nnz] pathway of entry e
  int n_pathways = 0, nnz = 0;
  int* cols_a = nullptr; int n_a = 0;
  int* cols_b = nullptr; int n_b = 0;
  int max_col = -1;           // largest referenced column
};

// Scratch for one evaluation on `rows` rows of width `cols`.
struct ConsWs {
  double* acc = nullptr;      // zeroed per call: [4*cols] column sums (recon sum, sumsq, true sum, sumsq) + pathway/ME sums
  int64_t acc_doubles = 0;
  float2* mi_r = nullptr;     // [cols] (mean, 1/std) of the reconstruction
  float2* mi_t = nullptr;     // [cols] of the true data
  float* s = nullptr;         // [rows][n_pathways]
  float* E = nullptr;         // [64][65] + rowterm[64] + colterm[64]
  float* coef = nullptr;      // [n_pathways]
};

int64_t cons_acc_doubles(const ConsPlan& p, int cols);
// carves the scratch out of `base` (null: sizes only) and returns the bytes needed
int64_t cons_carve(const ConsPlan& p, int64_t rows, int cols, char* base, ConsWs* w);
// validates the host description and uploads it (on failure the plan is left empty)
int cons_build_plan(const int32_t* off, const int32_t* mem, int n_pathways, const int32_t* ca, int na, const int32_t* cb, int nb, int cols,
                    ConsPlan* out);
void cons_free_plan(ConsPlan* p);

// loss_out (dev float[1]) += w_loss * L ; part_out (dev float[1], may be null) += L ; dx (dev [rows][ld], may be null) += w_grad * dL/dx
hipError_t cons_moments(hipStream_t s, const float* x, int ld, int64_t rows, int cols, double* sum2, float2* mi);
hipError_t cons_pathway(hipStream_t s, const ConsPlan& p, const ConsWs& w, const float* x, int ld, int64_t rows, int cols, float w_loss,
                        float w_grad, float* loss_out, float* part_out, float* dx);
hipError_t cons_mutexpr(hipStream_t s, const ConsPlan& p, const ConsWs& w, const float* x_recon, const float* x_true, int ld, int64_t rows,
                        int cols, float w_loss, float w_grad, float* loss_out, float* part_out, float* dx);

// building blocks shared with the validation metrics (validate.hip)
// sum2[0..cols) += column sums, sum2[cols..2cols) += column sums of squares (double, pre-zeroed by the caller)
hipError_t cons_column_sums(hipStream_t s, const float* x, int ld, int64_t rows, int cols, double* sum2);
// C[i][j] (64 x 64 doubles, pre-zeroed) += sum_r zA[r][i] zB[r][j] with z = (x - mi.x) * mi.y of the selected columns (device index arrays)
hipError_t cons_gram(hipStream_t s, const float* x, int ld, int64_t rows, const int* ca, int na, const int* cb, int nb, const float2* mi, double* C);

// x0_hat = (x_t - sqrt_1m[t] * eps_hat) / sqrt_ac[t]  in place over eps_hat (models/diffusion.py:405)
hipError_t launch_x0hat(hipStream_t s, const float* x_t, const int* t_idx, const float* sqrt_ac, const float* sqrt_1m, int64_t rows, int D,
                        float* eps_inout);
// d_eps += g_x0hat * (-sqrt_1m[t] / sqrt_ac[t])
hipError_t launch_x0hat_bwd(hipStream_t s, const float* g_x0, const int* t_idx, const float* sqrt_ac, const float* sqrt_1m, int64_t rows, int D,
                            float* d_eps);

}  // namespace osd
