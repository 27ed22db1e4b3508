// kernels.h -- host-callable wrappers around the device kernels (one .hip file each group).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm.h"
#include "epilogues.h"
#include "batch_src.h"

namespace osd {

// k_linear.hip ---------------------------------------------------------------------
// out[p][f] (+)= act( sum_k A(f,k) B(p,k) + bias[f] ); layouts: a_kc/b_kc as in gemm.h
hipError_t launch_linear(hipStream_t s, const GemmArgs& g, bool a_kc, bool b_kc, const float* bias,
                         float* out, int ldo, bool silu, bool accumulate);

hipError_t launch_wgrad_splitk(hipStream_t s, const GemmArgs& g, float* slabs, int ldo, long long slice_stride);

// k_fused.hip ----------------------------------------------------------------------
hipError_t launch_input(hipStream_t s, const GemmArgs& g, const EpiInput::Args& a, bool a_zero_padded);
hipError_t launch_input_splitk(hipStream_t s, const GemmArgs& g, const EpiInput::Args& a, float* slabs, int slices);
hipError_t launch_posterior(hipStream_t s, const GemmArgs& g, const EpiPosterior::Args& a);
hipError_t launch_mse(hipStream_t s, const GemmArgs& g, const EpiMse::Args& a);
// k_b3t.hip: precision = 1 (hipErrorInvalidValue: outside the kernel's preconditions -- run the fp32 launch)
hipError_t launch_mse_b3t(hipStream_t s, const GemmArgs& g, const EpiMse::Args& a);

// k_gn.hip / k_gn_drop.hip ------------------------------------------------------------
// Arguments common to every GW; the wrappers copy them into EpiGnSilu<GW,DROP>::Args.
struct GnArgs {
  const float* bias; const float* gamma; const float* beta;
  float* out; int ldo;
  float* z_out; int ldz; float* stats;
  int drop_mode; const float* mask; int ldm; float keep_scale; float p_drop;
  uint64_t seed; uint32_t row_offset; uint32_t step; uint32_t tag; const int* step_dev;
};
bool gn_width_supported(int gw);
hipError_t launch_gn_silu(hipStream_t s, const GemmArgs& g, int gw, const GnArgs& a);        // drop_mode == 0
hipError_t launch_gn_silu_drop(hipStream_t s, const GemmArgs& g, int gw, const GnArgs& a);   // drop_mode 1/2
// k_fused.hip: small batches -- K slices over workgroups + a reduce / GroupNorm / SiLU kernel (another fp32 summation order: opt-in)
hipError_t launch_gn_silu_splitk(hipStream_t s, const GemmArgs& g, const GnArgs& a, float* slabs, int slices);

// k_elem.hip -----------------------------------------------------------------------
hipError_t launch_set_int(hipStream_t s, int* p, int v);
hipError_t launch_add_int(hipStream_t s, int* p, int d);
hipError_t launch_fill_randn(hipStream_t s, float* out, int ld, int64_t rows, int cols, uint64_t seed,
                             uint32_t row_offset, uint32_t step, uint32_t tag);
hipError_t launch_copy2d(hipStream_t s, const float* src, int lds, float* dst, int ldd, int64_t rows, int cols);
hipError_t launch_q_sample(hipStream_t s, const float* x0, const int* t, const float* sqrt_ac, const float* sqrt_1m,
                           const float* noise_in, int64_t rows, int cols, uint64_t seed, uint32_t row_offset,
                           float* x_t, float* noise_out, int* t_out = nullptr, int T = 0, int ldxt = 0, const ZeroList* zl = nullptr);
hipError_t launch_q_sample_src(hipStream_t s, const BatchSrc& b, const int* t, const float* sqrt_ac, const float* sqrt_1m, const float* noise_in,
                               int64_t rows, int cols, int cd, uint64_t seed, uint32_t row_offset, float* x_t, float* noise_out, int* t_out, int T,
                               float* cond_out, float* x0_out, int ldxt = 0, const ZeroList* zl = nullptr);
hipError_t launch_clamp_int(hipStream_t s, const int* in, int64_t n, int lo, int hi, int* out);
hipError_t launch_randint(hipStream_t s, int* out, int64_t n, int hi, uint64_t seed, uint32_t row_offset);
hipError_t launch_mixup(hipStream_t s, const float* v, const int64_t* perm, double lam, int64_t rows, int cols, float* out);
hipError_t launch_mixup3(hipStream_t s, const float* d, const float* c, const float* sv, const int64_t* perm, double lam, int64_t rows, int D,
                         int cd, float* od, float* oc, float* os);
hipError_t launch_threshold(hipStream_t s, const float* x, int ldx, int64_t rows, int cols, float thr, float* out);

}  // namespace osd
