// split.h -- host entry points of the bf16x3 sampling engine (split.hip, gemm_bf3.h).
#pragma once
#include "handle.h"

namespace osd {

bool split_supported(const Arch& a);
int split_prepare(osd_handle* h, hipStream_t s);       // weight planes follow the current parameters (no-op when they are valid)
int split_denoiser_forward(osd_handle* h, const float* x, const int* t_idx, int32_t t_all, const float* cond, int64_t n, float* eps);
int split_p_sample_step(osd_handle* h, const float* x_t, int32_t t, const float* cond, const float* z, int64_t n, uint64_t seed, int64_t row_offset,
                        float* x_out);
int split_chain_chunk(osd_handle* h, Slot& sl, const float* cond, int64_t n_total, int64_t r0, int64_t m, const float* x_T, const float* noises,
                      uint64_t seed, int64_t row_offset, float* x_out, float* mut_mask_out, int flags);
int split_op_linear(osd_handle* h, const float* x, const float* w, const float* b, int64_t n, int K, int N, float* y);
void split_free(osd_handle* h);

}  // namespace osd
