// rng.h -- counter-based Philox4x32-10 and Box-Muller normals.
// A draw is addressed by (seed; global row, feature/4, step, tag), never by thread or
// block id, so results do not depend on launch geometry, chunking or GPU count.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osd {

enum : uint32_t {
  TAG_POSTERIOR = 0x50535400u,  // z of p_sample at step t; step == T is x_T
  TAG_QNOISE = 0x514e5300u,     // eps of q_sample
  TAG_TSTEP = 0x54535400u,      // randint timesteps
  TAG_DROPOUT = 0x44524f00u,    // + block index in the low byte
  TAG_USER = 0x55535200u,
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per word (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: integer multiplies are quarter-rate
    // VALU work, and VALU cycles come straight out of the fp32 matrix loop's time (tools/probes/valu_mfma.hip)
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

__device__ __forceinline__ uint4 philox_at(uint64_t seed, uint32_t row, uint32_t col4, uint32_t step, uint32_t tag) {
  return philox4x32_10(make_uint4(row, col4, step, tag), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

// u in (0,1): (x + 0.5) * 2^-32
__device__ __forceinline__ float u01_open(uint32_t x) { return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f); }
// u in [0,1): top 24 bits
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// 4 standard normals from one Philox block (two Box-Muller pairs).
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so sin(2*pi*u) is one op.
__device__ __forceinline__ float4 normal4(uint4 r) {
  // radius = sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u): v_log_f32, one multiply, v_sqrt_f32 (1 ulp; sqrtf() would add a
  // ten-instruction correction sequence that buys nothing for a random draw)
  const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01_open(r.x)));
  const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01_open(r.z)));
  const float a0 = u01(r.y), a1 = u01(r.w);
  return make_float4(r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0),
                     r1 * __builtin_amdgcn_cosf(a1), r1 * __builtin_amdgcn_sinf(a1));
}

__device__ __forceinline__ float4 randn4(uint64_t seed, uint32_t row, uint32_t col4, uint32_t step, uint32_t tag) {
  return normal4(philox_at(seed, row, col4, step, tag));
}

}  // namespace osd
