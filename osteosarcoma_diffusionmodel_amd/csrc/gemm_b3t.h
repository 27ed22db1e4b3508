// gemm_b3t.h -- the training step's long GEMMs on the bf16 matrix pipe at fp32 accuracy (osd_set_option("precision", 1)).
//
// gemm_bf3.h multiplies operands that already ARE bf16 planes (the eval path keeps its activations that way and packs its
// weights once).  A training step has neither: the weights change every step and the activations come from, and go to, the fp32
// kernels around.  So this kernel takes the fp32 operands of gemm_kernel<T, true, true, Epi> (both K-contiguous: C[p][f] =
// sum_k A[f][k] B[p][k]) and splits them where they are staged, as the grouped weight-gradient item does (wgrad_group.h): a thread
// loads float4 chunks (4 k of one row), makes the three bf16 planes of each (split4: ~5.5 VALU per element) and writes them as
// 8-byte halves of the 16-byte units v_mfma_f32_32x32x16_bf16 reads (unit = 8 consecutive k of one row; lane (l31, h) of a
// fragment read takes the unit of row l31, k-half h: ds_read_b128, lane-contiguous).  Six MFMAs per product (small terms first),
// fp32 accumulation in gemm.h's fragment layout -- so every epilogue of epilogues.h runs on the accumulators unchanged.
// Two LDS buffers, register prefetch two stages ahead, one barrier per stage (gemm_tile's pipeline).
//
// Used for output_proj + MSE of the training step (N = D = 2000: 512 tiles of 128 x 128, 64 x 64 accumulators per wave -- 24 MFMAs
// per 16-k stage against the split of 16 values per thread): 49.2 -> 43.0 us at batch 4096, the launch being bound by its epilogue's
// 64 MB of noise / gradient traffic rather than by its 16-stage K loop.  NOT used for the two K = D GEMMs with 256 outputs
// (input_proj forward, the first dgrad), where it was measured and lost (65.7 vs 45.6 us, 65.5 + 6.1 for the weight's transpose vs
// 49.2 us): 256 x 4096 outputs are 256 workgroups of 64 x 64, i.e. ONE 32 x 32 accumulator per wave -- 12 dependent MFMAs per
// 32-k stage against the same split work per thread, on one wave per SIMD: the VALU of the split (4 cycles per wave instruction)
// is 2 x the matrix time and nothing overlaps it.  Larger tiles there need split-K over workgroups plus a reduce pass that
// applies the epilogue (GroupNorm backward for the dgrad): DESIGN.md section 4.3.
#pragma once
#include "gemm.h"
#include "gemm_bf3.h"
#include "launch.h"

namespace osd {

// LDS image of one stage of one operand with R rows and KB 16-k blocks: unit (kb, plane, h, row) at ((kb * 3 + plane) * 2 + h) * R + row
template <class T, int BK3>
struct B3tTile {
  static constexpr int KB = BK3 / 16;                      // MFMA k blocks per stage
  static constexpr int CPR = BK3 / 4;                      // float4 chunks per row and stage
  static constexpr int NA = T::BF * CPR / NTHREADS, NB = T::BP * CPR / NTHREADS;      // chunks per thread
  static_assert(BK3 == 16 || BK3 == 32, "16 or 32 k per stage");
  static_assert(NA >= 1 && NB >= 1 && T::BF * CPR % NTHREADS == 0 && T::BP * CPR % NTHREADS == 0, "whole chunks per thread");
  static constexpr int A_U4 = KB * 6 * T::BF, B_U4 = KB * 6 * T::BP;                  // 16-byte units
  static constexpr int BUF_U4 = A_U4 + B_U4;
  static constexpr int LDS_BYTES = 2 * BUF_U4 * 16;
};

template <class T, class Epi, int BK3>
__device__ __forceinline__ void gemm_b3t_tile(const GemmArgs& g, const typename Epi::Args& ea, int f0, int p0, uint4* smem) {
  typedef B3tTile<T, BK3> L;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wf = (wave / T::NWP) * T::WF;
  const int wp = (wave % T::NWP) * T::WP;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[T::NFB][T::NPB];
#pragma unroll
  for (int i = 0; i < T::NFB; ++i)
#pragma unroll
    for (int j = 0; j < T::NPB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // chunk c = tid + 256 i of an operand: row c / CPR, k chunk c % CPR
  float4 va[2][L::NA], vb[2][L::NB];          // two register sets: stages are loaded two ahead
  auto gload = [&](int set, int k0) {
#pragma unroll
    for (int i = 0; i < L::NA; ++i) {
      const int c = tid + NTHREADS * i, row = f0 + c / L::CPR, k = k0 + 4 * (c % L::CPR);
      const int rc = row < g.F ? row : g.F - 1;
      va[set][i] = ldraw(g.A + (size_t)rc * g.lda, k, g.K);
    }
#pragma unroll
    for (int i = 0; i < L::NB; ++i) {
      const int c = tid + NTHREADS * i, row = p0 + c / L::CPR, k = k0 + 4 * (c % L::CPR);
      const int rc = row < g.P ? row : g.P - 1;
      vb[set][i] = ldraw(g.B0 + (size_t)rc * g.ldb0, k, g.K);
    }
  };
  // split + write: chunk (row, kc) -> the half (kc & 1) of unit (kb = kc / 4, h = (kc / 2) & 1, row) of each plane.  The K tail
  // of the A operand is written as zeros (a clamped chunk re-read valid data; 0 * finite = 0 makes the B side harmless).
  auto lstore = [&](int set, uint4* buf, int k0) {
    const int kvalid = g.K - k0;
    uint2* const a2 = reinterpret_cast<uint2*>(buf);
    uint2* const b2 = reinterpret_cast<uint2*>(buf + L::A_U4);
#pragma unroll
    for (int i = 0; i < L::NA; ++i) {
      const int c = tid + NTHREADS * i, row = c / L::CPR, kc = c % L::CPR;
      const bool ok = 4 * kc < kvalid;
      const float4 t = va[set][i];
      const Split4 s = split4(make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a2[(((((kc >> 2) * 3 + pl) * 2 + ((kc >> 1) & 1)) * T::BF + row) << 1) + (kc & 1)] = s.p[pl];
    }
#pragma unroll
    for (int i = 0; i < L::NB; ++i) {
      const int c = tid + NTHREADS * i, row = c / L::CPR, kc = c % L::CPR;
      const Split4 s = split4(vb[set][i]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b2[(((((kc >> 2) * 3 + pl) * 2 + ((kc >> 1) & 1)) * T::BP + row) << 1) + (kc & 1)] = s.p[pl];
    }
  };
  auto compute = [&](const uint4* buf) {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int kb = 0; kb < L::KB; ++kb) {
      uint4 fa[T::NFB][3], fb[T::NPB][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < T::NFB; ++i) fa[i][pl] = buf[((kb * 3 + pl) * 2 + h) * T::BF + wf + 32 * i + l31];
#pragma unroll
        for (int j = 0; j < T::NPB; ++j) fb[j][pl] = buf[L::A_U4 + ((kb * 3 + pl) * 2 + h) * T::BP + wp + 32 * j + l31];
      }
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < T::NFB; ++i)
#pragma unroll
          for (int j = 0; j < T::NPB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][PA[t]]), __builtin_bit_cast(bf16x8, fb[j][PB[t]]),
                                                                acc[i][j], 0, 0, 0);
    }
  };

  uint4* const buf0 = smem;
  uint4* const buf1 = smem + L::BUF_U4;
  const int nk = (g.K + BK3 - 1) / BK3;
  // stage st is loaded during stage st - 2 (register set st & 1) and written to buffer st & 1 during stage st - 1
  gload(0, 0);
  if (nk > 1) gload(1, BK3);
  lstore(0, buf0, 0);
  if (nk > 2) gload(0, 2 * BK3);
  __syncthreads();
  for (int st = 0; st < nk; ++st) {
    const uint4* cur = (st & 1) ? buf1 : buf0;
    uint4* nxt = (st & 1) ? buf0 : buf1;
    // the MFMAs of stage st with the split + write of stage st + 1 scheduled into their shadow (a bf16 MFMA holds the vector issue
    // for 8 of its 32 cycles), then the loads of stage st + 3: register set (st + 1) & 1 holds stage st + 1 and is free after the write
    __builtin_amdgcn_sched_barrier(0);
    compute(cur);
    if (st + 1 < nk) {
      if ((st + 1) & 1) lstore(1, nxt, (st + 1) * BK3); else lstore(0, nxt, (st + 1) * BK3);
    }
#pragma unroll
    for (int k = 0; k < L::KB * 6 * T::NFB * T::NPB; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // ... then a few VALU of the split
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st + 3 < nk) {
      if ((st + 1) & 1) gload(1, (st + 3) * BK3); else gload(0, (st + 3) * BK3);
    }
    __syncthreads();
  }

  const auto pre = Epi::template prefetch<T::NFB, true>(ea, f0 + wf, lane, g.F);
  if constexpr (Epi::XBUF) {
    static_assert(4 * T::NPB * 1024 * 4 <= L::LDS_BYTES, "transposer regions must fit the staging buffers");
    Epi::template apply<T::NFB, T::NPB, true>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P, reinterpret_cast<float*>(smem) + wave * (T::NPB * 1024));
  } else {
    Epi::template apply<T::NFB, T::NPB, true>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P);
  }
}

template <class T, class Epi, int BK3>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_b3t_kernel(GemmArgs g, typename Epi::Args ea) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem_u4[];
  // XCD-aware tile order of gemm_kernel: blocks b, b + 8, ... share an XCD and the feature tiles of one patient tile
  const int nft = (g.F + T::BF - 1) / T::BF;
  const int npt = (g.P + T::BP - 1) / T::BP;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int ft = idx % nft;
  const int pt = (idx / nft) * 8 + (b & 7);
  if (pt >= npt) return;
  gemm_b3t_tile<T, Epi, BK3>(g, ea, ft * T::BF, pt * T::BP, smem_u4);
}

template <class T, class Epi, int BK3>
struct B3tRegistrar {
  B3tRegistrar() { kernel_registry().push_back({reinterpret_cast<const void*>(gemm_b3t_kernel<T, Epi, BK3>), B3tTile<T, BK3>::LDS_BYTES}); }
  static B3tRegistrar instance;
};
template <class T, class Epi, int BK3>
B3tRegistrar<T, Epi, BK3> B3tRegistrar<T, Epi, BK3>::instance;

// hipErrorInvalidValue when the operands do not meet the preconditions of the branch-free loads (the caller runs the fp32 kernel)
template <class T, class Epi, int BK3>
hipError_t launch_gemm_b3t(hipStream_t s, const GemmArgs& g, const typename Epi::Args& ea) {
  if (g.F <= 0 || g.P <= 0) return hipSuccess;
  if (g.K0 < g.K || g.kchunk > 0 || g.a_kmax > 0 || !gemm_fast_ok(g, true, true) || !Epi::fast_ok(ea, g.F)) return hipErrorInvalidValue;
  (void)&B3tRegistrar<T, Epi, BK3>::instance;
  constexpr int lds = B3tTile<T, BK3>::LDS_BYTES;
  hipLaunchKernelGGL((gemm_b3t_kernel<T, Epi, BK3>), dim3(gemm_grid(g.F, g.P, T::BF, T::BP)), dim3(NTHREADS), lds, s, g, ea);
  return hipGetLastError();
}

}  // namespace osd
