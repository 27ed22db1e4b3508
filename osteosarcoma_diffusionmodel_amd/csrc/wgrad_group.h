// wgrad_group.h -- every weight gradient of a backward pass in ONE launch.
//
//   dW[n_out][k_in] = sum_m gz[m][n_out] * x[m][k_in]          (loss.backward() of a Linear, utils/train.py:239)
//
// The reduction runs over the batch (4096 rows at the BASELINE training shape), the outputs are small (<= 2 MB), and a
// step has ~17 of them: launched one by one, each is a handful of 128 x 128 tiles whose few K steps are dwarfed by the
// launch, the first staging round trip and the store tail.  Here the step's weight gradients form one list of work items
// (tensor, 128 x 128 output tile, row range): a workgroup per item, ~2 per CU in total, each running 30-60 K steps, the row
// ranges of a tile summed afterwards from slabs in a fixed order (deterministic, no float atomics).
//
// Both operands are "row contiguous, reduction index outermost" ([m][f]): a 32-row K stage of 128 columns is sixteen 1 KiB
// LDS-DMA pieces (two rows each) into a linear [32][128] image that ds_read_b32 reads conflict-free (lanes of a half wave
// walk 32 consecutive dwords; the halves take rows m and m + 1 in separate cycles) -- no register staging, no swizzle.
#pragma once
#include "gemm_glds.h"

namespace osd {

struct WgItem {
  const float* A; int lda;      // x  [rows][lda]: feature (k_in) tile read at columns f0..
  const float* B; int ldb;      // gz [rows][ldb]: feature (n_out) tile read at columns p0..
  int F, P;                     // k_in, n_out extents (F % 4 == 0)
  int f0, p0;                   // tile origin
  int k0, k1;                   // row range [k0, k1), both multiples of 32
  float* out; int ldo;          // out[p][f] at out + p * ldo + f: the gradient itself or this slice's slab
  float* bias[3];               // optional (items with f0 == 0): sum over this item's rows of gz[row][p0 + j] is ADDED (float atomics)
                                // to bias[i][p0 + j] -- the Linear's bias gradient (and its copies), which rides along with the
                                // weight gradient instead of taking a column-sum launch of its own; targets zeroed by the caller
};

struct WgReduce {
  float* out; int ldo;          // [P][F] destination
  const float* slab;            // slice s of the tensor at slab + s * stride, dense [P][F]
  long long stride;
  int P, F, n_slices;
};

constexpr int WG_BK = 32;
constexpr int WG_LDS_BYTES = 2 * 2 * WG_BK * 128 * 4;     // two stages of (A, B) [32][128] fp32

// One work item with all 256 threads of the workgroup; smem = WG_LDS_BYTES.  Ends with a barrier (the staging buffers are free).
// (Tried: the same 64 KB as four stages of 16 rows with three in flight and counted vmcnt waits -- the operands stream from HBM /
// Infinity Cache at an L2 hit rate of 0.43 --: 0.9245 vs 0.9212 ms per step on the same box, i.e. the K loop is not waiting for them.)
__device__ __forceinline__ void wgrad_item(const WgItem& it, float* smem) {
  constexpr int TILE = WG_BK * 128;
  float* As0 = smem;
  float* As1 = smem + TILE;
  float* Bs0 = smem + 2 * TILE;
  float* Bs1 = smem + 3 * TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = (wave >> 1) * 64, wp = (wave & 1) * 64;
  const int l31 = lane & 31, h = lane >> 5;

  // staging: piece q = 4 j + wave moves rows 2 q, 2 q + 1 of the stage; a lane carries 4 columns of one row
  const int a_col = min(it.f0 + 4 * l31, it.F - 4);       // columns beyond the extent re-read valid data: never stored
  const int b_col = min(it.p0 + 4 * l31, it.P - 4);
  auto stage = [&](int k, float* As, float* Bs) {
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(As) + (unsigned)wave * 1024u);
    const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs) + (unsigned)wave * 1024u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = k + 2 * (4 * j + wave) + h;
      glds16(it.A + (size_t)m * it.lda + a_col, __builtin_amdgcn_readfirstlane(la + (unsigned)j * 4096u));
      glds16(it.B + (size_t)m * it.ldb + b_col, __builtin_amdgcn_readfirstlane(lb + (unsigned)j * 4096u));
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = it.bias[0] != nullptr;                // uniform over the workgroup
  float csum = 0.f;                                          // thread: column tid & 127 of the B tile, rows 16 * (tid >> 7) ..

  stage(it.k0, As0, Bs0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nk = (it.k1 - it.k0) / WG_BK;
  for (int kt = 0; kt < nk; ++kt) {
    const float* Ac = (kt & 1) ? As1 : As0;
    const float* Bc = (kt & 1) ? Bs1 : Bs0;
    if (kt + 1 < nk) stage(it.k0 + (kt + 1) * WG_BK, (kt & 1) ? As0 : As1, (kt & 1) ? Bs0 : Bs1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float av[2][4], bv[2][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = 8 * i + 2 * e + h;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) av[fb][e] = Ac[m * 128 + wf + 32 * fb + l31];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) bv[pb][e] = Bc[m * 128 + wp + 32 * pb + l31];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
            acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[fb][e], bv[pb][e], acc[fb][pb], 0, 0, 0);
    }
    if (do_bias) {
      const float* col = Bc + (tid >> 7) * 16 * 128 + (tid & 127);
#pragma unroll
      for (int r = 0; r < 16; ++r) csum += col[r * 128];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (do_bias) {
    const int n = it.p0 + (tid & 127);
    if (n < it.P) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (it.bias[i]) __hip_atomic_fetch_add((gfloat1*)(it.bias[i] + n), csum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // out[p][f]: register quad q of block (fb, pb) holds features f .. f + 3 of patient-side index p
#pragma unroll
  for (int fb = 0; fb < 2; ++fb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = it.f0 + wf + 32 * fb + 8 * q + 4 * h;
        const int p = it.p0 + wp + 32 * pb + l31;
        if (p < it.P && f < it.F)
          stg4(it.out + (size_t)p * it.ldo + f, make_float4(acc[fb][pb][4 * q], acc[fb][pb][4 * q + 1], acc[fb][pb][4 * q + 2], acc[fb][pb][4 * q + 3]));
      }
  __syncthreads();                          // the next item restages buffer 0
}

// wgrad_group.hip: a workgroup runs items blockIdx.x, blockIdx.x + gridDim.x, ...; out[p][f] = sum over slices (fixed order)
__global__ void wgrad_group_kernel(const WgItem* __restrict__ items, int n_items);
__global__ void wgrad_group_reduce(const WgReduce* __restrict__ items);

}  // namespace osd
