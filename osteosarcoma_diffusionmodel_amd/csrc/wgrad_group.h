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
#include "gemm_bf3.h"

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

// ---- the same item on the bf16 matrix pipe at fp32 accuracy (osd_set_option("precision", 1); gemm_bf3.h) --------------------------
// The operands are the fp32 tensors the fp32 item reads -- nothing upstream changes -- and are split where they are staged: a thread
// loads 8 consecutive rows x 2 consecutive columns (eight 8-byte loads, each wave-instruction one 512-byte row segment), which is,
// per column, exactly the 8 reduction indices one lane of v_mfma_f32_32x32x16_bf16 carries: the transposition [row][col] ->
// [col][8 rows] costs no instruction (it is the choice of which registers are packed together), the three planes cost ~5.5 VALU
// operations per element, scheduled between the bf16 MFMAs (one holds the vector issue for 8 of its 32 cycles), and the 16-byte
// units go to LDS in fragment order (gemm_bf3.h: a fragment read is ds_read_b128 at lane * 16).
// A 16-row K stage: wave w stages operand w & 1 (A / B), rows 8 (w >> 1) .. + 7 -- every wave the same share of every stage;
// loads issued two stages ahead (two register sets of 16); two 24 KiB LDS buffers; one barrier per stage; 24 MFMAs per wave and stage.
constexpr int WG3_LDS_BYTES = 2 * 2 * B3_STAGE_BYTES;     // two buffers of (A stage | B stage)

__device__ __forceinline__ void wgrad_item_bf3(const WgItem& it, uint4* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rbA = (wave >> 1) * 2, rbB = (wave & 1) * 2;      // the wave's 64 x 64 accumulators: 32-row blocks of the A / B tile
  const int l31 = lane & 31, h = lane >> 5;
  // staging role: operand (A for even waves) and k group (rows 8 hs .. 8 hs + 7 of the 16-row stage)
  const bool stB = (wave & 1) != 0;
  const int hs = wave >> 1;
  const float* src = stB ? it.B : it.A;
  const int ld = stB ? it.ldb : it.lda;
  const int ext = stB ? it.P : it.F;
  const int col = min((stB ? it.p0 : it.f0) + 2 * lane, ext - 2);      // columns beyond the extent re-read valid data: never stored
  const float* base = src + (size_t)(it.k0 + 8 * hs) * ld + col;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = it.bias[0] != nullptr;                // uniform over the workgroup
  float cs0 = 0.f, cs1 = 0.f;                                // B-staging threads: column sums of their 8 rows x 2 columns

  float2 la[8], lb[8];                                       // the thread's 8 rows x 2 columns of an even / an odd stage
  auto gload = [&](float2 (&v)[8], int st) {
    const float* p = base + (size_t)st * 16 * ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ldg2(p + (size_t)j * ld);
  };
  // split + transposed write: column e of the block -> one 16-byte unit per plane (the thread's 8 rows = k group hs of the stage)
  const int cidx = 2 * lane;                                 // column of the tile; unit slot: row block cidx / 32, lane hs * 32 + cidx % 32
  auto lwrite = [&](const float2 (&v)[8], uint4* buf) {
    uint4* dst = buf + (stB ? B3_STAGE_U4 : 0) + (cidx >> 5) * 64 + hs * 32 + (cidx & 31);
    if (stB && do_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { cs0 += v[j].x; cs1 += v[j].y; }
    }
    const Split4 x0 = split4(make_float4(v[0].x, v[1].x, v[2].x, v[3].x)), x1 = split4(make_float4(v[4].x, v[5].x, v[6].x, v[7].x));
    const Split4 y0 = split4(make_float4(v[0].y, v[1].y, v[2].y, v[3].y)), y1 = split4(make_float4(v[4].y, v[5].y, v[6].y, v[7].y));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      dst[pl * 256] = make_uint4(x0.p[pl].x, x0.p[pl].y, x1.p[pl].x, x1.p[pl].y);
      dst[pl * 256 + 1] = make_uint4(y0.p[pl].x, y0.p[pl].y, y1.p[pl].x, y1.p[pl].y);
    }
  };
  auto read_frags = [&](B3Frags& f, const uint4* buf) {
    const uint4* sp = buf + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f.a[i][pl] = sp[(pl * 4 + rbA + i) * 64];
        f.b[i][pl] = sp[B3_STAGE_U4 + (pl * 4 + rbB + i) * 64];
      }
  };
  auto mfma_block = [&](const B3Frags& f) {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i][PA[t]]), __builtin_bit_cast(bf16x8, f.b[j][PB[t]]),
                                                              acc[i][j], 0, 0, 0);
  };

  uint4* const buf0 = smem;
  uint4* const buf1 = smem + 2 * B3_STAGE_U4;
  const int ns = (it.k1 - it.k0) / 16;                       // 16-row stages (k0, k1 are multiples of 32: ns is even)
  // stage st is loaded during stage st - 2 (register set st & 1) and written to buffer st & 1 during stage st - 1
  gload(la, 0);
  gload(lb, 1);
  lwrite(la, buf0);
  if (ns > 2) gload(la, 2);
  __syncthreads();
  B3Frags fr;
  // one stage: the MFMAs of stage st with the split + write of stage st + 1 scheduled into their shadow, then the loads of stage st + 3
  auto stage = [&](int st, const uint4* cur, uint4* nxt, float2 (&vn)[8]) {
    read_frags(fr, cur);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(fr);
    if (st + 1 < ns) lwrite(vn, nxt);
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // ... then a few VALU of the split
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st + 3 < ns) gload(vn, st + 3);
    __syncthreads();
  };
  for (int st = 0; st < ns; st += 2) {
    stage(st, buf0, buf1, lb);
    stage(st + 1, buf1, buf0, la);
  }
  if (do_bias && stB) {
    // a column's rows are spread over waves 1 and 3 (k groups) and over the stages: each staging thread adds its share
    const int n = it.p0 + cidx;
    if (n < it.P) {        // (a clamped column pair never has n < P: P % 4 == 0 and the clamp is to P - 2)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (it.bias[i]) {
          __hip_atomic_fetch_add((gfloat1*)(it.bias[i] + n), cs0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gfloat1*)(it.bias[i] + n + 1), cs1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
  }
  // out[p][f]: the fp32 item's store (same accumulator layout)
  const int wf = rbA * 32, wp = rbB * 32;
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

// (Tried in round 4: the fp32 item staged through registers like the bf16x3 one -- a thread's 8 rows x 2 columns written per column
// as two 16-byte units in fragment order, so that a fragment read is ds_read_b128: 16 LDS reads per 32-row K step instead of 64.
// Parity-green and slower, 0.945 vs 0.921 ms per step: under v_mfma_f32_32x32x2_f32 the loads / ds_writes a wave issues itself come
// out of the matrix loop's time, the LDS-DMA pieces do not.  Removed; the LDS-DMA item above stays the fp32 path.)

// wgrad_group.hip: a workgroup runs items blockIdx.x, blockIdx.x + gridDim.x, ...; out[p][f] = sum over slices (fixed order)
template <int MODE> __global__ void wgrad_group_kernel(const WgItem* __restrict__ items, int n_items);      // 0 fp32 (LDS-DMA), 1 bf16x3
__global__ void wgrad_group_reduce(const WgReduce* __restrict__ items);

}  // namespace osd
