// gemm_glds.h -- the forward (KC x KC) fp32 MFMA GEMM with direct global->LDS staging.
//
// Same contract and fragment layout as gemm_kernel (gemm.h) and the same epilogues, but the
// operand tiles never pass through VGPRs: every wave issues global_load_lds_dwordx4 (1 KiB per
// wave-instruction, LDS destination = wave-uniform base + lane*16) straight into an UNPADDED
// [rows][32] fp32 image.  An unpadded 128-byte row would make ds_read_b128 8-way bank
// conflicted, so the 16-byte chunks of row R are stored XOR-swizzled by (R>>1)&7; the DMA
// destination is linear, so the swizzle is applied to the per-lane SOURCE address and again
// on the read (both sides or neither).
//
// Preconditions (checked by the host, glds_ok()): 16-byte aligned operands, lda/ldb % 4 == 0,
// the A operand (weights) readable and ZERO for k in [K, roundup(K,32)) -- true when
// K % 32 == 0 or when the caller hands a zero-padded packed copy -- and K0 % 32 == 0 for a
// two-panel B.  B's K tail is clamped to valid addresses (finite data times zero weights).
// Rows beyond F / P re-read the last valid row: they only feed accumulators never stored.
#pragma once
#include "gemm.h"

namespace osd {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS bytes
// [lds_dst, lds_dst + 1024).  Written as asm so that hipcc does not count it: with the builtin
// the compiler cannot prove that the DMA target and the tile being read are different halves
// of the one LDS array and drains vmcnt(0) before the next ds_read, which serialises the
// prefetch behind the compute.  M0 carries the LDS base and is restored (hipcc reserves it).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <class T>
struct GldsTile {
  static constexpr int A_ELEMS = T::BF * BK;
  static constexpr int B_ELEMS = T::BP * BK;
  static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
  static constexpr int NA = T::BF / 32;   // wave-instructions per wave per tile, A
  static constexpr int NB = T::BP / 32;
};

template <class T, class Epi>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_glds_kernel(GemmArgs g, typename Epi::Args ea) {
  typedef GldsTile<T> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As0 = smem;
  float* As1 = smem + G::A_ELEMS;
  float* Bs0 = smem + 2 * G::A_ELEMS;
  float* Bs1 = smem + 2 * G::A_ELEMS + G::B_ELEMS;

  const int nft = (g.F + T::BF - 1) / T::BF;
  const int npt = (g.P + T::BP - 1) / T::BP;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int ft = idx % nft;
  const int pt = (idx / nft) * 8 + (b & 7);
  if (pt >= npt) return;
  const int f0 = ft * T::BF, p0 = pt * T::BP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // ---- staging addresses: wave-instruction (j*4 + wave) moves rows 8*(j*4+wave) .. +7 ----
  int a_off[G::NA], a_k4[G::NA];
#pragma unroll
  for (int j = 0; j < G::NA; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    int rg = f0 + row;
    rg = rg < g.F ? rg : g.F - 1;
    a_k4[j] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
    a_off[j] = rg * g.lda + a_k4[j];
  }
  int b_row[G::NB], b_k4[G::NB];
#pragma unroll
  for (int j = 0; j < G::NB; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    int rg = p0 + row;
    b_row[j] = rg < g.P ? rg : g.P - 1;
    b_k4[j] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
  }
  auto stage = [&](int k0, float* As, float* Bs) {
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(As) + (unsigned)wave * 1024u);
    const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs) + (unsigned)wave * 1024u);
#pragma unroll
    for (int j = 0; j < G::NA; ++j) glds16(g.A + (size_t)(a_off[j] + k0), la + (unsigned)j * 4096u);
    const bool first = k0 < g.K0;          // uniform: K0 is a multiple of BK (or >= K)
    const float* base = first ? g.B0 : g.B1;
    const int ld = first ? g.ldb0 : g.ldb1;
    const int kb = first ? k0 : k0 - g.K0;
    const int kend = first ? (g.K0 < g.K ? g.K0 : g.K) : g.K - g.K0;
#pragma unroll
    for (int j = 0; j < G::NB; ++j) {
      int k = kb + b_k4[j];
      k = k < kend - 4 ? k : kend - 4;
      glds16(base + (size_t)b_row[j] * ld + k, lb + (unsigned)j * 4096u);
    }
  };

  // ---- fragment read offsets: row R, 16-byte chunk (2i+h) ^ ((R>>1)&7) ----
  int a_rd[T::NFB], a_sw[T::NFB], b_rd[T::NPB], b_sw[T::NPB];
#pragma unroll
  for (int fb = 0; fb < T::NFB; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * BK; a_sw[fb] = h ^ ((R >> 1) & 7); }
#pragma unroll
  for (int pb = 0; pb < T::NPB; ++pb) { const int R = wp + 32 * pb + l31; b_rd[pb] = R * BK; b_sw[pb] = h ^ ((R >> 1) & 7); }

  auto compute = [&](const float* As, const float* Bs, int i) {
    float a[T::NFB][4], bb[T::NPB][4];
#pragma unroll
    for (int fb = 0; fb < T::NFB; ++fb) {
      const float4 t = *reinterpret_cast<const float4*>(&As[a_rd[fb] + 4 * (a_sw[fb] ^ (2 * i))]);
      a[fb][0] = t.x; a[fb][1] = t.y; a[fb][2] = t.z; a[fb][3] = t.w;
    }
#pragma unroll
    for (int pb = 0; pb < T::NPB; ++pb) {
      const float4 t = *reinterpret_cast<const float4*>(&Bs[b_rd[pb] + 4 * (b_sw[pb] ^ (2 * i))]);
      bb[pb][0] = t.x; bb[pb][1] = t.y; bb[pb][2] = t.z; bb[pb][3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int fb = 0; fb < T::NFB; ++fb)
#pragma unroll
        for (int pb = 0; pb < T::NPB; ++pb)
          acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[pb][e], acc[fb][pb], 0, 0, 0);
  };

  const int nk = (g.K + BK - 1) / BK;
  stage(0, As0, Bs0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the asm DMAs are invisible to hipcc's counters
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    float* Ac = (kt & 1) ? As1 : As0;
    float* Bc = (kt & 1) ? Bs1 : Bs0;
    if (kt + 1 < nk) stage((kt + 1) * BK, (kt & 1) ? As0 : As1, (kt & 1) ? Bs0 : Bs1);
    compute(Ac, Bc, 0);
    compute(Ac, Bc, 1);
    compute(Ac, Bc, 2);
    compute(Ac, Bc, 3);
    // the DMA of tile kt+1 had the whole MFMA block to land; publish it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  Epi::template apply<T::NFB, T::NPB, true>(acc, ea, f0 + wf, p0 + wp, lane, g.F, g.P);
}

}  // namespace osd
