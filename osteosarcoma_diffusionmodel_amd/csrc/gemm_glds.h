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
// K % 32 == 0 or when the caller hands a zero-padded packed copy; or (GemmArgs::a_kmax) A clamped
// at its true extent and the B operand zero beyond it -- and K0 % 32 == 0 for a two-panel B.  B's K tail is clamped to valid addresses (finite data times zero weights).
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
// The same piece with the address split into a wave-uniform base (SGPR pair) and a per-lane 32-bit byte offset: a K loop then
// advances the base with two scalar adds per operand instead of recomputing a 64-bit address per lane and piece.  That
// matters here more than elsewhere: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate, and every VALU instruction issued
// beside it comes out of the matrix loop's own time (an address costs ~8 VALU issues, a K step has 8 pieces per wave).
// s_nop 4: an SGPR pair freshly written by readfirstlane / VALU may not be read as a VMEM base for 5 states.
// a pointer every lane holds the same value of, moved to SGPRs (what an "s" asm operand needs to be provably uniform)
// (as a GLOBAL-address-space pointer: for a generic one hipcc emits shared-aperture checks around the integer casts, and one
// of them -- V_CMP_NE_U32 0, src_shared_base -- does not even assemble)
typedef const __attribute__((address_space(1))) float* gfloat_ptr;
__device__ __forceinline__ gfloat_ptr uniform_ptr(const float* p) {
  const gfloat_ptr q = (gfloat_ptr)p;
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<gfloat_ptr>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16s(gfloat_ptr sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 4\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
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

// Tile schedule.  g.persist == 0: one tile per workgroup, XCD-aware order (blocks b, b+8, ... share an
// XCD and take the feature tiles of one patient tile).  g.persist == 1 (host guarantees 64 % nft == 0 and
// gridDim.x % (8*nft) == 0): a workgroup keeps ONE feature tile and walks patient tiles
// pt = pg, pg + gridDim.x/nft, ...; the nft workgroups of a patient group sit on one XCD.  Persistence
// lets a workgroup (a) fetch its per-feature epilogue parameters once, before the first K loop,
// (b) issue the first DMA of the NEXT tile before the epilogue of the current one, and (c) leave the
// epilogue's stores in flight instead of draining them before its slot is reused.
//
// NG = 2 ("two wave groups", training-sized batches): a layer of 4096 rows x 256 features is 256 tiles -- one workgroup per CU, one
// wave per SIMD, and a lone wave cannot hide the barrier / wait / fragment-read latency of a K step (stamps: 1 660 cycles per step
// for 1 024 cycles of MFMA at 64 x 64).  Here a workgroup has EIGHT waves: waves 4-7 are a second copy of the four tile waves with
// their own staging buffers, and each group reduces one half of the K steps; at the end group 1 hands its accumulators to group 0
// through LDS and leaves, group 0 adds them (first half + second half: a different fp32 summation order than NG = 1, so the
// variant is opt-in per launch, GemmArgs::ksplit, and never used where the two sampling engines must agree bitwise) and runs the
// epilogue.  Two waves per SIMD on all 256 CUs without needing 512 tiles.
template <class T, class Epi, int NG = 1>
__global__ __launch_bounds__(NTHREADS * NG, NG == 1 ? 2 : 1) void gemm_glds_kernel(GemmArgs g, typename Epi::Args ea) {
  typedef GldsTile<T> G;
  static_assert(NG == 1 || NG == 2, "one or two wave groups");
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));      // wave group of this wave
  float* const smem = smem_all + grp * (G::LDS_BYTES / 4);
  float* As0 = smem;
  float* As1 = smem + G::A_ELEMS;
  float* Bs0 = smem + 2 * G::A_ELEMS;
  float* Bs1 = smem + 2 * G::A_ELEMS + G::B_ELEMS;

  const int nft = (g.F + T::BF - 1) / T::BF;
  const int npt = (g.P + T::BP - 1) / T::BP;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int ft = idx % nft;
  int pt, pstride;
  if (g.persist) {
    const int per_xcd = gridDim.x >> 3;
    pt = idx / nft + (per_xcd / nft) * (b & 7);
    pstride = gridDim.x / nft;
  } else {
    pt = (idx / nft) * 8 + (b & 7);
    pstride = npt;                       // exactly one tile
  }
  if (pt >= npt) return;
  if (g.tri && ft > pt) return;          // symmetric product (never with g.persist): the mirror tile carries this one's weight
  const int f0 = ft * T::BF;

  const int tid = threadIdx.x & (NTHREADS - 1);       // position inside the wave group
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (g.stamps) st0 = __builtin_amdgcn_s_memtime();
  const int wf = (wave / T::NWP) * T::WF;
  const int wp = (wave % T::NWP) * T::WP;
  const int l31 = lane & 31, h = lane >> 5;

  // per-feature epilogue parameters: in flight during the first K loop
  const auto pre = Epi::template prefetch<T::NFB, true>(ea, f0 + wf, lane, g.F);

  // ---- staging addresses: wave-instruction (j*4 + wave) moves rows 8*(j*4+wave) .. +7 ----
  int a_off[G::NA], a_k4[G::NA];
  int b_k4[G::NB], b_lrow[G::NB];
#pragma unroll
  for (int j = 0; j < G::NA; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    int rg = f0 + row;
    rg = rg < g.F ? rg : g.F - 1;
    a_k4[j] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
    a_off[j] = rg * g.lda + a_k4[j];
  }
#pragma unroll
  for (int j = 0; j < G::NB; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    b_lrow[j] = row;
    b_k4[j] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
  }
  // ---- DMA of one K tile, split into single wave-instructions so that each can be issued between
  // groups of MFMAs of the tile being computed.
  struct StageCtx { unsigned la, lb; const float* bbase; int bld, bk, bkend, ak, p0; };
  auto stage_begin = [&](int k0, int p0, float* As, float* Bs) {
    StageCtx c;
    c.la = __builtin_amdgcn_readfirstlane(lds_addr(As) + (unsigned)wave * 1024u);
    c.lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs) + (unsigned)wave * 1024u);
    const bool first = k0 < g.K0;          // uniform: K0 is a multiple of BK (or >= K)
    c.bbase = first ? g.B0 : g.B1;
    c.bld = first ? g.ldb0 : g.ldb1;
    c.bk = first ? k0 : k0 - g.K0;
    c.bkend = first ? (g.K0 < g.K ? g.K0 : g.K) : g.K - g.K0;
    c.ak = k0;
    c.p0 = p0;
    return c;
  };
  // piece j in [0, NA + NB): A pieces first
  auto stage_piece = [&](const StageCtx& c, int j) {
    if (j < G::NA) {
      int adj = 0;
      if (g.a_kmax > 0) {                        // uniform: clamp the 16-byte chunk into [0, a_kmax) (finite duplicates x B's zeros)
        const int k = c.ak + a_k4[j];
        adj = k < g.a_kmax - 4 ? 0 : g.a_kmax - 4 - k;
      }
      glds16(g.A + (size_t)(a_off[j] + c.ak + adj), __builtin_amdgcn_readfirstlane(c.la + (unsigned)j * 4096u));
    } else {
      const int jb = j - G::NA;
      int k = c.bk + b_k4[jb];
      k = k < c.bkend - 4 ? k : c.bkend - 4;
      int rg = c.p0 + b_lrow[jb];
      rg = rg < g.P ? rg : g.P - 1;
      glds16(c.bbase + (size_t)rg * c.bld + k, __builtin_amdgcn_readfirstlane(c.lb + (unsigned)jb * 4096u));
    }
  };
  constexpr int NPIECE = G::NA + G::NB;

  // ---- fragment read offsets: row R, 16-byte chunk (2i+h) ^ ((R>>1)&7) ----
  int a_rd[T::NFB], a_sw[T::NFB], b_rd[T::NPB], b_sw[T::NPB];
#pragma unroll
  for (int fb = 0; fb < T::NFB; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * BK; a_sw[fb] = h ^ ((R >> 1) & 7); }
#pragma unroll
  for (int pb = 0; pb < T::NPB; ++pb) { const int R = wp + 32 * pb + l31; b_rd[pb] = R * BK; b_sw[pb] = h ^ ((R >> 1) & 7); }

  f32x16 acc[T::NFB][T::NPB];

  // quarter i of a staged tile; with do_stage the DMA of the next K tile is issued during the FIRST quarter (two pieces
  // per k-pair group): measured, a piece issued late in the step does not land before the step's barrier and the wait for
  // it idles the matrix pipe (+2.3 % end to end against spreading the pieces over all four quarters)
  auto compute = [&](const float* As, const float* Bs, int i, const StageCtx& sc, bool do_stage) {
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
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int fb = 0; fb < T::NFB; ++fb)
#pragma unroll
        for (int pb = 0; pb < T::NPB; ++pb)
          acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[pb][e], acc[fb][pb], 0, 0, 0);
      // all DMA pieces in the first quarter of the K step (two per k-pair group): they then have three quarters of a step to land
      if (do_stage && i == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (2 * e < NPIECE) stage_piece(sc, 2 * e);
        if (2 * e + 1 < NPIECE) stage_piece(sc, 2 * e + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  const int nk_all = (g.K + BK - 1) / BK;
  // NG = 2: group 0 reduces K steps [0, nk), group 1 [kbeg, kbeg + kcnt); both walk nk iterations (the barrier counts every wave)
  const int nk = NG == 1 ? nk_all : (nk_all + 1) / 2;
  const int kbeg = grp ? nk : 0;
  const int kcnt = NG == 1 ? nk : (grp ? nk_all - nk : nk);
  // first K tile of the first patient tile
  {
    const StageCtx c0 = stage_begin(kbeg * BK, pt * T::BP, As0, Bs0);
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) stage_piece(c0, j);
  }
  // stores of a FULL tile's epilogue: a lower bound on the vector-memory operations a wave issues after
  // the prefetch DMA of the next tile, so vmcnt(that many) proves the (older) DMA has landed while the
  // stores stay in flight.  Edge tiles may skip stores under an empty exec mask: they wait for vmcnt(0).
  constexpr int FULL_TILE_STORES = T::NFB * T::NPB * 4;
  bool counted_wait = false;
  for (; pt < npt; pt += pstride) {
    const int p0 = pt * T::BP;
    if (counted_wait) {
      if constexpr (FULL_TILE_STORES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (FULL_TILE_STORES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (FULL_TILE_STORES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the asm DMAs are invisible to hipcc's counters
    }
    __syncthreads();
    if (g.stamps && st1 == 0) st1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < T::NFB; ++i)
#pragma unroll
      for (int j = 0; j < T::NPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      float* Ac = (kt & 1) ? As1 : As0;
      float* Bc = (kt & 1) ? Bs1 : Bs0;
      const bool more = kt + 1 < kcnt;
      const StageCtx sc = stage_begin(more ? (kbeg + kt + 1) * BK : 0, p0, (kt & 1) ? As0 : As1, (kt & 1) ? Bs0 : Bs1);
      if (NG == 1 || kt < kcnt) {            // uniform per wave group (group 1 may have one step less)
        compute(Ac, Bc, 0, sc, more);
        compute(Ac, Bc, 1, sc, more);
        compute(Ac, Bc, 2, sc, more);
        compute(Ac, Bc, 3, sc, more);
      }
      // the DMA of K tile kt+1 had the MFMA block to land; publish it
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (g.stamps && st2 == 0) st2 = __builtin_amdgcn_s_memtime();
    // first K tile of the next patient tile: every wave has left the K loop through its last barrier,
    // so buffer 0 is free; the DMA flies while the epilogue runs
    const int ptn = pt + pstride;
    if (ptn < npt) {
      const StageCtx cn = stage_begin(0, ptn * T::BP, As0, Bs0);
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) stage_piece(cn, j);
    }
    if constexpr (NG == 2) {
      // group 1's accumulators through its own staging buffers (free since the K loop's last barrier): [wave][register][lane]
      constexpr int NREG = T::NFB * T::NPB * 16;
      static_assert(4 * NREG * 64 * 4 <= G::LDS_BYTES, "accumulator hand-over must fit the group's staging buffers");
      float* const red = smem_all + G::LDS_BYTES / 4 + wave * (NREG * 64);
      if (grp == 1) {
#pragma unroll
        for (int i = 0; i < T::NFB; ++i)
#pragma unroll
          for (int j = 0; j < T::NPB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((i * T::NPB + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (grp == 1) return;                  // uniform per wave; finished waves no longer count at the workgroup barrier
#pragma unroll
      for (int i = 0; i < T::NFB; ++i)
#pragma unroll
        for (int j = 0; j < T::NPB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * T::NPB + j) * 16 + r) * 64 + lane];
    }
    counted_wait = Epi::COUNTED_STORES && (p0 + T::BP <= g.P) && (f0 + T::BF <= g.F);
    if constexpr (Epi::XBUF) {
      // the wave's share of the second operand buffers (idle from the K loop's last barrier until the next tile's second K stage)
      constexpr int XR = T::NPB * 1024;
      static_assert(G::A_ELEMS % XR == 0 && 4 * XR <= G::A_ELEMS + G::B_ELEMS, "transposer regions must tile the idle buffers");
      const int xo = wave * XR;
      float* const xb = xo < G::A_ELEMS ? As1 + xo : Bs1 + (xo - G::A_ELEMS);
      Epi::template apply<T::NFB, T::NPB, true>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P, xb);
    } else {
      Epi::template apply<T::NFB, T::NPB, true>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P);
    }
  }
  if (g.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // include the store drain in the last interval
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long* o = g.stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
  }
}

}  // namespace osd
