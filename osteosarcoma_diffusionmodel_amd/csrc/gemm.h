// gemm.h -- fp32 MFMA tile GEMM for gfx950 with fused epilogues.
//
//   out[p][f] (+)= epilogue( sum_k A(f,k) * B(p,k) )
//
// "f" (feature) is the MFMA M index and lands in the accumulator REGISTERS, "p"
// (patient / batch row) is the MFMA N index and lands on the LANES.  With
// v_mfma_f32_32x32x2_f32 a lane therefore holds, for ONE patient, 16 features of a
// 32-feature block, 4 consecutive features per register quad.  That orientation makes
//   * per-row GroupNorm statistics a register sum plus one lane^32 exchange, and
//   * the output store a float4 per register quad into row-major [p][f],
// which is what every epilogue here relies on.
//
// Operand storage:  KC  = [row][k]  (k contiguous:  Linear weights W[out][in], activations
//                                    X[m][k], upstream gradients dY[m][n] for dgrad)
//                   NKC = [k][row]  (row contiguous: W[n][k] read as A(k_feat, n) for dgrad,
//                                    X[m][k] / dY[m][n] reduced over m for wgrad)
// LDS images: KC tile [R][BK+4] read with ds_read_b128 (conflict-free at the +4 pad),
//             NKC tile [BK][R] read with ds_read_b32 (lanes walk consecutive addresses).
// The k order inside a BK=32 step is permuted identically for A and B (lane half h of
// read i covers k = 8i+4h..8i+4h+3), which leaves the sum unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace osd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTHREADS = 256;

struct GemmArgs {
  const float* A;  int lda;
  const float* B0; int ldb0;
  const float* B1; int ldb1;   // second K panel of a KC B operand (k >= K0); unused when K0 >= K
  int K0;
  int F, P, K;
  int kchunk;                  // gemm_kernel split-K: blockIdx.y reduces k in [y*kchunk, (y+1)*kchunk); 0 = no split
  int persist;                 // gemm_glds_kernel: persistent patient-tile walk (see gemm_glds.h)
  int ksplit;                  // gemm_glds_kernel: 1 = the two-wave-group variant (NG = 2, see gemm_glds.h) where it is instantiated
  int a_kmax;                  // gemm_glds_kernel: > 0 = the A operand is only readable for k < a_kmax (an unpadded weight whose K is not a
                               // multiple of 32): its staging addresses are clamped there, and the B operand must hold ZEROS for
                               // k in [a_kmax, K) instead (a padded activation buffer owned by the library)
  int tri;                     // 1: a symmetric product (A == B, F == P, square 128 x 128 tiles): only tiles on or above the diagonal
                               // (feature tile <= patient tile) are computed; the epilogue weighs the off-diagonal ones twice (EpiRbfSum)
  unsigned long long* stamps;  // diagnostic: per-wave s_memtime stamps [wave][4] (null in production)
};

template <int BF_, int BP_, int WF_, int WP_>
struct Tile {
  static constexpr int BF = BF_, BP = BP_, WF = WF_, WP = WP_;
  static constexpr int NWF = BF / WF, NWP = BP / WP;
  static constexpr int NFB = WF / 32, NPB = WP / 32;
  static_assert(NWF * NWP == 4, "4 waves per workgroup");
  static_assert(WF % 32 == 0 && WP % 32 == 0, "wave tile in 32x32 MFMA blocks");
  static constexpr int A_ELEMS = BF * LDK;   // >= BK*BF (NKC image)
  static constexpr int B_ELEMS = BP * LDK;
  static constexpr int LDS_BYTES = 2 * (A_ELEMS + B_ELEMS) * 4;
};

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Explicitly GLOBAL loads / stores.  Operand pointers that are kernel arguments are known to be global, but fields of an argument
// block or work descriptor that itself lives in memory (the chain kernel's ChainArgs, the grouped weight-gradient items, the
// persistent backward kernel's GEMM descriptors) are generic pointers to hipcc, and an access through one is a FLAT instruction:
// both counters, out-of-order return (every wait degrades to vmcnt(0) & lgkmcnt(0), so each LDS wait of a tile loop also waits
// for the global prefetch in flight), and the LDS pipe takes part in a global access.  Chain kernel, same box: 22.42 k -> 22.80 k
// patients/s; a dgrad tile of the persistent backward kernel: 3x.
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef float v2f32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v4f32 gfloat4;
typedef __attribute__((address_space(1))) v2f32 gfloat2;
typedef __attribute__((address_space(1))) float gfloat1;
__device__ __forceinline__ float4 ldg4(const float* p) { const v4f32 v = *(const gfloat4*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 ldg2(const float* p) { const v2f32 v = *(const gfloat2*)p; return make_float2(v.x, v.y); }
__device__ __forceinline__ float ldg1(const float* p) { return *(const gfloat1*)p; }
__device__ __forceinline__ void stg4(float* p, float4 v) { const v4f32 w = {v.x, v.y, v.z, v.w}; *(gfloat4*)p = w; }
__device__ __forceinline__ void stg1(float* p, float v) { *(gfloat1*)p = v; }

// guarded 4-element load of v[i..i+3] from an array of n elements
__device__ __forceinline__ float4 ld4g(const float* v, int i, int n) {
  const float* p = v + i;
  if (i + 3 < n && aligned16(p)) return *reinterpret_cast<const float4*>(p);
  float4 r;
  r.x = (i < n) ? p[0] : 0.f;
  r.y = (i + 1 < n) ? p[1] : 0.f;
  r.z = (i + 2 < n) ? p[2] : 0.f;
  r.w = (i + 3 < n) ? p[3] : 0.f;
  return r;
}
// guarded 4-element store to row[i..i+3], row has n valid elements
__device__ __forceinline__ void st4g(float* row, int i, int n, float4 v) {
  float* p = row + i;
  if (i + 3 < n && aligned16(p)) { *reinterpret_cast<float4*>(p) = v; return; }
  if (i < n) p[0] = v.x;
  if (i + 1 < n) p[1] = v.y;
  if (i + 2 < n) p[2] = v.z;
  if (i + 3 < n) p[3] = v.w;
}

// FAST = every pointer 16-byte aligned, leading dimensions and extents multiples of 4: loads
// become unconditional float4 loads from a clamped (always valid) address with the value
// zeroed by a select, so the compiler can keep a whole tile of loads in flight instead of
// waiting behind each guarded load.
template <bool FAST>
__device__ __forceinline__ float4 ldq(const float* v, int i, int n) {
  if constexpr (FAST) {
    const int ic = (i < n - 4) ? i : n - 4;
    const float4 r = ldg4(v + ic);
    const bool ok = i < n;
    return make_float4(ok ? r.x : 0.f, ok ? r.y : 0.f, ok ? r.z : 0.f, ok ? r.w : 0.f);
  } else {
    return ld4g(v, i, n);
  }
}
// FAST staging load: v[min(i, n-4) .. +3], no zeroing.  Out-of-range chunks re-read valid
// (finite) data; the kernel zeroes the K tail of the A operand when it writes LDS, and rows or
// features beyond the extents only ever feed accumulators that are never stored.
__device__ __forceinline__ float4 ldraw(const float* v, int i, int n) {
  const int ic = (i < n - 4) ? i : n - 4;
  return ldg4(v + ic);
}
template <bool FAST>
__device__ __forceinline__ void stq(float* row, int i, int n, float4 v) {
  if constexpr (FAST) {
    if (i < n) stg4(row + i, v);
  } else {
    st4g(row, i, n, v);
  }
}

// ---- global -> register staging of one BK-deep tile ------------------------------
// KC operand, R rows: thread owns k-chunk kc = tid&7 of rows (tid>>3) + 32*i.
template <int R>
struct StageKC {
  static constexpr int N = R / 32;
  float4 v[N];
  template <bool FAST>
  __device__ __forceinline__ void load(const GemmArgs& g, bool isB, int r0, int nrows, int k0, int tid) {
    const int kc = tid & 7;
    int k = k0 + 4 * kc;
    const float* base; int ld; int kend;
    if (isB) {
      // FAST: K0 is a multiple of BK (or >= K), so the panel choice is uniform over the tile
      const bool first = FAST ? (k0 < g.K0) : (k < g.K0);
      if (first) { base = g.B0; ld = g.ldb0; kend = (g.K0 < g.K ? g.K0 : g.K); }
      else { base = g.B1; ld = g.ldb1; kend = g.K - g.K0; k -= g.K0; }
    } else { base = g.A; ld = g.lda; kend = g.K; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int row = r0 + (tid >> 3) + 32 * i;
      if constexpr (FAST) {
        // clamped, unconditional: out-of-range rows / k re-read valid data (see ldraw)
        const int rc = (row < nrows) ? row : nrows - 1;
        v[i] = ldraw(base + (size_t)rc * ld, k, kend);
      } else {
        if (row < nrows && k < kend) v[i] = ld4g(base + (size_t)row * ld, k, kend);
        else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // kvalid = number of in-range k of this tile; chunks at or beyond it are stored as zeros
  // (only needed for ONE operand: 0 * finite = 0).  Pass BK to skip the zeroing.
  __device__ __forceinline__ void store(float* lds, int tid, int kvalid) const {
    const int kc = tid & 7;
    const bool ok = 4 * kc < kvalid;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float4 t = v[i];
      *reinterpret_cast<float4*>(&lds[((tid >> 3) + 32 * i) * LDK + 4 * kc]) =
          make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
};
// NKC operand, tile [BK][R]: chunk c = tid + 256*i -> k = c / (R/4), rc = c % (R/4).
template <int R>
struct StageNK {
  static constexpr int CPR = R / 4;
  static constexpr int N = (BK * CPR) / NTHREADS;
  static_assert(N >= 1, "tile too small");
  float4 v[N];
  template <bool FAST>
  __device__ __forceinline__ void load(const float* base, int ld, int r0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + NTHREADS * i;
      const int k = k0 + c / CPR;
      const int r = r0 + 4 * (c % CPR);
      if constexpr (FAST) {
        const int kc = (k < K) ? k : K - 1;
        v[i] = ldraw(base + (size_t)kc * ld, r, nrows);
      } else {
        if (k < K && r < nrows) v[i] = ld4g(base + (size_t)k * ld, r, nrows);
        else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid, int kvalid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + NTHREADS * i;
      const bool ok = (c / CPR) < kvalid;
      const float4 t = v[i];
      *reinterpret_cast<float4*>(&lds[(c / CPR) * R + 4 * (c % CPR)]) =
          make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
};

// ---- one output tile ---------------------------------------------------------------
// Epi::apply<NFB,NPB,FAST>(acc, args, f_wave, p_wave, lane, F, P) consumes the wave's accumulators.
// gemm_tile computes the BF x BP tile at (f0, p0) with all 256 threads of the workgroup; `smem` = T::LDS_BYTES of LDS.
// NG = 2 (gemm_kernel<..., 2>: 512 threads): two wave groups, each with its own staging buffers, reduce one half of the K tiles each;
// group 1 hands its accumulators to group 0 through LDS and leaves, group 0 adds them and runs the epilogue (the two-wave-group
// idea of gemm_glds.h for the register-staged kernel: the first dgrad of a training step is 256 tiles x 63 K tiles).
template <class T, bool AKC, bool BKC, class Epi, bool FAST, int NG = 1>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const typename Epi::Args& ea, int f0, int p0, float* smem_all) {
  const int grp = NG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  float* const smem = smem_all + grp * (T::LDS_BYTES / 4);
  float* As0 = smem;
  float* As1 = smem + T::A_ELEMS;
  float* Bs0 = smem + 2 * T::A_ELEMS;
  float* Bs1 = smem + 2 * T::A_ELEMS + T::B_ELEMS;

  const int tid = NG == 1 ? threadIdx.x : (threadIdx.x & (NTHREADS - 1));      // position inside the wave group
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wf = (wave / T::NWP) * T::WF;   // wave's feature offset inside the block tile
  const int wp = (wave % T::NWP) * T::WP;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[T::NFB][T::NPB];
#pragma unroll
  for (int i = 0; i < T::NFB; ++i)
#pragma unroll
    for (int j = 0; j < T::NPB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  StageKC<T::BF> sAk; StageNK<T::BF> sAn;
  StageKC<T::BP> sBk; StageNK<T::BP> sBn;

  auto gload = [&](int k0) {
    if constexpr (AKC) sAk.template load<FAST>(g, false, f0, g.F, k0, tid);
    else sAn.template load<FAST>(g.A, g.lda, f0, g.F, k0, g.K, tid);
    if constexpr (BKC) sBk.template load<FAST>(g, true, p0, g.P, k0, tid);
    else sBn.template load<FAST>(g.B0, g.ldb0, p0, g.P, k0, g.K, tid);
  };
  auto lstore = [&](float* As, float* Bs, int k0) {
    const int kvalid = FAST ? g.K - k0 : BK;     // slow path zero-fills at load time
    if constexpr (AKC) sAk.store(As, tid, kvalid); else sAn.store(As, tid, kvalid);
    if constexpr (BKC) sBk.store(Bs, tid, BK); else sBn.store(Bs, tid, BK);
  };
  // one quarter (8 of the 32 k) of a staged tile: 4 LDS fragment reads, 16 * NFB * NPB / 4 MFMAs
  auto compute = [&](const float* As, const float* Bs, int i) {
    float a[T::NFB][4], bb[T::NPB][4];
#pragma unroll
    for (int fb = 0; fb < T::NFB; ++fb) {
      if constexpr (AKC) {
        const float4 t = *reinterpret_cast<const float4*>(&As[(wf + 32 * fb + l31) * LDK + 8 * i + 4 * h]);
        a[fb][0] = t.x; a[fb][1] = t.y; a[fb][2] = t.z; a[fb][3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[fb][e] = As[(8 * i + 4 * h + e) * T::BF + wf + 32 * fb + l31];
      }
    }
#pragma unroll
    for (int pb = 0; pb < T::NPB; ++pb) {
      if constexpr (BKC) {
        const float4 t = *reinterpret_cast<const float4*>(&Bs[(wp + 32 * pb + l31) * LDK + 8 * i + 4 * h]);
        bb[pb][0] = t.x; bb[pb][1] = t.y; bb[pb][2] = t.z; bb[pb][3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bb[pb][e] = Bs[(8 * i + 4 * h + e) * T::BP + wp + 32 * pb + l31];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int fb = 0; fb < T::NFB; ++fb)
#pragma unroll
        for (int pb = 0; pb < T::NPB; ++pb)
          acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[pb][e], acc[fb][pb], 0, 0, 0);
  };

  // Software pipeline, two LDS buffers, register prefetch two tiles ahead:
  //   iteration kt computes on LDS tile kt, writes register tile kt+1 into the other buffer and
  //   issues the global loads of tile kt+2 -- both placed BETWEEN MFMA groups so they issue in the
  //   shadow of the matrix pipe; one barrier per iteration publishes tile kt+1.
  // (writing buffer (kt+1)&1 during iteration kt is safe: its tile kt-1 was last read in
  //  iteration kt-1, which every wave left through the barrier.)
  const int nk_all = (g.K + BK - 1) / BK;
  // NG = 2: group 0 reduces K tiles [0, nk), group 1 [kb, kb + kc); both walk nk iterations (the barrier counts every wave)
  const int nk = NG == 1 ? nk_all : (nk_all + 1) / 2;
  const int kb = grp ? nk : 0;
  const int kc = NG == 1 ? nk : (grp ? nk_all - nk : nk);
  if (kc > 0) {
    gload(kb * BK);
    lstore(As0, Bs0, kb * BK);
    if (kc > 1) gload((kb + 1) * BK);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    float* Ac = (kt & 1) ? As1 : As0;
    float* Bc = (kt & 1) ? Bs1 : Bs0;
    float* An = (kt & 1) ? As0 : As1;
    float* Bn = (kt & 1) ? Bs0 : Bs1;
    if (NG == 1 || kt < kc) {             // uniform per wave group (group 1 may have one tile less)
      compute(Ac, Bc, 0);
      if (kt + 1 < kc) lstore(An, Bn, (kb + kt + 1) * BK);
      compute(Ac, Bc, 1);
      if (kt + 2 < kc) gload((kb + kt + 2) * BK);
      compute(Ac, Bc, 2);
      compute(Ac, Bc, 3);
    }
    __syncthreads();
  }
  if constexpr (NG == 2) {
    // group 1's accumulators through its own staging buffers (free since the K loop's last barrier): [wave][register][lane]
    constexpr int NREG = T::NFB * T::NPB * 16;
    static_assert(4 * NREG * 64 * 4 <= T::LDS_BYTES, "accumulator hand-over must fit the group's staging buffers");
    float* const red = smem_all + T::LDS_BYTES / 4 + wave * (NREG * 64);
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

  const auto pre = Epi::template prefetch<T::NFB, FAST>(ea, f0 + wf, lane, g.F);
  if constexpr (Epi::XBUF) {
    // every wave has left the K loop through its last barrier: the staging buffers are free for the epilogue's row transposer
    static_assert(4 * T::NPB * 1024 * 4 <= T::LDS_BYTES, "transposer regions must fit the staging buffers");
    Epi::template apply<T::NFB, T::NPB, FAST>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P, FAST ? smem + wave * (T::NPB * 1024) : nullptr);
  } else {
    Epi::template apply<T::NFB, T::NPB, FAST>(acc, ea, pre, f0 + wf, p0 + wp, lane, g.F, g.P);
  }
}

// ---- the kernel -----------------------------------------------------------------
template <class T, bool AKC, bool BKC, class Epi, bool FAST, int NG = 1>
__global__ __launch_bounds__(NTHREADS * NG, NG == 1 ? 2 : 1) void gemm_kernel(GemmArgs g, typename Epi::Args ea) {
  // split-K (wgrad: the reduction runs over the batch): slice y owns k in [y*kchunk, (y+1)*kchunk) and
  // writes its partial tile to its own slab (Epi::slice moves the output pointer); single K panel only.
  if (g.kchunk > 0) {
    const int kb = blockIdx.y * g.kchunk;
    g.A += AKC ? (size_t)kb : (size_t)kb * g.lda;
    g.B0 += BKC ? (size_t)kb : (size_t)kb * g.ldb0;
    g.K = (g.K - kb < g.kchunk) ? g.K - kb : g.kchunk;
    g.K0 = g.K;
    Epi::slice(ea, blockIdx.y);
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // XCD-aware tile order: blocks b, b+8, b+16 ... share an XCD (and its L2); give them
  // the feature tiles of ONE patient tile so the activation panel is fetched once per XCD.
  const int nft = (g.F + T::BF - 1) / T::BF;
  const int npt = (g.P + T::BP - 1) / T::BP;
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int ft = idx % nft;
  const int pt = (idx / nft) * 8 + (b & 7);
  if (pt >= npt) return;
  if (g.tri && ft > pt) return;          // symmetric product: the mirror tile (pt, ft) carries this one's weight
  gemm_tile<T, AKC, BKC, Epi, FAST, NG>(g, ea, ft * T::BF, pt * T::BP, smem);
}

// element (fb, reg) of a lane's fragment is feature  f_wave + 32*fb + 8*(reg>>2) + 4*h + (reg&3)
// and patient p_wave + 32*pb + (lane&31).
#define OSD_FOR_QUADS(fb, pb, q)                         \
  _Pragma("unroll") for (int fb = 0; fb < NFB; ++fb)     \
  _Pragma("unroll") for (int pb = 0; pb < NPB; ++pb)     \
  _Pragma("unroll") for (int q = 0; q < 4; ++q)

}  // namespace osd
