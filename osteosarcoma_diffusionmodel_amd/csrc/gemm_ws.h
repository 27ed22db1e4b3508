// gemm_ws.h -- wave-specialised variant of the direct-to-LDS forward GEMM (gemm_glds.h) for the sampling-sized
// 128 x 128 tile: ONE workgroup of TWELVE waves per CU, persistent over its tiles.
//
//   waves 0-7   (two per SIMD, each a 64-feature x 32-patient sub-tile) run the K loop of gemm_glds_kernel -- LDS-DMA
//               staging, swizzled fragment reads, 32 MFMAs per K step each -- and never execute an epilogue: after the
//               last K step of a tile they copy their 32 accumulator registers to an LDS hand-off area and start
//               the next tile, whose first K stage was streamed in during that last K step.
//   waves 8-11  (one per SIMD) are the shadows: wave 8+e reads the hand-offs of waves 2e and 2e+1 back into the SAME
//               register layout and runs the unchanged epilogue (Epi::apply) for tile i while the MFMA waves compute
//               tile i+1.
//
// In gemm_glds_kernel the epilogue's VALU / transcendental / store work overlaps MFMAs only when the two co-resident
// workgroups of a CU happen to be out of phase; here it always issues in the shadow of the MFMA waves.
//
// gfx950 has one workgroup barrier and no named barriers, so every wave executes exactly nk barriers per tile: the
// MFMA waves one per K step (the last one also publishes the hand-off), the epilogue waves that publishing one plus
// nk-1 at the sync.tick() points inside Epi::apply (TickSync turns a few ticks into barriers, the rest are drained
// after the epilogue).
//
// Results are bit-identical to gemm_glds_kernel: same K order per accumulator, same epilogue code.
#pragma once
#include "gemm_glds.h"

namespace osd {

// s_barrier alone: no waitcnt, no fence instructions (callers wait for exactly what they need); the empty asm
// statements keep the compiler from moving memory operations across it.
__device__ __forceinline__ void ws_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct TickSync {
  int budget;   // barriers still owed in this phase
  int stride;   // one barrier every `stride` ticks
  int cnt;
  __device__ __forceinline__ void tick() {
    if (++cnt >= stride) {
      cnt = 0;
      if (budget > 0) { --budget; ws_barrier(); }
    }
  }
  __device__ __forceinline__ void drain() {
    while (budget > 0) { --budget; ws_barrier(); }
  }
};

struct WsCfg {
  static constexpr int BF = 128, BP = 128;          // workgroup tile
  static constexpr int NFB = 2, NPB = 1;            // an MFMA wave's sub-tile: 64 features x 32 patients
  static constexpr int N_MFMA = 8, N_EPI = 4;
  static constexpr int THREADS = 64 * (N_MFMA + N_EPI);
  static constexpr int A_ELEMS = BF * BK, B_ELEMS = BP * BK;
  static constexpr int STAGE_FLOATS = 3 * (A_ELEMS + B_ELEMS);             // three stages: the DMA runs two K steps ahead
  static constexpr int WAVE_HAND = NFB * NPB * 16 * 64;                    // floats one MFMA wave hands over
  static constexpr int LDS_BYTES = (STAGE_FLOATS + N_MFMA * WAVE_HAND) * 4;
  static constexpr int TICKS = 2 * NFB * NPB * 4;                          // lower bound on tick() calls per epilogue wave per tile
};

template <class Epi>
__global__ __launch_bounds__(WsCfg::THREADS) void gemm_ws_kernel(GemmArgs g, typename Epi::Args ea) {
  typedef WsCfg W;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hand = smem + W::STAGE_FLOATS;

  const int nft = (g.F + W::BF - 1) / W::BF;
  const int npt = (g.P + W::BP - 1) / W::BP;
  const int nvb = 8 * nft * ((npt + 7) / 8);          // virtual one-tile workgroups of gemm_glds_kernel's XCD-aware order
  const int nk = (g.K + BK - 1) / BK;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // virtual block v -> (feature tile, patient tile); blocks v, v+8, ... of one XCD take the feature tiles of one patient tile
  auto tile_of = [&](int v, int& ft, int& pt) { const int idx = v >> 3; ft = idx % nft; pt = (idx / nft) * 8 + (v & 7); };
  auto next_valid = [&](int v) {
    for (; v < nvb; v += gridDim.x) { int ft, pt; tile_of(v, ft, pt); if (pt < npt) return v; }
    return -1;
  };
  int v = next_valid(blockIdx.x);
  if (v < 0) return;

  if (wave12 < W::N_MFMA) {
    // ================================ MFMA waves ================================
    const int wave = wave12;
    const int wf = (wave >> 2) * 64;                  // waves 0-3: features 0-63, waves 4-7: 64-127
    const int wp = (wave & 3) * 32;
    float* myhand = hand + wave * W::WAVE_HAND;
    // staging: a K tile is 16 + 16 wave-instructions of 8 rows; wave w moves rows 8*(j*8+w) .. +7 of A and of B, j = 0, 1
    int s_row[2], s_k4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (j * 8 + wave) * 8 + (lane >> 3);
      s_row[j] = row;
      s_k4[j] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    struct StageCtx { unsigned la, lb; const float* bbase; int bld, bk, bkend, ak, p0, f0; };
    auto stage_begin = [&](int k0, int f0, int p0, float* As, float* Bs) {
      StageCtx c;
      c.la = __builtin_amdgcn_readfirstlane(lds_addr(As) + (unsigned)wave * 1024u);
      c.lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs) + (unsigned)wave * 1024u);
      const bool first = k0 < g.K0;
      c.bbase = first ? g.B0 : g.B1;
      c.bld = first ? g.ldb0 : g.ldb1;
      c.bk = first ? k0 : k0 - g.K0;
      c.bkend = first ? (g.K0 < g.K ? g.K0 : g.K) : g.K - g.K0;
      c.ak = k0;
      c.p0 = p0;
      c.f0 = f0;
      return c;
    };
    // piece 0, 1: A rows; piece 2, 3: B rows
    auto stage_piece = [&](const StageCtx& c, int j) {
      if (j < 2) {
        int rg = c.f0 + s_row[j];
        rg = rg < g.F ? rg : g.F - 1;
        glds16(g.A + ((size_t)rg * g.lda + s_k4[j] + c.ak), __builtin_amdgcn_readfirstlane(c.la + (unsigned)j * 8192u));
      } else {
        const int jb = j - 2;
        int k = c.bk + s_k4[jb];
        k = k < c.bkend - 4 ? k : c.bkend - 4;
        int rg = c.p0 + s_row[jb];
        rg = rg < g.P ? rg : g.P - 1;
        glds16(c.bbase + (size_t)rg * c.bld + k, __builtin_amdgcn_readfirstlane(c.lb + (unsigned)jb * 8192u));
      }
    };
    int a_rd[W::NFB], a_sw[W::NFB], b_rd, b_sw;
#pragma unroll
    for (int fb = 0; fb < W::NFB; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * BK; a_sw[fb] = h ^ ((R >> 1) & 7); }
    { const int R = wp + l31; b_rd = R * BK; b_sw = h ^ ((R >> 1) & 7); }

    f32x16 acc[W::NFB][W::NPB];
    // quarter i (8 k) of a staged tile: 8 MFMAs, one DMA wave-instruction of the stage being prefetched in their middle
    auto compute = [&](const float* As, const float* Bs, int i, const StageCtx& sc, bool do_stage) {
      float a[W::NFB][4], bb[4];
#pragma unroll
      for (int fb = 0; fb < W::NFB; ++fb) {
        const float4 t = *reinterpret_cast<const float4*>(&As[a_rd[fb] + 4 * (a_sw[fb] ^ (2 * i))]);
        a[fb][0] = t.x; a[fb][1] = t.y; a[fb][2] = t.z; a[fb][3] = t.w;
      }
      {
        const float4 t = *reinterpret_cast<const float4*>(&Bs[b_rd + 4 * (b_sw ^ (2 * i))]);
        bb[0] = t.x; bb[1] = t.y; bb[2] = t.z; bb[3] = t.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int fb = 0; fb < W::NFB; ++fb)
          acc[fb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fb][e], bb[e], acc[fb][0], 0, 0, 0);
        if (do_stage && e == 1) {
          __builtin_amdgcn_sched_barrier(0);
          stage_piece(sc, i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    // All eight MFMA waves meet at one barrier per K step, so nothing covers a wave that waits for its DMA (in
    // gemm_glds_kernel the other co-resident workgroup does): the DMA therefore runs TWO stages ahead in three LDS
    // buffers, and the wait before the barrier is a counted one that leaves the newest stage's loads in flight.
    auto bufA = [&](int b) { return smem + b * W::A_ELEMS; };
    auto bufB = [&](int b) { return smem + 3 * W::A_ELEMS + b * W::B_ELEMS; };
    struct Cursor { int v, ft, pt, kt; };               // the stream of stages: (tile, kt) in order
    auto advance = [&](Cursor c) {
      if (c.v < 0) return c;
      if (c.kt + 1 < nk) { ++c.kt; return c; }
      c.v = next_valid(c.v + gridDim.x);
      c.kt = 0;
      if (c.v >= 0) tile_of(c.v, c.ft, c.pt);
      return c;
    };
    auto issue_all = [&](const Cursor& c, int b) {
      if (c.v < 0) return;
      const StageCtx sc = stage_begin(c.kt * BK, c.ft * W::BF, c.pt * W::BP, bufA(b), bufB(b));
#pragma unroll
      for (int j = 0; j < 4; ++j) stage_piece(sc, j);
    };
    Cursor cur;
    cur.v = v; cur.kt = 0;
    tile_of(v, cur.ft, cur.pt);
    Cursor c1 = advance(cur);
    issue_all(cur, 0);
    issue_all(c1, 1);
    if (c1.v >= 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws_barrier();                                       // barrier 0 of the kernel: the first stage is visible
    unsigned long long ms[4] = {0, 0, 0, 0};
    int ntile = 0;
    if (g.stamps) ms[0] = __builtin_amdgcn_s_memtime();
    int b0 = 0;                                         // LDS buffer of the stage being computed
#pragma unroll
    for (int i = 0; i < W::NFB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    while (true) {
      const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
      const Cursor c2 = advance(c1);                    // the stage to prefetch during this K step
      const bool more = c2.v >= 0;
      const StageCtx sc = stage_begin(more ? c2.kt * BK : 0, c2.ft * W::BF, c2.pt * W::BP, bufA(b2), bufB(b2));
      const float* Ac = bufA(b0);
      const float* Bc = bufB(b0);
      compute(Ac, Bc, 0, sc, more);
      compute(Ac, Bc, 1, sc, more);
      compute(Ac, Bc, 2, sc, more);
      compute(Ac, Bc, 3, sc, more);
      const bool tile_done = cur.kt + 1 == nk;
      if (tile_done) {
        // hand the accumulators to the shadow wave: [fb][quad][lane] float4, conflict-free
#pragma unroll
        for (int fb = 0; fb < W::NFB; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(&myhand[((fb * 4 + q) * 64 + lane) * 4]) =
                make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
#pragma unroll
        for (int i = 0; i < W::NFB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        if (g.stamps) { ++ntile; if (ntile <= 2) ms[ntile] = __builtin_amdgcn_s_memtime(); }
      }
      // stage s+1 (issued one K step ago) has landed; the four loads of stage s+2 may still be in flight
      if (more) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      ws_barrier();
      if (c1.v < 0) break;
      cur = c1; c1 = c2; b0 = b1;
    }
    if (g.stamps && wave == 0) {
      ms[3] = __builtin_amdgcn_s_memtime();
      if (lane == 0) { unsigned long long* o = g.stamps + (size_t)blockIdx.x * 16; o[0] = ms[0]; o[1] = ms[1]; o[2] = ms[2]; o[3] = ms[3]; }
    }
  } else {
    // ================================ epilogue waves ================================
    const int ew = wave12 - W::N_MFMA;                  // shadows MFMA waves 2*ew and 2*ew + 1
    const int wf = ((2 * ew) >> 2) * 64;                // both have the same feature range
    f32x16 acc[W::NFB][W::NPB];
    typedef decltype(Epi::template prefetch<W::NFB, true>(ea, 0, 0, 0)) PreT;
    auto pre_of = [&](int vt) {
      int ft, pt;
      tile_of(vt < 0 ? 0 : vt, ft, pt);
      return Epi::template prefetch<W::NFB, true>(ea, ft * W::BF + wf, lane, g.F);
    };
    auto finish = [&](int vt, const PreT& pre, TickSync& sync) {
      int ft, pt;
      tile_of(vt, ft, pt);
      const int f0 = ft * W::BF, p0 = pt * W::BP;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int mw = 2 * ew + s;
        const float* src = hand + mw * W::WAVE_HAND;
#pragma unroll
        for (int fb = 0; fb < W::NFB; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&src[((fb * 4 + q) * 64 + lane) * 4]);
            acc[fb][0][4 * q] = t.x; acc[fb][0][4 * q + 1] = t.y; acc[fb][0][4 * q + 2] = t.z; acc[fb][0][4 * q + 3] = t.w;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // in registers before this wave's next barrier
        Epi::template apply<W::NFB, W::NPB, true>(acc, ea, pre, f0 + wf, p0 + (mw & 3) * 32, lane, g.F, g.P, sync);
      }
    };
    ws_barrier();                                       // barrier 0 of the kernel
    int vprev = -1;
    // barriers placed INSIDE an epilogue: enough that no slice outlasts a K step, few enough that the second hand-off
    // is read well before the MFMA waves overwrite it; the rest of the nk-1 are drained afterwards
    int inside = nk - 3 < Epi::WS_SLICES ? nk - 3 : Epi::WS_SLICES;
    inside = inside < 1 ? 1 : inside;
    PreT pre = pre_of(v);
    bool stamped = false;
    while (true) {
      // phase: the MFMA waves run the nk K steps of tile v (the last barrier publishes its hand-off); this wave finishes vprev
      TickSync sync;
      sync.budget = v >= 0 ? inside : 0;
      sync.stride = (W::TICKS + inside - 1) / inside;
      sync.cnt = 0;
      unsigned long long t0 = 0, t1 = 0;
      const bool stamp = g.stamps && ew == 0 && vprev >= 0 && !stamped;
      if (stamp) t0 = __builtin_amdgcn_s_memtime();
      if (vprev >= 0) finish(vprev, pre, sync);
      if (stamp) t1 = __builtin_amdgcn_s_memtime();
      if (v < 0) break;                                 // that was the last tile's epilogue
      sync.budget += nk - 1 - inside;
      sync.drain();
      if (stamp) {
        stamped = true;
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { unsigned long long* o = g.stamps + (size_t)blockIdx.x * 16 + 4; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = (unsigned long long)sync.cnt; }
      }
      pre = pre_of(v);                                  // per-feature parameters of tile v: in flight across the barrier
      ws_barrier();                                     // publishes tile v's hand-off
      vprev = v;
      v = next_valid(v + gridDim.x);
    }
  }
}

}  // namespace osd
