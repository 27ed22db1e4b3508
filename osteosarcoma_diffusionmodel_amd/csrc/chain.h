// chain.h -- the persistent reverse-chain kernel: all layers of the denoiser and all T steps in ONE launch.
//
// models/diffusion.py:427-449 runs T sequential p_sample steps over rows that never interact (GroupNorm is per
// row).  The per-layer kernels of gemm_glds.h pay, 12 times per step, a launch boundary, a wave of workgroups that all
// sit in their prologue (then their epilogue) at the same moment, and a partial last round.  Here the unit of work is
// (128-row tile, one step): a workgroup carries its tile through input_proj, the ten Linear+GroupNorm+SiLU layers and
// output_proj + posterior update with the same MFMA tile loop and the same epilogue arithmetic as the per-layer kernels
// (the two engines agree bitwise); the
// activations of the tile live in a workspace PRIVATE to the workgroup's slot (written and re-read by one CU: L2 /
// Infinity-Cache traffic, never another workgroup's business), the chain state x stays in the caller's [n][D] tensor.
//
// Units are taken from ONE atomic counter in global order, unit u = (tile u % n_tiles, step u / n_tiles): whoever is free takes
// the next one (the two workgroups of a CU do not run at the same speed: the SIMD arbiter favours the older waves).  A unit needs
// x_t of its tile, written by the unit n_tiles earlier (another workgroup, possibly another XCD): the producer publishes with an
// agent-scope release behind a drained workgroup barrier and a relaxed agent store of progress[tile]; the consumer polls that
// one word relaxed from one wave (wave-uniform control flow), then ONE agent-scope acquire, s_waitcnt, workgroup barrier,
// plain loads (cdna_hip_programming.md Guideline 16).  Every dependency points to a unit that was taken earlier by a
// workgroup that is running, so there is no cycle and no wait on a workgroup that is not resident; with n_tiles >= G the
// producer is a full round ahead and the poll normally passes at once (measured: 0.13 % of a workgroup's cycles).  Every spin
// is bounded by a wall-clock budget: on expiry the workgroup raises status[0] and leaves; every other workgroup sees the flag
// at its next unit (or in its own spin) and leaves too.
//
// Epilogues (xpose.h, below): per-feature parameters arrive in LDS by DMA with the tile's first weight stage; outputs leave and
// x_t / cond_proj arrive as full 128-byte row segments through a per-wave LDS transposer in the idle second operand buffer;
// global accesses are explicit (the argument block lives in memory, so its pointers are generic to hipcc).
//
// `stagger` (the second workgroup to arrive on a CU waits that many cycles once) is kept as a knob: 0 ... 120 000 cycles all
// measure within +-0.3 % -- the phases of the two workgroups drift anyway (DESIGN.md section 3.2 for what does and does not
// move this kernel: per SIMD its time is MFMA cycles + VALU cycles + the stalls that hit both workgroups at once).
#pragma once
#include "gemm_glds.h"
#include "epilogues.h"
#include "xpose.h"
#include "launch.h"

namespace osd {

constexpr int CHAIN_MAX_LAYERS = 24;
enum : int { CK_INPUT = 0, CK_GN32 = 1, CK_GN64 = 2, CK_POST = 3 };
enum : unsigned { CHAIN_OK = 0, CHAIN_TIMEOUT = 1, CHAIN_ABORT = 2 };     // ABORT: written by the host (wall-clock budget)

struct ChainLayer {
  const float* A; int lda;        // weights [F][K], K contiguous, readable and zero for k in [K, roundup(K, 32))
  int K, K0;                      // reduction length; first-panel width (K0 >= K: one panel)
  int F;                          // output features
  int in0, ld0, in1, ld1;         // input panels: float offsets into the slot workspace (in0 < 0: the chain state x)
  int out, ldo;                   // output: float offset into the slot workspace (unused by CK_POST)
  int kind;
  const float* bias; const float* gamma; const float* beta;
};

struct ChainArgs {
  ChainLayer L[CHAIN_MAX_LAYERS];
  int n_layers;
  float* ws; long long ws_stride;            // slot s owns ws[s * ws_stride, (s + 1) * ws_stride)
  float* x; int D;                           // chain state [n][D], in place
  int n, n_tiles;
  int t_first, n_steps;                      // this launch runs t = t_first, t_first - 1, ..., t_first - n_steps + 1
  unsigned base_done;                        // steps of the chain completed before this launch
  const float* cproj; int ldc;               // [n][H0]  cond_proj(condition_embed(c)), loop-invariant
  const float* temb; int ldt;                // [T][H0]  time_proj(TimeEmbedding(t / T))
  const float* coef;                         // [T][4]   (A_t, B_t, C_t, 0)
  const float* z; int ldzz; long long z_step_stride; int z_t_first;      // injected draws or null -> Philox
  uint64_t seed; uint32_t row_offset;
  float* mut_mask; int mutation_dim;
  unsigned* progress;                        // [n_tiles] steps completed per tile (monotonic over the launches of a chain)
  unsigned* status;                          // [0] CHAIN_OK / CHAIN_TIMEOUT
  unsigned* queue;                           // [0] next unit of this launch (zeroed per launch): workgroups take units in global order
  unsigned* cu_arrivals;                     // [2048] zeroed per launch; null = no stagger
  int stagger;                               // shader cycles the second workgroup of a CU waits before its first unit
  unsigned long long spin_budget;            // s_memrealtime ticks (100 MHz) a dependency wait may take
  unsigned long long* stamps;                // diagnostic (null in production): per workgroup 32 counters of shader cycles --
                                             // dependency wait, tile prologue, K loop, epilogue + drain, whole kernel, units run,
                                             // placement x 2, then 6 per layer kind (see kcnt in the kernel)
};

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ unsigned ld_relaxed_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_relaxed_agent(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0 waits until *word >= want (or the chain failed elsewhere, or the budget ran out); false on failure.  Executed by
// the WHOLE wave with wave-uniform control flow: every lane loads the same word and the decision is taken on lane 0's copy
// (readfirstlane), so no branch here depends on a lane id.  A lane-divergent `if (tid == 0) { poll ... }` next to workgroup
// barriers lets hipcc lay the two lane groups of wave 0 on different paths to the same s_barrier (seen: the non-leader
// lanes' path ran the barrier on its own and the workgroup deadlocked).
__device__ __forceinline__ bool chain_wait(const unsigned* word, unsigned want, unsigned* status, unsigned long long budget, int lane) {
  if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(word)) >= want) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(32);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(word)) >= want) return true;
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(status)) != CHAIN_OK) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
      if (lane == 0) st_relaxed_agent(status, CHAIN_TIMEOUT);
      return false;
    }
  }
}

// ---- epilogue plumbing of the chain kernel: the LDS row transposer of xpose.h (outputs leave, and x_t / cond_proj arrive,
// as full 128-byte row segments) and per-feature parameters staged in LDS --------------------------------------------------
// Per-feature parameters of a tile in LDS: prm[0..127] bias, [128..255] gamma (input_proj: the time-embedding row), [256..383]
// beta, for features f0 .. f0 + 127.  They arrive by DMA together with the tile's first weight stage (issued under the
// previous tile's epilogue), so no epilogue waits for a parameter load from L2 (stamps: two round trips of ~4 500 cycles each).
constexpr int CHAIN_PRM_FLOATS = 512;

// Where an epilogue's outputs go / its row-wise side input comes from.  The workspace chain (this file) turns every 32-feature
// block through the wave's LDS transposer and moves full row segments to / from global memory; the LDS-resident chain
// (chain_panel.h) writes the fragments straight into the next layer's operand panel.  The arithmetic around them is shared.
template <int NPB>
struct XposeRowsOut {
  const WaveXpose<NPB>& xp; float* out; int ldo;
  __device__ __forceinline__ void put(int fb, int pb, int q, int l31, int h, float4 v) const { (void)fb; xp.put(pb, q, l31, h, v); }
  __device__ __forceinline__ void flush(int fb, int lane) const { xp.template store_rows<false>(out + 32 * fb, ldo, lane, 0, 0); }
};
template <int NPB>
struct XposeRowsIn {
  const WaveXpose<NPB>& xp; const float* src; int ld;
  __device__ __forceinline__ void load(int fb, int lane) const { xp.template load_rows<false>(src + 32 * fb, ld, lane, 0, 0); }
  __device__ __forceinline__ float4 get(int fb, int pb, int q, int l31, int h) const { (void)fb; return xp.get(pb, q, l31, h); }
};

// Linear -> GroupNorm(8) -> SiLU epilogue on a full private tile: the arithmetic of EpiGnSilu<GW, false>::apply, operation
// for operation (the two engines agree bitwise).  `fl` = first feature of the wave inside the tile, `out` = row 0 of the
// wave's rows at feature f0 + fl of the output buffer (all 128 rows exist: no guards).
// PS = floats between the bias | gamma | beta arrays of the parameter block.
template <int GW, int NFB, int NPB, int PS, class Out>
__device__ __forceinline__ void chain_gn_silu(f32x16 (&acc)[NFB][NPB], const float* __restrict__ prm, int fl, const Out& o, int lane,
                                              unsigned long long* ts = nullptr) {
  static_assert(GW >= 8 && NFB * 32 >= GW, "wave must own whole groups");
  constexpr int RPG = GW / 2;                 // registers of one group in this lane
  constexpr int NG = NFB * 16 / RPG;
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(prm + fl + 32 * fb + 8 * q + 4 * h);
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        acc[fb][pb][4 * q] += bv.x; acc[fb][pb][4 * q + 1] += bv.y;
        acc[fb][pb][4 * q + 2] += bv.z; acc[fb][pb][4 * q + 3] += bv.w;
      }
    }
  if (ts) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[0] = __builtin_amdgcn_s_memtime(); }     // diagnostic builds only
  float mean[NPB][NG], rstd[NPB][NG];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < RPG; ++j) { const int Li = g * RPG + j; s += acc[Li / 16][pb][Li % 16]; }
      s += swap_halves(s);
      const float m = s * (1.0f / GW);
      float qs = 0.f;
#pragma unroll
      for (int j = 0; j < RPG; ++j) { const int Li = g * RPG + j; const float d = acc[Li / 16][pb][Li % 16] - m; qs = fmaf(d, d, qs); }
      qs += swap_halves(qs);
      mean[pb][g] = m;
      rstd[pb][g] = 1.0f / sqrtf(qs * (1.0f / GW) + GN_EPS);
    }
  if (ts) { asm volatile("s_nop 0" ::: "memory"); ts[1] = __builtin_amdgcn_s_memtime(); ts[2] = ts[1]; }
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = fl + 32 * fb + 8 * q + 4 * h;
      const float4 gv = *reinterpret_cast<const float4*>(prm + PS + f);
      const float4 bev = *reinterpret_cast<const float4*>(prm + 2 * PS + f);
      const int g = (fb * 16 + 4 * q) / RPG;
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const float m = mean[pb][g], r = rstd[pb][g];
        float4 y;
        y.x = silu_f(fmaf((acc[fb][pb][4 * q] - m) * r, gv.x, bev.x));
        y.y = silu_f(fmaf((acc[fb][pb][4 * q + 1] - m) * r, gv.y, bev.y));
        y.z = silu_f(fmaf((acc[fb][pb][4 * q + 2] - m) * r, gv.z, bev.z));
        y.w = silu_f(fmaf((acc[fb][pb][4 * q + 3] - m) * r, gv.w, bev.w));
        o.put(fb, pb, q, l31, h, y);
      }
    }
    o.flush(fb, lane);
  }
}

// input_proj epilogue, EpiInput::apply's arithmetic: h = ((acc + b) + t_emb[t]) + c_proj.  The time-embedding row segment sits
// in the gamma slot of the parameter block; cond_proj comes in (and h goes out) as full row segments through the transposer.
// Rows beyond the valid ones of a partial tile hold finite copies (the host pads cproj to whole tiles) and stay private.
template <int NFB, int NPB, int PS, class In, class Out>
__device__ __forceinline__ void chain_input(f32x16 (&acc)[NFB][NPB], const float* __restrict__ prm, int fl, const In& in, const Out& o, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    in.load(fb, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = fl + 32 * fb + 8 * q + 4 * h;
      const float4 bv = *reinterpret_cast<const float4*>(prm + f);
      const float4 tv = *reinterpret_cast<const float4*>(prm + PS + f);
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const float4 cv = in.get(fb, pb, q, l31, h);   // a lane reads and rewrites only its own fragment slots
        float4 v;
        v.x = ((acc[fb][pb][4 * q] + bv.x) + tv.x) + cv.x;
        v.y = ((acc[fb][pb][4 * q + 1] + bv.y) + tv.y) + cv.y;
        v.z = ((acc[fb][pb][4 * q + 2] + bv.z) + tv.z) + cv.z;
        v.w = ((acc[fb][pb][4 * q + 3] + bv.w) + tv.w) + cv.w;
        o.put(fb, pb, q, l31, h, v);
      }
    }
    o.flush(fb, lane);
  }
}

// output_proj + DDPM posterior update, EpiPosterior::apply's arithmetic: x' = A_t x + B_t (acc + b) + C_t z.  `x` = the wave's
// first row at feature f0 + fl of the chain state (read and written in place: every element is read before this wave
// overwrites it and no other wave touches it); prow / pcol = valid rows / features from there (<= 0: nothing to do).
#ifndef OSD_EXP
#define OSD_EXP 0      // timing experiments only (make CXXFLAGS+=-DOSD_EXP=n, garbage results): 1 no Philox / Box-Muller, 2 no x_t loads, 4 no stores
#endif
// PRE (chain_panel.h): the caller has already requested block 0's x_t rows into *pre (WaveXpose::issue_rows, same guards); every
// block then requests the next one's rows before it computes, so no block waits for a global round trip.  Data movement only.
template <int NFB, int NPB, bool PRE = false, bool GUARD = true>
__device__ __forceinline__ void chain_posterior(f32x16 (&acc)[NFB][NPB], const float* __restrict__ prm, int fl, float* __restrict__ x, int ldx,
                                                int prow, int pcol, float cA, float cB, float cC, int t, const float* __restrict__ zrow, int ldzz,
                                                uint64_t seed, uint32_t row_id0, int f_glob, float* __restrict__ mut_mask, int mutation_dim,
                                                const WaveXpose<NPB>& xp, int lane, float4 (*pre)[4 * NPB] = nullptr) {
  const int l31 = lane & 31, h = lane >> 5;
  // GUARD = false: every row and every feature of the wave's block exists (chain.h measured a guard-free variant of its own epilogue:
  // +0.3 % for 4 spills and 14 KB of code, so it always guards; chain_panel.h picks per pass)
  if (prow <= 0) return;                      // uniform over the wave
  const bool do_mask = t == 0 && mut_mask != nullptr;      // uniform
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    const int cols = pcol - 32 * fb;          // valid features of this block
    if (GUARD && cols <= 0) break;            // uniform
    // x_t of the block in row segments; a lane then reads and rewrites only its own fragment slots, so each get() can sit
    // right before its use (no 32-register copy of the block)
    if constexpr (PRE) {
      if (!(OSD_EXP & 2)) {
        xp.commit_rows(*pre, lane);
        if (fb + 1 < NFB && cols - 32 > 0) xp.template issue_rows<GUARD>(*pre, x + 32 * (fb + 1), ldx, lane, prow, cols - 32);
      }
    } else {
      xp.template load_rows<GUARD>(x + 32 * fb, ldx, lane, prow, cols);
    }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int p = 32 * pb + l31;            // row inside the wave's rows
      const int pc = (!GUARD || p < prow) ? p : prow - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int fo = 32 * fb + 8 * q + 4 * h;           // feature offset from the wave's first
        const float4 bv = *reinterpret_cast<const float4*>(prm + fl + fo);
        const float e[4] = {acc[fb][pb][4 * q] + bv.x, acc[fb][pb][4 * q + 1] + bv.y, acc[fb][pb][4 * q + 2] + bv.z, acc[fb][pb][4 * q + 3] + bv.w};
        const float4 xq = xp.get(pb, q, l31, h);
        const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
        float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t > 0) {
          if (zrow) {
            const int fc = (!GUARD || fo < pcol - 4) ? fo : pcol - 4;
            zz = ldg4(zrow + (size_t)pc * ldzz + fc);
          } else if (!(OSD_EXP & 1)) {
            zz = randn4(seed, row_id0 + (uint32_t)p, (uint32_t)((f_glob + fo) >> 2), (uint32_t)t, TAG_POSTERIOR);
          }
        }
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaf(cA, xv[r], fmaf(cB, e[r], cC * zv[r]));
        if (do_mask && p < prow && fo < pcol && f_glob + fo < mutation_dim) {
          float* mrow = mut_mask + (size_t)p * mutation_dim;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f_glob + fo + r < mutation_dim) stg1(mrow + f_glob + fo + r, (o[r] > 0.5f) ? 1.0f : 0.0f);
        }
        xp.put(pb, q, l31, h, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
    if (!(OSD_EXP & 4)) xp.template store_rows<GUARD>(x + 32 * fb, ldx, lane, prow, cols);
  }
}

#ifndef CHAIN_EPI_PRIO
#define CHAIN_EPI_PRIO 0
#endif
#ifndef CHAIN_BP
#define CHAIN_BP 128
#endif
#if CHAIN_BP == 128
typedef Tile<128, 128, 64, 64> ChainTile;            // 4 waves, 64 x 64 accumulators each (64 VGPRs), 2 waves per SIMD
constexpr int CHAIN_WPS = 2;                         // workgroups per CU = waves per SIMD
#else
// make CXXFLAGS+=-DCHAIN_BP=64: 64-row tiles, 64 x 32 per wave (167 VGPRs, 52 KB of LDS), THREE workgroups per CU.  Bit-identical;
// measured 22.25 k against 23.17 k patients/s for the default on the same box: half the MFMAs per tile against the same
// per-tile latencies, twice the weight traffic per row.  Kept as a build option for that experiment only.
typedef Tile<128, 64, 64, 32> ChainTile;
constexpr int CHAIN_WPS = 3;
#endif
// tiles | flag word (16 B) | 40 per-kind counters of the diagnostic builds | two per-feature parameter blocks (double-buffered over tiles)
constexpr int CHAIN_PRM_OFF = 2 * (ChainTile::BF * BK + ChainTile::BP * BK) + 4 + 80;      // floats
constexpr int CHAIN_LDS_BYTES = (CHAIN_PRM_OFF + 2 * CHAIN_PRM_FLOATS) * 4;
static_assert(CHAIN_PRM_OFF % 4 == 0, "parameter blocks are read as float4");

// STAMP (diagnostic builds only, make DIAG=1): per-workgroup cycle counters; the product kernel carries none of that state.
template <bool STAMP>
__global__ __launch_bounds__(NTHREADS, CHAIN_WPS) void chain_kernel(const ChainArgs* __restrict__ gp) {
  // The argument block lives in device memory (uniform scalar loads).  Passed by value, hipcc hoists the loads of all ~60
  // fields to the kernel entry and keeps them in SGPRs for the whole kernel; the spills of that end up in VGPRs.  Each phase
  // therefore re-derives its pointer to the block through an empty asm, so a field is loaded where it is used.
  const ChainArgs& a = *gp;
  typedef ChainTile T;
  typedef GldsTile<T> G;
  static_assert(T::BF == 128 && T::BP % 32 == 0, "staging below assumes a 128-row weight image");
  constexpr int NW = NTHREADS / 64;         // waves
  constexpr int NPW = 16 / NW;              // 1 KiB DMA pieces (8 rows x 32 k) per wave of the weight image: wave w moves pieces w, w + NW, ...
  constexpr int NPWB = T::BP / 8 / NW;      // ... and of the activation image
  constexpr int PROWS = 8 * NW;             // rows between two pieces of a wave
  constexpr unsigned PBYTES = 1024u * NW;   // LDS bytes between them
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As0 = smem;
  float* As1 = smem + G::A_ELEMS;
  float* Bs0 = smem + 2 * G::A_ELEMS;
  float* Bs1 = smem + 2 * G::A_ELEMS + G::B_ELEMS;
  // leader -> workgroup: 1 = go on, 0 = leave.  In the dynamic region behind the tiles: a static __shared__ would shift
  // the dynamic base off its 16-byte alignment (ds_read_b128 replays; Guideline 17)
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + 2 * (G::A_ELEMS + G::B_ELEMS));   // also carries the unit number

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = (wave / T::NWP) * T::WF;
  const int wp = (wave % T::NWP) * T::WP;

  float* const ws = a.ws + (long long)blockIdx.x * a.ws_stride;
  // epilogue transposer of this wave: its share of the K loop's second operand buffers (idle between a tile's last K step and
  // the next tile's first), 32 * NPB rows x 32 floats
  WaveXpose<T::NPB> xp;
  {
    const int xo = wave * (T::NPB * 1024);
    xp.buf = xo < G::A_ELEMS ? As1 + xo : Bs1 + (xo - G::A_ELEMS);
  }
  float* const prm_base = smem + CHAIN_PRM_OFF;
  int pcur = 0;                               // parameter block of the tile being computed (uniform)

  // ---- stagger: the second workgroup to arrive on a CU starts `stagger` cycles late, once ----
  if (a.cu_arrivals && a.stagger > 0) {
    if (wave == 0) {                      // wave-uniform branch; only the atomic itself is under a lane mask
      const unsigned hw = __builtin_amdgcn_s_getreg(0xF804);      // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
      const unsigned xcc = __builtin_amdgcn_s_getreg(0xF814) & 7; // HW_REG_XCC_ID
      const unsigned key = (xcc << 8) | ((hw >> 8) & 0xFF);
      unsigned r = 0;
      if (lane == 0) r = atomicAdd(a.cu_arrivals + key, 1u);
      s_flag = __builtin_amdgcn_readfirstlane((int)(r & 1u));     // every lane stores lane 0's value
    }
    __syncthreads();
    const int late = __builtin_amdgcn_readfirstlane(s_flag);
    __syncthreads();
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(16);
    }
  }

  unsigned long long c_dep = 0, c_pro = 0, c_k = 0, c_epi = 0, c_units = 0;
  // STAMP: per layer kind {K loop, epilogue until its last store is issued, store drain, barrier, layer-boundary first stage, tiles}
  unsigned long long* const kcnt = reinterpret_cast<unsigned long long*>(smem + 2 * (G::A_ELEMS + G::B_ELEMS) + 4);
  if constexpr (STAMP) {
    if (tid < 40) kcnt[tid] = 0;
    __syncthreads();
  }
  const unsigned long long c_start = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  const long long n_units = (long long)a.n_tiles * a.n_steps;
  int c_iter = 0;
  (void)c_iter;
  for (;;) {
    const unsigned long long td0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    // next unit, in global order: whoever is free takes it (a slow workgroup simply takes fewer), and every dependency of a
    // unit was taken earlier by a workgroup that is running, so no wait can depend on a workgroup that is not resident
#ifdef CHAIN_STATIC_UNITS
    static_assert(true, "");
    const long long u = (long long)blockIdx.x + (long long)c_iter * gridDim.x;
    ++c_iter;
#else
    if (wave == 0) {
      unsigned nu = 0;
      if (lane == 0) nu = atomicAdd(a.queue, 1u);
      s_flag = __builtin_amdgcn_readfirstlane((int)nu);
    }
    __syncthreads();
    const long long u = (unsigned)__builtin_amdgcn_readfirstlane(s_flag);      // wave-uniform by construction: keep it in SGPRs
    __syncthreads();
#endif
    if (u >= n_units) break;
    const int tile = (int)(u % a.n_tiles);
    const int si = (int)(u / a.n_tiles);
    const int t = a.t_first - si;
    const int p0 = tile * T::BP;
    const int P = (a.n - p0 < T::BP) ? a.n - p0 : T::BP;      // valid rows of this tile

    // ---- dependency: x_t of this tile (written by the unit n_tiles earlier, another workgroup) ----
    if (wave == 0) {
      bool ok = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(a.status)) == CHAIN_OK;
      if (ok && si > 0) {
        ok = chain_wait(a.progress + tile, a.base_done + (unsigned)si, a.status, a.spin_budget, lane);
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 after the match
      }
      s_flag = ok ? 1 : 0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the barrier must not release before the invalidate is done
    }
    __syncthreads();
    const int go = __builtin_amdgcn_readfirstlane(s_flag);
    if (!go) return;                      // uniform over the workgroup: every wave leaves
    if constexpr (STAMP) { c_dep += __builtin_amdgcn_s_memtime() - td0; ++c_units; }

    // First K stage of a tile (k in [0, 32)) into LDS buffer 0: the A part (weights: no dependency, so it is issued BEFORE the
    // previous tile's epilogue and lands under it) and the B part (activations: behind the producing layer's drained stores).
    // With the A part go the tile's per-feature parameters, into parameter block `pb`: wave 0 moves bias | second array (gamma,
    // the time-embedding row of input_proj, or bias again), wave 1 beta; 32 lanes x 16 B per array.
    auto first_stage = [&](const ChainLayer& Lr, int f0n, bool do_a, bool do_b, int pb) {
      struct { const float* A; int lda, F, in0, ld0, K0, K; } Ln{Lr.A, Lr.lda, Lr.F, Lr.in0, Lr.ld0, Lr.K0, Lr.K};   // values, not re-loads
      if (do_a && wave < 2) {
        const int kind = Lr.kind;
        const float* src = Lr.bias;
        if (wave == 1) src = Lr.beta;
        else if (lane >= 32) src = kind == CK_INPUT ? a.temb + (size_t)t * a.ldt : (kind == CK_POST ? Lr.bias : Lr.gamma);
        if (wave == 0 || (kind != CK_INPUT && kind != CK_POST)) {       // uniform
          int f = f0n + 4 * (lane & 31);
          f = f < Ln.F - 4 ? f : Ln.F - 4;
          glds16(src + f, __builtin_amdgcn_readfirstlane(lds_addr(prm_base + pb * CHAIN_PRM_FLOATS) + (unsigned)wave * 1024u));
        }
      }
      const float* const xrows = a.x + (size_t)p0 * a.D;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int r0 = wave * 8 + (ln >> 3);                        // piece j moves rows r0 + PROWS j
      const int k4 = 4 * ((ln & 7) ^ ((r0 >> 1) & 7));            // (row >> 1) & 7 does not depend on j (PROWS % 16 == 0)
      const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(As0) + (unsigned)wave * 1024u);
      const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs0) + (unsigned)wave * 1024u);
      if (do_a) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          int rg = f0n + r0 + PROWS * j;
          rg = rg < Ln.F ? rg : Ln.F - 1;
          glds16(Ln.A + (size_t)rg * Ln.lda + k4, __builtin_amdgcn_readfirstlane(la + (unsigned)j * PBYTES));
        }
      }
      if (do_b) {
        const bool from_x = Ln.in0 < 0;
        const float* bb = from_x ? xrows : ws + Ln.in0;
        const int rows = from_x ? P : T::BP;
        const int kend = Ln.K0 < Ln.K ? Ln.K0 : Ln.K;
        const int k = k4 < kend - 4 ? k4 : kend - 4;
#pragma unroll
        for (int j = 0; j < NPWB; ++j) {
          int rg = r0 + PROWS * j;
          rg = rg < rows ? rg : rows - 1;
          glds16(bb + (size_t)rg * Ln.ld0 + k, __builtin_amdgcn_readfirstlane(lb + (unsigned)j * PBYTES));
        }
      }
    };
    unsigned long long tt0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    first_stage(a.L[0], 0, true, true, pcur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the asm DMAs are invisible to hipcc's counters
    __syncthreads();

    for (int l = 0; l < a.n_layers; ++l) {
      const ChainLayer& L = a.L[l];
      // the K loop's operands as plain values: the DMA asm statements clobber "memory", and a field read through the argument
      // block after one of them would be re-loaded from device memory every time (measured: the K loop ran at half speed)
      const gfloat_ptr LA = uniform_ptr(L.A);
      const int Llda = L.lda, LK0 = L.K0, Lld0 = L.ld0, Lld1 = L.ld1;
      const int F = L.F, K = L.K;
      const float* const B0g = (L.in0 < 0) ? a.x + (size_t)p0 * a.D : ws + L.in0;      // generic copies for the K-tail path
      const float* const B1g = ws + L.in1;
      const gfloat_ptr B0 = uniform_ptr(B0g), B1 = uniform_ptr(B1g);
      const int rowsB = (L.in0 < 0) ? P : T::BP;                // the workspace always holds a full tile
      const int nk = (K + BK - 1) / BK;
      const int nft = (F + T::BF - 1) / T::BF;
      for (int ft = 0; ft < nft; ++ft) {
        const int f0 = ft * T::BF;
        // staging / fragment geometry, re-derived per tile from an opaque copy of the lane id: a dozen VALU instructions, and
        // the registers are free during the epilogue instead of being carried through the whole kernel
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int l31 = ln & 31, h = ln >> 5;
        // piece j of a wave moves rows 8 * (NW j + wave) .. + 7, 16 B per lane
        int st_row[NPW], st_k4[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          st_row[j] = (j * NW + wave) * 8 + (ln >> 3);
          st_k4[j] = 4 * ((ln & 7) ^ ((st_row[j] >> 1) & 7));
        }
        int a_rd[T::NFB], a_sw[T::NFB], b_rd[T::NPB], b_sw[T::NPB];
#pragma unroll
        for (int fb = 0; fb < T::NFB; ++fb) { const int R = wf + 32 * fb + l31; a_rd[fb] = R * BK; a_sw[fb] = h ^ ((R >> 1) & 7); }
#pragma unroll
        for (int pb = 0; pb < T::NPB; ++pb) { const int R = wp + 32 * pb + l31; b_rd[pb] = R * BK; b_sw[pb] = h ^ ((R >> 1) & 7); }
        // ---- staging: direct global -> LDS DMA, XOR-swizzled 16-byte chunks (gemm_glds.h).  Per-lane byte offsets of the four
        // pieces are fixed for the tile (row clamps folded in); a K step only moves the wave-uniform bases (scalar adds) ----
        unsigned offA[NPW], offB[NPWB];
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          int rg = f0 + st_row[j];
          rg = rg < F ? rg : F - 1;
          offA[j] = (unsigned)(rg * Llda + st_k4[j]) * 4u;
        }
#pragma unroll
        for (int j = 0; j < NPWB; ++j) {
          int rb = st_row[j];
          rb = rb < rowsB ? rb : rowsB - 1;
          offB[j] = (unsigned)(rb * Lld0 + st_k4[j]) * 4u;
        }
        int b_panel = 0;                          // which input panel offB was built for
        auto stage = [&](int k0, float* As, float* Bs, int j) {
          const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr(As) + (unsigned)wave * 1024u);
          const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr(Bs) + (unsigned)wave * 1024u);
          if (j < NPW) {
            glds16s(LA + __builtin_amdgcn_readfirstlane(k0), offA[j], __builtin_amdgcn_readfirstlane(la + (unsigned)j * PBYTES));
          } else {
            const int jb = j - NPW;
            const bool first = k0 < LK0;            // uniform: K0 is a multiple of BK (or >= K)
            const int kend = first ? (LK0 < K ? LK0 : K) : K - LK0;
            const int kl = first ? k0 : k0 - LK0;   // k inside the panel
            if (kl + BK <= kend) {
              glds16s((first ? B0 : B1) + __builtin_amdgcn_readfirstlane(kl), offB[jb], __builtin_amdgcn_readfirstlane(lb + (unsigned)jb * PBYTES));
            } else {
              // K tail of a panel whose width is not a multiple of 32 (input_proj, K = D): per-lane clamp to valid floats; the
              // weights are zero there, so the re-read values do not matter as long as they are finite
              const float* bb = first ? B0g : B1g;
              const int ld = first ? Lld0 : Lld1;
              int k = kl + st_k4[jb];
              k = k < kend - 4 ? k : kend - 4;
              int rg = st_row[jb];
              rg = rg < rowsB ? rg : rowsB - 1;
              glds16(bb + (size_t)rg * ld + k, __builtin_amdgcn_readfirstlane(lb + (unsigned)jb * PBYTES));
            }
          }
        };
        // the B offsets follow the panel: re-derived (uniform branch, four multiplies) when the K loop crosses K0
        auto b_offsets_for = [&](int k0) {
          const int want = k0 < LK0 ? 0 : 1;
          if (want != b_panel) {
            b_panel = want;
            const int ld = want ? Lld1 : Lld0;
#pragma unroll
            for (int j = 0; j < NPWB; ++j) {
              int rb = st_row[j];
              rb = rb < rowsB ? rb : rowsB - 1;
              offB[j] = (unsigned)(rb * ld + st_k4[j]) * 4u;
            }
          }
        };
        const unsigned long long tt1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;     // buffer 0 holds this tile's first K stage

        f32x16 acc[T::NFB][T::NPB];
#pragma unroll
        for (int i = 0; i < T::NFB; ++i)
#pragma unroll
          for (int j = 0; j < T::NPB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < nk; ++kt) {
          const float* Ac = (kt & 1) ? As1 : As0;
          const float* Bc = (kt & 1) ? Bs1 : Bs0;
          float* An = (kt & 1) ? As0 : As1;
          float* Bn = (kt & 1) ? Bs0 : Bs1;
          const bool more = kt + 1 < nk;
          const int kn = (kt + 1) * BK;
          if (more) b_offsets_for(kn);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float av[T::NFB][4], bv[T::NPB][4];
#pragma unroll
            for (int fb = 0; fb < T::NFB; ++fb) {
              const float4 tq = *reinterpret_cast<const float4*>(&Ac[a_rd[fb] + 4 * (a_sw[fb] ^ (2 * i))]);
              av[fb][0] = tq.x; av[fb][1] = tq.y; av[fb][2] = tq.z; av[fb][3] = tq.w;
            }
#pragma unroll
            for (int pb = 0; pb < T::NPB; ++pb) {
              const float4 tq = *reinterpret_cast<const float4*>(&Bc[b_rd[pb] + 4 * (b_sw[pb] ^ (2 * i))]);
              bv[pb][0] = tq.x; bv[pb][1] = tq.y; bv[pb][2] = tq.z; bv[pb][3] = tq.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
              for (int fb = 0; fb < T::NFB; ++fb)
#pragma unroll
                for (int pb = 0; pb < T::NPB; ++pb)
                  acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[fb][e], bv[pb][e], acc[fb][pb], 0, 0, 0);
              // the DMA of the next K tile goes out in the first quarter of the step (two pieces per k-pair group): it then
              // has three quarters of the step to land before the barrier (gemm_glds.h)
              if (i == 0 && more && 2 * e < NPW + NPWB) {
                __builtin_amdgcn_sched_barrier(0);
                stage(kn, An, Bn, 2 * e);
                if (2 * e + 1 < NPW + NPWB) stage(kn, An, Bn, 2 * e + 1);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }

        // ---- epilogue (the per-layer kernels' own, on local row coordinates of the tile) ----
        const unsigned long long tt2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        // the next tile of this unit: its weights' first K stage flies under this epilogue (every wave has left the K loop
        // through its last barrier, so buffer 0 is free)
        const bool same_layer = ft + 1 < nft;
        const bool has_next = same_layer || l + 1 < a.n_layers;
        const int ln_next = same_layer ? l : l + 1;
        const int f0_next = same_layer ? f0 + T::BF : 0;
        // ... and, inside a layer, its activation stage too (the layer's own input panel, not touched by this epilogue): both DMAs
        // are then OLDER than the epilogue's stores, so a counted wait can release the tile while the stores are still draining.
        // (Staging the NEXT layer's first activation stage early as well -- it only depends on feature tile 0 of this layer --
        // was measured: no gain, so a layer boundary keeps its plain drain + barrier + stage sequence.)
        const bool early_b = same_layer;
        if (has_next) first_stage(a.L[ln_next], f0_next, true, early_b, pcur ^ 1);
        const int fw = f0 + wf;
        const float* const prm = prm_base + pcur * CHAIN_PRM_FLOATS;
        pcur ^= 1;
        if (CHAIN_EPI_PRIO) __builtin_amdgcn_s_setprio(CHAIN_EPI_PRIO);
        const ChainArgs* ep = gp;
        asm volatile("" : "+s"(ep));          // see the kernel head: epilogue-only fields are loaded here, not at kernel entry
        const ChainArgs& e = *ep;
        unsigned long long ets[3] = {0, 0, 0};
        if (L.kind == CK_GN64) {
          const XposeRowsOut<T::NPB> o{xp, ws + L.out + (size_t)wp * L.ldo + fw, L.ldo};
          chain_gn_silu<64, T::NFB, T::NPB, 128>(acc, prm, wf, o, lane, STAMP ? ets : nullptr);
        } else if (L.kind == CK_GN32) {
          const XposeRowsOut<T::NPB> o{xp, ws + L.out + (size_t)wp * L.ldo + fw, L.ldo};
          chain_gn_silu<32, T::NFB, T::NPB, 128>(acc, prm, wf, o, lane, STAMP ? ets : nullptr);
        } else if (L.kind == CK_INPUT) {
          // rows beyond P hold a clamped copy of the last valid row: computed and stored to the private tile like the others (the
          // host pads cproj to whole tiles), never published (the posterior epilogue stores rows < P only)
          const XposeRowsIn<T::NPB> ci{xp, e.cproj + (size_t)(p0 + wp) * e.ldc + fw, e.ldc};
          const XposeRowsOut<T::NPB> o{xp, ws + L.out + (size_t)wp * L.ldo + fw, L.ldo};
          chain_input<T::NFB, T::NPB, 128>(acc, prm, wf, ci, o, lane);
        } else {
          const float* c = e.coef + 4 * t;
          const float cA = c[0], cB = c[1], cC = c[2];
          const float* zrow = e.z ? e.z + (long long)(e.z_t_first - t) * e.z_step_stride + (size_t)(p0 + wp) * e.ldzz + fw : nullptr;
          float* const mm = e.mut_mask ? e.mut_mask + (size_t)(p0 + wp) * e.mutation_dim : nullptr;
          chain_posterior<T::NFB, T::NPB>(acc, prm, wf, e.x + (size_t)(p0 + wp) * e.D + fw, e.D, P - wp, F - fw, cA, cB, cC, t, zrow, e.ldzz,
                                          e.seed, e.row_offset + (uint32_t)(p0 + wp), fw, mm, e.mutation_dim, xp, lane);
        }
        if (CHAIN_EPI_PRIO) __builtin_amdgcn_s_setprio(0);
        unsigned long long tt3 = 0, te1 = 0, te2 = 0, te3 = 0, te4 = 0;
        if constexpr (STAMP) te1 = __builtin_amdgcn_s_memtime();
        if (early_b) {
          if constexpr (STAMP) tt3 = __builtin_amdgcn_s_memtime();
          // the GroupNorm and input_proj epilogues issue at least 16 unguarded vector-memory operations per wave after the DMAs
          // (8 * NFB * NPB / 2 stores; input_proj also 16 loads): vmcnt(16) = the DMAs have landed; the stores drain under the
          // next K step, whose closing vmcnt(0) collects them.  The posterior epilogue's count depends on guards: full wait.
          static_assert(4 * T::NFB * T::NPB == 16 || 4 * T::NFB * T::NPB == 8, "counted wait below is written for 16 or 8 stores");
          if (L.kind == CK_POST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (4 * T::NFB * T::NPB == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          if constexpr (STAMP) te2 = __builtin_amdgcn_s_memtime();
          __syncthreads();
          if constexpr (STAMP) te3 = te4 = __builtin_amdgcn_s_memtime();
        } else {
          // layer boundary (or the unit's last tile): every wave's stores have left before any wave stages the next layer's
          // input, which is this output
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (STAMP) te2 = __builtin_amdgcn_s_memtime();
          __syncthreads();
          if constexpr (STAMP) tt3 = te3 = __builtin_amdgcn_s_memtime();
          if (has_next) {
            first_stage(a.L[ln_next], 0, false, true, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
          }
          if constexpr (STAMP) te4 = __builtin_amdgcn_s_memtime();
        }
        if constexpr (STAMP) {
          c_pro += tt1 - tt0; c_k += tt2 - tt1; c_epi += tt3 - tt2;
          tt0 = tt3;
          if (tid == 0) {
            unsigned long long* kc = kcnt + 10 * L.kind;
            kc[0] += tt2 - tt1; kc[1] += te1 - tt2; kc[2] += te2 - te1; kc[3] += te3 - te2; kc[4] += te4 - te3; kc[5] += 1;
            if (ets[0]) { kc[6] += ets[0] - tt2; kc[7] += ets[1] - ets[0]; kc[8] += ets[2] - ets[1]; kc[9] += te1 - ets[2]; }
          }
        }
      }
    }

    // ---- publish x_{t-1} of this tile: all waves drained (above), then ONE agent-scope release and the progress word ----
    if (wave == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // hipcc may drop the fence's own wait (Guideline 16, pitfall 12)
      if (lane == 0) st_relaxed_agent(a.progress + tile, a.base_done + (unsigned)si + 1u);
    }
  }
  if (STAMP && a.stamps && tid == 0) {
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 64;
    o[0] = c_dep; o[1] = c_pro; o[2] = c_k; o[3] = c_epi; o[4] = __builtin_amdgcn_s_memtime() - c_start; o[5] = c_units;
    o[6] = __builtin_amdgcn_s_getreg(0xF804);      // HW_REG_HW_ID: where this workgroup ran
    o[7] = __builtin_amdgcn_s_getreg(0xF814) & 7;  // HW_REG_XCC_ID
    for (int i = 0; i < 40; ++i) o[8 + i] = kcnt[i];
  }
}

}  // namespace osd
