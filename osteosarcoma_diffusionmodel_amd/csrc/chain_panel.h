// chain_panel.h -- the LDS-resident reverse-chain kernel: a workgroup owns 64 patients for one whole step and their activations
// never leave the CU.
//
// Same job, queue, hand-off protocol and epilogue arithmetic as chain.h (models/diffusion.py:427-449; the engines agree
// bitwise), different data flow.  chain.h computes 128 x 128 output tiles and moves every layer's activations through a
// private workspace in L2 / Infinity Cache: per tile a DMA round trip, a barrier per 32-k stage, an epilogue that transposes
// through LDS and drains its stores -- 30 % of a workgroup's cycles are epilogue + drain, and two workgroups per CU only
// partly hide each other's bubbles (fp32 MFMA and VALU share the SIMD's issue).  Here:
//
//   * unit = (64-patient tile, step); ONE 4-wave workgroup per CU (141 KB of LDS) carries the tile through all layers;
//   * the layer input sits in LDS as a panel [64 patients][K (+4 pad)]: it is the MFMA B operand, read with conflict-free
//     ds_read_b128 (row stride = 4 mod 64 dwords); a wave owns F/4 output features x all 64 patients, so GroupNorm's row
//     statistics stay inside a wave exactly as in the tile kernels;
//   * the weights (A operand) never touch LDS: a fragment-ordered copy (chain_panel.hip: pack) makes every wave load one
//     contiguous 1 KiB -- lane (l31, h) gets W[f0 + l31][8 i + 4 h .. + 3], which is precisely the k-pair order of the tile
//     kernels' swizzled ds_read_b128 -- straight into registers, PC_DEPTH (8-k) blocks ahead.  No barrier and no s_waitcnt
//     vmcnt(0) inside a layer's K loop: a lone wave per SIMD streams MFMAs at 0.90 of the matrix peak with all 256 CUs
//     pulling the 10.7 MB of weights through L2 (tools/probes/lds_panel.hip; the tile loop of a lone workgroup: 0.62);
//   * an epilogue writes its output fragments straight into the next layer's panel (ds_write_b128, the same conflict-free
//     pattern): two workgroup barriers per layer, no global store, no drain, no transposer;
//   * the encoder output a late decoder block needs again (512 wide: it does not fit beside the running panel) is spilled
//     to a private slot in fragment order (coalesced 1 KiB stores, read back by the very lanes that wrote it); the narrower
//     one stays in unused panel columns;
//   * x_t streams in for input_proj as [64][256]-k chunks (LDS-DMA, double-buffered), cond_proj's tile comes in under the
//     last chunk, and output_proj + posterior runs as passes of 512 features straight out of the panel with chain.h's
//     posterior epilogue (x_t / x_{t-1} through the per-wave row transposer: global traffic in full 128-byte segments).
//
// The host (chain_panel.hip) lays the panels out per layer and refuses architectures that do not fit; those run on chain.h.
#pragma once
#include <type_traits>
#include "chain.h"

namespace osd {

// 1: full output blocks take a guard-free instantiation of the posterior epilogue.  Measured 0.740 -> 0.709 of peak: the kernel is
// 51.6 KB of code (chain.h: 38 KB) and the second instantiation pushes it past the 64 KB instruction cache two CUs share.
#ifndef PC_POST_NOGUARD
#define PC_POST_NOGUARD 0
#endif
constexpr int PC_BP = 64;                         // patients per unit
#ifndef PC_NW
#define PC_NW 8
#endif
// Waves per workgroup.  8 (two per SIMD): a wave owns F/8 output features x all 64 patients, i.e. the 64 x 64 accumulator of the
// tile kernels for 512-wide layers and 32 x 64 for 256-wide ones; the second wave of a SIMD fills the first one's LDS / memory /
// transcendental latencies and MFMA issue bubbles (one wave per SIMD: 72 cycles per 64-cycle MFMA in the K loops and every
// epilogue stall paid in full -- profiles/r03_chain_panel.md).  4 (make CXXFLAGS+=-DPC_NW=4) is the first version, kept as a build option.
constexpr int PC_THREADS = 64 * PC_NW;
constexpr int PC_NFB_WIDE = 512 / (32 * PC_NW);   // feature blocks per wave: 512-wide layers and output_proj passes
constexpr int PC_NFB_NARROW = 256 / (32 * PC_NW); // 256-wide layers and input_proj
static_assert(PC_NW == 4 || PC_NW == 8, "4 or 8 waves");
constexpr int PC_N8_MIN = 16;                     // 8-k blocks of a K segment: a multiple of 8, at least 16 (two groups of the 64-feature waves' weight stream)
constexpr int PC_CHUNK = 256;                     // k per staged chunk of x_t
constexpr int PC_HALF = PC_BP * (PC_CHUNK + 4);   // floats of one chunk buffer [64][260]
constexpr int PC_REGION = 2 * PC_HALF;            // panel / chunk region (>= [64][516])
constexpr int PC_PS = 512;                        // parameter block: bias | gamma (temb row) | beta, 512 floats each; output_proj: 128 bias floats per wave
constexpr int PC_PRM = 3 * PC_PS;
constexpr int PC_LDS_BYTES = (PC_REGION + PC_PRM + 16) * 4;
constexpr int PC_MAX_LAYERS = 16;

struct PanelSeg {
  int col;        // first panel column of the segment
  int n8;         // 8-k blocks (multiple of PC_DEPTH)
  int reload;     // >= 0: float offset (slot workspace) of a fragment-ordered spill that is brought into the panel columns first
};
struct PanelLayer {
  const float* wpk;             // fragment-ordered weights [ceil(F/32)][K8][64 lanes][4]
  int K8;                       // 8-k blocks per feature block, all segments
  int F;                        // output features (CK_POST: true extent; the packed copy is zero-padded to whole 128s)
  int kind;                     // CK_INPUT / CK_GN32 / CK_GN64 / CK_POST
  int nseg; PanelSeg seg[2];
  int in_base, in_ld;           // input panel: float offset into the region, row stride
  int out_base, out_ld, out_col;
  int spill;                    // >= 0: the output is also written to this float offset of the slot workspace, fragment order
  const float* bias; const float* gamma; const float* beta;
};

struct PanelArgs {
  PanelLayer L[PC_MAX_LAYERS];
  int n_layers;
  float* ws; long long ws_stride;
  float* x; int ldx; int D;                  // chain state [n][ldx], D valid columns
  int n, n_tiles;
  int t_first, n_steps;
  unsigned base_done;
  const float* cproj; int ldc;               // [n_tiles * 64][H0]
  const float* temb; int ldt;
  const float* coef;
  const float* z; int ldzz; long long z_step_stride; int z_t_first;
  uint64_t seed; uint32_t row_offset;
  float* mut_mask; int mutation_dim;
  unsigned* progress; unsigned* status; unsigned* queue;
  unsigned long long spin_budget;
  int cp_base;                               // region offset of the staged cond_proj tile [64][H0 + 4]
  int xp_base;                               // region offset of the four posterior transposers (8 KB each)
  unsigned long long* stamps;                // diagnostic builds: per workgroup {dependency wait, input layer, GN layers, output layer, total, units}
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* gv4f_ptr;

// ---- the weight stream ----------------------------------------------------------------------------------------------------
// A lane keeps PC_SLOTS float4 registers of weights in flight: DEPTH = PC_SLOTS / NFB 8-k blocks ahead of the MFMAs (8 blocks in the
// 256-wide layers, 4 in the 512-wide ones), slot j = d * NFB + fb.  The stream never stops: the last
// group of a K segment refills its slots from whatever comes next -- the same layer's next segment, the next layer, the next
// output_proj pass, the next unit's input_proj -- so the only exposed round trip of a workgroup is its very first one.
constexpr int PC_SLOTS = 64 / PC_NW;     // 16 float4 per lane with four waves, 8 with eight (256 registers per wave)
struct WStream {
  gv4f_ptr w;       // the lane's pointer at (feature block 0 of its wave, 8-k block 0)
  int fbs;          // float4s between feature blocks
  int shift;        // log2(NFB) of the consumer
};
__device__ __forceinline__ v4f ws_slot(const WStream& s, int j) {
  const int fb = j & ((1 << s.shift) - 1), d = j >> s.shift;
  return s.w[(size_t)fb * s.fbs + d * 64];
}
__device__ __forceinline__ void ws_prime(v4f (&aq)[PC_SLOTS], const WStream& s) {
#pragma unroll
  for (int j = 0; j < PC_SLOTS; ++j) aq[j] = ws_slot(s, j);
}

// One K segment: acc[fb][pb] += W[f][k] * act[p][k] over n8 8-k blocks (n8 a multiple of DEPTH, n8 >= 2 DEPTH).  On entry aq holds
// the segment's first DEPTH blocks, on exit the first blocks of `nxt`.  wl = the lane's pointer at (feature block 0, the
// segment's block 0), bl = the lane's panel row at its k-half.  Branch-free groups of DEPTH blocks (with the refill loads in
// conditional blocks hipcc's wait insertion falls back to vmcnt(0) per group: no prefetch at all); first and last group peeled.
// `mid` runs once, after the first block's MFMAs: LDS-DMA pieces issued there are older than every later weight load -- vmcnt
// counts in order -- so the first wait that includes them is DEPTH blocks away.  After the segment exactly PC_SLOTS loads are
// the youngest outstanding memory operations: s_waitcnt vmcnt(16) then means "the DMA pieces have landed".
// MFMA order = the tile kernels' order: 8-k blocks ascending, inside a block the pairs (e, e + 4).
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// FRESH: the accumulators start at zero -- the segment's first MFMA of each takes the constant 0 as its C operand instead of a
// register block that 64-128 v_accvgpr_write would have to clear first (same bits: 0 + a b either way).
template <int NFB, bool FRESH = false, class Mid = NoMid>
__device__ __forceinline__ void panel_kseg(f32x16 (&acc)[NFB][2], v4f (&aq)[PC_SLOTS], gv4f_ptr wl, int fbs, int n8, const float* bl, int ldb,
                                           const WStream& nxt, const Mid& mid = Mid()) {
  // (Tried, with eight waves: the two waves of a SIMD taking turns at the higher issue priority -- left alone the arbiter favours
  // the older one, which reaches the layer's barrier early while its partner finishes on its own at a lone wave's rate: 5 600 of a
  // chunk's 38 000 cycles, 21 000 of a 512 x 512 layer's 141 000.  s_setprio swapped every 8-k block: 0.740 -> 0.703 of peak; swapped
  // once in the middle of a segment: 0.730.  Streaming one wave's MFMAs is what the SIMD does best; removed.)
  constexpr int DEPTH = PC_SLOTS / NFB;
  static_assert(DEPTH % 2 == 0, "the B fragments alternate between two register sets");
  v4f bq[2][2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) bq[0][pb] = *reinterpret_cast<const v4f*>(bl + pb * 32 * ldb);
  auto group = [&](int i0, bool own, bool first) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int i = i0 + d;
      const int in = i + 1 < n8 ? i + 1 : i;          // scalar clamp: the last block re-reads itself
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) bq[(d + 1) & 1][pb] = *reinterpret_cast<const v4f*>(bl + pb * 32 * ldb + 8 * in);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            if (FRESH && first && d == 0 && e == 0) {
              const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d * NFB + fb][e], bq[d & 1][pb][e], zero, 0, 0, 0);
            } else {
              acc[fb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d * NFB + fb][e], bq[d & 1][pb][e], acc[fb][pb], 0, 0, 0);
            }
          }
      if (first && d == 0) mid();
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        if (own) aq[d * NFB + fb] = wl[(size_t)fb * fbs + (size_t)(i + DEPTH) * 64];
        else aq[d * NFB + fb] = ws_slot(nxt, d * NFB + fb);
      }
    }
  };
  group(0, true, true);
  int i0 = DEPTH;
  for (; i0 < n8 - DEPTH; i0 += DEPTH) group(i0, true, false);
  group(i0, false, false);
}
#if PC_NW == 4
#define PC_WAIT_DMA() asm volatile("s_waitcnt vmcnt(16)" ::: "memory")
#else
#define PC_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#endif
// Workgroup barrier for LDS traffic only.  __syncthreads() is a workgroup-scope release/acquire over ALL address spaces: hipcc
// puts s_waitcnt vmcnt(0) in front of it, which drains the weight stream's prefetch at every barrier (measured: the round trip
// then shows up in the epilogues instead of the K loops).  Everything the barriers of this kernel order is LDS -- panels,
// parameters, flags; DMA pieces are waited for explicitly (PC_WAIT_DMA), global hand-offs have their own fences.
#define PC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
static_assert(PC_SLOTS == (PC_NW == 4 ? 16 : 8), "PC_WAIT_DMA counts the slots");

template <int NFB>
__device__ __forceinline__ void panel_zero(f32x16 (&acc)[NFB][2]) {
#pragma unroll
  for (int i = 0; i < NFB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Output fragments -> the next layer's panel (and, optionally, a fragment-ordered spill in the slot workspace).
// base = the lane's position: panel + out_base + l31 * ld + out_col + (wave's first feature) + 4 h
struct PanelOut {
  float* base; int ld;
  float* spill;           // the lane's float4 slot 0 of this wave's spill block, or null
  __device__ __forceinline__ void put(int fb, int pb, int q, int l31, int h, float4 v) const {
    (void)l31; (void)h;
    *reinterpret_cast<float4*>(base + pb * 32 * ld + 32 * fb + 8 * q) = v;
    if (spill) stg4(spill + (size_t)(((fb * 2 + pb) * 4 + q) * 256), v);
  }
  __device__ __forceinline__ void flush(int fb, int lane) const { (void)fb; (void)lane; }
};
// cond_proj's tile staged in LDS as [64][ld]: the lane reads its own fragment positions
struct PanelIn {
  const float* base;      // tile + l31 * ld + (wave's first feature) + 4 h
  int ld;
  __device__ __forceinline__ void load(int fb, int lane) const { (void)fb; (void)lane; }
  __device__ __forceinline__ float4 get(int fb, int pb, int q, int l31, int h) const {
    (void)l31; (void)h;
    return *reinterpret_cast<const float4*>(base + pb * 32 * ld + 32 * fb + 8 * q);
  }
};

template <bool STAMP>
__global__ __launch_bounds__(PC_THREADS, 1) void panel_chain_kernel(const PanelArgs* __restrict__ gp) {
  const PanelArgs& a = *gp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const region = smem;
  float* const prm = smem + PC_REGION;
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + PC_REGION + PC_PRM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  float* const ws = a.ws + (long long)blockIdx.x * a.ws_stride;

  unsigned long long c_dep = 0, c_in = 0, c_gn = 0, c_post = 0, c_units = 0;
  unsigned long long fin[3] = {0, 0, 0};                 // STAMP: this wave's chunk 3 of input_proj: K loop, DMA wait, barrier
  unsigned long long fine[6] = {0, 0, 0, 0, 0, 0};      // STAMP: this wave's phases of layer 8 (512 -> 512): zero+setup, K loop, DMA wait, barrier, epilogue, barrier
  unsigned long long c_in_e = 0, c_gn_k[2] = {0, 0}, c_gn_e[2] = {0, 0}, c_post_k = 0, c_post_e = 0, c_reload = 0;      // STAMP: K loops / epilogues by layer class
  const unsigned long long c_start = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  const long long n_units = (long long)a.n_tiles * a.n_steps;

  // one 1 KiB DMA piece: 256 floats from src (16 B per lane, clamped to the last whole float4 below `valid`) to LDS floats dst
  auto dma_row = [&](const float* src, int valid, float* dst) {
    int k = 4 * lane;
    k = k < valid - 4 ? k : valid - 4;
    glds16(src + k, __builtin_amdgcn_readfirstlane(lds_addr(dst)));
  };

  // the wave's weight stream of a layer (input_proj / GroupNorm layers: F / PC_NW features per wave; output_proj: pass 0)
  auto layer_stream = [&](const PanelLayer& Ls) {
    const int nfb = (Ls.kind == CK_POST || Ls.F == 512) ? PC_NFB_WIDE : PC_NFB_NARROW;
    const int fbs = Ls.K8 * 64;
    return WStream{(gv4f_ptr)(Ls.wpk) + (size_t)(wave * nfb) * fbs + lane, fbs, nfb == 4 ? 2 : (nfb == 2 ? 1 : 0)};
  };
  const WStream s_in = layer_stream(a.L[0]);
  v4f aq[PC_SLOTS];
  ws_prime(aq, s_in);

  for (;;) {
    const unsigned long long td0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    if (wave == 0) {
      unsigned nu = 0;
      if (lane == 0) nu = atomicAdd(a.queue, 1u);
      s_flag = __builtin_amdgcn_readfirstlane((int)nu);
    }
    PC_BARRIER();
    const long long u = (unsigned)__builtin_amdgcn_readfirstlane(s_flag);
    PC_BARRIER();
    if (u >= n_units) break;
    const int tile = (int)(u % a.n_tiles);
    const int si = (int)(u / a.n_tiles);
    const int t = a.t_first - si;
    const int p0 = tile * PC_BP;
    const int P = (a.n - p0 < PC_BP) ? a.n - p0 : PC_BP;

    // ---- dependency: x_t of this tile (chain.h's protocol) ----
    if (wave == 0) {
      bool ok = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(a.status)) == CHAIN_OK;
      if (ok && si > 0) {
        ok = chain_wait(a.progress + tile, a.base_done + (unsigned)si, a.status, a.spin_budget, lane);
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_flag = ok ? 1 : 0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    PC_BARRIER();
    const int go = __builtin_amdgcn_readfirstlane(s_flag);
    if (!go) return;
    const unsigned long long tu0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    if constexpr (STAMP) { c_dep += tu0 - td0; ++c_units; }

    // =============================== input_proj: x_t in chunks of 256 k ===============================
    {
      const PanelLayer& L = a.L[0];
      const int K8 = L.K8, F = L.F;
      const int fbs = K8 * 64;
      const float* const xrows = a.x + (size_t)p0 * a.ldx;
      const int ldx = a.ldx, D = a.D;
      const int nchunk = (K8 * 8 + PC_CHUNK - 1) / PC_CHUNK;
      const float* const temb_row = a.temb + (size_t)t * a.ldt;
      const float* const in_bias = L.bias;
      auto stage_x = [&](int c, float* buf) {
        const int k0 = c * PC_CHUNK;
#pragma unroll 4
        for (int j = 0; j < PC_BP / PC_NW; ++j) {
          int r = wave * (PC_BP / PC_NW) + j;
          const int rg = r < P ? r : P - 1;
          dma_row(xrows + (size_t)rg * ldx + k0, D - k0, buf + r * (PC_CHUNK + 4));
        }
      };
      auto stage_cproj = [&](float* buf) {
        const float* cp = a.cproj + (size_t)p0 * a.ldc;
#pragma unroll 4
        for (int j = 0; j < PC_BP / PC_NW; ++j) {
          const int r = wave * (PC_BP / PC_NW) + j;
          dma_row(cp + (size_t)r * a.ldc, F, buf + r * (PC_CHUNK + 4));      // the host pads cproj to whole tiles
        }
      };
      stage_x(0, region);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PC_BARRIER();
      constexpr int NFI = PC_NFB_NARROW;
      f32x16 acc[NFI][2];
      panel_zero<NFI>(acc);
      gv4f_ptr wl = (gv4f_ptr)(L.wpk) + (size_t)(wave * NFI) * fbs + lane;
      const WStream s_l1 = layer_stream(a.L[1]);
      for (int c = 0; c < nchunk; ++c) {
        float* const cur = region + (c & 1) * PC_HALF;
        float* const nxt = region + ((c + 1) & 1) * PC_HALF;
        const int n8 = (c + 1 < nchunk) ? PC_CHUNK / 8 : K8 - c * (PC_CHUNK / 8);
        auto mid = [&]() {
          if (c + 1 < nchunk) stage_x(c + 1, nxt);
          else stage_cproj(nxt);
          if (c == 0) {       // parameters: bias (wave 0) and the time-embedding row (wave 1), F <= 256 floats each
            if (wave == 0) dma_row(in_bias, F, prm);
            else if (wave == 1) dma_row(temb_row, F, prm + PC_PS);
          }
        };
        const WStream wnext = (c + 1 < nchunk) ? WStream{wl + (PC_CHUNK / 8) * 64, fbs, NFI == 2 ? 1 : 0} : s_l1;
        const unsigned long long ti0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        panel_kseg<NFI, false>(acc, aq, wl, fbs, n8, cur + l31 * (PC_CHUNK + 4) + 4 * h, PC_CHUNK + 4, wnext, mid);
        wl += (PC_CHUNK / 8) * 64;
        const unsigned long long ti1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        PC_WAIT_DMA();
        const unsigned long long ti2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        PC_BARRIER();
        if constexpr (STAMP) if (c == 3) { fin[0] += ti1 - ti0; fin[1] += ti2 - ti1; fin[2] += __builtin_amdgcn_s_memtime() - ti2; }
      }
      // h0 = ((acc + b) + temb[t]) + cproj: cproj sits in the buffer the last chunk did not use, h0 goes where the host says
      // (the last chunk's buffer: every wave has left its K loop)
      const unsigned long long tie = STAMP ? __builtin_amdgcn_s_memtime() : 0;
      const int fl = wave * 32 * NFI;
      const PanelIn ci{region + (nchunk & 1) * PC_HALF + l31 * (PC_CHUNK + 4) + fl + 4 * h, PC_CHUNK + 4};
      const PanelOut o{region + L.out_base + l31 * L.out_ld + L.out_col + fl + 4 * h, L.out_ld, nullptr};
      chain_input<NFI, 2, PC_PS>(acc, prm, fl, ci, o, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PC_BARRIER();
      if constexpr (STAMP) c_in_e += __builtin_amdgcn_s_memtime() - tie;
    }
    const unsigned long long tu1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;

    // =============================== Linear + GroupNorm + SiLU layers ===============================
    const int nl = a.n_layers;
    for (int l = 1; l + 1 < nl; ++l) {
      const PanelLayer& L = a.L[l];
      const int F = L.F;
      // parameters: 3 arrays x F / 256 pieces, dealt round-robin to the waves (the previous epilogue is behind a barrier)
      const float* const pb_ = L.bias; const float* const pg_ = L.gamma; const float* const pbe_ = L.beta;
      auto params = [&]() {
        const int per = F / 256;
        for (int j = wave; j < 3 * per; j += PC_NW) {
          const int arr = j / per, piece = j % per;
          const float* src = arr == 0 ? pb_ : (arr == 1 ? pg_ : pbe_);
          dma_row(src + piece * 256, F - piece * 256, prm + arr * PC_PS + piece * 256);
        }
      };
      auto reload = [&](const PanelSeg& sg) {
        // the spill's lanes are this workgroup's own: wave w wrote its 32 NFB features of the 512 as NFB x 2 x 4 float4 per lane
        constexpr int NFS = PC_NFB_WIDE;
        const float* sp = ws + sg.reload + (size_t)wave * (NFS * 2 * 4 * 256) + 4 * lane;
        float* dst = region + L.in_base + l31 * L.in_ld + sg.col + wave * 32 * NFS + 4 * h;
#pragma unroll
        for (int fb = 0; fb < NFS; ++fb) {
          float4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ldg4(sp + (size_t)((fb * 8 + j) * 256));
#pragma unroll
          for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(dst + (j >> 2) * 32 * L.in_ld + 32 * fb + 8 * (j & 3)) = v[j];
        }
      };
      auto run = [&](auto nfb_tag) {
        constexpr int NFB = decltype(nfb_tag)::value;
        const unsigned long long tk0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        const int fbs = L.K8 * 64;
        f32x16 acc[NFB][2];
        gv4f_ptr wl = (gv4f_ptr)(L.wpk) + (size_t)(wave * NFB) * fbs + lane;
        const WStream s_next = layer_stream(a.L[l + 1]);
        for (int s = 0; s < L.nseg; ++s) {
          const PanelSeg sg = L.seg[s];
          if (sg.reload >= 0) {
            const unsigned long long tr0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
            PC_BARRIER();                  // every wave has finished the segment that lived in these columns
            reload(sg);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            PC_BARRIER();
            if constexpr (STAMP) c_reload += __builtin_amdgcn_s_memtime() - tr0;
          }
          const float* bl = region + L.in_base + l31 * L.in_ld + sg.col + 4 * h;
          const WStream nxt = (s + 1 < L.nseg) ? WStream{wl + (size_t)sg.n8 * 64, fbs, NFB == 4 ? 2 : (NFB == 2 ? 1 : 0)} : s_next;
          const unsigned long long tf0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
          if (s == 0) panel_kseg<NFB, true>(acc, aq, wl, fbs, sg.n8, bl, L.in_ld, nxt, params);
          else panel_kseg<NFB>(acc, aq, wl, fbs, sg.n8, bl, L.in_ld, nxt);
          wl += (size_t)sg.n8 * 64;
          if constexpr (STAMP) if (l == 8) { asm volatile("s_nop 0" ::: "memory"); fine[0] += tf0 - tk0; fine[1] += __builtin_amdgcn_s_memtime() - tf0; }
        }
        const unsigned long long tf1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        PC_WAIT_DMA();                        // the parameter DMA (a whole segment old); the next layer's first blocks stay in flight
        const unsigned long long tf2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        PC_BARRIER();                      // every wave has read its operands: the panel may be overwritten
        const unsigned long long tk1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        if constexpr (STAMP) if (l == 8) { fine[2] += tf2 - tf1; fine[3] += tk1 - tf2; }
        const int fl = wave * 32 * NFB;
        float* sp = L.spill >= 0 ? ws + L.spill + (size_t)wave * (NFB * 2 * 4 * 256) + 4 * lane : nullptr;
        const PanelOut o{region + L.out_base + l31 * L.out_ld + L.out_col + fl + 4 * h, L.out_ld, sp};
        // GroupNorm(8, F): groups of 64 channels in the 512-wide layers, 32 in the 256-wide ones (the host checks gw == F / 8)
        if constexpr (NFB == PC_NFB_WIDE) chain_gn_silu<64, NFB, 2, PC_PS>(acc, prm, fl, o, lane);
        else chain_gn_silu<32, NFB, 2, PC_PS>(acc, prm, fl, o, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long tf3 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        PC_BARRIER();
        if constexpr (STAMP) if (l == 8) { fine[4] += tf3 - tk1; fine[5] += __builtin_amdgcn_s_memtime() - tf3; }
        if constexpr (STAMP) { const unsigned long long tk2 = __builtin_amdgcn_s_memtime(); c_gn_k[NFB == PC_NFB_WIDE] += tk1 - tk0; c_gn_e[NFB == PC_NFB_WIDE] += tk2 - tk1; }
      };
      if (F == 512) run(std::integral_constant<int, PC_NFB_WIDE>{});
      else run(std::integral_constant<int, PC_NFB_NARROW>{});
    }
    const unsigned long long tu2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;

    // =============================== output_proj + posterior: passes of 512 features ===============================
    {
      const PanelLayer& L = a.L[nl - 1];
      const int F = L.F, K8 = L.K8;
      const int fbs = K8 * 64;
      const int Fpad = (F + 127) / 128 * 128;
      const float* const out_bias = L.bias;
      const PanelArgs* ep = gp;
      asm volatile("" : "+s"(ep));
      const PanelArgs& e = *ep;
      const float* c = e.coef + 4 * t;
      const float cA = c[0], cB = c[1], cC = c[2];
      WaveXpose<2> xp;
      xp.buf = region + e.xp_base + wave * 2048;
      constexpr int NFP = PC_NFB_WIDE;
      float* const prm_w = prm + wave * 32 * NFP;     // this wave's bias floats of the pass: private, so no barrier between passes
      const int npass = (Fpad + 511) / 512;
      for (int ps = 0; ps < npass; ++ps) {
        const int fw = ps * 512 + wave * 32 * NFP;
        if (fw >= Fpad) break;                // uniform per wave; the workgroup meets again at the barrier behind the loop
        auto bias_dma = [&]() {
          if (lane < 8 * NFP) {               // 8 NFP lanes x 16 B; the DMA honours the exec mask
            int k = 4 * lane;
            const int valid = F - fw > 4 ? F - fw : 4;
            k = k < valid - 4 ? k : valid - 4;
            glds16(out_bias + fw + k, __builtin_amdgcn_readfirstlane(lds_addr(prm_w)));
          }
        };
        const unsigned long long tp0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        f32x16 acc[NFP][2];
        gv4f_ptr wl = (gv4f_ptr)(L.wpk) + (size_t)(fw / 32) * fbs + lane;
        const float* bl = region + L.in_base + l31 * L.in_ld + L.seg[0].col + 4 * h;
        float* const xw = e.x + (size_t)p0 * e.ldx + fw;
        float4 xpre[8];
        if (F - fw > 0) xp.template issue_rows<true>(xpre, xw, e.ldx, lane, P, F - fw);      // block 0's x_t rows fly under the K loop
        const WStream nxt = (fw + 512 < Fpad) ? WStream{wl + (size_t)16 * fbs, fbs, NFP == 4 ? 2 : 1} : s_in;      // this wave's next pass, or the next unit's input_proj
        panel_kseg<NFP, true>(acc, aq, wl, fbs, K8, bl, L.in_ld, nxt, bias_dma);
        PC_WAIT_DMA();                        // the bias DMA (a whole K loop old)
        const unsigned long long tp1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
        const float* zrow = e.z ? e.z + (long long)(e.z_t_first - t) * e.z_step_stride + (size_t)p0 * e.ldzz + fw : nullptr;
        float* const mm = e.mut_mask ? e.mut_mask + (size_t)p0 * e.mutation_dim : nullptr;
        if (PC_POST_NOGUARD && P == PC_BP && F - fw >= 32 * NFP)      // a full block: no clamps, no per-element bounds
          chain_posterior<NFP, 2, true, false>(acc, prm_w, 0, xw, e.ldx, P, F - fw, cA, cB, cC, t, zrow, e.ldzz,
                                               e.seed, e.row_offset + (uint32_t)p0, fw, mm, e.mutation_dim, xp, lane, &xpre);
        else
          chain_posterior<NFP, 2, true>(acc, prm_w, 0, xw, e.ldx, P, F - fw, cA, cB, cC, t, zrow, e.ldzz,
                                        e.seed, e.row_offset + (uint32_t)p0, fw, mm, e.mutation_dim, xp, lane, &xpre);
        if constexpr (STAMP) { const unsigned long long tp2 = __builtin_amdgcn_s_memtime(); c_post_k += tp1 - tp0; c_post_e += tp2 - tp1; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PC_BARRIER();
    }
    // ---- publish x_{t-1} of this tile ----
    if (wave == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) st_relaxed_agent(a.progress + tile, a.base_done + (unsigned)si + 1u);
    }
    if constexpr (STAMP) {
      const unsigned long long tu3 = __builtin_amdgcn_s_memtime();
      c_in += tu1 - tu0; c_gn += tu2 - tu1; c_post += tu3 - tu2;
    }
  }
  if (STAMP && a.stamps && lane == 0 && wave < 4) {
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 64 + 16 + wave * 8;
    for (int i = 0; i < 6; ++i) o[i] = fine[i];
    o[6] = fin[0]; o[7] = fin[1];
    a.stamps[(size_t)blockIdx.x * 64 + 48 + wave] = fin[2];
  }
  if (STAMP && a.stamps && tid == 0) {
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 64;
    o[0] = c_dep; o[1] = c_in; o[2] = c_gn; o[3] = c_post; o[4] = __builtin_amdgcn_s_memtime() - c_start; o[5] = c_units;
    o[6] = __builtin_amdgcn_s_getreg(0xF804);
    o[7] = __builtin_amdgcn_s_getreg(0xF814) & 7;
    o[8] = c_in_e; o[9] = c_gn_k[0]; o[10] = c_gn_e[0]; o[11] = c_gn_k[1]; o[12] = c_gn_e[1]; o[13] = c_post_k; o[14] = c_post_e; o[15] = c_reload;
  }
}

}  // namespace osd
