// chain_squad.h -- the small-batch reverse-chain kernel: eight workgroups (a "squad") carry 32 patients through a whole chain.
//
// models/diffusion.py:427-449 at the reference's own generation sizes (utils/generate.py: 3 scenarios x 333 ... 1000 patients) is
// a few thousand rows: 8-24 tiles of the 128 x 128 kernels, a queue of 64-row units for 16-47 of 256 CUs.  The per-layer engine
// spreads such a step over the chip with split-K launches and pays ~20 launch boundaries of 4-5 us per step for ~50 us of matrix
// work (profiles/r04_refw_kernel_stats.csv); a barrier over ALL workgroups of a resident kernel is no cheaper than a launch
// boundary once it carries data (tools/probes/grid_barrier.hip: 1.8 us for the barrier, 7-26 us with agent-scope fences).
//
// Rows of the chain never interact, GroupNorm(8, C) has exactly 8 groups per layer, and a workgroup's share of a layer is small.
// So the unit of parallelism here is the squad: 8 workgroups x 4 waves own one 32-patient panel for all T steps.
//   * Linear + GroupNorm + SiLU layer: workgroup g computes GroupNorm group g (C / 8 = 32 or 64 features) for the 32 patients;
//     its four waves split K four ways (v_mfma_f32_32x32x2_f32, 32 x 32 accumulators), the three partial accumulators meet
//     wave 0's in LDS, wave 0 applies bias + GroupNorm + SiLU (the row statistics of a group never leave the wave) and stores the
//     32 x C/8 result in the operand order of the next layer;
//   * activations travel between the squad's workgroups in "unit" order -- [8-k block][lane][4 floats], lane (l31, h) holding
//     patient l31, k = 8 i + 4 h .. + 3: one coalesced 1 KiB per wave instruction for the writer (its accumulator fragments) and
//     the reader (its MFMA B operand), no LDS staging, no transposer -- with agent-scope (sc1) loads and stores, so no cache
//     write-back or invalidate is needed around the hand-off: s_waitcnt vmcnt(0), one relaxed atomic arrive on the squad's
//     counter, a poll by wave 0 (the probe's mode 3: no stale reads, ~1 us per 8-workgroup barrier);
//   * weights come from the fragment-ordered copies the LDS-resident chain uses (chain_panel.hip: k_pack_fragments), straight
//     into registers, SQ_DEPTH 8-k blocks ahead; workgroup g of every squad reads the same slices, and blockIdx % 8 = g puts
//     them all on one XCD: a slice is L2-resident in the one L2 that needs it;
//   * input_proj splits K instead (workgroup g takes the state features it owns, all H0 outputs): the partials meet in a
//     reduce phase that adds bias, time embedding and cond_proj;
//   * output_proj + posterior: workgroup g owns the same D/8 features of the chain state that it reads in input_proj, so x_t
//     never crosses a workgroup boundary: it lives in unit order in a private slice (L2) and is written out row-major once, at
//     the end of the launch.
// 12 squad barriers per step; squads never wait for each other.  Arithmetic is the per-layer kernels' (same MFMA, same epilogue
// formulas, same Philox addressing); the K split differs, so results agree with the other engines to fp32 rounding, not bitwise.
// Every spin is bounded (chain.h's budget and status word): on expiry the kernel drains and the host re-runs the chain on the
// per-layer kernels.
#pragma once
#include "chain.h"
#include "chain_panel.h"

namespace osd {

constexpr int SQ_RP = 32;            // patients per panel
constexpr int SQ_S = 8;              // workgroups per squad = GroupNorm groups
constexpr int SQ_THREADS = 256;
constexpr int SQ_MAX_LAYERS = 16;
// 8-k blocks a wave keeps in flight: 8 when it has a SIMD to itself (one workgroup per CU: 999 rows), 4 with two or three
// waves per SIMD (their registers are a half or a third, and the other waves cover the rest of the latency)
template <int WPC> struct SquadDepth { static constexpr int value = WPC == 1 ? 8 : 4; };
constexpr int SQ_PRM = 3 * 64;       // per layer in LDS: bias | gamma | beta of this workgroup's group (<= 64 features)
constexpr int SQ_STAGE_FLOATS = 32 * 256 + 12 * 256;      // output_proj's operand (last width 256, unit order) + the partials of a K-split tile;
                                                         // the layers' partial accumulators [4 waves][32 patients][64 + 4] share it
static_assert(SQ_STAGE_FLOATS >= 4 * 32 * 68, "partials");
// dynamic LDS: stage | n_layers parameter blocks | flag.  10 layers: 53 056 bytes -- three workgroups per CU (3 072 rows on 256 CUs)
__host__ __device__ constexpr int sq_lds_bytes(int n_layers) { return (SQ_STAGE_FLOATS + n_layers * SQ_PRM + 16) * 4; }

// Every memory operation of the wave done, then the workgroup barrier.  The wait is the builtin, so that hipcc's own book of
// outstanding loads is cleared with it (an asm wait leaves stale entries that cost a vmcnt(0) inside a later loop); the barrier is asm,
// because hipcc drains every counter in front of a barrier it can see.
#define SQ_DRAIN_BARRIER() do { __builtin_amdgcn_s_waitcnt(0x0070); asm volatile("s_barrier" ::: "memory"); } while (0)

struct SquadLayer {
  int w_off;                         // float offset (SquadArgs::wpk) of the fragment-ordered weights [F / 32][K8][64][4]
  int K8;                            // 8-k blocks of the whole input (both sources)
  int F;                             // output features: 256 or 512
  int in0, n8_0;                     // first source: float offset of its buffer inside the panel's activation region, its 8-k blocks
  int in1;                           // second source (decoder skip) or -1
  int out;                           // float offset of the output buffer
  const float* bias; const float* gamma; const float* beta;
};

struct SquadArgs {
  SquadLayer L[SQ_MAX_LAYERS];
  int n_layers;
  const float* wpk; long long wpk_floats;                // every packed weight of the chain (one buffer descriptor)
  int in_off; const float* bias_in; int H0;              // input_proj [H0 / 32][4 T32][64][4] at float offset in_off
  int out_off; const float* bias_out;                    // output_proj [T32][hl / 8][64][4]; bias padded to 32 T32 floats
  int T32;                           // 32-feature tiles of the chain state: ceil(D / 32)
  int h0_out;                        // float offset of input_proj's output buffer
  int last_in;                       // float offset of output_proj's input buffer
  int K8_out;                        // hl / 8
  float* x; int ldx; int D;          // chain state [n][ldx], row-major: read at the start, written at the end of the launch
  int n;
  int t_first, n_steps;
  const float* cproj; int ldc;       // [n][H0]
  const float* temb; int ldt;        // [T][H0]
  const float* coef;                 // [T][4]
  const float* z; int ldzz; long long z_step_stride; int z_t_first;
  uint64_t seed; uint32_t row_offset;
  float* mut_mask; int mutation_dim;
  float* xs; long long xs_stride;    // per panel: the state in unit order [4 T32][64][4]
  float* slab; long long slab_stride;// per panel: input_proj partials [8][H0 / 32][4][64][4]
  float* act; long long act_stride;  // per panel: the layers' outputs in unit order
  unsigned* bar;                     // [n_panels][16] arrival counters (64 B apart), zero at launch
  unsigned* status;
  unsigned long long spin_budget;
  unsigned long long* stamps;        // diagnostic builds: per wave 16 counters of shader cycles (tools/squad_stamps.py), else null
};

typedef int v4i32 __attribute__((ext_vector_type(4)));

// 16-byte accesses through a buffer descriptor: the lane's byte offset in ONE register (16 lane for unit order), the unit's position
// as a scalar offset -- no per-load 64-bit VALU address (VALU work comes out of the fp32 matrix loop's time) and no address
// registers per slot in flight.  _sc1: agent scope.  The compiler keeps its own vmcnt book for these.
__device__ __forceinline__ v4f sq_ld(__amdgpu_buffer_rsrc_t r, int lane_off, int s_off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, s_off, 0));
}
__device__ __forceinline__ v4f sq_ld_sc1(__amdgpu_buffer_rsrc_t r, int lane_off, int s_off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, s_off, 16));      // aux 16 = sc1
}
__device__ __forceinline__ void sq_st(__amdgpu_buffer_rsrc_t r, int lane_off, int s_off, v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, v), r, lane_off, s_off, 0);
}
__device__ __forceinline__ void sq_st_sc1(__amdgpu_buffer_rsrc_t r, int lane_off, int s_off, v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, v), r, lane_off, s_off, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sq_rsrc(const float* p, long long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// One wave's K loop: acc[fb] += A(fb, i) x B(i) over i in [0, n8), DEPTH blocks in flight.  la / lb must be callable for any
// i in [0, n8) and free of side effects (a refill beyond the end re-reads the last block).  aq arrives primed (sq_prime_a: the
// weights do not depend on the barrier in front of a phase and fly while wave 0 polls), bq is primed here.
template <int NFB, int SQ_DEPTH, class LA>
__device__ __forceinline__ void sq_prime_a(v4f (&aq)[SQ_DEPTH][2], int n8, const LA& la) {
#pragma unroll
  for (int d = 0; d < SQ_DEPTH; ++d) {
    const int i = d < n8 ? d : n8 - 1;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = la(fb, i);
  }
}
// hipcc's wait insertion handles this shape well and little else: the group that consumes the primed slots is peeled, so that both
// ways into the loop (from the peeled group, from the back edge) leave the same loads outstanding in the same order and every
// wait inside is "all but the youngest 3 DEPTH - 3" -- with the primes meeting the refills at the loop header it merges the two
// histories into vmcnt(7) or vmcnt(0) per group and the stream stops DEPTH times per phase (seen in the ISA of the first version).
template <int NFB, int SQ_DEPTH, class LA, class LB>
__device__ __forceinline__ void sq_kloop(f32x16 (&acc)[NFB][1], v4f (&aq)[SQ_DEPTH][2], int n8, const LA& la, const LB& lb) {
  v4f bq[SQ_DEPTH];
#pragma unroll
  for (int d = 0; d < SQ_DEPTH; ++d) bq[d] = lb(d < n8 ? d : n8 - 1);
  auto group = [&](int i0) {                          // a whole group with a successor: MFMAs, then the slot's refill (unconditional loads)
#pragma unroll
    for (int d = 0; d < SQ_DEPTH; ++d) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          acc[fb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d][fb][e], bq[d][e], acc[fb][0], 0, 0, 0);
      const int in = i0 + d + SQ_DEPTH < n8 ? i0 + d + SQ_DEPTH : n8 - 1;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = la(fb, in);
      bq[d] = lb(in);
      __builtin_amdgcn_sched_barrier(0);              // the refills stay in slot order: every wait in the loop is "all but the youngest ones"
    }
  };
  int i0 = 0;
  if (SQ_DEPTH < n8) {
    group(0);
    for (i0 = SQ_DEPTH; i0 + SQ_DEPTH < n8; i0 += SQ_DEPTH) group(i0);
  }
#pragma unroll
  for (int d = 0; d < SQ_DEPTH; ++d) {               // the last group (possibly partial): no refill
    if (i0 + d < n8) {                               // uniform
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          acc[fb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[d][fb][e], bq[d][e], acc[fb][0], 0, 0, 0);
    }
  }
}

// chain_wait with a short nap: a squad barrier is passed 12 times per step and its partners arrive within a microsecond or two
// (chain.h's 32 x 64-cycle sleep between polls is most of such a wait); 8 pollers per counter, every squad its own cache line.
__device__ __forceinline__ bool squad_wait(const unsigned* word, unsigned want, unsigned* status, unsigned long long budget, int lane) {
  if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(word)) >= want) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned it = 1;; ++it) {
    __builtin_amdgcn_s_sleep(1);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(word)) >= want) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
      if (lane == 0) st_relaxed_agent(status, CHAIN_TIMEOUT);
      return false;
    }
    if ((it & 31u) == 0 && (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(status)) != CHAIN_OK) return false;      // a partner gave up / the host's abort
  }
}

// sum over the 8 lanes of an aligned group (DPP: xor 1, xor 2, mirror within 8), every lane gets the total
__device__ __forceinline__ float sq_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
}

template <int WPC, bool STAMP = false>      // WPC: workgroups per CU the build is sized for (registers)
__global__ __launch_bounds__(SQ_THREADS, WPC) void squad_chain_kernel(const SquadArgs* __restrict__ gp) {
  constexpr int SQ_DEPTH = SquadDepth<WPC>::value;
  const SquadArgs& a = *gp;
  unsigned long long cyc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long { if constexpr (STAMP) { asm volatile("s_nop 0" ::: "memory"); return __builtin_amdgcn_s_memtime(); } else return 0ull; };
  const unsigned long long c_start = now();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const stage = smem;
  float* const prm = smem + SQ_STAGE_FLOATS;
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + SQ_STAGE_FLOATS + a.n_layers * SQ_PRM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int panel = blockIdx.x >> 3, g = blockIdx.x & 7;
  const int p0 = panel * SQ_RP;
  const int row = p0 + l31;
  const int rowc = row < a.n ? row : a.n - 1;
  unsigned* const bar = a.bar + (size_t)panel * 16;
  unsigned nb = 0;                                    // squad barriers passed

  const int T32 = a.T32, D = a.D;
  const int t0 = g * T32 / SQ_S, t1 = (g + 1) * T32 / SQ_S;      // this workgroup's 32-feature tiles of the state
  float* const xs = a.xs + (size_t)panel * a.xs_stride;
  const __amdgpu_buffer_rsrc_t r_act = sq_rsrc(a.act + (size_t)panel * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_xs = sq_rsrc(xs, a.xs_stride);
  const __amdgpu_buffer_rsrc_t r_w = sq_rsrc(a.wpk, a.wpk_floats);
  const int l16 = 16 * lane;
  const __amdgpu_buffer_rsrc_t r_slab = sq_rsrc(a.slab + (size_t)panel * a.slab_stride, a.slab_stride);

  // All waves: own stores done, workgroup barrier; wave 0 arrives for the workgroup; `prime` (the next phase's weight loads: they
  // do not depend on the squad) is issued by every wave; wave 0 waits for the squad.
  auto squad_sync = [&](auto&& prime) -> bool {
    SQ_DRAIN_BARRIER();
    ++nb;
    if (wave == 0 && lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prime();
    if (wave == 0) {
      const bool ok = squad_wait(bar, SQ_S * nb, a.status, a.spin_budget, lane);
      s_flag = ok ? 1 : 0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int go = __builtin_amdgcn_readfirstlane(s_flag);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // s_flag may be rewritten only after everyone has read it
    return go != 0;
  };

  // ---- once per launch: this workgroup's parameters into LDS, loop invariants into registers, the state into unit order ----
  for (int l = 0; l < a.n_layers; ++l) {
    const SquadLayer& L = a.L[l];
    const int fs = L.F / SQ_S;
    if (tid < 3 * fs) {
      const int arr = tid / fs, j = tid % fs;
      const float* src = arr == 0 ? L.bias : (arr == 1 ? L.gamma : L.beta);
      prm[l * SQ_PRM + arr * 64 + j] = src[g * fs + j];
    }
  }
  // reduce phase: thread (wave = q, lane) owns the float4 at features 32 g + 8 q + 4 h of its row (H0 = 256: one 32-feature block per workgroup)
  const int fr = 32 * g + 8 * wave + 4 * h;
  const float4 r_bias = ldg4(a.bias_in + fr);
  const float4 r_cproj = ldg4(a.cproj + (size_t)rowc * a.ldc + fr);
  {
    const float* xrow = a.x + (size_t)rowc * a.ldx;
    for (int u = 4 * t0 + wave; u < 4 * t1; u += 4) {      // unit u: features 8 u + 4 h .. + 3
      const int f = 8 * u + 4 * h;
      float4 v;
      v.x = f < D ? xrow[f] : 0.f;
      v.y = f + 1 < D ? xrow[f + 1] : 0.f;
      v.z = f + 2 < D ? xrow[f + 2] : 0.f;
      v.w = f + 3 < D ? xrow[f + 3] : 0.f;
      sq_st(r_xs, l16, u * 1024, v4f{v.x, v.y, v.z, v.w});
    }
  }
  SQ_DRAIN_BARRIER();

  v4f aq[SQ_DEPTH][2];
  // input_proj's weight stream of this wave: feature blocks 2 wave, 2 wave + 1; the workgroup's 8-k blocks 4 t0 .. 4 t1
  const int K8i = 4 * T32;
  const int n8i = 4 * (t1 - t0);
  const int wi = a.in_off * 4 + ((2 * wave) * K8i + 4 * t0) * 1024;      // bytes
  auto la_in = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wi + (fb * K8i + i) * 1024); };
  sq_prime_a<2, SQ_DEPTH>(aq, n8i, la_in);

  for (int si = 0; si < a.n_steps; ++si) {
    const int t = a.t_first - si;
    unsigned long long tc = now(), tn;
#define SQ_STAMP(i) do { if constexpr (STAMP) { tn = now(); cyc[i] += tn - tc; tc = tn; } } while (0)
    // =============================== input_proj: partial sums over this workgroup's state features ===============================
    {
      f32x16 acc[2][1];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
      auto lb = [&](int i) -> v4f { return sq_ld(r_xs, l16, (4 * t0 + i) * 1024); };
      sq_kloop<2, SQ_DEPTH>(acc, aq, n8i, la_in, lb);
      // slab [g][fb 0..7][q][lane]
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f v = {acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]};
          sq_st_sc1(r_slab, l16, ((g * 8 + 2 * wave + fb) * 4 + q) * 1024, v);
        }
    }
    const float4 r_temb = ldg4(a.temb + (size_t)t * a.ldt + fr);
    auto layer_a = [&](const SquadLayer& L, int nfb) {
      const int n8q = L.K8 / 4;
      return L.w_off * 4 + ((g * nfb) * L.K8 + wave * n8q) * 1024;          // bytes, uniform
    };
    auto prime_layer = [&](int l) {
      const SquadLayer& L = a.L[l];
      const int wl = layer_a(L, L.F / 256);
      const int K8 = L.K8;
      auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
      if (L.F == 512) sq_prime_a<2, SQ_DEPTH>(aq, K8 / 4, la); else sq_prime_a<1, SQ_DEPTH>(aq, K8 / 4, la);
    };
    auto no_prime = [] {};
    SQ_STAMP(0);
    if (!squad_sync([&] { prime_layer(0); })) return;       // the first layer's weights stay in flight over the reduce phase
    SQ_STAMP(1);
    // =============================== reduce: h0 = ((sum + b) + temb[t]) + cproj ===============================
    {
      v4f p[SQ_S];
#pragma unroll
      for (int s = 0; s < SQ_S; ++s) p[s] = sq_ld_sc1(r_slab, l16, ((s * 8 + g) * 4 + wave) * 1024);
      v4f sum = p[0];
#pragma unroll
      for (int s = 1; s < SQ_S; ++s) sum += p[s];
      v4f o;
      o.x = ((sum.x + r_bias.x) + r_temb.x) + r_cproj.x;
      o.y = ((sum.y + r_bias.y) + r_temb.y) + r_cproj.y;
      o.z = ((sum.z + r_bias.z) + r_temb.z) + r_cproj.z;
      o.w = ((sum.w + r_bias.w) + r_temb.w) + r_cproj.w;
      sq_st_sc1(r_act, l16, a.h0_out * 4 + (4 * g + wave) * 1024, o);
    }
    SQ_STAMP(2);
    if (!squad_sync(no_prime)) return;
    SQ_STAMP(3);

    // =============================== Linear + GroupNorm + SiLU layers ===============================
    for (int l = 0; l < a.n_layers; ++l) {
      const SquadLayer& L = a.L[l];
      auto run = [&](auto nfb_tag) {
        constexpr int NFB = decltype(nfb_tag)::value;
        const int K8 = L.K8, n8q = K8 / 4;
        const int wl = layer_a(L, NFB);
        auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
        const int i_first = wave * n8q;
        const int n8_0 = L.n8_0, in0 = L.in0, in1 = L.in1;
        auto lb = [&](int i) -> v4f {
          const int ig = i_first + i;
          const int off = ig < n8_0 ? in0 + ig * 256 : in1 + (ig - n8_0) * 256;      // uniform
          return sq_ld_sc1(r_act, l16, off * 4);
        };
        f32x16 acc[NFB][1];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
        sq_kloop<NFB, SQ_DEPTH>(acc, aq, n8q, la, lb);
        SQ_STAMP(4);
        // All four partial accumulators -> LDS as [wave][patient][feature] (row stride 4 mod 32 dwords: the b128 accesses of both
        // sides spread over the banks).  Then the epilogue runs on all 256 threads, 8 threads per patient: thread (row, c) owns
        // features 4 c .. + 3 of every 32-feature block of the group -- sum of the partials in wave order, bias, the row's GroupNorm statistics over
        // its 8 lanes (DPP), SiLU (chain_gn_silu's formulas; one wave doing the whole tile measured 1.0 us per layer).
        constexpr int LDP = 32 * NFB + 4, NE = 4 * NFB, GW = 32 * NFB;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(stage + (wave * 32 + l31) * LDP + 32 * fb + 8 * q + 4 * h) =
                make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
          const int erow = tid >> 3, c = tid & 7, f0 = 4 * c;      // the thread's j-th float4: features 32 j + 4 c .. + 3 (8 lanes = one 128-byte row: no bank conflicts)
          const float* pl = prm + l * SQ_PRM;
          float v[NE];
#pragma unroll
          for (int j = 0; j < NFB; ++j) {
            float4 sum = *reinterpret_cast<const float4*>(stage + erow * LDP + f0 + 32 * j);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
              const float4 pv = *reinterpret_cast<const float4*>(stage + (w * 32 + erow) * LDP + f0 + 32 * j);
              sum.x += pv.x; sum.y += pv.y; sum.z += pv.z; sum.w += pv.w;
            }
            const float4 bv = *reinterpret_cast<const float4*>(pl + f0 + 32 * j);
            v[4 * j] = sum.x + bv.x; v[4 * j + 1] = sum.y + bv.y; v[4 * j + 2] = sum.z + bv.z; v[4 * j + 3] = sum.w + bv.w;
          }
          float sm = 0.f;
#pragma unroll
          for (int e = 0; e < NE; ++e) sm += v[e];
          const float mean = sq_sum8(sm) * (1.0f / GW);
          float qs = 0.f;
#pragma unroll
          for (int e = 0; e < NE; ++e) { const float d = v[e] - mean; qs = fmaf(d, d, qs); }
          const float rstd = 1.0f / sqrtf(sq_sum8(qs) * (1.0f / GW) + GN_EPS);
#pragma unroll
          for (int j = 0; j < NFB; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(pl + 64 + f0 + 32 * j);
            const float4 bev = *reinterpret_cast<const float4*>(pl + 128 + f0 + 32 * j);
            v4f y;
            y.x = silu_f(fmaf((v[4 * j] - mean) * rstd, gv.x, bev.x));
            y.y = silu_f(fmaf((v[4 * j + 1] - mean) * rstd, gv.y, bev.y));
            y.z = silu_f(fmaf((v[4 * j + 2] - mean) * rstd, gv.z, bev.z));
            y.w = silu_f(fmaf((v[4 * j + 3] - mean) * rstd, gv.w, bev.w));
            const int f = f0 + 32 * j;                      // feature inside the group: unit (f / 8), half (f / 4) & 1
            const int unit = g * NFB * 4 + (f >> 3), ln = erow + 32 * ((f >> 2) & 1);
            sq_st_sc1(r_act, 16 * ln, L.out * 4 + unit * 1024, y);
          }
        }
      };
      if (L.F == 512) run(std::integral_constant<int, 2>{});
      else run(std::integral_constant<int, 1>{});
      const bool more = l + 1 < a.n_layers;
      SQ_STAMP(5);
      if (!squad_sync([&] { if (more) prime_layer(l + 1); })) return;
      SQ_STAMP(6);
    }

    // =============================== output_proj + posterior on this workgroup's state tiles ===============================
    {
      const int K8o = a.K8_out;                 // 32
      // the operand (32 patients x hl) into LDS in unit order: 256 threads x K8o / 4 float4
      for (int u = wave; u < K8o; u += 4) {
        const v4f v = sq_ld_sc1(r_act, l16, a.last_in * 4 + u * 1024);
        *reinterpret_cast<v4f*>(stage + (u * 64 + lane) * 4) = v;
      }
      const float* c = a.coef + 4 * t;
      const float cA = c[0], cB = c[1], cC = c[2];
      const bool last_step = si + 1 == a.n_steps;
      const bool do_mask = t == 0 && a.mut_mask != nullptr;
      const float* zbase = a.z ? a.z + (long long)(a.z_t_first - t) * a.z_step_stride : nullptr;
      SQ_DRAIN_BARRIER();
      SQ_STAMP(7);
      // one float4 of the posterior update (EpiPosterior::apply's arithmetic): features 32 tile + 8 q + 4 h .. + 3 of this lane's patient
      auto post_unit = [&](int tile, int q, float e0, float e1, float e2, float e3, float4 xv4, float4 bv) {
        const int f = 32 * tile + 8 * q + 4 * h;
        const float e[4] = {e0 + bv.x, e1 + bv.y, e2 + bv.z, e3 + bv.w};
        const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        float zv[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
          if (zbase) {
            const float* zr = zbase + (size_t)rowc * a.ldzz;
#pragma unroll
            for (int r = 0; r < 4; ++r) zv[r] = f + r < D ? zr[f + r] : 0.f;
          } else {
            const float4 zz = randn4(a.seed, a.row_offset + (uint32_t)row, (uint32_t)(f >> 2), (uint32_t)t, TAG_POSTERIOR);
            zv[0] = zz.x; zv[1] = zz.y; zv[2] = zz.z; zv[3] = zz.w;
          }
        }
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = fmaf(cA, xv[r], fmaf(cB, e[r], cC * zv[r]));
          if (f + r >= D) o[r] = 0.f;           // pad features of the last tile stay zero
        }
        sq_st(r_xs, l16, (4 * tile + q) * 1024, v4f{o[0], o[1], o[2], o[3]});
        if (row < a.n) {
          if (do_mask && f < a.mutation_dim) {
            float* mrow = a.mut_mask + (size_t)row * a.mutation_dim;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < a.mutation_dim) stg1(mrow + f + r, (o[r] > 0.5f) ? 1.0f : 0.0f);
          }
          if (last_step) {
            float* xrow = a.x + (size_t)row * a.ldx;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < D) stg1(xrow + f + r, o[r]);
          }
        }
      };
      auto lb = [&](int i) -> v4f { return *reinterpret_cast<const v4f*>(stage + (i * 64 + lane) * 4); };
      // Whole rounds: a tile per wave, full K.  One or two left-over tiles are split four ways over K instead (a fifth round for
      // one wave would keep the other seven workgroups of the squad waiting at the next barrier: 6 us per step at 161 tiles);
      // three left-over tiles are a round of their own.
      const int nt = t1 - t0, rem = nt & 3;
      const int rounds = (nt >> 2) + (rem == 3 ? 1 : 0), n_split = rem == 3 ? 0 : rem;
      auto tile_a = [&](int tile) { return a.out_off * 4 + tile * K8o * 1024; };      // bytes, uniform
      const int n8s = K8o / 4;                    // a wave's 8-k blocks of a K-split tile
      auto split_prime = [&](int r) {
        const int wo = tile_a(t0 + 4 * rounds + r) + (wave * n8s) * 1024;
        sq_prime_a<1, SQ_DEPTH>(aq, n8s, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wo + i * 1024); });
      };
      if (rounds == 0 && n_split > 0) split_prime(0);
      {
        const int tile0 = t0 + wave;
        if (rounds > 0 && tile0 < t1) {
          const int wo = tile_a(tile0);
          sq_prime_a<1, SQ_DEPTH>(aq, K8o, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wo + i * 1024); });
        }
      }
      for (int j = 0; j < rounds; ++j) {
        const int tile = t0 + wave + 4 * j;
        if (tile >= t0 + 4 * rounds || tile >= t1) break;                  // rem == 3: wave 3 sits the last round out (uniform per wave)
        const int wo = tile_a(tile);
        auto la = [&](int fb, int i) -> v4f { (void)fb; return sq_ld(r_w, l16, wo + i * 1024); };
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        // x_t of the tile (this wave wrote it one step ago) and the bias fly under the K loop
        float4 xq[4], bq4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          { const v4f xv_ = sq_ld(r_xs, l16, (4 * tile + q) * 1024); xq[q] = make_float4(xv_.x, xv_.y, xv_.z, xv_.w); }
          bq4[q] = ldg4(a.bias_out + 32 * tile + 8 * q + 4 * h);
        }
        sq_kloop<1, SQ_DEPTH>(acc, aq, K8o, la, lb);
        // the next tile's first weights fly under this tile's epilogue
        const int tile_n = tile + 4;
        if (j + 1 < rounds && tile_n < t1) {
          const int wn = tile_a(tile_n);
          sq_prime_a<1, SQ_DEPTH>(aq, K8o, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wn + i * 1024); });
        } else if (j + 1 == rounds && n_split > 0) {
          split_prime(0);
        }
        SQ_STAMP(8);
#pragma unroll
        for (int q = 0; q < 4; ++q) post_unit(tile, q, acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3], xq[q], bq4[q]);
        SQ_STAMP(9);
      }
      for (int r = 0; r < n_split; ++r) {
        const int tile = t0 + 4 * rounds + r;
        const int n8q = n8s;
        const int wo = tile_a(tile) + (wave * n8q) * 1024;
        auto la = [&](int fb, int i) -> v4f { (void)fb; return sq_ld(r_w, l16, wo + i * 1024); };
        auto lbq = [&](int i) -> v4f { return *reinterpret_cast<const v4f*>(stage + ((wave * n8q + i) * 64 + lane) * 4); };
        const v4f xw_ = sq_ld(r_xs, l16, (4 * tile + wave) * 1024);                      // unit q = wave is this wave's to finish
        const float4 xw = make_float4(xw_.x, xw_.y, xw_.z, xw_.w);
        const float4 bw = ldg4(a.bias_out + 32 * tile + 8 * wave + 4 * h);
        f32x16 acc[1][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][0][i] = 0.f;
        sq_kloop<1, SQ_DEPTH>(acc, aq, n8q, la, lbq);
        if (r + 1 < n_split) split_prime(r + 1);
        // every wave hands the three units it does not finish to their owners: red[q][slot of the writer among the other three][lane]
        float* const red = stage + 32 * 256;
        if (r > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous left-over tile's partials have been read
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q != wave)
            *reinterpret_cast<float4*>(red + ((q * 3 + (wave - (wave > q ? 1 : 0))) * 64 + lane) * 4) =
                make_float4(acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // sum in wave order 0..3 (own partial in its place)
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float4 pv;
          if (w == wave) {
            float own[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q == wave) { own[0] = acc[0][0][4 * q]; own[1] = acc[0][0][4 * q + 1]; own[2] = acc[0][0][4 * q + 2]; own[3] = acc[0][0][4 * q + 3]; }
            pv = make_float4(own[0], own[1], own[2], own[3]);
          } else {
            pv = *reinterpret_cast<const float4*>(red + ((wave * 3 + (w - (w > wave ? 1 : 0))) * 64 + lane) * 4);
          }
          if (w == 0) { sum[0] = pv.x; sum[1] = pv.y; sum[2] = pv.z; sum[3] = pv.w; }
          else { sum[0] += pv.x; sum[1] += pv.y; sum[2] += pv.z; sum[3] += pv.w; }
        }
        SQ_STAMP(8);
        post_unit(tile, wave, sum[0], sum[1], sum[2], sum[3], xw, bw);
        SQ_STAMP(9);
      }
      // the state this workgroup just wrote is read by all four waves; then the next step's input_proj stream
      SQ_DRAIN_BARRIER();
      sq_prime_a<2, SQ_DEPTH>(aq, n8i, la_in);
      SQ_STAMP(10);
    }
  }
#undef SQ_STAMP
  if constexpr (STAMP) {
    if (a.stamps && lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
      for (int i = 0; i < 11; ++i) o[i] = cyc[i];
      o[11] = now() - c_start;
      o[12] = (unsigned long long)a.n_steps;
    }
  }
}

}  // namespace osd
