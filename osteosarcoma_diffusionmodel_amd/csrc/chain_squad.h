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
#ifndef SQ_DEPTH
#define SQ_DEPTH 8                   // 8-k blocks a wave keeps in flight
#endif
constexpr int SQ_PRM = 3 * 64;       // per layer in LDS: bias | gamma | beta of this workgroup's group (<= 64 features)
constexpr int SQ_STAGE_FLOATS = 32 * 256;      // output_proj's operand (last width 256) in unit order; the partial accumulators (3 x 2 x 1024) share it
constexpr int SQ_LDS_FLOATS = SQ_STAGE_FLOATS + SQ_MAX_LAYERS * SQ_PRM + 64 + 16;
constexpr int SQ_LDS_BYTES = SQ_LDS_FLOATS * 4;

struct SquadLayer {
  const float* wpk;                  // fragment-ordered weights [F / 32][K8][64][4]
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
  const float* wpk_in; const float* bias_in; int H0;     // [H0 / 32][4 T32][64][4]
  const float* wpk_out; const float* bias_out;           // [T32][hl / 8][64][4]; bias padded to 32 T32 floats
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
};

typedef int v4i32 __attribute__((ext_vector_type(4)));

// agent-scope 16-byte accesses through a buffer descriptor (the compiler keeps its own vmcnt book for these)
__device__ __forceinline__ v4f sq_ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);      // aux 16 = sc1
  return __builtin_bit_cast(v4f, v);
}
__device__ __forceinline__ void sq_st_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, v4f v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sq_rsrc(const float* p, long long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// One wave's K loop: acc[fb] += A(fb, i) x B(i) over i in [0, n8), DEPTH blocks in flight.  la / lb must be callable for any
// i in [0, n8) and free of side effects (a refill beyond the end re-reads the last block).  aq arrives primed (sq_prime_a: the
// weights do not depend on the barrier in front of a phase and fly while wave 0 polls), bq is primed here.
template <int NFB, class LA>
__device__ __forceinline__ void sq_prime_a(v4f (&aq)[SQ_DEPTH][2], int n8, const LA& la) {
#pragma unroll
  for (int d = 0; d < SQ_DEPTH; ++d) {
    const int i = d < n8 ? d : n8 - 1;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = la(fb, i);
  }
}
template <int NFB, class LA, class LB>
__device__ __forceinline__ void sq_kloop(f32x16 (&acc)[NFB][1], v4f (&aq)[SQ_DEPTH][2], int n8, const LA& la, const LB& lb) {
  v4f bq[SQ_DEPTH];
#pragma unroll
  for (int d = 0; d < SQ_DEPTH; ++d) bq[d] = lb(d < n8 ? d : n8 - 1);
  int i0 = 0;
  for (; i0 + SQ_DEPTH < n8; i0 += SQ_DEPTH) {      // whole groups with a successor: MFMAs, then the slot's refill (unconditional loads)
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
    }
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

struct SquadOut {                   // chain_gn_silu's sink: fragments -> the output buffer in unit order (agent-scope stores)
  __amdgpu_buffer_rsrc_t r; int byte0;           // the lane's slot of (this workgroup's first feature block, q = 0)
  __device__ __forceinline__ void put(int fb, int pb, int q, int l31, int h, float4 v) const {
    (void)pb; (void)l31; (void)h;
    const v4f w = {v.x, v.y, v.z, v.w};
    sq_st_sc1(r, byte0 + (fb * 4 + q) * 1024, w);
  }
  __device__ __forceinline__ void flush(int fb, int lane) const { (void)fb; (void)lane; }
};

template <int WPC>                   // workgroups per CU the build is sized for (registers)
__global__ __launch_bounds__(SQ_THREADS, WPC) void squad_chain_kernel(const SquadArgs* __restrict__ gp) {
  const SquadArgs& a = *gp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const stage = smem;
  float* const prm = smem + SQ_STAGE_FLOATS;
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + SQ_STAGE_FLOATS + SQ_MAX_LAYERS * SQ_PRM);

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
  const __amdgpu_buffer_rsrc_t r_slab = sq_rsrc(a.slab + (size_t)panel * a.slab_stride, a.slab_stride);

  // All waves: own stores done, workgroup barrier; wave 0 arrives for the workgroup; `prime` (the next phase's weight loads: they
  // do not depend on the squad) is issued by every wave; wave 0 waits for the squad.
  auto squad_sync = [&](auto&& prime) -> bool {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ++nb;
    if (wave == 0 && lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prime();
    if (wave == 0) {
      const bool ok = chain_wait(bar, SQ_S * nb, a.status, a.spin_budget, lane);
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
      stg4(xs + (size_t)u * 256 + 4 * lane, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  v4f aq[SQ_DEPTH][2];
  // input_proj's weight stream of this wave: feature blocks 2 wave, 2 wave + 1; the workgroup's 8-k blocks 4 t0 .. 4 t1
  const int K8i = 4 * T32;
  const int n8i = 4 * (t1 - t0);
  const gv4f_ptr wi = (gv4f_ptr)(a.wpk_in) + ((size_t)(2 * wave) * K8i + 4 * t0) * 64 + lane;
  auto la_in = [&](int fb, int i) -> v4f { return wi[((size_t)fb * K8i + i) * 64]; };
  sq_prime_a<2>(aq, n8i, la_in);

  for (int si = 0; si < a.n_steps; ++si) {
    const int t = a.t_first - si;
    // =============================== input_proj: partial sums over this workgroup's state features ===============================
    {
      f32x16 acc[2][1];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
      const gv4f_ptr xb = (gv4f_ptr)(xs) + (size_t)(4 * t0) * 64 + lane;
      auto lb = [&](int i) -> v4f { return xb[(size_t)i * 64]; };
      sq_kloop<2>(acc, aq, n8i, la_in, lb);
      // slab [g][fb 0..7][q][lane]
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f v = {acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]};
          sq_st_sc1(r_slab, (((g * 8 + 2 * wave + fb) * 4 + q) * 64 + lane) * 16, v);
        }
    }
    const float4 r_temb = ldg4(a.temb + (size_t)t * a.ldt + fr);
    auto layer_a = [&](const SquadLayer& L, int nfb) {
      const int n8q = L.K8 / 4;
      return (gv4f_ptr)(L.wpk) + ((size_t)(g * nfb) * L.K8 + wave * n8q) * 64 + lane;
    };
    auto prime_layer = [&](int l) {
      const SquadLayer& L = a.L[l];
      const gv4f_ptr wl = layer_a(L, L.F / 256);
      const int K8 = L.K8;
      auto la = [&](int fb, int i) -> v4f { return wl[((size_t)fb * K8 + i) * 64]; };
      if (L.F == 512) sq_prime_a<2>(aq, K8 / 4, la); else sq_prime_a<1>(aq, K8 / 4, la);
    };
    auto no_prime = [] {};
    if (!squad_sync([&] { prime_layer(0); })) return;       // the first layer's weights stay in flight over the reduce phase
    // =============================== reduce: h0 = ((sum + b) + temb[t]) + cproj ===============================
    {
      v4f p[SQ_S];
#pragma unroll
      for (int s = 0; s < SQ_S; ++s) p[s] = sq_ld_sc1(r_slab, (((s * 8 + g) * 4 + wave) * 64 + lane) * 16);
      v4f sum = p[0];
#pragma unroll
      for (int s = 1; s < SQ_S; ++s) sum += p[s];
      v4f o;
      o.x = ((sum.x + r_bias.x) + r_temb.x) + r_cproj.x;
      o.y = ((sum.y + r_bias.y) + r_temb.y) + r_cproj.y;
      o.z = ((sum.z + r_bias.z) + r_temb.z) + r_cproj.z;
      o.w = ((sum.w + r_bias.w) + r_temb.w) + r_cproj.w;
      sq_st_sc1(r_act, (a.h0_out + ((4 * g + wave) * 64 + lane) * 4) * 4, o);
    }
    if (!squad_sync(no_prime)) return;

    // =============================== Linear + GroupNorm + SiLU layers ===============================
    for (int l = 0; l < a.n_layers; ++l) {
      const SquadLayer& L = a.L[l];
      auto run = [&](auto nfb_tag) {
        constexpr int NFB = decltype(nfb_tag)::value;
        const int K8 = L.K8, n8q = K8 / 4;
        const gv4f_ptr wl = layer_a(L, NFB);
        auto la = [&](int fb, int i) -> v4f { return wl[((size_t)fb * K8 + i) * 64]; };
        const int i_first = wave * n8q;
        const int n8_0 = L.n8_0, in0 = L.in0, in1 = L.in1;
        auto lb = [&](int i) -> v4f {
          const int ig = i_first + i;
          const int off = ig < n8_0 ? in0 + ig * 256 : in1 + (ig - n8_0) * 256;      // uniform
          return sq_ld_sc1(r_act, (off + 4 * lane) * 4);
        };
        f32x16 acc[NFB][1];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
        sq_kloop<NFB>(acc, aq, n8q, la, lb);
        // partial accumulators of waves 1..3 -> LDS [(w - 1) NFB + fb][q][lane]
        if (wave > 0) {
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<float4*>(stage + ((((wave - 1) * NFB + fb) * 4 + q) * 64 + lane) * 4) =
                  make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave == 0) {
#pragma unroll
          for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(stage + (((w * NFB + fb) * 4 + q) * 64 + lane) * 4);
                acc[fb][0][4 * q] += v.x; acc[fb][0][4 * q + 1] += v.y; acc[fb][0][4 * q + 2] += v.z; acc[fb][0][4 * q + 3] += v.w;
              }
          const SquadOut o{r_act, (L.out + ((g * NFB * 4) * 64 + lane) * 4) * 4};
          chain_gn_silu<32 * NFB, NFB, 1, 64>(acc, prm + l * SQ_PRM, 0, o, lane);
        }
      };
      if (L.F == 512) run(std::integral_constant<int, 2>{});
      else run(std::integral_constant<int, 1>{});
      const bool more = l + 1 < a.n_layers;
      if (!squad_sync([&] { if (more) prime_layer(l + 1); })) return;
    }

    // =============================== output_proj + posterior on this workgroup's state tiles ===============================
    {
      const int K8o = a.K8_out;                 // 32
      // the operand (32 patients x hl) into LDS in unit order: 256 threads x K8o / 4 float4
      for (int u = wave; u < K8o; u += 4) {
        const v4f v = sq_ld_sc1(r_act, (a.last_in + (u * 64 + lane) * 4) * 4);
        *reinterpret_cast<v4f*>(stage + (u * 64 + lane) * 4) = v;
      }
      const float* c = a.coef + 4 * t;
      const float cA = c[0], cB = c[1], cC = c[2];
      const bool last_step = si + 1 == a.n_steps;
      const bool do_mask = t == 0 && a.mut_mask != nullptr;
      const float* zbase = a.z ? a.z + (long long)(a.z_t_first - t) * a.z_step_stride : nullptr;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      for (int tile = t0 + wave; tile < t1; tile += 4) {
        const gv4f_ptr wo = (gv4f_ptr)(a.wpk_out) + (size_t)tile * K8o * 64 + lane;
        auto la = [&](int fb, int i) -> v4f { (void)fb; return wo[(size_t)i * 64]; };
        auto lb = [&](int i) -> v4f { return *reinterpret_cast<const v4f*>(stage + (i * 64 + lane) * 4); };
        sq_prime_a<1>(aq, K8o, la);
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        // x_t of the tile (this wave wrote it one step ago) and the bias fly under the K loop
        float4 xq[4], bq4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xq[q] = ldg4(xs + (size_t)(4 * tile + q) * 256 + 4 * lane);
          bq4[q] = ldg4(a.bias_out + 32 * tile + 8 * q + 4 * h);
        }
        sq_kloop<1>(acc, aq, K8o, la, lb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * tile + 8 * q + 4 * h;
          const float e[4] = {acc[0][0][4 * q] + bq4[q].x, acc[0][0][4 * q + 1] + bq4[q].y, acc[0][0][4 * q + 2] + bq4[q].z, acc[0][0][4 * q + 3] + bq4[q].w};
          const float xv[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
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
          stg4(xs + (size_t)(4 * tile + q) * 256 + 4 * lane, make_float4(o[0], o[1], o[2], o[3]));
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
        }
      }
      // the state this workgroup just wrote is read by all four waves; then the next step's input_proj stream
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      sq_prime_a<2>(aq, n8i, la_in);
    }
  }
}

}  // namespace osd
