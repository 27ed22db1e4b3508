// train_squad.h -- the trunk of a TRAINING forward pass (models/diffusion.py:232-251: the ten Linear + GroupNorm + SiLU (+ Dropout)
// layers between input_proj and output_proj) as ONE launch of squads (chain_squad.h): eight workgroups per 64 patients, workgroup g =
// GroupNorm group g of every layer, activations handed between them in MFMA operand order with agent-scope loads / stores behind a
// per-squad barrier.
//
// At batch 4096 the per-layer forward is ten launches of 12-28 us for 3.4-14 us of matrix work each (256-1024 workgroups of 8-32
// K steps: launch boundary, ramp, prologue and epilogue of every workgroup at the same moment): 196 of the step's 920 us.  Here a
// squad walks all ten layers: 512 workgroups = two per CU, one squad's hand-off (~2.5 us) under the other's matrix work.
//   * 64-patient panels = two 32-patient sub-panels with chain_squad.h's unit-order buffers; wave w of a workgroup takes sub-panel
//     w & 1 and K-half w >> 1: 32 x (C / 8) accumulators over half of K, both halves meet in LDS;
//   * weights change every step: their fragment-ordered copies (6.3 MB) are remade by ONE launch per step (k_pack_fragments_multi).
//     Reading the parameters where they are -- lane (l31, h) loading W[f0 + l31][8 i + 4 h .. + 3], a different row per lane, 32
//     cache lines per wave instruction -- was tried first: 205 us for the launch against the per-layer kernels' 196;
//   * the epilogue (256 threads, 8 per patient: sum of the two K-halves, bias, the row's GroupNorm statistics by DPP, SiLU,
//     dropout from an injected mask or from Philox at the per-layer kernels' address (row, feature / 4, step, block tag)) writes
//     what the backward pass reads -- z = pre-norm activations, (mean, rstd) per (row, group), the layer's output, all row-major
//     as EpiGnSilu leaves them -- and the output once more in unit order for the squad's next layer.
// input_proj stays the launch it is (its output h0, row-major, is turned into units by the squad's first phase); output_proj + MSE
// reads the last layer's row-major output as before.  Another fp32 summation order than the per-layer kernels (two K-halves): the
// training tests' tolerances, not bit equality.  A squad that cannot finish (a partner not resident within the spin budget) writes
// NaN into the loss accumulator: the step's loss is NaN instead of silently wrong.
#pragma once
#include "chain_squad.h"

namespace osd {

constexpr int TS_RP = 64;                        // patients per workgroup
constexpr int TS_DEPTH = 8;
constexpr int TS_STAGE_FLOATS = 4 * 32 * 68;     // partial accumulators [K-half][sub-panel][32 patients][64 + 4]
__host__ __device__ constexpr int ts_lds_bytes(int n_layers) { return (TS_STAGE_FLOATS + n_layers * SQ_PRM + 16) * 4; }

struct TrainSquadLayer {
  int w_off; int K;                  // fragment-ordered weight [F / 32][K / 8][64][4] at float offset w_off of TrainSquadArgs::wpk (repacked every step); K = all inputs
  int F;                             // 256 or 512
  int in0, n8_0, in1, out;           // unit-order buffers: float offsets inside a sub-panel's activation region (SquadPlan, 32-patient panels)
  const float* bias; const float* gamma; const float* beta;
  float* y; int ldy;                 // row-major output [n][F]
  float* z; float* stats;            // pre-norm [n][F] and (mean, rstd) [n][8][2], or null (no backward)
  int drop_mode;                     // 0 none, 1 injected keep-mask, 2 Philox
  const float* mask; int ldm;
  uint32_t tag;
};

struct TrainSquadArgs {
  TrainSquadLayer L[SQ_MAX_LAYERS];
  int n_layers;
  const float* wpk; long long wpk_floats;    // this step's fragment-ordered copies of the ten weights (k_pack_fragments_multi: one launch)
  const float* h0; int ldh; int h0_out;      // input_proj's output [n][256] row-major; its unit-order buffer
  int n;
  float* act; long long act_stride;          // per 32-patient sub-panel: the layers' outputs in unit order
  unsigned* bar;                             // [panels][16], zero at launch
  unsigned* status;                          // [0]: raised on a timeout
  float* loss_poison;                        // the step's loss accumulator: NaN on a timeout
  unsigned long long spin_budget;
  float keep_scale, p_drop;
  uint64_t seed; uint32_t row_offset; uint32_t step;
  unsigned long long* stamps;                // diagnostic (null in production): per wave 8 cycle counters (tools/train_squad_stamps.py)
};

// lane (l31, h) of a weight fragment: W[f0 + l31][8 i + 4 h .. + 3]
template <int NFB, class LA>
__device__ __forceinline__ void ts_prime_a(v4f (&aq)[TS_DEPTH][2], int n8, const LA& la) { sq_prime_a<NFB, TS_DEPTH>(aq, n8, la); }

__global__ __launch_bounds__(SQ_THREADS, 2) void train_squad_fwd_kernel(const TrainSquadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const stage = smem;
  float* const prm = smem + TS_STAGE_FLOATS;
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + TS_STAGE_FLOATS + a.n_layers * SQ_PRM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // (Tried: a start offset of 6-12 us for every second / fourth / ... / 64th panel, so that the two workgroups of a CU are not in the
  // same phase: 169-176 us for the launch at 4 096 rows against 172 -- whatever keeps two workgroups per CU at 1.65 x the time of
  // one (103 us at 2 048 rows), it is not lockstep.)
  // (Tried: squads of ONE XCD -- members blockIdx % 8 equal, placement checked through XCC_ID behind an agent-scope first barrier --
  // handing over through that XCD's L2 with plain loads / stores and an L2-atomic barrier: 0.886 vs 0.882 ms per step.  The
  // memory-side hand-off is not what bounds this launch; removed.)
  const int panel = blockIdx.x >> 3, g = blockIdx.x & 7;
  const int p0 = panel * TS_RP;
  const int rb = wave & 1, kh = wave >> 1;            // this wave's sub-panel and K-half
  unsigned* const bar = a.bar + (size_t)panel * 16;
  unsigned nb = 0;
  const int l16 = 16 * lane;
  unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0};      // K loops | partials -> LDS + barrier | epilogue | arrive .. released | h0 units | whole kernel
  const bool stamp = a.stamps != nullptr;
  auto now = [&]() -> unsigned long long { return stamp ? __builtin_amdgcn_s_memtime() : 0ull; };
  const unsigned long long c_start = now();
  unsigned long long tc = c_start, tn;
#define TS_STAMP(i) do { if (stamp) { asm volatile("s_nop 0" ::: "memory"); tn = now(); cyc[i] += tn - tc; tc = tn; } } while (0)
  // unit-order regions of the two sub-panels
  const __amdgpu_buffer_rsrc_t r_act0 = sq_rsrc(a.act + (size_t)(2 * panel) * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_act1 = sq_rsrc(a.act + (size_t)(2 * panel + 1) * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_w = sq_rsrc(a.wpk, a.wpk_floats);

  // `after` (work that the squad does not wait for: the layer's row-major outputs, the next layer's first weights) runs between the
  // arrive and the poll
  auto squad_sync = [&](auto&& after) -> bool {
    SQ_DRAIN_BARRIER();
    ++nb;
    if (wave == 0 && lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    after();
    if (wave == 0) {
      const bool ok = squad_wait(bar, SQ_S * nb, a.status, a.spin_budget, lane);
      if (!ok && lane == 0) __hip_atomic_store(a.loss_poison, __builtin_nanf(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = ok ? 1 : 0;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int go = __builtin_amdgcn_readfirstlane(s_flag);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return go != 0;
  };

  // this workgroup's per-feature parameters of every layer
  for (int l = 0; l < a.n_layers; ++l) {
    const TrainSquadLayer& L = a.L[l];
    const int fs = L.F / SQ_S;
    if (tid < 3 * fs) {
      const int arr = tid / fs, j = tid % fs;
      const float* src = arr == 0 ? L.bias : (arr == 1 ? L.gamma : L.beta);
      prm[l * SQ_PRM + arr * 64 + j] = src[g * fs + j];
    }
  }
  // ---- h0 (row-major) -> units: features 32 g .. + 31 of this workgroup's 64 patients; thread (wave = q, lane), both sub-panels ----
  {
    const int f = 32 * g + 8 * wave + 4 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = p0 + 32 * s + l31;
      const int rc = row < a.n ? row : a.n - 1;
      const float4 v = ldg4(a.h0 + (size_t)rc * a.ldh + f);
      sq_st_sc1(s ? r_act1 : r_act0, l16, a.h0_out * 4 + (4 * g + wave) * 1024, v4f{v.x, v.y, v.z, v.w});
    }
  }
  v4f aq[TS_DEPTH][2];
  auto layer_w = [&](const TrainSquadLayer& L, int nfb) { return L.w_off * 4 + ((g * nfb) * (L.K / 8) + kh * (L.K / 16)) * 1024; };      // bytes, uniform
  auto prime_layer = [&](int l) {
    const TrainSquadLayer& L = a.L[l];
    const int K8 = L.K / 8, wl = layer_w(L, L.F / 256);
    auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
    if (L.F == 512) sq_prime_a<2, TS_DEPTH>(aq, L.K / 16, la); else sq_prime_a<1, TS_DEPTH>(aq, L.K / 16, la);
  };
  TS_STAMP(4);
  if (!squad_sync([&] { prime_layer(0); })) return;
  TS_STAMP(3);

  // a layer's row-major outputs of this thread, stored after the arrive: [sub-panel][32-feature block]
  float4 zq[2][2], yq[2][2];
  float st_mean[2], st_rstd[2];
  for (int l = 0; l < a.n_layers; ++l) {
    const TrainSquadLayer& L = a.L[l];
    auto run = [&](auto nfb_tag) {
      constexpr int NFB = decltype(nfb_tag)::value;
      constexpr int LDP = 32 * NFB + 4, GW = 32 * NFB;
      const int K = L.K, n8h = K / 16;                 // 8-k blocks of this wave's K-half
      // weights: feature blocks g * NFB + fb, 8-k blocks kh * n8h + i of the fragment-ordered copy
      const int K8 = K / 8;
      const int wl = layer_w(L, NFB);
      auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
      const int i_first = kh * n8h;
      const int n8_0 = L.n8_0, in0 = L.in0, in1 = L.in1;
      auto lb = [&](int i) -> v4f {
        const int ig = i_first + i;
        const int off = ig < n8_0 ? in0 + ig * 256 : in1 + (ig - n8_0) * 256;      // uniform
        return sq_ld_sc1(rb ? r_act1 : r_act0, l16, off * 4);
      };
      f32x16 acc[NFB][1];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
      sq_kloop<NFB, TS_DEPTH>(acc, aq, n8h, la, lb);        // aq was primed behind the previous barrier's arrive
      TS_STAMP(0);
      // partial accumulators -> LDS [K-half][sub-panel][patient][feature (+4)]
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(stage + ((kh * 2 + rb) * 32 + l31) * LDP + 32 * fb + 8 * q + 4 * h) =
              make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      TS_STAMP(1);
      // epilogue: 8 threads per patient, two passes (sub-panels); thread (erow, c): features 32 j + 4 c .. + 3 of the group
      const float* pl = prm + l * SQ_PRM;
      const int erow = tid >> 3, c = tid & 7, f0 = 4 * c;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = p0 + 32 * s + erow;
        const bool rok = row < a.n;
        float v[4 * NFB];
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
          const float4 p0v = *reinterpret_cast<const float4*>(stage + (s * 32 + erow) * LDP + f0 + 32 * j);
          const float4 p1v = *reinterpret_cast<const float4*>(stage + ((2 + s) * 32 + erow) * LDP + f0 + 32 * j);
          const float4 bv = *reinterpret_cast<const float4*>(pl + f0 + 32 * j);
          v[4 * j] = (p0v.x + p1v.x) + bv.x; v[4 * j + 1] = (p0v.y + p1v.y) + bv.y;
          v[4 * j + 2] = (p0v.z + p1v.z) + bv.z; v[4 * j + 3] = (p0v.w + p1v.w) + bv.w;
        }
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 4 * NFB; ++e) sm += v[e];
        const float mean = sq_sum8(sm) * (1.0f / GW);
        float qs = 0.f;
#pragma unroll
        for (int e = 0; e < 4 * NFB; ++e) { const float d = v[e] - mean; qs = fmaf(d, d, qs); }
        const float rstd = 1.0f / sqrtf(sq_sum8(qs) * (1.0f / GW) + GN_EPS);
        st_mean[s] = mean; st_rstd[s] = rstd;
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
          const int f = f0 + 32 * j;                    // feature inside the group
          const int gf = g * GW + f;                    // ... of the layer
          zq[s][j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
          const float4 gv = *reinterpret_cast<const float4*>(pl + 64 + f);
          const float4 bev = *reinterpret_cast<const float4*>(pl + 128 + f);
          float4 y;
          y.x = silu_f(fmaf((v[4 * j] - mean) * rstd, gv.x, bev.x));
          y.y = silu_f(fmaf((v[4 * j + 1] - mean) * rstd, gv.y, bev.y));
          y.z = silu_f(fmaf((v[4 * j + 2] - mean) * rstd, gv.z, bev.z));
          y.w = silu_f(fmaf((v[4 * j + 3] - mean) * rstd, gv.w, bev.w));
          if (L.drop_mode == 1) {
            const int rc = rok ? row : a.n - 1;
            const float4 mk = ldg4(L.mask + (size_t)rc * L.ldm + gf);
            y.x *= mk.x * a.keep_scale; y.y *= mk.y * a.keep_scale; y.z *= mk.z * a.keep_scale; y.w *= mk.w * a.keep_scale;
          } else if (L.drop_mode == 2) {
            const uint4 r = philox_at(a.seed, a.row_offset + (uint32_t)row, (uint32_t)(gf >> 2), a.step, L.tag);
            y.x *= (u01(r.x) >= a.p_drop) ? a.keep_scale : 0.f;
            y.y *= (u01(r.y) >= a.p_drop) ? a.keep_scale : 0.f;
            y.z *= (u01(r.z) >= a.p_drop) ? a.keep_scale : 0.f;
            y.w *= (u01(r.w) >= a.p_drop) ? a.keep_scale : 0.f;
          }
          yq[s][j] = y;
          const int unit = g * NFB * 4 + (f >> 3), ln = erow + 32 * ((f >> 2) & 1);
          sq_st_sc1(s ? r_act1 : r_act0, 16 * ln, L.out * 4 + unit * 1024, v4f{y.x, y.y, y.z, y.w});
        }
      }
    };
    if (L.F == 512) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
    // what the backward pass (and output_proj) reads, row-major as EpiGnSilu leaves it: nobody in the squad waits for it
    auto row_major = [&]() {
      const int erow = tid >> 3, c = tid & 7, nfb = L.F / 256, gw = 32 * nfb;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = p0 + 32 * s + erow;
        if (row >= a.n) continue;
        if (L.z && c == 0) {
          float* sp = L.stats + ((size_t)row * SQ_S + g) * 2;
          sp[0] = st_mean[s]; sp[1] = st_rstd[s];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j >= nfb) break;
          const int gf = g * gw + 4 * c + 32 * j;
          if (L.z) stg4(L.z + (size_t)row * L.F + gf, zq[s][j]);
          stg4(L.y + (size_t)row * L.ldy + gf, yq[s][j]);
        }
      }
    };
    TS_STAMP(2);
    if (l + 1 < a.n_layers) {
      if (!squad_sync([&] { row_major(); prime_layer(l + 1); })) return;
      TS_STAMP(3);
    } else {
      row_major();
    }
  }
#undef TS_STAMP
  if (stamp && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 5; ++i) o[i] = cyc[i];
    o[5] = __builtin_amdgcn_s_memtime() - c_start;
  }
}

}  // namespace osd
