// train_squad_bwd.h -- the dgrad chain of a training step's backward pass (loss.backward() through the ten Linear + GroupNorm +
// SiLU (+ Dropout) layers, utils/train.py:239) as ONE launch of squads: train_squad.h's decomposition run backwards.
//
// Per-layer, that chain is nine launches of 14-35 us (gemm_kernel / gemm_dual_kernel with EpiGnBwd, k_gnbwd.hip) behind the first
// dgrad (output_proj, K = D: stays the launch it is) and the last one (into h0: no GroupNorm below, a plain phase here).  Here a squad --
// eight workgroups per 64 patients -- walks them: a phase is "the gradient through one Linear into the layer below", i.e.
//     g[:, c] = sum_j gz_up[:, j] W[j][c]      (dgrad: K = the upper layer's features j, output = its input columns c)
// followed by the GroupNorm + SiLU (+ dropout) backward of the layer that produced those columns (EpiGnBwd's arithmetic: per
// (row, group) sums only, and workgroup g owns group g's columns of every layer, exactly as in the forward squad).  A decoder
// layer's input is [current | skip]: its skip columns are a second K loop of the same phase, stored plainly into the encoder
// output's gradient buffer, where the phase that later reaches that encoder layer adds them (same workgroup: the slice it wrote).
//   * gradients travel between the squad's workgroups in unit order (chain_squad.h) through agent-scope loads / stores behind the
//     squad's barrier; what the weight-gradient launch and the column sums read afterwards -- dL/dz and dL/dy of every layer --
//     is written row-major as the per-layer kernels leave it, after the arrive;
//   * the weights are needed transposed (A[m = input column][k = output feature]): this step's fragment-ordered copies of W^T
//     come from the same kind of single pack launch as the forward's (k_pack_fragments_multi_t);
//   * wave w of a workgroup: sub-panel w & 1 (32 patients), K-half w >> 1; the halves meet in LDS; epilogue on 256 threads, 8 per
//     patient, the two group sums of the GroupNorm backward by DPP over those 8 lanes.
// Another fp32 summation order than the per-layer kernels: the training tests' tolerances.  Single-GPU steps only (data parallel
// flushes weight gradients mid-pass between those launches and keeps them).
#pragma once
#include "train_squad.h"

namespace osd {

struct TrainSquadBwdPhase {
  int w_off, K;                      // W^T fragments [F / 32][K / 8][64][4] at float offset w_off: F = columns of this phase, K = features of the upper layer
  int F;                             // main columns = features of the producer layer (256 or 512)
  int in, out;                       // unit-order buffers (float offsets in a sub-panel's region): dL/dz of the upper layer in, of the producer out
  int prm;                           // the producer layer's index (its gamma / beta in LDS)
  const float* z; const float* stats;// the producer's pre-norm activations [n][F] and (mean, rstd) [n][8][2]
  float* gy; float* gz;              // row-major outputs [n][F]: dL/dy (read first when accumulate) and dL/dz
  int accumulate;
  int plain;                         // 1: no GroupNorm below (the last phase, into h0): the sum of the K-halves goes to gz as it is
  int drop_mode; const float* mask; int ldm; uint32_t tag;
  int skip_w_off, skip_F; float* skip_out;     // second K loop: skip columns (0 = none), plain store [n][skip_F]
};

struct TrainSquadBwdArgs {
  TrainSquadBwdPhase P[SQ_MAX_LAYERS];
  int n_phases;
  const float* gamma[SQ_MAX_LAYERS]; const float* beta[SQ_MAX_LAYERS]; int width[SQ_MAX_LAYERS]; int n_layers;
  const float* wpk; long long wpk_floats;
  const float* gz_top; int top_F; int top_out;      // dL/dz of the last layer [n][top_F] row-major (left by the first dgrad) and its unit-order buffer
  int n;
  float* act; long long act_stride;
  unsigned* bar; unsigned* status; float* loss_poison;
  unsigned long long spin_budget;
  float keep_scale, p_drop;
  uint64_t seed; uint32_t row_offset; uint32_t step;
};

__global__ __launch_bounds__(SQ_THREADS, 2) void train_squad_bwd_kernel(const TrainSquadBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const stage = smem;
  float* const prm = smem + TS_STAGE_FLOATS;           // per layer: gamma at [0, 64), beta at [64, 128) of this workgroup's group
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + TS_STAGE_FLOATS + a.n_layers * SQ_PRM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int panel = blockIdx.x >> 3, g = blockIdx.x & 7;
  const int p0 = panel * TS_RP;
  const int rb = wave & 1, kh = wave >> 1;
  unsigned* const bar = a.bar + (size_t)panel * 16;
  unsigned nb = 0;
  const int l16 = 16 * lane;
  const __amdgpu_buffer_rsrc_t r_act0 = sq_rsrc(a.act + (size_t)(2 * panel) * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_act1 = sq_rsrc(a.act + (size_t)(2 * panel + 1) * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_w = sq_rsrc(a.wpk, a.wpk_floats);

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

  for (int l = 0; l < a.n_layers; ++l) {
    const int fs = a.width[l] / SQ_S;
    if (tid < 2 * fs) {
      const int arr = tid / fs, j = tid % fs;
      prm[l * SQ_PRM + arr * 64 + j] = (arr == 0 ? a.gamma[l] : a.beta[l])[g * fs + j];
    }
  }
  // ---- dL/dz of the last layer (row-major) -> units: features of group g, both sub-panels ----
  {
    const int fs = a.top_F / SQ_S;                     // 32 or 64: 4 or 8 units per sub-panel
    for (int idx = wave; idx < 2 * (fs / 8); idx += 4) {
      const int sp = idx / (fs / 8), q = idx % (fs / 8);
      const int f = g * fs + 8 * q + 4 * h;
      const int row = p0 + 32 * sp + l31;
      const int rc = row < a.n ? row : a.n - 1;
      const float4 v = ldg4(a.gz_top + (size_t)rc * a.top_F + f);
      sq_st_sc1(sp ? r_act1 : r_act0, l16, a.top_out * 4 + (g * (fs / 8) + q) * 1024, v4f{v.x, v.y, v.z, v.w});
    }
  }
  v4f aq[TS_DEPTH][2];
  auto phase_w = [&](int w_off, int nfb, int K) { return w_off * 4 + ((g * nfb) * (K / 8) + kh * (K / 16)) * 1024; };      // bytes, uniform
  auto prime_phase = [&](int ph) {
    const TrainSquadBwdPhase& P = a.P[ph];
    const int K8 = P.K / 8, wl = phase_w(P.w_off, P.F / 256, P.K);
    auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
    if (P.F == 512) sq_prime_a<2, TS_DEPTH>(aq, P.K / 16, la); else sq_prime_a<1, TS_DEPTH>(aq, P.K / 16, la);
  };
  if (!squad_sync([&] { prime_phase(0); })) return;

  float4 gyq[2][2], gzq[2][2], skq[2][2];            // this thread's row-major outputs of a phase: [sub-panel][32-column block]
  for (int ph = 0; ph < a.n_phases; ++ph) {
    const TrainSquadBwdPhase& P = a.P[ph];
    const int K = P.K, K8 = K / 8, n8h = K / 16;
    const int in_off = P.in;
    auto lb = [&](int i) -> v4f { return sq_ld_sc1(rb ? r_act1 : r_act0, l16, (in_off + (kh * n8h + i) * 256) * 4); };
    const int erow = tid >> 3, c = tid & 7, f0 = 4 * c;
    // ---- main columns: dgrad + GroupNorm / SiLU (/ dropout) backward of the producer (EpiGnBwd::apply's arithmetic) ----
    auto main_part = [&](auto nfb_tag) {
      constexpr int NFB = decltype(nfb_tag)::value;
      constexpr int LDP = 32 * NFB + 4, GW = 32 * NFB;
      {   // one K loop + the hand-over of the partial accumulators through LDS: [K-half][sub-panel][patient][column (+4)]
        const int wl = phase_w(P.w_off, NFB, K);
        auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
        f32x16 acc[NFB][1];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
        // aq was primed behind the previous barrier's arrive
        sq_kloop<NFB, TS_DEPTH>(acc, aq, n8h, la, lb);
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(stage + ((kh * 2 + rb) * 32 + l31) * LDP + 32 * fb + 8 * q + 4 * h) =
                make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (P.plain) {                                // uniform: dL/dh0 = the plain dgrad
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < NFB; ++j) {
            const int f = f0 + 32 * j;
            const float4 p0v = *reinterpret_cast<const float4*>(stage + (s * 32 + erow) * LDP + f);
            const float4 p1v = *reinterpret_cast<const float4*>(stage + ((2 + s) * 32 + erow) * LDP + f);
            gzq[s][j] = make_float4(p0v.x + p1v.x, p0v.y + p1v.y, p0v.z + p1v.z, p0v.w + p1v.w);
          }
        return;
      }
      const float* pl = prm + P.prm * SQ_PRM;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = p0 + 32 * s + erow;
        const bool rok = row < a.n;
        const int rc = rok ? row : a.n - 1;
        const float2 st = ldg2(P.stats + ((size_t)rc * SQ_S + g) * 2);
        float gyv[4 * NFB], zh[4 * NFB];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
          const int f = f0 + 32 * j, gf = g * GW + f;
          const float4 p0v = *reinterpret_cast<const float4*>(stage + (s * 32 + erow) * LDP + f);
          const float4 p1v = *reinterpret_cast<const float4*>(stage + ((2 + s) * 32 + erow) * LDP + f);
          const float4 z4 = ldg4(P.z + (size_t)rc * P.F + gf);
          float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
          if (P.accumulate) add = ldg4(P.gy + (size_t)rc * P.F + gf);
          const float4 gv = *reinterpret_cast<const float4*>(pl + f);
          const float4 bev = *reinterpret_cast<const float4*>(pl + 64 + f);
          float keep[4] = {1.f, 1.f, 1.f, 1.f};
          if (P.drop_mode == 1) {
            const float4 mk = ldg4(P.mask + (size_t)rc * P.ldm + gf);
            keep[0] = mk.x * a.keep_scale; keep[1] = mk.y * a.keep_scale; keep[2] = mk.z * a.keep_scale; keep[3] = mk.w * a.keep_scale;
          } else if (P.drop_mode == 2) {
            const uint4 rr = philox_at(a.seed, a.row_offset + (uint32_t)row, (uint32_t)(gf >> 2), a.step, P.tag);
            keep[0] = (u01(rr.x) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[1] = (u01(rr.y) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[2] = (u01(rr.z) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[3] = (u01(rr.w) >= a.p_drop) ? a.keep_scale : 0.f;
          }
          const float gin[4] = {p0v.x + p1v.x, p0v.y + p1v.y, p0v.z + p1v.z, p0v.w + p1v.w};
          const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
          const float av[4] = {add.x, add.y, add.z, add.w};
          const float gm[4] = {gv.x, gv.y, gv.z, gv.w};
          const float bt[4] = {bev.x, bev.y, bev.z, bev.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zhat = (zv[e] - st.x) * st.y;
            const float y = zhat * gm[e] + bt[e];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
            const float gv_ = (gin[e] + av[e]) * keep[e] * (sg * (1.0f + y * (1.0f - sg)));
            const float gzh = gv_ * gm[e];
            zh[4 * j + e] = zhat;
            gyv[4 * j + e] = gv_;
            s1 += gzh;
            s2 += gzh * zhat;
          }
          gyq[s][j] = make_float4(gyv[4 * j], gyv[4 * j + 1], gyv[4 * j + 2], gyv[4 * j + 3]);
        }
        s1 = sq_sum8(s1) * (1.0f / GW);
        s2 = sq_sum8(s2) * (1.0f / GW);
#pragma unroll
        for (int j = 0; j < NFB; ++j) {
          const int f = f0 + 32 * j;
          const float4 gv = *reinterpret_cast<const float4*>(pl + f);
          const float gm[4] = {gv.x, gv.y, gv.z, gv.w};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = st.y * (gyv[4 * j + e] * gm[e] - s1 - zh[4 * j + e] * s2);
          gzq[s][j] = make_float4(o[0], o[1], o[2], o[3]);
          const int unit = g * NFB * 4 + (f >> 3), ln = erow + 32 * ((f >> 2) & 1);
          sq_st_sc1(s ? r_act1 : r_act0, 16 * ln, P.out * 4 + unit * 1024, v4f{o[0], o[1], o[2], o[3]});
        }
      }
    };
    if (P.F == 512) main_part(std::integral_constant<int, 2>{});
    else main_part(std::integral_constant<int, 1>{});
    // ---- skip columns: a second K loop over the same input, plain result ----
    if (P.skip_F > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every thread has read the main part's partials
      auto skip_part = [&](auto nfb_tag) {
        constexpr int NFB = decltype(nfb_tag)::value;
        constexpr int LDP = 32 * NFB + 4;
      {   // one K loop + the hand-over of the partial accumulators through LDS: [K-half][sub-panel][patient][column (+4)]
          const int wl = phase_w(P.skip_w_off, NFB, K);
          auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K8 + i) * 1024); };
          f32x16 acc[NFB][1];
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[fb][0][r] = 0.f;
          sq_prime_a<NFB, TS_DEPTH>(aq, n8h, la);
          sq_kloop<NFB, TS_DEPTH>(acc, aq, n8h, la, lb);
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<float4*>(stage + ((kh * 2 + rb) * 32 + l31) * LDP + 32 * fb + 8 * q + 4 * h) =
                  make_float4(acc[fb][0][4 * q], acc[fb][0][4 * q + 1], acc[fb][0][4 * q + 2], acc[fb][0][4 * q + 3]);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < NFB; ++j) {
            const int f = f0 + 32 * j;
            const float4 p0v = *reinterpret_cast<const float4*>(stage + (s * 32 + erow) * LDP + f);
            const float4 p1v = *reinterpret_cast<const float4*>(stage + ((2 + s) * 32 + erow) * LDP + f);
            skq[s][j] = make_float4(p0v.x + p1v.x, p0v.y + p1v.y, p0v.z + p1v.z, p0v.w + p1v.w);
          }
      };
      if (P.skip_F == 512) skip_part(std::integral_constant<int, 2>{});
      else skip_part(std::integral_constant<int, 1>{});
    }
    // what the weight-gradient launch, the column sums and later phases read, row-major as the per-layer kernels leave it
    auto row_major = [&]() {
      const int nfb = P.F / 256, gw = 32 * nfb, snfb = P.skip_F / 256, sgw = 32 * snfb;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = p0 + 32 * s + erow;
        if (row >= a.n) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j < nfb) {
            const int gf = g * gw + f0 + 32 * j;
            if (!P.plain) stg4(P.gy + (size_t)row * P.F + gf, gyq[s][j]);
            stg4(P.gz + (size_t)row * P.F + gf, gzq[s][j]);
          }
          if (j < snfb) stg4(P.skip_out + (size_t)row * P.skip_F + g * sgw + f0 + 32 * j, skq[s][j]);
        }
      }
    };
    if (ph + 1 < a.n_phases) {
      if (!squad_sync([&] { row_major(); prime_phase(ph + 1); })) return;
    } else {
      row_major();
    }
  }
}

}  // namespace osd
