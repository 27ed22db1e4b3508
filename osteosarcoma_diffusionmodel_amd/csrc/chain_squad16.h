// chain_squad16.h -- the squad chain kernel (chain_squad.h) on 16-patient panels: v_mfma_f32_16x16x4_f32 instead of 32x32x2.
//
// Up to ~1 000 patients the 32-patient squads are at most one workgroup per CU, and a workgroup spends half of a reverse step
// waiting: 12 hand-offs of ~2.5 us through the memory side, epilogues, the first loads of every phase (profiles/r04_squad_chain.md:
// 59 us of matrix work in a 112 us step -- and the same 112-119 us for 240 or 496 rows, where half or three quarters of the CUs
// idle).  Halving the panel halves the matrix work per phase at the same latencies and uses twice the CUs: 79 us per step up to
// 512 rows (one workgroup per CU; 32-patient panels: 115-119).  From 513 to 1 024 rows two workgroups of different squads share a
// CU and its SIMDs are busy -- matrix work + VALU of two squads, which fp32 MFMA does not overlap: 109-111 us against 112-115, a
// start offset between the two changes nothing --, beyond that the 32-patient kernel (less weight traffic per row) takes over.
// Same decomposition, same hand-off protocol, same formulas (the 16x16x4 MFMA sums k in groups of four of a 16-k block instead
// of pairs of an 8-k block: another fp32 summation order again, chain tolerance).
//   unit = [16-k block][lane][4 floats]: lane (l15, kg) holds patient l15, k = 16 i + 4 kg .. + 3 -- the accumulator fragment of the
//   writer (lane (l15, mg): features 4 mg .. + 3 of a 16-feature block) and the B operand of the reader; weights in the matching
//   fragment order [F / 16][K / 16][64 lanes][4] (k_pack_fragments16).
// Host side, arguments, barrier, failure handling: chain_squad.hip / chain_squad.h (SquadArgs: K8 counts 16-k blocks, T32 counts
// 16-feature state tiles here).
#pragma once
#include "chain_squad.h"

namespace osd {

constexpr int SQ16_RP = 16;
constexpr int SQ16_DEPTH = 4;                    // 16-k blocks a wave keeps in flight
constexpr int SQ16_STAGE_FLOATS = 16 * 256 + 4 * 256;      // output_proj's operand (16 blocks) + the partials of a K-split tile; the layers' partials [4][16][68] share it
static_assert(SQ16_STAGE_FLOATS >= 4 * 16 * 68, "partials");
__host__ __device__ constexpr int sq16_lds_bytes(int n_layers) { return (SQ16_STAGE_FLOATS + n_layers * SQ_PRM + 16) * 4; }

template <int NFB, class LA>
__device__ __forceinline__ void sq16_prime_a(v4f (&aq)[SQ16_DEPTH][4], int n, const LA& la) {
#pragma unroll
  for (int d = 0; d < SQ16_DEPTH; ++d) {
    const int i = d < n ? d : n - 1;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = la(fb, i);
  }
}
// chain_squad.h's sq_kloop: peeled first group, refills in slot order
template <int NFB, class LA, class LB>
__device__ __forceinline__ void sq16_kloop(v4f (&acc)[NFB], v4f (&aq)[SQ16_DEPTH][4], int n, const LA& la, const LB& lb) {
  v4f bq[SQ16_DEPTH];
#pragma unroll
  for (int d = 0; d < SQ16_DEPTH; ++d) bq[d] = lb(d < n ? d : n - 1);
  auto group = [&](int i0) {
#pragma unroll
    for (int d = 0; d < SQ16_DEPTH; ++d) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          acc[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d][fb][e], bq[d][e], acc[fb], 0, 0, 0);
      const int in = i0 + d + SQ16_DEPTH < n ? i0 + d + SQ16_DEPTH : n - 1;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) aq[d][fb] = la(fb, in);
      bq[d] = lb(in);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int i0 = 0;
  if (SQ16_DEPTH < n) {
    group(0);
    for (i0 = SQ16_DEPTH; i0 + SQ16_DEPTH < n; i0 += SQ16_DEPTH) group(i0);
  }
#pragma unroll
  for (int d = 0; d < SQ16_DEPTH; ++d) {
    if (i0 + d < n) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          acc[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d][fb][e], bq[d][e], acc[fb], 0, 0, 0);
    }
  }
}

template <bool STAMP = false>
__global__ __launch_bounds__(SQ_THREADS, 2) void squad16_chain_kernel(const SquadArgs* __restrict__ gp) {
  const SquadArgs& a = *gp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const stage = smem;
  float* const prm = smem + SQ16_STAGE_FLOATS;
  volatile int& s_flag = *reinterpret_cast<volatile int*>(smem + SQ16_STAGE_FLOATS + a.n_layers * SQ_PRM);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kg = lane >> 4;
  const int panel = blockIdx.x >> 3, g = blockIdx.x & 7;
  const int p0 = panel * SQ16_RP;
  const int row = p0 + l15;
  const int rowc = row < a.n ? row : a.n - 1;
  unsigned* const bar = a.bar + (size_t)panel * 16;
  unsigned nb = 0;

  const int T16 = a.T32, D = a.D;
  const int t0 = g * T16 / SQ_S, t1 = (g + 1) * T16 / SQ_S;      // this workgroup's 16-feature tiles of the state
  float* const xs = a.xs + (size_t)panel * a.xs_stride;
  const __amdgpu_buffer_rsrc_t r_act = sq_rsrc(a.act + (size_t)panel * a.act_stride, a.act_stride);
  const __amdgpu_buffer_rsrc_t r_xs = sq_rsrc(xs, a.xs_stride);
  const __amdgpu_buffer_rsrc_t r_w = sq_rsrc(a.wpk, a.wpk_floats);
  const __amdgpu_buffer_rsrc_t r_slab = sq_rsrc(a.slab + (size_t)panel * a.slab_stride, a.slab_stride);
  const int l16 = 16 * lane;

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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return go != 0;
  };

  // ---- once per launch ----
  for (int l = 0; l < a.n_layers; ++l) {
    const SquadLayer& L = a.L[l];
    const int fs = L.F / SQ_S;
    if (tid < 3 * fs) {
      const int arr = tid / fs, j = tid % fs;
      const float* src = arr == 0 ? L.bias : (arr == 1 ? L.gamma : L.beta);
      prm[l * SQ_PRM + arr * 64 + j] = src[g * fs + j];
    }
  }
  // reduce phase: waves 0 and 1; thread (wave = j, lane) owns the float4 at features 32 g + 16 j + 4 kg of its row
  const int fr = 32 * g + 16 * (wave & 1) + 4 * kg;
  const float4 r_bias = ldg4(a.bias_in + fr);
  const float4 r_cproj = ldg4(a.cproj + (size_t)rowc * a.ldc + fr);
  {
    const float* xrow = a.x + (size_t)rowc * a.ldx;
    for (int u = t0 + wave; u < t1; u += 4) {             // unit u: features 16 u + 4 kg .. + 3
      const int f = 16 * u + 4 * kg;
      v4f v;
      v.x = f < D ? xrow[f] : 0.f;
      v.y = f + 1 < D ? xrow[f + 1] : 0.f;
      v.z = f + 2 < D ? xrow[f + 2] : 0.f;
      v.w = f + 3 < D ? xrow[f + 3] : 0.f;
      sq_st(r_xs, l16, u * 1024, v);
    }
  }
  SQ_DRAIN_BARRIER();

  v4f aq[SQ16_DEPTH][4];
  // input_proj: the wave's feature blocks 4 wave .. 4 wave + 3 (of H0 / 16 = 16), the workgroup's 16-k blocks t0 .. t1
  const int K16i = T16;
  const int n16i = t1 - t0;
  const int wi = a.in_off * 4 + ((4 * wave) * K16i + t0) * 1024;
  auto la_in = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wi + (fb * K16i + i) * 1024); };
  sq16_prime_a<4>(aq, n16i, la_in);

  for (int si = 0; si < a.n_steps; ++si) {
    const int t = a.t_first - si;
    // =============================== input_proj: partial sums over this workgroup's state features ===============================
    {
      v4f acc[4];
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) acc[fb] = v4f{0.f, 0.f, 0.f, 0.f};
      auto lb = [&](int i) -> v4f { return sq_ld(r_xs, l16, (t0 + i) * 1024); };
      sq16_kloop<4>(acc, aq, n16i, la_in, lb);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) sq_st_sc1(r_slab, l16, (g * 16 + 4 * wave + fb) * 1024, acc[fb]);      // slab [g][16 blocks][lane]
    }
    const float4 r_temb = ldg4(a.temb + (size_t)t * a.ldt + fr);
    auto layer_a = [&](const SquadLayer& L, int nfb) {
      const int nq = L.K8 / 4;
      return L.w_off * 4 + ((g * nfb) * L.K8 + wave * nq) * 1024;
    };
    auto prime_layer = [&](int l) {
      const SquadLayer& L = a.L[l];
      const int wl = layer_a(L, L.F / 128);
      const int K16 = L.K8;
      auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K16 + i) * 1024); };
      if (L.F == 512) sq16_prime_a<4>(aq, K16 / 4, la); else sq16_prime_a<2>(aq, K16 / 4, la);
    };
    auto no_prime = [] {};
    if (!squad_sync([&] { prime_layer(0); })) return;
    // =============================== reduce: h0 = ((sum + b) + temb[t]) + cproj ===============================
    if (wave < 2) {
      v4f p[SQ_S];
#pragma unroll
      for (int s = 0; s < SQ_S; ++s) p[s] = sq_ld_sc1(r_slab, l16, (s * 16 + 2 * g + wave) * 1024);
      v4f sum = p[0];
#pragma unroll
      for (int s = 1; s < SQ_S; ++s) sum += p[s];
      v4f o;
      o.x = ((sum.x + r_bias.x) + r_temb.x) + r_cproj.x;
      o.y = ((sum.y + r_bias.y) + r_temb.y) + r_cproj.y;
      o.z = ((sum.z + r_bias.z) + r_temb.z) + r_cproj.z;
      o.w = ((sum.w + r_bias.w) + r_temb.w) + r_cproj.w;
      sq_st_sc1(r_act, l16, a.h0_out * 4 + (2 * g + wave) * 1024, o);
    }
    if (!squad_sync(no_prime)) return;

    // =============================== Linear + GroupNorm + SiLU layers ===============================
    for (int l = 0; l < a.n_layers; ++l) {
      const SquadLayer& L = a.L[l];
      auto run = [&](auto nfb_tag) {
        constexpr int NFB = decltype(nfb_tag)::value;          // 16-feature blocks of this workgroup's group: 2 or 4
        const int K16 = L.K8, nq = K16 / 4;
        const int wl = layer_a(L, NFB);
        auto la = [&](int fb, int i) -> v4f { return sq_ld(r_w, l16, wl + (fb * K16 + i) * 1024); };
        const int i_first = wave * nq;
        const int n0 = L.n8_0, in0 = L.in0, in1 = L.in1;
        auto lb = [&](int i) -> v4f {
          const int ig = i_first + i;
          const int off = ig < n0 ? in0 + ig * 256 : in1 + (ig - n0) * 256;
          return sq_ld_sc1(r_act, l16, off * 4);
        };
        v4f acc[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc[fb] = v4f{0.f, 0.f, 0.f, 0.f};
        sq16_kloop<NFB>(acc, aq, nq, la, lb);
        // partials -> LDS [wave][patient][feature (+4)]; the epilogue of chain_squad.h on 128 threads (8 per patient)
        constexpr int LDP = 16 * NFB + 4, GW = 16 * NFB, NJ = NFB / 2;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          *reinterpret_cast<v4f*>(stage + (wave * 16 + l15) * LDP + 16 * fb + 4 * kg) = acc[fb];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave < 2) {
          const int erow = tid >> 3, c = tid & 7, f0 = 4 * c;
          const float* pl = prm + l * SQ_PRM;
          float v[4 * NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            float4 sum = *reinterpret_cast<const float4*>(stage + erow * LDP + f0 + 32 * j);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
              const float4 pv = *reinterpret_cast<const float4*>(stage + (w * 16 + erow) * LDP + f0 + 32 * j);
              sum.x += pv.x; sum.y += pv.y; sum.z += pv.z; sum.w += pv.w;
            }
            const float4 bv = *reinterpret_cast<const float4*>(pl + f0 + 32 * j);
            v[4 * j] = sum.x + bv.x; v[4 * j + 1] = sum.y + bv.y; v[4 * j + 2] = sum.z + bv.z; v[4 * j + 3] = sum.w + bv.w;
          }
          float sm = 0.f;
#pragma unroll
          for (int e = 0; e < 4 * NJ; ++e) sm += v[e];
          const float mean = sq_sum8(sm) * (1.0f / GW);
          float qs = 0.f;
#pragma unroll
          for (int e = 0; e < 4 * NJ; ++e) { const float d = v[e] - mean; qs = fmaf(d, d, qs); }
          const float rstd = 1.0f / sqrtf(sq_sum8(qs) * (1.0f / GW) + GN_EPS);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(pl + 64 + f0 + 32 * j);
            const float4 bev = *reinterpret_cast<const float4*>(pl + 128 + f0 + 32 * j);
            v4f y;
            y.x = silu_f(fmaf((v[4 * j] - mean) * rstd, gv.x, bev.x));
            y.y = silu_f(fmaf((v[4 * j + 1] - mean) * rstd, gv.y, bev.y));
            y.z = silu_f(fmaf((v[4 * j + 2] - mean) * rstd, gv.z, bev.z));
            y.w = silu_f(fmaf((v[4 * j + 3] - mean) * rstd, gv.w, bev.w));
            const int f = f0 + 32 * j;                    // feature inside the group: 16-block f / 16, k group (f / 4) & 3
            const int unit = g * NFB + (f >> 4), ln = erow + 16 * ((f >> 2) & 3);
            sq_st_sc1(r_act, 16 * ln, L.out * 4 + unit * 1024, y);
          }
        }
      };
      if (L.F == 512) run(std::integral_constant<int, 4>{});
      else run(std::integral_constant<int, 2>{});
      const bool more = l + 1 < a.n_layers;
      if (!squad_sync([&] { if (more) prime_layer(l + 1); })) return;
    }

    // =============================== output_proj + posterior on this workgroup's state tiles ===============================
    {
      const int K16o = a.K8_out;                // 16
      for (int u = wave; u < K16o; u += 4) {
        const v4f v = sq_ld_sc1(r_act, l16, a.last_in * 4 + u * 1024);
        *reinterpret_cast<v4f*>(stage + (u * 64 + lane) * 4) = v;
      }
      const float* c = a.coef + 4 * t;
      const float cA = c[0], cB = c[1], cC = c[2];
      const bool last_step = si + 1 == a.n_steps;
      const bool do_mask = t == 0 && a.mut_mask != nullptr;
      const float* zbase = a.z ? a.z + (long long)(a.z_t_first - t) * a.z_step_stride : nullptr;
      SQ_DRAIN_BARRIER();
      // the posterior update of one 16-feature tile: this lane's patient, features 16 tile + 4 kg .. + 3
      auto post_tile = [&](int tile, v4f e4, v4f xv4, float4 bv) {
        const int f = 16 * tile + 4 * kg;
        const float e[4] = {e4.x + bv.x, e4.y + bv.y, e4.z + bv.z, e4.w + bv.w};
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
          if (f + r >= D) o[r] = 0.f;
        }
        sq_st(r_xs, l16, tile * 1024, v4f{o[0], o[1], o[2], o[3]});
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
      const int nt = t1 - t0, rem = nt & 3;
      const int rounds = (nt >> 2) + (rem == 3 ? 1 : 0), n_split = rem == 3 ? 0 : rem;
      auto tile_a = [&](int tile) { return a.out_off * 4 + tile * K16o * 1024; };
      const int nqs = K16o / 4;
      auto split_prime = [&](int r) {
        const int wo = tile_a(t0 + 4 * rounds + r) + (wave * nqs) * 1024;
        sq16_prime_a<1>(aq, nqs, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wo + i * 1024); });
      };
      if (rounds == 0 && n_split > 0) split_prime(0);
      {
        const int tile0 = t0 + wave;
        if (rounds > 0 && tile0 < t1) {
          const int wo = tile_a(tile0);
          sq16_prime_a<1>(aq, K16o, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wo + i * 1024); });
        }
      }
      for (int j = 0; j < rounds; ++j) {
        const int tile = t0 + wave + 4 * j;
        if (tile >= t1) break;                              // rem == 3: wave 3 sits the last round out
        const int wo = tile_a(tile);
        auto la = [&](int fb, int i) -> v4f { (void)fb; return sq_ld(r_w, l16, wo + i * 1024); };
        v4f acc[1] = {v4f{0.f, 0.f, 0.f, 0.f}};
        const v4f xq = sq_ld(r_xs, l16, tile * 1024);
        const float4 bq4 = ldg4(a.bias_out + 16 * tile + 4 * kg);
        sq16_kloop<1>(acc, aq, K16o, la, lb);
        const int tile_n = tile + 4;
        if (j + 1 < rounds && tile_n < t1) {
          const int wn = tile_a(tile_n);
          sq16_prime_a<1>(aq, K16o, [&](int, int i) -> v4f { return sq_ld(r_w, l16, wn + i * 1024); });
        } else if (j + 1 == rounds && n_split > 0) {
          split_prime(0);
        }
        post_tile(tile, acc[0], xq, bq4);
      }
      for (int r = 0; r < n_split; ++r) {
        const int tile = t0 + 4 * rounds + r;
        const int wo = tile_a(tile) + (wave * nqs) * 1024;
        auto la = [&](int fb, int i) -> v4f { (void)fb; return sq_ld(r_w, l16, wo + i * 1024); };
        auto lbq = [&](int i) -> v4f { return *reinterpret_cast<const v4f*>(stage + ((wave * nqs + i) * 64 + lane) * 4); };
        v4f acc[1] = {v4f{0.f, 0.f, 0.f, 0.f}};
        sq16_kloop<1>(acc, aq, nqs, la, lbq);
        if (r + 1 < n_split) split_prime(r + 1);
        float* const red = stage + 16 * 256;               // [wave][lane] float4
        if (r > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        *reinterpret_cast<v4f*>(red + (wave * 64 + lane) * 4) = acc[0];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave == 0) {
          v4f sum = *reinterpret_cast<const v4f*>(red + lane * 4);
#pragma unroll
          for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const v4f*>(red + (w * 64 + lane) * 4);
          const v4f xq = sq_ld(r_xs, l16, tile * 1024);
          const float4 bq4 = ldg4(a.bias_out + 16 * tile + 4 * kg);
          post_tile(tile, sum, xq, bq4);
        }
      }
      SQ_DRAIN_BARRIER();
      sq16_prime_a<4>(aq, n16i, la_in);
    }
  }
}

}  // namespace osd
