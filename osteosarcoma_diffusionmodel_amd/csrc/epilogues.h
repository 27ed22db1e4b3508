// epilogues.h -- fused epilogues for gemm_kernel (see gemm.h for the fragment layout).
//
// Every epilogue loads its side inputs from CLAMPED indices (always in range) and guards only
// the stores, so in FAST mode no load sits behind a branch and the compiler issues them all
// before a single wait.  fast_ok() tells the host whether FAST's alignment/divisibility
// preconditions hold for the epilogue's own pointers.
#pragma once
#include "gemm.h"
#include "rng.h"
#include "xpose.h"

namespace osd {

constexpr float GN_EPS = 1e-5f;

// SiLU with the hardware exp2 / rcp (v_exp_f32, v_rcp_f32; ~1 ulp each): a handful of VALU issues per
// element instead of the ~30 of expf() + an IEEE divide.  Relative error <= ~4e-7 of |silu(x)|.
__device__ __forceinline__ float silu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// value of lane^32 (the other half-wave), through v_permlane32_swap instead of an LDS bpermute
__device__ __forceinline__ float swap_halves(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  // after the swap: r[0] holds {own low half | partner's low half -> upper lanes ...}; lane < 32 reads r[1]'s low, lane >= 32 reads r[0]'s high
  return __uint_as_float((__lane_id() < 32) ? r[1] : r[0]);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Hook called by every epilogue after each unit of work (an accumulator quad, a group statistic): a kernel whose
// epilogue waves must keep rendezvousing with other waves passes a Sync that turns some calls into barriers; the
// product kernels pass this no-op.
struct NoSync {
  __device__ __forceinline__ void tick() {}
};

// ---- bias (+ optional SiLU, + optional accumulate into out) -----------------------
template <bool SILU, bool ACCUM>
struct EpiBias {
  static constexpr bool COUNTED_STORES = true;    // one float4 store per accumulator quad on a full tile
  static constexpr bool XBUF = false;             // true: apply() takes the wave's LDS transposer region (xpose.h)
  struct Args { const float* bias; float* out; int ldo; long long slice_stride; };
  static bool fast_ok(const Args& a, int F) { return F % 4 == 0 && al16(a.bias) && al16(a.out) && a.ldo % 4 == 0 && a.slice_stride % 4 == 0; }
  static __device__ __forceinline__ void slice(Args& a, int y) { a.out += (long long)y * a.slice_stride; }
  template <int NFB> struct Pre { float4 bias[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        r.bias[fb][q] = a.bias ? ldq<FAST>(a.bias, fw + 32 * fb + 8 * q + 4 * h, F) : make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
  }
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    const int l31 = lane & 31, h = lane >> 5;
    OSD_FOR_QUADS(fb, pb, q) {
      const int f = fw + 32 * fb + 8 * q + 4 * h;
      const int p = pw + 32 * pb + l31;
      const int pc = p < P ? p : P - 1;
      float4 v = make_float4(acc[fb][pb][4 * q], acc[fb][pb][4 * q + 1], acc[fb][pb][4 * q + 2], acc[fb][pb][4 * q + 3]);
      { const float4 bv = pre.bias[fb][q]; v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
      float* row = a.out + (size_t)pc * a.ldo;
      if (ACCUM) { const float4 o = ldq<FAST>(row, f, F); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      if (SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
      if (p < P) stq<FAST>(row, f, F, v);
      sync.tick();
    }
  }
};

// ---- input_proj: h = ((x W^T + b) + t_emb[t]) + c_proj   (models/diffusion.py:229-232) ----
struct EpiInput {
  static constexpr bool KSPLIT2 = true;            // launch.h: the two-wave-group variant of the LDS-DMA kernel is instantiated for it
  static constexpr bool COUNTED_STORES = true;
  static constexpr bool XBUF = false;
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  struct Args {
    const float* bias;
    const float* temb; int ldt;   // [T][F] time_proj(TimeEmbedding(t/T))
    const int* t_index;           // per-row t or null
    const int* t_dev; int t_imm;  // shared t: *t_dev if non-null else t_imm
    const float* cproj; int ldc;  // [P][F]
    float* out; int ldo;
  };
  static bool fast_ok(const Args& a, int F) {
    return F % 4 == 0 && al16(a.bias) && al16(a.temb) && al16(a.cproj) && al16(a.out) && a.ldt % 4 == 0 && a.ldc % 4 == 0 && a.ldo % 4 == 0;
  }
  template <int NFB> struct Pre { float4 bias[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) r.bias[fb][q] = ldq<FAST>(a.bias, fw + 32 * fb + 8 * q + 4 * h, F);
    return r;
  }
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    const int l31 = lane & 31, h = lane >> 5;
    const int t_shared = a.t_dev ? *a.t_dev : a.t_imm;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int p = pw + 32 * pb + l31;
      const int pc = p < P ? p : P - 1;
      const int t = a.t_index ? a.t_index[pc] : t_shared;
      const float* trow = a.temb + (size_t)t * a.ldt;
      const float* crow = a.cproj + (size_t)pc * a.ldc;
      float* orow = a.out + (size_t)pc * a.ldo;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const float4 bv = pre.bias[fb][q], tv = ldq<FAST>(trow, f, F), cv = ldq<FAST>(crow, f, F);
          float4 v;
          v.x = ((acc[fb][pb][4 * q] + bv.x) + tv.x) + cv.x;
          v.y = ((acc[fb][pb][4 * q + 1] + bv.y) + tv.y) + cv.y;
          v.z = ((acc[fb][pb][4 * q + 2] + bv.z) + tv.z) + cv.z;
          v.w = ((acc[fb][pb][4 * q + 3] + bv.w) + tv.w) + cv.w;
          if (p < P) stq<FAST>(orow, f, F, v);
          sync.tick();
        }
    }
  }
};

// ---- Linear -> GroupNorm(8) -> SiLU [-> Dropout]   (models/diffusion.py:200-204) ----
// GW = channels per group (C/8), a power of two in [4,128]; the wave's feature extent
// covers whole groups, so a group's statistics are a sum over this lane's registers plus
// (GW >= 8) one exchange with lane^32.
// DROP compiles the dropout code in; drop_mode then selects 0 none, 1 keep-mask from
// memory, 2 Philox keep-mask.  z_out != null also stores the pre-norm activations and
// (mean, rstd) for backward.
template <int GW, bool DROP>
struct EpiGnSilu {
  static constexpr bool KSPLIT2 = GW == 32 || GW == 64;      // the widths of the BASELINE trunk (256 / 512): two-wave-group variant instantiated
  static constexpr bool COUNTED_STORES = true;
  static constexpr bool XBUF = false;
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  struct Args {
    const float* bias; const float* gamma; const float* beta;
    float* out; int ldo;
    float* z_out; int ldz;            // optional: pre-norm [P][F]
    float* stats;                     // with z_out: [P][F/GW][2] = (mean, rstd)
    int drop_mode;
    const float* mask; int ldm;       // drop_mode 1
    float keep_scale; float p_drop;   // 1/(1-p), p
    uint64_t seed; uint32_t row_offset; uint32_t step; uint32_t tag;   // drop_mode 2
    const int* step_dev;              // drop_mode 2: step = *step_dev when non-null
  };
  static bool fast_ok(const Args& a, int F) {
    return F % 4 == 0 && al16(a.bias) && al16(a.gamma) && al16(a.beta) && al16(a.out) && a.ldo % 4 == 0 &&
           (!a.z_out || (al16(a.z_out) && a.ldz % 4 == 0)) && (a.drop_mode != 1 || (al16(a.mask) && a.ldm % 4 == 0));
  }
  // per-feature parameters of this lane's quads: fetched before the K loop so their latency is hidden
  template <int NFB> struct Pre { float4 bias[NFB][4], gamma[NFB][4], beta[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = fw + 32 * fb + 8 * q + 4 * h;
        r.bias[fb][q] = ldq<FAST>(a.bias, f, F);
        r.gamma[fb][q] = ldq<FAST>(a.gamma, f, F);
        r.beta[fb][q] = ldq<FAST>(a.beta, f, F);
      }
    return r;
  }
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    static_assert(NFB * 32 >= GW, "wave must own whole groups");
    constexpr int RPG = (GW >= 8) ? GW / 2 : 4;   // registers of one group in this lane
    constexpr int NG = NFB * 16 / RPG;
    const int l31 = lane & 31, h = lane >> 5;
    // bias
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = pre.bias[fb][q];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          acc[fb][pb][4 * q] += bv.x; acc[fb][pb][4 * q + 1] += bv.y;
          acc[fb][pb][4 * q + 2] += bv.z; acc[fb][pb][4 * q + 3] += bv.w;
        }
      }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int p = pw + 32 * pb + l31;
      const bool prow = p < P;
      const int pc = prow ? p : P - 1;
      float mean[NG], rstd[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RPG; ++j) { const int L = g * RPG + j; s += acc[L / 16][pb][L % 16]; }
        if (GW >= 8) s += swap_halves(s);
        const float m = s * (1.0f / GW);
        float qs = 0.f;
#pragma unroll
        for (int j = 0; j < RPG; ++j) { const int L = g * RPG + j; const float d = acc[L / 16][pb][L % 16] - m; qs = fmaf(d, d, qs); }
        if (GW >= 8) qs += swap_halves(qs);
        mean[g] = m;
        rstd[g] = 1.0f / sqrtf(qs * (1.0f / GW) + GN_EPS);
        sync.tick();
      }
      float* orow = a.out + (size_t)pc * a.ldo;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const int g = (fb * 16 + 4 * q) / RPG;
          const float4 gv = pre.gamma[fb][q], bev = pre.beta[fb][q];
          const float4 z = make_float4(acc[fb][pb][4 * q], acc[fb][pb][4 * q + 1], acc[fb][pb][4 * q + 2], acc[fb][pb][4 * q + 3]);
          if (a.z_out && prow) {
            stq<FAST>(a.z_out + (size_t)p * a.ldz, f, F, z);
            if ((f % GW) == 0 && f < F) {   // first quad of the group in this lane-pair writes the stats
              float* sp = a.stats + ((size_t)p * (F / GW) + f / GW) * 2;
              sp[0] = mean[g]; sp[1] = rstd[g];
            }
          }
          float4 y;
          y.x = silu_f(fmaf((z.x - mean[g]) * rstd[g], gv.x, bev.x));
          y.y = silu_f(fmaf((z.y - mean[g]) * rstd[g], gv.y, bev.y));
          y.z = silu_f(fmaf((z.z - mean[g]) * rstd[g], gv.z, bev.z));
          y.w = silu_f(fmaf((z.w - mean[g]) * rstd[g], gv.w, bev.w));
          if (DROP && a.drop_mode == 1) {
            const float4 mk = ldq<FAST>(a.mask + (size_t)pc * a.ldm, f, F);
            y.x *= mk.x * a.keep_scale; y.y *= mk.y * a.keep_scale; y.z *= mk.z * a.keep_scale; y.w *= mk.w * a.keep_scale;
          } else if (DROP && a.drop_mode == 2) {
            const uint32_t step = a.step_dev ? (uint32_t)*a.step_dev : a.step;
            const uint4 r = philox_at(a.seed, a.row_offset + (uint32_t)p, (uint32_t)(f >> 2), step, a.tag);
            y.x *= (u01(r.x) >= a.p_drop) ? a.keep_scale : 0.f;
            y.y *= (u01(r.y) >= a.p_drop) ? a.keep_scale : 0.f;
            y.z *= (u01(r.z) >= a.p_drop) ? a.keep_scale : 0.f;
            y.w *= (u01(r.w) >= a.p_drop) ? a.keep_scale : 0.f;
          }
          if (prow) stq<FAST>(orow, f, F, y);
          sync.tick();
        }
    }
  }
};

// ---- dgrad fused with the GroupNorm(8) + SiLU (+ dropout) BACKWARD of the layer it feeds (loss.backward(), utils/train.py:239) ----
// The GEMM is a dgrad: acc = g = dL/d(out) of layer L, where out = dropout(silu(y)), y = zhat * gamma + beta,
// zhat = (z - mean) * rstd (models/diffusion.py:200-204).  A lane holds one row and whole groups of it (the forward epilogue's
// layout), so the per-(row, group) sums of the GroupNorm backward are a register sum + one lane^32 exchange and dL/dz leaves
// the kernel directly -- no separate pass over g:
//   gy   = g * keep * silu'(y),  gzh = gy * gamma
//   gz   = rstd * (gzh - mean_group(gzh) - zhat * mean_group(gzh * zhat))
// `gy` is stored too (into the buffer g would have gone to): d gamma = colsum(gy * zhat), d beta = colsum(gy) are taken from
// it by one grouped column-sum launch per flush (k_gn_colsums); d bias = colsum(gz) rides with the weight-gradient launch.
// g_add: a partial gradient already in the gy buffer (the skip connection's share) is added first, in place.
template <int GW, bool DROP>
struct EpiGnBwd {
  static constexpr bool KSPLIT2 = true;            // launch.h: the two-wave-group variant of gemm_kernel is instantiated for the 64 x 64 tile
  static constexpr bool COUNTED_STORES = false;
  static constexpr bool XBUF = false;
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  struct Args {
    const float* z; int ldz;          // pre-norm activations of layer L  [P][F]
    const float* stats;               // (mean, rstd) [P][F/GW][2]
    const float* gamma; const float* beta;
    float* gz; int ldg;               // dL/dz                            [P][F]
    float* gy; int ldy;               // dL/dy (and, when accumulate, the incoming partial g)  [P][F]
    int accumulate;
    int drop_mode; const float* mask; int ldm; float keep_scale; float p_drop;
    uint64_t seed; uint32_t row_offset; uint32_t step; uint32_t tag;
  };
  static bool fast_ok(const Args& a, int F) {
    return F % 4 == 0 && F % GW == 0 && al16(a.z) && a.ldz % 4 == 0 && al16(a.gamma) && al16(a.beta) && al16(a.gz) && a.ldg % 4 == 0 &&
           al16(a.gy) && a.ldy % 4 == 0 && (a.drop_mode != 1 || (al16(a.mask) && a.ldm % 4 == 0));
  }
  template <int NFB> struct Pre { float4 gamma[NFB][4], beta[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = fw + 32 * fb + 8 * q + 4 * h;
        r.gamma[fb][q] = ldq<FAST>(a.gamma, f, F);
        r.beta[fb][q] = ldq<FAST>(a.beta, f, F);
      }
    return r;
  }
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    static_assert(GW >= 8 && NFB * 32 >= GW, "wave must own whole groups");
    constexpr int RPG = GW / 2;                 // registers of one group in this lane
    constexpr int NG = NFB * 16 / RPG;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int p = pw + 32 * pb + l31;
      const bool prow = p < P;
      const int pc = prow ? p : P - 1;
      const float* zrow = a.z + (size_t)pc * a.ldz;
      float* gyrow = a.gy + (size_t)pc * a.ldy;
      float* gzrow = a.gz + (size_t)pc * a.ldg;
      // pass 1: gy (kept in acc), zhat recomputed in pass 2 from z; group sums
      float zh[NFB][16];
      float s1[NG], s2[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) s1[g] = s2[g] = 0.f;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const int g = (fb * 16 + 4 * q) / RPG;
          const int fc = f < F ? f : F - 4;
          const float2 st = ldg2(a.stats + ((size_t)pc * (F / GW) + fc / GW) * 2);
          const float4 z4 = ldq<FAST>(zrow, f, F);
          float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.accumulate) add = ldq<FAST>(gyrow, f, F);
          const float4 gv = pre.gamma[fb][q], bev = pre.beta[fb][q];
          float keep[4] = {1.f, 1.f, 1.f, 1.f};
          if (DROP && a.drop_mode == 1) {
            const float4 mk = ldq<FAST>(a.mask + (size_t)pc * a.ldm, f, F);
            keep[0] = mk.x * a.keep_scale; keep[1] = mk.y * a.keep_scale; keep[2] = mk.z * a.keep_scale; keep[3] = mk.w * a.keep_scale;
          } else if (DROP && a.drop_mode == 2) {
            const uint4 rr = philox_at(a.seed, a.row_offset + (uint32_t)p, (uint32_t)(f >> 2), a.step, a.tag);
            keep[0] = (u01(rr.x) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[1] = (u01(rr.y) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[2] = (u01(rr.z) >= a.p_drop) ? a.keep_scale : 0.f;
            keep[3] = (u01(rr.w) >= a.p_drop) ? a.keep_scale : 0.f;
          }
          const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
          const float av[4] = {add.x, add.y, add.z, add.w};
          const float gm[4] = {gv.x, gv.y, gv.z, gv.w};
          const float bt[4] = {bev.x, bev.y, bev.z, bev.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float zhat = (zv[e] - st.x) * st.y;
            const float y = zhat * gm[e] + bt[e];
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));   // hardware exp2 / rcp, as silu_f
            const float gyv = (acc[fb][pb][r] + av[e]) * keep[e] * (sg * (1.0f + y * (1.0f - sg)));
            const float gzh = gyv * gm[e];
            zh[fb][r] = zhat;
            acc[fb][pb][r] = gyv;
            s1[g] += gzh;
            s2[g] += gzh * zhat;
          }
          if (prow) stq<FAST>(gyrow, f, F, make_float4(acc[fb][pb][4 * q], acc[fb][pb][4 * q + 1], acc[fb][pb][4 * q + 2], acc[fb][pb][4 * q + 3]));
        }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        s1[g] += swap_halves(s1[g]);
        s2[g] += swap_halves(s2[g]);
        s1[g] *= (1.0f / GW);
        s2[g] *= (1.0f / GW);
      }
      // pass 2: dL/dz
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const int g = (fb * 16 + 4 * q) / RPG;
          const int fc = f < F ? f : F - 4;
          const float rstd = ldg1(a.stats + ((size_t)pc * (F / GW) + fc / GW) * 2 + 1);
          const float4 gv = pre.gamma[fb][q];
          const float gm[4] = {gv.x, gv.y, gv.z, gv.w};
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            o[e] = rstd * (acc[fb][pb][r] * gm[e] - s1[g] - zh[fb][r] * s2[g]);
          }
          if (prow) stq<FAST>(gzrow, f, F, make_float4(o[0], o[1], o[2], o[3]));
          sync.tick();
        }
    }
  }
};

// ---- output_proj fused with the DDPM posterior update (models/diffusion.py:398-425) ----
// eps = acc + bias; the reference's x0 = (x - c0*eps)/c1, mean = c2*x0/c3 + c4*x/c3, x' = mean + c5*z
// is linear in (x, eps, z):   x' = A_t*x + B_t*eps + C_t*z   with
//   A_t = c4/c3 + c2/(c1*c3),  B_t = -c0*c2/(c1*c3),  C_t = c5        (t > 0)
//   A_0 = 1/c1,                B_0 = -c0/c1,          C_0 = 0         (t == 0: x' = x0)
// A, B, C are formed in double on the host from the reference's fp32 scalars and rounded once
// (osd_set_schedule), which replaces three IEEE divides per element by two FMAs; the result
// differs from the reference's op order by a few ulp of the same intermediate magnitudes.
struct EpiPosterior {
  static constexpr bool COUNTED_STORES = true;
  static constexpr bool XBUF = false;
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  struct Args {
    const float* bias;
    const float* xin; int ldx;
    float* xout; int ldo;
    const float* coef;              // dev [T][4] = (A_t, B_t, C_t, 0)
    const int* t_dev; int t_imm;
    const float* z; int ldzz;       // injected noise for draw 0 (t = t_first), [P][F]; null -> Philox
    long long z_step_stride; int t_first;   // draw for step t sits at z + (t_first - t) * stride
    uint64_t seed; uint32_t row_offset;
    float* mut_mask; int mutation_dim;   // written at t == 0 when non-null: (x' > 0.5)
  };
  static bool fast_ok(const Args& a, int F) {
    return F % 4 == 0 && al16(a.bias) && al16(a.xin) && al16(a.xout) && a.ldx % 4 == 0 && a.ldo % 4 == 0 &&
           (!a.z || (al16(a.z) && a.ldzz % 4 == 0 && a.z_step_stride % 4 == 0));
  }
  template <int NFB> struct Pre { float4 bias[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) r.bias[fb][q] = ldq<FAST>(a.bias, fw + 32 * fb + 8 * q + 4 * h, F);
    return r;
  }
  // WIDE: every x_t quad of the wave's tile is requested before the first one is used (one latency, not sixteen; 64
  // registers).  !WIDE (the chain kernel, which has fewer registers to spare): the four quads of one 32 x 32 block at a time.
  template <int NFB, int NPB, bool FAST, class Sync = NoSync, bool WIDE = true>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    const int l31 = lane & 31, h = lane >> 5;
    const int t = a.t_dev ? *a.t_dev : a.t_imm;
    const float* c = a.coef + 4 * t;
    const float cA = c[0], cB = c[1], cC = c[2];
    const float* zbase = a.z ? a.z + (long long)(a.t_first - t) * a.z_step_stride : nullptr;
    float4 xq[WIDE ? NFB : 1][WIDE ? NPB : 1][4];
    if constexpr (WIDE) {
      OSD_FOR_QUADS(fb, pb, q) {
        const int p = pw + 32 * pb + l31;
        xq[fb][pb][q] = ldq<FAST>(a.xin + (size_t)(p < P ? p : P - 1) * a.ldx, fw + 32 * fb + 8 * q + 4 * h, F);
      }
    }
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) {
        const int p = pw + 32 * pb + l31;
        const int pc = p < P ? p : P - 1;
        if constexpr (!WIDE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xq[0][0][q] = ldq<FAST>(a.xin + (size_t)pc * a.ldx, fw + 32 * fb + 8 * q + 4 * h, F);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const bool ok = p < P && f < F;
          const float4 bv = pre.bias[fb][q];
          const float4 x = xq[WIDE ? fb : 0][WIDE ? pb : 0][q];
          const float e[4] = {acc[fb][pb][4 * q] + bv.x, acc[fb][pb][4 * q + 1] + bv.y, acc[fb][pb][4 * q + 2] + bv.z, acc[fb][pb][4 * q + 3] + bv.w};
          const float xv[4] = {x.x, x.y, x.z, x.w};
          float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t > 0) {
            if (zbase) zz = ldq<FAST>(zbase + (size_t)pc * a.ldzz, f, F);
            else zz = randn4(a.seed, a.row_offset + (uint32_t)p, (uint32_t)(f >> 2), (uint32_t)t, TAG_POSTERIOR);
          }
          const float zv[4] = {zz.x, zz.y, zz.z, zz.w};
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf(cA, xv[r], fmaf(cB, e[r], cC * zv[r]));
          if (t == 0 && a.mut_mask && ok && f < a.mutation_dim) {
            float* mrow = a.mut_mask + (size_t)p * a.mutation_dim;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < a.mutation_dim) mrow[f + r] = (o[r] > 0.5f) ? 1.0f : 0.0f;
          }
          if (p < P) stq<FAST>(a.xout + (size_t)pc * a.ldo, f, F, make_float4(o[0], o[1], o[2], o[3]));
          sync.tick();
        }
      }
  }
};

// ---- output_proj fused with the MSE loss (models/diffusion.py:373-377) and its gradient ----
// d = (acc + bias) - noise;  loss += sum d^2 * inv_count;  dout = d * gscale
struct EpiMse {
  static constexpr bool COUNTED_STORES = false;   // dout / pred are optional
  static constexpr bool XBUF = true;              // the noise target comes in, dL/d eps goes out, as full row segments
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  struct Args {
    const float* bias; const float* noise; int ldn;
    float* dout; int ldd;         // may be null (validation: loss only)
    float* pred; int ldp;         // may be null
    float* loss;                  // dev float[1], atomically accumulated
    float inv_count; float gscale;
  };
  static bool fast_ok(const Args& a, int F) {
    return F % 4 == 0 && al16(a.bias) && al16(a.noise) && a.ldn % 4 == 0 && (!a.dout || (al16(a.dout) && a.ldd % 4 == 0)) &&
           (!a.pred || (al16(a.pred) && a.ldp % 4 == 0));
  }
  template <int NFB> struct Pre { float4 bias[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) r.bias[fb][q] = ldq<FAST>(a.bias, fw + 32 * fb + 8 * q + 4 * h, F);
    return r;
  }
  // xbuf (FAST only): the wave's transposer region; the 32-feature blocks of the noise target are read, and those of dL/d eps
  // written, as full 128-byte row segments (xpose.h) instead of 32 rows x 32 bytes per wave-instruction
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               float* xbuf = nullptr, Sync&& sync = Sync()) {
    const int l31 = lane & 31, h = lane >> 5;
    float part = 0.f;
    if (FAST && xbuf) {
      const WaveXpose<NPB> xp{xbuf};
      const int rows = P - pw;                         // valid rows of the wave's block (uniform)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        const int cols = F - (fw + 32 * fb);            // valid features of this block (uniform)
        if (rows <= 0 || cols <= 0) continue;
        xp.template load_rows<true>(a.noise + (size_t)pw * a.ldn + fw + 32 * fb, a.ldn, lane, rows, cols);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int fo = 8 * q + 4 * h;
            const int p = 32 * pb + l31;
            const bool prow = p < rows;
            const float4 bv = pre.bias[fb][q];
            const float4 nz = xp.get(pb, q, l31, h);
            const float4 e = make_float4(acc[fb][pb][4 * q] + bv.x, acc[fb][pb][4 * q + 1] + bv.y, acc[fb][pb][4 * q + 2] + bv.z, acc[fb][pb][4 * q + 3] + bv.w);
            if (a.pred && prow) stq<FAST>(a.pred + (size_t)(pw + p) * a.ldp, fw + 32 * fb + fo, F, e);
            float4 d = make_float4(e.x - nz.x, e.y - nz.y, e.z - nz.z, e.w - nz.w);
            if (!prow || fo >= cols) d.x = 0.f;
            if (!prow || fo + 1 >= cols) d.y = 0.f;
            if (!prow || fo + 2 >= cols) d.z = 0.f;
            if (!prow || fo + 3 >= cols) d.w = 0.f;
            part += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            xp.put(pb, q, l31, h, make_float4(d.x * a.gscale, d.y * a.gscale, d.z * a.gscale, d.w * a.gscale));
          }
        if (a.dout) xp.template store_rows<true>(a.dout + (size_t)pw * a.ldd + fw + 32 * fb, a.ldd, lane, rows, cols);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
      // one atomic per workgroup (the four waves meet in LDS): same-address float atomics serialise in L2
      __shared__ float wave_part[4];               // 16 bytes: keeps the dynamic LDS base 16-byte aligned
      const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
      if (lane == 0) wave_part[wv] = part;
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(a.loss, ((wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3])) * a.inv_count);
      return;
    }
    OSD_FOR_QUADS(fb, pb, q) {
      const int f = fw + 32 * fb + 8 * q + 4 * h;
      const int p = pw + 32 * pb + l31;
      const int pc = p < P ? p : P - 1;
      const bool prow = p < P;
      const float4 bv = pre.bias[fb][q];
      const float4 nz = ldq<FAST>(a.noise + (size_t)pc * a.ldn, f, F);
      const float4 e = make_float4(acc[fb][pb][4 * q] + bv.x, acc[fb][pb][4 * q + 1] + bv.y, acc[fb][pb][4 * q + 2] + bv.z, acc[fb][pb][4 * q + 3] + bv.w);
      if (a.pred && prow) stq<FAST>(a.pred + (size_t)p * a.ldp, f, F, e);
      float4 d = make_float4(e.x - nz.x, e.y - nz.y, e.z - nz.z, e.w - nz.w);
      if (!prow || f >= F) d.x = 0.f;
      if (!prow || f + 1 >= F) d.y = 0.f;
      if (!prow || f + 2 >= F) d.z = 0.f;
      if (!prow || f + 3 >= F) d.w = 0.f;
      part += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      if (a.dout && prow) stq<FAST>(a.dout + (size_t)p * a.ldd, f, F, make_float4(d.x * a.gscale, d.y * a.gscale, d.z * a.gscale, d.w * a.gscale));
      sync.tick();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) atomicAdd(a.loss, part * a.inv_count);
  }
};

// ---- RBF-kernel sum for the MMD metric (utils/validation.py:287-296) -------------------------------
// acc = x_f . y_p;  d2 = |x_f|^2 + |y_p|^2 - 2 acc;  sum += exp(-gamma * max(d2, 0)) over the valid tile
struct EpiRbfSum {
  static constexpr bool COUNTED_STORES = false;   // stores nothing
  static constexpr bool XBUF = false;
  template <class A> static __device__ __forceinline__ void slice(A&, int) {}
  // sum: RBF_SLOTS doubles; a workgroup adds into slot blockIdx.x % RBF_SLOTS (same-address atomics serialise in L2: 153 000
  // tiles of a 50 000 x 50 000 block on ONE address cost ~2 % of the kernel), the host adds the slots up
  static constexpr int RBF_SLOTS = 256;
  // tri: the product is symmetric and only tiles with feature tile <= patient tile run (GemmArgs::tri, 128 x 128 tiles): a tile
  // strictly above the diagonal stands for its mirror image too and counts twice; the diagonal tiles are computed whole
  // (utils/validation.py:286-296 sums the full matrix, diagonal included)
  struct Args { const float* sqa; const float* sqb; float gamma; double* sum; int tri; };
  static bool fast_ok(const Args& a, int F) { return F % 4 == 0 && al16(a.sqa); }
  template <int NFB> struct Pre { float4 sqa[NFB][4]; };
  template <int NFB, bool FAST>
  static __device__ __forceinline__ Pre<NFB> prefetch(const Args& a, int fw, int lane, int F) {
    Pre<NFB> r;
    const int h = lane >> 5;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) r.sqa[fb][q] = ldq<FAST>(a.sqa, fw + 32 * fb + 8 * q + 4 * h, F);
    return r;
  }
  template <int NFB, int NPB, bool FAST, class Sync = NoSync>
  static __device__ __forceinline__ void apply(f32x16 (&acc)[NFB][NPB], const Args& a, const Pre<NFB>& pre, int fw, int pw, int lane, int F, int P,
                                               Sync&& sync = Sync()) {
    const int l31 = lane & 31, h = lane >> 5;
    float part = 0.f;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      const int p = pw + 32 * pb + l31;
      const float sb = a.sqb[p < P ? p : P - 1];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = fw + 32 * fb + 8 * q + 4 * h;
          const float4 sa = pre.sqa[fb][q];
          const float sav[4] = {sa.x, sa.y, sa.z, sa.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d2 = fmaxf(sav[r] + sb - 2.0f * acc[fb][pb][4 * q + r], 0.f);
            const float k = expf(-a.gamma * d2);
            part += (p < P && f + r < F) ? k : 0.f;
          }
        }
    }
    double dp = (double)part;
    if (a.tri && (fw >> 7) < (pw >> 7)) dp *= 2.0;       // uniform over the workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dp += __shfl_xor(dp, o);
    // one atomic per workgroup: a 50 000 x 50 000 Gram block is 153 000 tiles, and same-address atomics serialise in L2
    __shared__ double wave_sum[4];                 // 32 bytes: the dynamic LDS base stays 16-byte aligned
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane == 0) wave_sum[wv] = dp;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.sum + (blockIdx.x % RBF_SLOTS), (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]));
  }
};

}  // namespace osd
