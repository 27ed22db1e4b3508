// gemm_bf3.h -- fp32-accuracy GEMM on the bf16 matrix pipe ("bf16x3 split"), opt-in (osd_set_option("precision", 1)).
//
//   out[p][f] = epilogue( sum_k W[f][k] * X[p][k] )          models/diffusion.py:198-256 (F.linear in fp32)
//
// v_mfma_f32_32x32x2_f32 issues at the fp32 VECTOR rate (64 cycles per 32x32x2, 1/16 of the bf16 rate) and blocks the SIMD's
// vector issue while it runs.  Here every fp32 operand is carried as THREE bf16 planes
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// which is EXACT (3 x 8 significand bits = fp32's 24; each residual is exactly representable), and a product is formed from six
// of the nine cross terms, a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (the dropped ones are below 2^-24 of the product), each a
// v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation): 6 x 32 = 192 matrix-pipe cycles per 16 k against 8 x 64 = 512,
// and a bf16 MFMA holds the vector issue for 8 of its 32 cycles only, so the other workgroup's epilogue runs beside it.
//
// Operands are pre-split where they are PRODUCED (weights once per osd_load_weights, activations by the epilogue that writes
// them), never in the K loop, and stored fragment-major:
//
//   planes buffer of an operand with R rows (features of a weight, patients of an activation) and K columns, in 16-byte units:
//     unit(((tile * nkb + kb) * 3 + plane) * 4 + rb) * 64 + lane,   tile = row / 128, rb = (row % 128) / 32,
//     lane = 32 * ((k % 16) / 8) + row % 32,   8 bf16 of k % 8 inside the unit,   nkb = ceil(K / 16)
//   i.e. one (tile, 16-k block) is 12 KiB: plane-major, then the four 32-row blocks, each exactly the 64-lane A/B operand of one
//   v_mfma_f32_32x32x16_bf16 (lane l: row l % 32, k 8 (l / 32) .. + 7).
//
// Consequences: a K stage arrives by LDS-DMA as linear 1 KiB pieces (global address = base + lane * 16, LDS image = the same
// bytes) and a fragment read is ds_read_b128 at lane * 16: no swizzle, no bank conflict, no address arithmetic in the loop; an
// epilogue store is 1 KiB contiguous per wave-instruction.
//
// Tile 128 features x 128 patients, 4 waves of 64 x 64 (the fp32 kernels' accumulator layout, gemm.h: the epilogue arithmetic is
// theirs), two workgroups per CU.  K loop: a ring of three 24 KiB LDS slots (A 12 KiB | B 12 KiB per 16-k stage); in step s a
// wave reads the fragments of stage s + 1, issues the DMA of stage s + 3 into the slot stage s has left, runs the 24 MFMAs of
// stage s and waits with a COUNTED vmcnt (the six newest pieces stay in flight): a stage has two steps to land.
#pragma once
#include "gemm_glds.h"
#include "epilogues.h"

namespace osd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u32 gv4u32;

constexpr int B3_ROWS = 128;                 // rows per tile of a planes buffer
constexpr int B3_KB = 16;                    // k per block
constexpr int B3_STAGE_U4 = 768;             // 16-byte units per (tile, k block): 3 planes x 4 row blocks x 64 lanes
constexpr int B3_STAGE_BYTES = B3_STAGE_U4 * 16;
constexpr int B3_SLOT_BYTES = 2 * B3_STAGE_BYTES;          // A stage | B stage
#ifndef B3_RING
#define B3_RING 3                            // slots of the K ring.  2 (experiment: 50 KB of LDS, THREE workgroups per CU, a stage has one step to land)
#endif
constexpr int B3_SLOTS = B3_RING;
constexpr int B3_LDS_RING_BYTES = B3_SLOTS * B3_SLOT_BYTES;             // 73 728
constexpr int B3_LDS_BYTES = B3_LDS_RING_BYTES + 384 * 4;               // + the tile's per-feature parameters: 75 264, two workgroups per CU

inline int64_t b3_tiles(int64_t rows) { return (rows + B3_ROWS - 1) / B3_ROWS; }
inline int b3_nkb(int K) { return (K + B3_KB - 1) / B3_KB; }
// 16-byte units of a planes buffer
inline int64_t b3_units(int64_t rows, int K) { return b3_tiles(rows) * b3_nkb(K) * B3_STAGE_U4; }

struct Bf3Args {
  const uint4* A; int nkb;        // weight planes [ceil(F/128)][nkb][768]; rows beyond F and k beyond K are zero
  const uint4* B0; int nkb0;      // input panel 0: k blocks [0, nkb0) of the reduction, buffer [ceil(P/128)][nkb0][768]
  const uint4* B1; int nkb1;      // input panel 1: k blocks [nkb0, nkb), taken from blocks [0, nkb - nkb0) of a buffer with nkb1 blocks per tile
  int F, P;                       // valid features / rows (tiles are always computed whole; rows beyond P hold finite leftovers)
  unsigned long long* stamps;     // diagnostic (null in production): per workgroup 4 words: K loop in shader cycles / in 100 MHz ticks, epilogue cycles
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const bf16x2 p = {(__bf16)a, (__bf16)b};         // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// four fp32 values -> three planes of four bf16 (two dwords each); v == p0 + p1 + p2 exactly for finite v
struct Split4 { uint2 p[3]; };
__device__ __forceinline__ Split4 split4(float4 v) {
  Split4 s;
  const unsigned a0 = pk_bf16(v.x, v.y), a1 = pk_bf16(v.z, v.w);
  const float rx = v.x - bf_lo(a0), ry = v.y - bf_hi(a0), rz = v.z - bf_lo(a1), rw = v.w - bf_hi(a1);
  const unsigned b0 = pk_bf16(rx, ry), b1 = pk_bf16(rz, rw);
  const float sx = rx - bf_lo(b0), sy = ry - bf_hi(b0), sz = rz - bf_lo(b1), sw = rw - bf_hi(b1);
  s.p[0] = make_uint2(a0, a1);
  s.p[1] = make_uint2(b0, b1);
  s.p[2] = make_uint2(pk_bf16(sx, sy), pk_bf16(sz, sw));
  return s;
}

__device__ __forceinline__ void stg_u4(uint4* p, uint4 v) { const v4u32 w = {v.x, v.y, v.z, v.w}; *(gv4u32*)p = w; }

// Two accumulator quads of one patient -- X = features 16 j + 4 h .. + 3 (quad 2 j), Y = features 16 j + 8 + 4 h .. + 3 (quad 2 j + 1)
// of a 32-feature block -- become the lane's 16-byte unit of k block `kb` of the output planes: lane (l31, h = 0) stores features
// 16 j .. + 7 (its own X and the partner's X), lane (l31, h = 1) features 16 j + 8 .. + 15 (the partner's Y and its own): one
// v_permlane32_swap per dword, then a wave-instruction stores 1 KiB contiguous.  `o` = unit 0 of the output tile + lane.
__device__ __forceinline__ void b3_put_pair(uint4* __restrict__ o, int kb, int rb, float4 X, float4 Y) {
  const Split4 sx = split4(X), sy = split4(Y);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(sx.p[pl].x, sy.p[pl].x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(sx.p[pl].y, sy.p[pl].y, false, false);
    stg_u4(o + ((kb * 3 + pl) * 4 + rb) * 64, make_uint4(r0[0], r1[0], r0[1], r1[1]));
  }
}

// ---- epilogues ------------------------------------------------------------------------------------------------------------------
// An epilogue consumes the wave's 64 x 64 accumulators (fragment layout of gemm.h).  Per-feature parameters of the 128-feature tile
// arrive in LDS with the first K stage (B3Ctx::prm: bias | second array | third array, 128 floats each; which arrays: Epi::prm_ptr),
// so no epilogue starts with a round trip to L2.
struct B3Ctx {
  int f0;            // first feature of the tile
  int fl;            // first feature of the wave inside the tile (0 / 64)
  int pt, rb0;       // row tile; first 32-row block of the wave inside it
  int p0w;           // first row of the wave
  int lane, wave;
  int F, P;
  const float* prm;  // LDS: [0,128) array 0 | [128,256) array 1 | [256,384) array 2, features f0 .. f0 + 127
  float* xreg;       // LDS: the wave's 16 KiB of the (free) K ring, EpiB3Post's x tile
};
constexpr int B3_PRM_FLOATS = 384;

// planes destination of a layer output
struct B3Out { uint4* planes; int nkb; };     // [tiles][nkb][768]

struct B3NoState {};
// defaults shared by the epilogues
struct EpiB3Base {
  typedef B3NoState State;
  static constexpr bool LOOP_RNG = false;      // true: kstep() runs inside the K loop (needs the unrolled, NKB > 0 kernel)
  static constexpr bool X_TILE = false;        // true: xtile_issue() at the top of the last K step, into the ring the loop has left
};

// fp32 rows out (+ bias): the noise prediction of osd_denoiser_forward, osd_op_linear
struct EpiB3Bias : EpiB3Base {
  struct Args { const float* bias; float* out; int ldo; };
  static __device__ __forceinline__ const float* prm_ptr(const Args& a, int k) { return k == 0 ? a.bias : nullptr; }
  static __device__ __forceinline__ void apply(f32x16 (&acc)[2][2], const Args& a, const B3Ctx& c, State&) {
    const int l31 = c.lane & 31, h = c.lane >> 5;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int fo = c.fl + 32 * fb + 8 * q + 4 * h;
        const int f = c.f0 + fo;
        const float4 bv = a.bias ? *reinterpret_cast<const float4*>(c.prm + fo) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int p = c.p0w + 32 * pb + l31;
          if (p < c.P) {
            float* row = a.out + (size_t)p * a.ldo;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < c.F) row[f + r] = acc[fb][pb][4 * q + r] + b4[r];
          }
        }
      }
  }
};

// input_proj: h = ((acc + b) + t_emb[t]) + c_proj   (models/diffusion.py:229-232; EpiInput's arithmetic), planes out.  F % 128 == 0.
struct EpiB3Input : EpiB3Base {
  struct Args {
    const float* bias; const float* temb; int ldt; const int* t_index; const int* t_dev; int t_imm;
    const float* cproj; int ldc; B3Out o;
  };
  // shared t (the reverse chain): the time-embedding row rides in the parameter block; per-row t: gathered per row below
  static __device__ __forceinline__ const float* prm_ptr(const Args& a, int k) {
    if (k == 0) return a.bias;
    if (k == 1 && !a.t_index) return a.temb + (size_t)(a.t_dev ? *a.t_dev : a.t_imm) * a.ldt;
    return nullptr;
  }
  static __device__ __forceinline__ void apply(f32x16 (&acc)[2][2], const Args& a, const B3Ctx& c, State&) {
    const int l31 = c.lane & 31, h = c.lane >> 5;
    uint4* const ob = a.o.planes + (size_t)c.pt * a.o.nkb * B3_STAGE_U4 + c.lane;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int p = c.p0w + 32 * pb + l31;
      const int pc = p < c.P ? p : c.P - 1;
      const float* trow = a.t_index ? a.temb + (size_t)a.t_index[pc] * a.ldt + c.f0 : nullptr;
      const float* crow = a.cproj + (size_t)pc * a.ldc + c.f0;
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 v[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * j + e;
            const int fo = c.fl + 32 * fb + 8 * q + 4 * h;
            const float4 bv = *reinterpret_cast<const float4*>(c.prm + fo);
            const float4 tv = trow ? ldg4(trow + fo) : *reinterpret_cast<const float4*>(c.prm + 128 + fo);
            const float4 cv = ldg4(crow + fo);
            v[e].x = ((acc[fb][pb][4 * q] + bv.x) + tv.x) + cv.x;
            v[e].y = ((acc[fb][pb][4 * q + 1] + bv.y) + tv.y) + cv.y;
            v[e].z = ((acc[fb][pb][4 * q + 2] + bv.z) + tv.z) + cv.z;
            v[e].w = ((acc[fb][pb][4 * q + 3] + bv.w) + tv.w) + cv.w;
          }
          b3_put_pair(ob, (c.f0 + c.fl + 32 * fb) / 16 + j, c.rb0 + pb, v[0], v[1]);
        }
    }
  }
};

// Linear -> GroupNorm(8) -> SiLU   (models/diffusion.py:200-204; EpiGnSilu<GW, false>'s arithmetic), planes out.  F % 128 == 0.
template <int GW>
struct EpiB3Gn : EpiB3Base {
  struct Args { const float* bias; const float* gamma; const float* beta; B3Out o; };
  static __device__ __forceinline__ const float* prm_ptr(const Args& a, int k) { return k == 0 ? a.bias : (k == 1 ? a.gamma : a.beta); }
  static __device__ __forceinline__ void apply(f32x16 (&acc)[2][2], const Args& a, const B3Ctx& c, B3NoState&) {
    static_assert(GW == 32 || GW == 64, "group widths of the 256 / 512 wide trunk");
    constexpr int RPG = GW / 2;                 // registers of one group in this lane
    constexpr int NG = 32 / RPG;                // groups of the wave's 64 features
    const int h = c.lane >> 5;
    uint4* const ob = a.o.planes + (size_t)c.pt * a.o.nkb * B3_STAGE_U4 + c.lane;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = *reinterpret_cast<const float4*>(c.prm + c.fl + 32 * fb + 8 * q + 4 * h);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          acc[fb][pb][4 * q] += bv.x; acc[fb][pb][4 * q + 1] += bv.y;
          acc[fb][pb][4 * q + 2] += bv.z; acc[fb][pb][4 * q + 3] += bv.w;
        }
      }
    float mean[2][NG], rstd[2][NG];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
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
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 gv[2], bev[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int fo = c.fl + 32 * fb + 8 * (2 * j + e) + 4 * h;
          gv[e] = *reinterpret_cast<const float4*>(c.prm + 128 + fo);
          bev[e] = *reinterpret_cast<const float4*>(c.prm + 256 + fo);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          float4 y[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * j + e;
            const int g = (fb * 16 + 4 * q) / RPG;
            const float m = mean[pb][g], r = rstd[pb][g];
            y[e].x = silu_f(fmaf((acc[fb][pb][4 * q] - m) * r, gv[e].x, bev[e].x));
            y[e].y = silu_f(fmaf((acc[fb][pb][4 * q + 1] - m) * r, gv[e].y, bev[e].y));
            y[e].z = silu_f(fmaf((acc[fb][pb][4 * q + 2] - m) * r, gv[e].z, bev[e].z));
            y[e].w = silu_f(fmaf((acc[fb][pb][4 * q + 3] - m) * r, gv[e].w, bev[e].w));
          }
          b3_put_pair(ob, (c.f0 + c.fl + 32 * fb) / 16 + j, c.rb0 + pb, y[0], y[1]);
        }
      }
  }
};

// output_proj + DDPM posterior update (models/diffusion.py:398-425; EpiPosterior's arithmetic): x' = A_t x + B_t (acc + b) + C_t z.
// The fp32 chain state [n][ldx] is read and written in place (what the caller gets back); the planes of x' feed the next step's
// input_proj.  Features beyond F = D inside the last 16-k block are written as zeros (they meet zero weights).
//   * z: the sixteen Philox blocks of a lane (4 normals each, ~100 VALU instructions) are generated INSIDE the K loop, one block per
//     16-k step -- a bf16 MFMA holds the vector issue for 8 of its 32 cycles, so the generator runs in the matrix loop's shadow
//     instead of being 60 % of the epilogue (LOOP_RNG; needs the kernel unrolled over NKB = 16 or 32 steps).
//   * x_t: the wave's 64 x 64 tile arrives by LDS-DMA as 256-byte row segments during the last K step, into the ring the loop has
//     left (16-byte chunks XOR-swizzled by row & 15, applied to the source address and again on the fragment-side access), x' goes
//     back through the same image and leaves as row segments (X_TILE; needs 16-byte aligned rows, else the direct accesses below).
struct EpiB3Post : EpiB3Base {
  static constexpr bool LOOP_RNG = true;
  static constexpr bool X_TILE = true;
  struct Args {
    const float* bias; float* x; int ldx; const float* coef; const int* t_dev; int t_imm;
    const float* z; int ldzz; long long z_step_stride; int t_first;
    uint64_t seed; uint32_t row_offset; float* mut_mask; int mutation_dim; B3Out o;
    int x_tile;                   // host: rows of x 16-byte aligned (ldx % 4 == 0, aligned base, F % 4 == 0)
  };
  struct State { float4 z[16]; int t; };
  static __device__ __forceinline__ const float* prm_ptr(const Args& a, int k) { return k == 0 ? a.bias : nullptr; }
  static __device__ __forceinline__ void init(State& st, const Args& a) { st.t = a.t_dev ? *a.t_dev : a.t_imm; }
  // block i = (fb * 2 + pb) * 4 + q of the wave's tile
  static __device__ __forceinline__ float4 draw(const Args& a, const B3Ctx& c, int t, int i) {
    const int fb = i >> 3, pb = (i >> 2) & 1, q = i & 3;
    const int p = c.p0w + 32 * pb + (c.lane & 31);
    const int f = c.f0 + c.fl + 32 * fb + 8 * q + 4 * (c.lane >> 5);
    return randn4(a.seed, a.row_offset + (uint32_t)p, (uint32_t)(f >> 2), (uint32_t)t, TAG_POSTERIOR);
  }
  static __device__ __forceinline__ bool rng_on(const Args& a, const State& st) { return st.t > 0 && !a.z; }      // uniform
  static __device__ __forceinline__ void kstep(State& st, const Args& a, const B3Ctx& c, int i) { st.z[i] = draw(a, c, st.t, i); }
  // x_t tile of the wave -> LDS: piece i = rows 4 i .. 4 i + 3, 256 bytes each; lane L holds position L % 16 of row 4 i + L / 16,
  // which is source chunk (L % 16) ^ (row & 15).  Rows beyond P / chunks beyond F re-read valid ones (never stored).
  static __device__ __forceinline__ void xtile_issue(const Args& a, const B3Ctx& c) {
    const int rows = c.P - c.p0w;                 // valid rows of the wave (may be <= 0: the wave then loads row P - 1 copies)
    const int cmax = (c.F - (c.f0 + c.fl)) / 4 - 1;   // last valid 16-byte chunk of the wave's 64 features (may be < 0)
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(c.xreg));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 4 * i + (c.lane >> 4);
      int ch = (c.lane & 15) ^ (r & 15);
      ch = ch < cmax ? ch : (cmax > 0 ? cmax : 0);
      long long row = (long long)c.p0w + (r < rows ? r : rows - 1);
      row = row > 0 ? row : 0;
      int f = c.f0 + c.fl + 4 * ch;
      f = f < c.F - 4 ? f : c.F - 4;
      glds16(a.x + row * a.ldx + f, __builtin_amdgcn_readfirstlane(dst + (unsigned)i * 1024u));
    }
  }
  static __device__ __forceinline__ void apply(f32x16 (&acc)[2][2], const Args& a, const B3Ctx& c, State& st) {
    const int l31 = c.lane & 31, h = c.lane >> 5;
    const int t = st.t;
    const float* cf = a.coef + 4 * t;
    const float cA = cf[0], cB = cf[1], cC = cf[2];
    const float* zbase = a.z ? a.z + (long long)(a.t_first - t) * a.z_step_stride : nullptr;
    const bool do_mask = t == 0 && a.mut_mask != nullptr;
    const bool zal = zbase && (a.ldzz & 3) == 0 && (reinterpret_cast<uintptr_t>(zbase) & 15) == 0;
    const int f0w = c.f0 + c.fl;
    uint4* const ob = a.o.planes + (size_t)c.pt * a.o.nkb * B3_STAGE_U4 + c.lane;
    // B_t (acc + b) + C_t z first -- it needs no x_t and frees the generator's registers -- while the x tile is still landing
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int fo = c.fl + 32 * fb + 8 * q + 4 * h;
        const int f = c.f0 + fo;
        const float4 b4 = *reinterpret_cast<const float4*>(c.prm + fo);
        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          float zv[4] = {0.f, 0.f, 0.f, 0.f};
          if (t > 0) {
            if (zbase) {
              const int p = c.p0w + 32 * pb + l31;
              const size_t zo = (size_t)(p < c.P ? p : c.P - 1) * a.ldzz;
              if (zal && f + 3 < c.F) { const float4 z4 = ldg4(zbase + zo + f); zv[0] = z4.x; zv[1] = z4.y; zv[2] = z4.z; zv[3] = z4.w; }
              else {
#pragma unroll
                for (int r = 0; r < 4; ++r) zv[r] = f + r < c.F ? zbase[zo + f + r] : 0.f;
              }
            } else {
              const float4 z4 = st.z[(fb * 2 + pb) * 4 + q];
              zv[0] = z4.x; zv[1] = z4.y; zv[2] = z4.z; zv[3] = z4.w;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[fb][pb][4 * q + r] = fmaf(cB, acc[fb][pb][4 * q + r] + bv[r], cC * zv[r]);
        }
      }
    if (a.x_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the x tile has landed (asm DMAs are invisible to hipcc)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
      if (f0w + 32 * fb >= c.F) break;                 // uniform
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int rl = 32 * pb + l31;                  // row inside the wave's rows
        const int p = c.p0w + rl;
        const bool prow = p < c.P;
        const int pc = prow ? p : c.P - 1;
        float* xrow = a.x + (size_t)pc * a.ldx;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (f0w + 32 * fb + 16 * j >= c.F) break;    // uniform: no k block of the planes there
          float4 o[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * j + e;
            const int fo = c.fl + 32 * fb + 8 * q + 4 * h;
            const int f = c.f0 + fo;
            float xv[4];
            float* xs = nullptr;
            if (a.x_tile) {
              xs = c.xreg + rl * 64 + 4 * ((8 * fb + 2 * q + h) ^ (rl & 15));
              const float4 x4 = *reinterpret_cast<const float4*>(xs);
              xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) xv[r] = f + r < c.F ? xrow[f + r] : 0.f;
            }
            float ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ov[r] = fmaf(cA, xv[r], acc[fb][pb][4 * q + r]);
              if (f + r >= c.F) ov[r] = 0.f;
            }
            if (do_mask && prow && f < a.mutation_dim) {
              float* mrow = a.mut_mask + (size_t)p * a.mutation_dim;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (f + r < a.mutation_dim) mrow[f + r] = (ov[r] > 0.5f) ? 1.0f : 0.0f;
            }
            o[e] = make_float4(ov[0], ov[1], ov[2], ov[3]);
            if (a.x_tile) *reinterpret_cast<float4*>(xs) = o[e];
            else if (prow) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (f + r < c.F) xrow[f + r] = ov[r];
            }
          }
          b3_put_pair(ob, (f0w + 32 * fb) / 16 + j, c.rb0 + pb, o[0], o[1]);
        }
      }
    }
    if (a.x_tile) {
      // x' leaves as row segments: instruction i stores rows 4 i .. 4 i + 3 (LDS accesses of one wave execute in order)
      const int rows = c.P - c.p0w;
      const int nch = (c.F - f0w) / 4;                 // valid chunks of the wave's features
#pragma unroll
      for (int i0 = 0; i0 < 16; i0 += 4) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(c.xreg + (i0 + k) * 256 + c.lane * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 4 * (i0 + k) + (c.lane >> 4);
          const int ch = (c.lane & 15) ^ (r & 15);
          if (r < rows && ch < nch) stg4(a.x + (size_t)(c.p0w + r) * a.ldx + f0w + 4 * ch, v[k]);
        }
      }
    }
  }
};

// ---- the kernel --------------------------------------------------------------------------------------------------------------------
struct B3Frags { uint4 a[2][3], b[2][3]; };     // [32-row block][plane]
#ifndef B3_LD
#define B3_LD 0       // loader waves of the product build (0: every wave moves its share of the DMA; 2: measured no faster, see the kernel)
#endif
#ifndef B3_SGB_VALU
#define B3_SGB_VALU 5
#endif
#ifndef B3_EXP
#define B3_EXP 0      // timing experiments only (tools/probes/split_probe.hip, garbage results): 1 no DMA after the prologue, 2 every workgroup
#endif                // stages tile 0 (operands L2-resident), 4 no fragment reads after the first, 8 no barriers in the K loop

// N consecutive 1 KiB LDS-DMA pieces behind ONE M0 setup: the instruction offset advances the global AND the LDS address, and a
// stage's global image and LDS image are the same linear bytes, so pieces i = 0 .. N - 1 are `offset:1024 i` of one base pair.
template <int N>
__device__ __forceinline__ void glds16s_run(gfloat_ptr sbase, unsigned voff, unsigned lds_dst) {
  static_assert(N == 3 || N == 4, "runs of 3 or 4 pieces (instruction offsets up to 3072)");
  unsigned keep;
  if constexpr (N == 3)
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// one dword per lane (256 bytes per wave-instruction): the per-feature parameters, whatever their alignment
__device__ __forceinline__ void glds4(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void b3_wait_vm() {
  static_assert(N == 0 || N == 6 || N == 12 || N == 24, "");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
}

// LD = 0: four waves, each computes 64 x 64 and moves 6 of a stage's 24 DMA pieces.
// LD = 2: six waves -- waves 0-3 compute and never touch vector memory inside the K loop, waves 4 / 5 do nothing but the DMA of
// the A / B stages (12 pieces each per step) and the counted wait for them.  Measured (tools/probes/split_probe.hip): no faster --
// the K loop does not lose its time to the DMA *issue* -- so the product build uses LD = 0.
// NKB > 0: the reduction has exactly NKB 16-k steps and the loop is fully unrolled (step indices are compile-time constants, which
// is what lets an epilogue keep per-step results in registers: EpiB3Post's in-loop generator).  NKB = 0: any length, rolled loop.
template <class Epi, int LD, int NKB = 0>
__global__ __launch_bounds__(NTHREADS + 64 * LD, (LD || B3_RING == 2) ? 3 : 2) void gemm_bf3_kernel(Bf3Args g, typename Epi::Args ea) {
  static_assert(LD == 0 || LD == 2, "no loader waves, or one per operand");
  static_assert(NKB == 0 || NKB == 16 || NKB == 32, "unrolled lengths: 256 / 512 deep reductions");
  extern __shared__ __attribute__((aligned(16))) uint4 b3smem[];
  const int nft = (g.F + B3_ROWS - 1) / B3_ROWS;
  const int npt = (g.P + B3_ROWS - 1) / B3_ROWS;
  // XCD-aware order (gemm.h): blocks b, b + 8, ... share an XCD and take the feature tiles of one patient tile
  const int b = blockIdx.x;
  const int idx = b >> 3;
  const int ft = idx % nft;
  const int pt = (idx / nft) * 8 + (b & 7);
  if (pt >= npt) return;
  const int pt_src = (B3_EXP & 2) ? 0 : pt;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = LD > 0 && wave >= 4;
  const int rbA = ((wave >> 1) & 1) * 2;    // the wave's first 32-row block of the A tile (features) ...
  const int rbB = (wave & 1) * 2;           // ... and of the B tile (patients)

  const int nkb = NKB ? NKB : g.nkb;
  const gfloat_ptr Ag = uniform_ptr(reinterpret_cast<const float*>(g.A + (size_t)ft * nkb * B3_STAGE_U4));
  const gfloat_ptr B0g = uniform_ptr(reinterpret_cast<const float*>(g.B0 + (size_t)pt_src * g.nkb0 * B3_STAGE_U4));
  const gfloat_ptr B1g = uniform_ptr(reinterpret_cast<const float*>(g.B1 ? g.B1 + (size_t)pt_src * g.nkb1 * B3_STAGE_U4 : g.B0));
  const int nkb0 = g.nkb0;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(b3smem)));
  float* const prm = reinterpret_cast<float*>(b3smem) + B3_LDS_RING_BYTES / 4;
  unsigned long long st_c = 0, st_r = 0;
  if (g.stamps) { st_c = __builtin_amdgcn_s_memtime(); st_r = __builtin_amdgcn_s_memrealtime(); }

  // per-feature parameters of the tile: wave k < 3 moves array k (two 256-byte pieces); oldest DMAs of the workgroup, so every
  // later counted wait covers them
  if (wave < 3) {
    const float* src = Epi::prm_ptr(ea, wave);
    if (src) {                               // uniform
      const int f0 = ft * B3_ROWS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int f = f0 + 64 * i + lane;
        f = f < g.F ? f : g.F - 1;
        glds4(src + f, __builtin_amdgcn_readfirstlane(lds_addr(prm) + (unsigned)wave * 512u + (unsigned)i * 256u));
      }
    }
  }

  // DMA of stage `st` into slot `sl`: 12 A pieces + 12 B pieces of 1 KiB
  auto issue_stage = [&](int st, int sl) {
    const gfloat_ptr as = Ag + (size_t)st * (B3_STAGE_BYTES / 4);
    const gfloat_ptr bs = st < nkb0 ? B0g + (size_t)st * (B3_STAGE_BYTES / 4) : B1g + (size_t)(st - nkb0) * (B3_STAGE_BYTES / 4);
    const unsigned dst = lds0 + (unsigned)sl * (unsigned)B3_SLOT_BYTES;
    if constexpr (LD == 0) {               // wave w: pieces 3 w .. 3 w + 2 of each operand
      glds16s_run<3>(as + wave * 768, voff, __builtin_amdgcn_readfirstlane(dst + (unsigned)wave * 3072u));
      glds16s_run<3>(bs + wave * 768, voff, __builtin_amdgcn_readfirstlane(dst + (unsigned)B3_STAGE_BYTES + (unsigned)wave * 3072u));
    } else {                               // loader 4: the A stage, loader 5: the B stage
      const gfloat_ptr src = wave == 4 ? as : bs;
      const unsigned d = __builtin_amdgcn_readfirstlane(dst + (wave == 4 ? 0u : (unsigned)B3_STAGE_BYTES));
#pragma unroll
      for (int i = 0; i < 3; ++i) glds16s_run<4>(src + i * 1024, voff, __builtin_amdgcn_readfirstlane(d + (unsigned)i * 4096u));
    }
  };
  constexpr int PPS = LD ? 12 : 6;          // pieces per stage and issuing wave
  auto read_frags = [&](B3Frags& f, int sl) {
    const uint4* s = b3smem + sl * (B3_SLOT_BYTES / 16) + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f.a[i][pl] = s[(pl * 4 + rbA + i) * 64];
        f.b[i][pl] = s[B3_STAGE_U4 + (pl * 4 + rbB + i) * 64];
      }
  };

  int s0 = 0, s1 = 1, s2 = 2;        // slots of stages s, s + 1, s + 2 (uniform)
  if (loader) {
    // ---- loader waves: the whole DMA schedule, nothing else ----
    issue_stage(0, 0);
    if (nkb > 1) issue_stage(1, 1);
    if (nkb > 2) issue_stage(2, 2);
    if (nkb > 2) b3_wait_vm<2 * PPS>(); else if (nkb > 1) b3_wait_vm<PPS>(); else b3_wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    if (nkb > 2) b3_wait_vm<PPS>(); else b3_wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    for (int s = 0; s < nkb; ++s) {
      const bool more = s + 3 < nkb && !(B3_EXP & 1);
      if (more) { issue_stage(s + 3, s0); b3_wait_vm<PPS>(); } else b3_wait_vm<0>();
      if (!(B3_EXP & 8)) asm volatile("s_barrier" ::: "memory");
      const int k = s0; s0 = s1; s1 = s2; s2 = k;
    }
    return;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  B3Ctx ctx;
  ctx.f0 = ft * B3_ROWS; ctx.fl = rbA * 32; ctx.pt = pt; ctx.rb0 = rbB; ctx.p0w = pt * B3_ROWS + rbB * 32;
  ctx.lane = lane; ctx.wave = wave; ctx.F = g.F; ctx.P = g.P; ctx.prm = prm;
  ctx.xreg = reinterpret_cast<float*>(b3smem) + wave * 4096;
  typename Epi::State est;
  bool rng = false;                  // uniform: this launch draws its normals in the K loop
  if constexpr (Epi::LOOP_RNG) { Epi::init(est, ea); rng = NKB > 0 && Epi::rng_on(ea, est); }
  bool xt = false;
  if constexpr (Epi::X_TILE) xt = ea.x_tile != 0 && B3_RING == 3;       // the x tile needs 64 KiB of ring

  // the 24 MFMAs of a stage; with `rs` >= 0 one generator block (Epi::kstep) is scheduled into their shadow: a bf16 MFMA holds
  // the vector issue for 8 of its 32 cycles, ~5 VALU instructions fit behind each
  auto mfma_block = [&](const B3Frags& f, bool draw, int rs) {
    // small cross terms first: (a3 b1), (a2 b2), (a1 b3), (a2 b1), (a1 b2), (a1 b1)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i][PA[t]]), __builtin_bit_cast(bf16x8, f.b[j][PB[t]]),
                                                              acc[i][j], 0, 0, 0);
    if constexpr (NKB > 0 && Epi::LOOP_RNG) {
      if (draw) {                       // uniform; rs is a compile-time constant of the unrolled loop
        Epi::kstep(est, ea, ctx, rs);
#ifndef B3_NO_SGB
#pragma unroll
        for (int k = 0; k < 24; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
          __builtin_amdgcn_sched_group_barrier(0x002, B3_SGB_VALU, 0);      // ... then a few VALU
        }
#endif
      }
    }
  };

  // ---- prologue: stages 0, 1, 2 in flight; fragments of stage 0 in registers ----
  B3Frags f0, f1;
  if constexpr (LD == 0) {
    issue_stage(0, 0);
    if (nkb > 1) issue_stage(1, 1);
    if (nkb > 2 && B3_RING == 3) issue_stage(2, 2);
    if (nkb > 2 && B3_RING == 3) b3_wait_vm<2 * PPS>(); else if (nkb > 1) b3_wait_vm<PPS>(); else b3_wait_vm<0>();
  }
  asm volatile("s_barrier" ::: "memory");
  read_frags(f0, 0);
  if constexpr (LD == 0) { if (nkb > 2 && B3_RING == 3) b3_wait_vm<PPS>(); else b3_wait_vm<0>(); }
  // lgkmcnt waits through the BUILTIN, not asm: hipcc's waitcnt pass does not see an asm wait, believes the fragment reads of the
  // previous step still pending where the next MFMAs use them, and puts its own lgkmcnt(0) in front of those MFMAs -- which also
  // waits for the reads issued a moment ago for the step after (seen in the ISA).  0xC07F = lgkmcnt(0), vmcnt / expcnt untouched.
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("s_barrier" ::: "memory");

  auto step = [&](const B3Frags& cur, B3Frags& nxt, int s) {
    // sched_barriers: without them hipcc sinks the MFMAs of stage s behind the next step's fragment reads (one register set
    // instead of two) and every step waits for its own LDS reads
    __builtin_amdgcn_sched_barrier(0);
    const bool last = s + 1 >= nkb;
    const bool more = LD == 0 && s + B3_RING < nkb && !(B3_EXP & 1);
    // the DMA statements FIRST: hipcc follows an asm statement with s_waitcnt lgkmcnt(0) when LDS reads are pending, which would
    // put the fragment reads of stage s + 1 in front of the MFMAs of stage s (seen in the ISA: every step waited for its LDS reads)
    if (more) issue_stage(s + B3_RING, s0);
    if constexpr (Epi::X_TILE) {
      // every stage has landed (the step before this one waited for vmcnt(0)) and every fragment read of the ring is complete:
      // the wave's x_t tile flies into its 16 KiB of the ring under the last MFMAs
      if (last && xt) Epi::xtile_issue(ea, ctx);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!last && !(B3_EXP & 4)) read_frags(nxt, s1);
    __builtin_amdgcn_sched_barrier(0);
    // one generator block per step (16 steps) or per other step (32 steps): sixteen per tile
    mfma_block((B3_EXP & 4) ? f0 : cur, rng && (NKB == 16 || !(s & 1)), NKB == 16 ? s : s / 2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LD == 0) {
      if (more && B3_RING == 3) b3_wait_vm<PPS>();
      else if (!(Epi::X_TILE && last && xt)) b3_wait_vm<0>();      // the x tile is waited for where the epilogue first needs it
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the fragments of stage s + 1 are in registers, every read of the ring is complete
    if (!(B3_EXP & 8)) asm volatile("s_barrier" ::: "memory");
    if (B3_RING == 3) { const int k = s0; s0 = s1; s1 = s2; s2 = k; }
    else { const int k = s0; s0 = s1; s1 = k; }
  };
  if constexpr (NKB > 0) {
#pragma unroll
    for (int s = 0; s < NKB; s += 2) {
      step(f0, f1, s);
      step(f1, f0, s + 1);
    }
  } else {
    for (int s = 0; s < nkb; s += 2) {
      step(f0, f1, s);
      if (s + 1 < nkb) step(f1, f0, s + 1);
    }
  }
  unsigned long long st_k = 0;
  if (g.stamps) {
    st_k = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* o = g.stamps + (size_t)blockIdx.x * 4;
      o[0] = st_k - st_c;                                   // prologue + K loop, shader cycles
      o[1] = __builtin_amdgcn_s_memrealtime() - st_r;       // the same in 100 MHz ticks
    }
  }
  if constexpr (Epi::LOOP_RNG) {
    // no unrolled loop for this length (or a single step): the draws the loop did not make
    if (!rng && Epi::rng_on(ea, est)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Epi::kstep(est, ea, ctx, i);
    }
  }
  Epi::apply(acc, ea, ctx, est);
  if (g.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // include the store drain
    if (tid == 0) g.stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - st_k;     // epilogue, shader cycles
  }
}

}  // namespace osd
