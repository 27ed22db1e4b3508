// bwd_persist.h -- the backward pass of the denoiser trunk (loss.backward(), utils/train.py:239) as ONE persistent launch.
//
// At the BASELINE training batch (4096 rows) a dgrad GEMM of a 256/512-wide layer is 256 output tiles of 8-16 K steps: launched
// on its own it is one workgroup per CU whose run time is launch ramp + first staging round trip + epilogue + store drain, with
// the matrix pipes busy a quarter of the time (profiles/r03_train_pmc.md), and the weight gradients -- leaves of the graph, a
// third of the step's FLOP -- ran as one grouped launch at the very end, when nothing else is left to run beside them.  Here
// both are work units of one kernel whose workgroups (two per CU) stay resident:
//
//   * dgrad queue: every output tile of every dgrad GEMM of the chain, in backward order, row-block major inside a layer.  A tile
//     needs the full K extent of its rows of the layer above: per (tensor, 64-row block) counters count finished feature tiles.
//   * wgrad queue: the grouped weight-gradient items of wgrad_group.h (tensor x 128 x 128 tile x row range), in the order the
//     chain finalises their gz operand; an item needs the row blocks of its range.
//   * policy (software priority instead of the SIMD arbiter, which starves whichever launch is younger): a free workgroup takes the
//     dgrad queue's head whenever that head is runnable; otherwise it runs a weight-gradient item.  The critical path -- the chain
//     of dgrads -- therefore always finds a free workgroup, and the weight gradients fill every slot the chain leaves idle.
//
// Deadlock freedom: a dgrad ticket may overshoot to a tile that is not runnable yet (only the head was inspected); its owner
// waits, but only for dgrad tiles that are earlier in the queue and hence already owned by a running workgroup (induction on
// queue order; every workgroup of the grid is resident: grid <= occupancy x CUs).  A weight-gradient ticket whose rows are not
// final yet is never waited for: it is parked in the owner (`pending`) while the owner keeps serving the dgrad queue, and is
// run when it becomes runnable -- at the latest when the dgrad queue is exhausted and its producers, all owned, finish.  Every
// poll loop has a wall-clock budget and observes a status word; on expiry all workgroups leave and the host reports OSD_EHIP.
//
// Control flow of the scheduler is wave-uniform (wave 0 decides for the workgroup; decisions travel through LDS behind a
// barrier): see chain.h for the hang a lane-divergent poll next to barriers produced.
#pragma once
#include "chain.h"          // ld_relaxed_agent / st_relaxed_agent
#include "epilogues.h"
#include "kernels_train.h"
#include "wgrad_group.h"

namespace osd {

enum : int { BU_DG_GN32 = 0, BU_DG_GN32D = 1, BU_DG_GN64 = 2, BU_DG_GN64D = 3, BU_DG_PLAIN = 4, BU_WG = 5 };
enum : unsigned { BWD_OK = 0, BWD_TIMEOUT = 1 };
constexpr int BWD_RB = 64;                       // rows per dependency block

typedef Tile<64, 64, 32, 32> BwdTile32;          // group width 32 (256-wide layers) and plain dgrads: waves of 32 features own whole groups
typedef Tile<64, 128, 64, 32> BwdTile64;         // group width 64 (512-wide layers)

struct BwdOp {                                   // one dgrad GEMM; its tiles are units of the dgrad queue
  GemmArgs g;                                    // A = W [n_out][k_in] (lda), B0 = gz [rows][n_out], K = K0 = n_out, F = k_in, P = rows
  GnBwdEpi e;                                    // GroupNorm+SiLU(+dropout) backward of the layer the GEMM feeds; BU_DG_PLAIN: e.gz / e.ldg = output
};
struct BwdUnit {
  int type, op;                                  // BU_*; op = index into ops (dgrad) or items (wgrad)
  int f0, p0;                                    // dgrad: tile origin
  int dep_ctr[2], dep_n[2], dep_cnt[2];          // runnable when ctr[dep_ctr[i] + j] >= dep_cnt[i] for all j < dep_n[i], i = 0, 1
  int sig_ctr, sig_n;                            // on completion: ctr[sig_ctr + j] += 1 for j < sig_n   (sig_n <= 2)
};
struct BwdPlan {
  const BwdUnit* dq; const BwdUnit* wq;
  const BwdOp* ops; const WgItem* items;
  int n_dq, n_wq;
  unsigned* ctr;                                 // dependency counters, zero at launch
  unsigned* ctl;                                 // [0] dgrad queue head, [1] wgrad queue head; zero at launch
  unsigned* status;                              // BWD_OK / BWD_TIMEOUT; sticky (the host resets it when it reports the failure)
  unsigned long long spin_budget;                // s_memrealtime ticks (100 MHz) a workgroup may look for work / wait for a dependency
  int flags;                                     // experiments: 1 = no acquire fence, 2 = no release fence
  unsigned long long* stamps;                    // diagnostic (null in production): per workgroup {scheduler, dgrad, wgrad, total} shader cycles, {dgrad, wgrad} unit counts
};
constexpr int BWD_LDS_BYTES = WG_LDS_BYTES;      // 64 KB: the weight-gradient stages; the dgrad tiles need less
static_assert(BwdTile32::LDS_BYTES <= BWD_LDS_BYTES && BwdTile64::LDS_BYTES <= BWD_LDS_BYTES, "dgrad tiles must fit");

// true when every dependency counter of the unit has reached its count.  Wave-uniform: every lane runs the same loop, the
// lanes spread over the counters of a range (index clamped, no exec masking), the verdict is a ballot.
__device__ __forceinline__ bool bwd_unit_ready(const BwdUnit* u, const unsigned* ctr, int lane) {
  bool mine = true;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = __builtin_amdgcn_readfirstlane(u->dep_n[i]);
    if (n > 0) {                                 // uniform
      const int base = __builtin_amdgcn_readfirstlane(u->dep_ctr[i]);
      const unsigned want = (unsigned)__builtin_amdgcn_readfirstlane(u->dep_cnt[i]);
      for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane < n ? j0 + lane : n - 1;
        mine = mine && (ld_relaxed_agent(ctr + base + j) >= want);
      }
    }
  }
  return __builtin_amdgcn_ballot_w64(!mine) == 0ull;
}

// A pointer that was LOADED from memory (a field of a work descriptor) is generic to hipcc, and every access through it is a FLAT
// instruction: both counters, out-of-order return -- each LDS wait of the tile loop then also waits for the global prefetch in
// flight, and the software pipeline collapses (measured: a dgrad tile took 3x its stand-alone time).  A round trip through the
// global address space tells InferAddressSpaces what the pointer is.
template <class P>
__device__ __forceinline__ P* as_global(P* p) {
  typedef __attribute__((address_space(1))) P GP;
  // through integers and readfirstlane (descriptor fields are wave-uniform): the pair of casts cannot fold back into the generic
  // pointer, and the base lands in SGPRs
  const unsigned long long v = reinterpret_cast<unsigned long long>((GP*)p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (P*)reinterpret_cast<GP*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ GemmArgs global_args(const GemmArgs& g) {
  GemmArgs r = g;
  r.A = as_global(g.A); r.B0 = as_global(g.B0); r.B1 = as_global(g.B1);
  return r;
}

template <class T, int GW, bool DROP>
__device__ __forceinline__ void bwd_run_gn(const BwdOp& op, int f0, int p0, uint64_t seed, uint32_t row_offset, float* smem) {
  typedef EpiGnBwd<GW, DROP> E;
  const GnBwdEpi& a = op.e;
  const typename E::Args ea{as_global(a.z), a.ldz, as_global(a.stats), as_global(a.gamma), as_global(a.beta), as_global(a.gz), a.ldg,
                            as_global(a.gy), a.ldy, a.accumulate, a.drop_mode, as_global(a.mask), a.ldm,
                            a.keep_scale, a.p_drop, seed, row_offset, a.step, a.tag};
  const GemmArgs g = global_args(op.g);
  gemm_tile<T, false, true, E, true>(g, ea, f0, p0, smem);
}

__global__ __launch_bounds__(NTHREADS, 2) void bwd_persist_kernel(BwdPlan pl, uint64_t seed, uint32_t row_offset) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_ctl[2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  enum { ACT_EXIT = 0, ACT_DG = 1, ACT_WG = 2 };
  int pending = -1;                              // wave 0 only: a claimed weight-gradient item whose rows are not final yet
  unsigned long long c_sched = 0, c_dg = 0, c_wg = 0, n_dg = 0, n_wg = 0, c_pub = 0;
  unsigned long long c_type[5] = {0, 0, 0, 0, 0}, n_type[5] = {0, 0, 0, 0, 0};
  int utype = 0;
  const unsigned long long c_start = pl.stamps ? __builtin_amdgcn_s_memtime() : 0;
  for (;;) {
    const unsigned long long ts0 = pl.stamps ? __builtin_amdgcn_s_memtime() : 0;
    // ---- scheduler (wave 0, wave-uniform) ----
    if (wave == 0) {
      int act = ACT_EXIT, idx = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      // experiments (flags 4 / 8): workgroups of a CU share a role -- odd CUs (4) or every fourth CU (8) only run weight-gradient
      // items while the dgrad queue has tiles left, the other CUs only dgrad tiles while weight-gradient items are left
      const unsigned hwid = __builtin_amdgcn_s_getreg(0xF804);
      const unsigned cu_id = (hwid >> 8) & 15u;
      const bool split = (pl.flags & 12) != 0;
      const bool role_w = split && ((pl.flags & 4) ? (cu_id & 1u) != 0 : (cu_id & 3u) == 3u);
      int naps = 0;
      for (;;) {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.status)) != BWD_OK) break;
        const int hd = __builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.ctl));
        const bool wq_left = split && __builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.ctl + 1)) < pl.n_wq;
        const bool may_dg = !role_w || !wq_left;
        const bool may_wg = !split || role_w || hd >= pl.n_dq;
        if (may_dg && hd < pl.n_dq && bwd_unit_ready(pl.dq + hd, pl.ctr, lane)) {
          unsigned tk = 0;
          if (lane == 0) tk = atomicAdd(pl.ctl, 1u);
          tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
          if ((int)tk < pl.n_dq) { act = ACT_DG; idx = (int)tk; break; }
          continue;                              // the queue ran out between the look and the claim
        }
        if (pending >= 0) {
          if (bwd_unit_ready(pl.wq + pending, pl.ctr, lane)) { act = ACT_WG; idx = pending; pending = -1; break; }
        } else if (may_wg && __builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.ctl + 1)) < pl.n_wq) {
          unsigned tk = 0;
          if (lane == 0) tk = atomicAdd(pl.ctl + 1, 1u);
          tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
          if ((int)tk < pl.n_wq) { pending = (int)tk; continue; }      // parked; look at the dgrad queue again first
        }
        if (hd >= pl.n_dq && pending < 0 && __builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.ctl + 1)) >= pl.n_wq) break;      // both queues drained
        // back off: idle workgroups polling two words in lock step slow the memory channel that holds them for everybody
        if (naps < 4) __builtin_amdgcn_s_sleep(20); else if (naps < 16) __builtin_amdgcn_s_sleep(60); else __builtin_amdgcn_s_sleep(127);
        ++naps;
        if (__builtin_amdgcn_s_memrealtime() - t0 > pl.spin_budget) {
          if (lane == 0) st_relaxed_agent(pl.status, BWD_TIMEOUT);
          break;
        }
      }
      if (act == ACT_DG) {
        // the ticket may be ahead of the head that was inspected: wait for this tile's own dependencies (earlier dgrad tiles,
        // all owned by running workgroups)
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        while (!bwd_unit_ready(pl.dq + idx, pl.ctr, lane)) {
          __builtin_amdgcn_s_sleep(8);
          bool bad = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed_agent(pl.status)) != BWD_OK;
          if (!bad && __builtin_amdgcn_s_memrealtime() - t1 > pl.spin_budget) {
            if (lane == 0) st_relaxed_agent(pl.status, BWD_TIMEOUT);
            bad = true;
          }
          if (bad) { act = ACT_EXIT; break; }
        }
      }
      if (act != ACT_EXIT && !(pl.flags & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the producers' rows, written on other XCDs
      if (lane == 0) { s_ctl[0] = act; s_ctl[1] = idx; }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  // invalidate and LDS write done before the barrier releases
    }
    __syncthreads();
    const int act = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    const int idx = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    __syncthreads();                             // s_ctl is rewritten only after every wave has read it
    if (act == ACT_EXIT) {                       // uniform over the workgroup
      if (pl.stamps && tid == 0) {
        unsigned long long* o = pl.stamps + (size_t)blockIdx.x * 24;
        o[0] = c_sched; o[1] = c_dg; o[2] = c_wg; o[3] = __builtin_amdgcn_s_memtime() - c_start; o[4] = n_dg; o[5] = n_wg;
        o[6] = __builtin_amdgcn_s_getreg(0xF814) & 7; o[7] = c_pub;
        for (int i = 0; i < 5; ++i) { o[8 + i] = c_type[i]; o[13 + i] = n_type[i]; }
      }
      return;
    }
    const unsigned long long ts1 = pl.stamps ? __builtin_amdgcn_s_memtime() : 0;

    // ---- the unit ----
    int sig_ctr = 0, sig_n = 0;
    if (act == ACT_WG) {
      const BwdUnit* u = pl.wq + idx;
      sig_ctr = __builtin_amdgcn_readfirstlane(u->sig_ctr); sig_n = __builtin_amdgcn_readfirstlane(u->sig_n);
      const WgItem it = pl.items[__builtin_amdgcn_readfirstlane(u->op)];       // by value: the DMA asm statements clobber "memory"
      wgrad_item(it, smem);
    } else {
      const BwdUnit* u = pl.dq + idx;
      sig_ctr = __builtin_amdgcn_readfirstlane(u->sig_ctr); sig_n = __builtin_amdgcn_readfirstlane(u->sig_n);
      const int type = __builtin_amdgcn_readfirstlane(u->type);
      utype = type;
      const int f0 = __builtin_amdgcn_readfirstlane(u->f0), p0 = __builtin_amdgcn_readfirstlane(u->p0);
      const BwdOp op = pl.ops[__builtin_amdgcn_readfirstlane(u->op)];
      switch (type) {                            // uniform
        case BU_DG_GN32: bwd_run_gn<BwdTile32, 32, false>(op, f0, p0, seed, row_offset, smem); break;
        case BU_DG_GN32D: bwd_run_gn<BwdTile32, 32, true>(op, f0, p0, seed, row_offset, smem); break;
        case BU_DG_GN64: bwd_run_gn<BwdTile64, 64, false>(op, f0, p0, seed, row_offset, smem); break;
        case BU_DG_GN64D: bwd_run_gn<BwdTile64, 64, true>(op, f0, p0, seed, row_offset, smem); break;
        default: {
          const EpiBias<false, false>::Args ea{nullptr, as_global(op.e.gz), op.e.ldg, 0};
          const GemmArgs g = global_args(op.g);
          gemm_tile<BwdTile32, false, true, EpiBias<false, false>, true>(g, ea, f0, p0, smem);
        }
      }
    }
    // ---- publish: every wave's stores drained, then ONE agent-scope release and the counters ----
    const unsigned long long tp0 = pl.stamps ? __builtin_amdgcn_s_memtime() : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                             // also: the staging buffers are free for the next unit
    if (wave == 0 && sig_n > 0) {
      if (!(pl.flags & 2)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane < sig_n) atomicAdd(pl.ctr + sig_ctr + lane, 1u);
    }
    if (pl.stamps) {
      const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
      c_sched += ts1 - ts0;
      c_pub += ts2 - tp0;
      if (act == ACT_WG) { c_wg += ts2 - ts1; ++n_wg; } else { c_dg += ts2 - ts1; ++n_dg; c_type[utype] += ts2 - ts1; ++n_type[utype]; }
    }
  }
}

}  // namespace osd
