// bwd_persist.hip -- host side of the persistent backward launch: work queues from the pass description (bwd_host.h), upload,
// launch, status.
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include "bwd_persist.h"
#include "bwd_host.h"
#include "handle.h"
#include "launch.h"

namespace osd {

static bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool same(const GemmArgs& a, const GemmArgs& b) {
  return a.A == b.A && a.lda == b.lda && a.B0 == b.B0 && a.ldb0 == b.ldb0 && a.B1 == b.B1 && a.ldb1 == b.ldb1 && a.K0 == b.K0 && a.F == b.F &&
         a.P == b.P && a.K == b.K && a.kchunk == b.kchunk;
}
// seed / row_offset are launch arguments of the kernel, not part of the uploaded description
static bool same(const GnBwdEpi& a, const GnBwdEpi& b) {
  return a.z == b.z && a.ldz == b.ldz && a.stats == b.stats && a.gamma == b.gamma && a.beta == b.beta && a.gz == b.gz && a.ldg == b.ldg &&
         a.gy == b.gy && a.ldy == b.ldy && a.accumulate == b.accumulate && a.drop_mode == b.drop_mode && a.mask == b.mask && a.ldm == b.ldm &&
         a.keep_scale == b.keep_scale && a.p_drop == b.p_drop && a.step == b.step && a.tag == b.tag;
}
static bool same(const BwdDgradIn& a, const BwdDgradIn& b) {
  return a.gw == b.gw && a.drop == b.drop && same(a.g, b.g) && same(a.e, b.e) && a.dep0 == b.dep0 && a.dep1 == b.dep1 && a.sig == b.sig;
}
static bool same(const BwdWgradIn& a, const BwdWgradIn& b) {
  const WgPending &x = a.w, &y = b.w;
  return a.dep == b.dep && x.x == y.x && x.ldx == y.ldx && x.kin == y.kin && x.gz == y.gz && x.ldg == y.ldg && x.nout == y.nout && x.rows == y.rows &&
         x.dw == y.dw && x.lddw == y.lddw && x.bias[0] == y.bias[0] && x.bias[1] == y.bias[1] && x.bias[2] == y.bias[2];
}

bool BwdBuilder::add_dgrad(const BwdDgradIn& d) {
  const GemmArgs& g = d.g;
  if (d.gw != 0 && d.gw != 32 && d.gw != 64) return false;
  if (g.P != (int)rows_ || g.K0 < g.K || g.kchunk != 0) return false;
  if (!gemm_fast_ok(g, false, true)) return false;
  const GnBwdEpi& a = d.e;
  if (d.gw == 0) {
    if (!al16p(a.gz) || a.ldg % 4 || g.F % 4) return false;
  } else {
    const EpiGnBwd<32, false>::Args ea{a.z, a.ldz, a.stats, a.gamma, a.beta, a.gz, a.ldg, a.gy, a.ldy, a.accumulate, a.drop_mode, a.mask, a.ldm,
                                       a.keep_scale, a.p_drop, 0, 0, a.step, a.tag};
    if (!EpiGnBwd<32, false>::fast_ok(ea, g.F) || g.F % d.gw) return false;
  }
  dg_.push_back(d);
  return true;
}

bool BwdBuilder::add_wgrad(const WgPending& w, int dep) {
  if (w.rows != rows_ || !wgrad_group_ok(w)) return false;
  wg_.push_back({w, dep});
  return true;
}

struct BwdPlanDev {
  std::vector<BwdDgradIn> dg;
  std::vector<BwdWgradIn> wg;
  int64_t rows = -1;
  float* slabs = nullptr; int64_t slab_floats = 0;
  // device image: [ops | dq | wq | items | reds | ctr + ctl]
  char* dev = nullptr; size_t cap = 0;
  BwdPlan plan{};
  WgReduce* d_reds = nullptr; int n_reds = 0;
  size_t ctr_bytes = 0;
  int grid = 0;
  unsigned* host_status = nullptr;       // pinned: the status word of the previous launch lands here
  hipEvent_t status_ev = nullptr;
  bool status_pending = false;
  unsigned long long* stamps = nullptr;
};

static BwdPlanDev* plan_of(osd_handle* h) {
  if (!h->bwd_plan) h->bwd_plan = new BwdPlanDev();
  return static_cast<BwdPlanDev*>(h->bwd_plan);
}

void bwd_persist_free(osd_handle* h) {
  if (!h->bwd_plan) return;
  BwdPlanDev* pl = static_cast<BwdPlanDev*>(h->bwd_plan);
  hipError_t e = hipSuccess;
  if (pl->stamps) {
    if (const char* path = getenv("OSD_BWD_STAMPS")) {
      std::vector<unsigned long long> st((size_t)pl->grid * 24);
      if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(st.data(), pl->stamps, st.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(path, "w")) {
          for (int i = 0; i < pl->grid; ++i) {
            for (int j = 0; j < 24; ++j) fprintf(f, "%llu ", st[(size_t)i * 24 + j]);
            fprintf(f, "\n");
          }
          fclose(f);
        }
      }
    }
    e = hipFree(pl->stamps);
  }
  if (pl->dev) e = hipFree(pl->dev);
  if (pl->host_status) e = hipHostFree(pl->host_status);
  if (pl->status_ev) e = hipEventDestroy(pl->status_ev);
  (void)e;
  delete pl;
  h->bwd_plan = nullptr;
}

static int status_error(unsigned st) {
  set_error("the persistent backward kernel gave up in a dependency wait (status %u): the gradients of that step are invalid; "
            "osd_set_option(\"persistent_bwd\", 0) selects the per-launch backward", st);
  return OSD_EHIP;
}

int bwd_persist_check(osd_handle* h, hipStream_t s) {
  if (!h->bwd_plan) return OSD_OK;
  BwdPlanDev* pl = static_cast<BwdPlanDev*>(h->bwd_plan);
  if (!pl->status_pending) return OSD_OK;
  OSD_HIP(hipStreamSynchronize(s));
  pl->status_pending = false;
  if (*pl->host_status != BWD_OK) {
    const unsigned st = *pl->host_status;
    *pl->host_status = BWD_OK;
    if (pl->plan.status) OSD_HIP(hipMemsetAsync(pl->plan.status, 0, 4, s));
    return status_error(st);
  }
  return OSD_OK;
}

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

int BwdBuilder::launch(osd_handle* h, hipStream_t s, float* slabs, int64_t slab_floats, uint64_t seed, uint32_t row_offset) {
  BwdPlanDev* pl = plan_of(h);
  const int dev_id = h->cfg.device;
  if (!pl->host_status) {
    OSD_HIP(hipHostMalloc((void**)&pl->host_status, 64, hipHostMallocDefault));
    *pl->host_status = BWD_OK;
    OSD_HIP(hipEventCreateWithFlags(&pl->status_ev, hipEventDisableTiming));
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS_BYTES));
    int occ = 0;
    OSD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bwd_persist_kernel, NTHREADS, BWD_LDS_BYTES));
    hipDeviceProp_t prop;
    OSD_HIP(hipGetDeviceProperties(&prop, dev_id));
    pl->grid = std::min(occ, 2) * prop.multiProcessorCount;       // every workgroup resident: the scheduler's waits rely on it
    if (pl->grid < 1) { set_error("the persistent backward kernel does not fit this device"); return OSD_EUNSUPPORTED; }
    if (const char* e = getenv("OSD_BWD_GRID")) { const int v = atoi(e); if (v > 0 && v < pl->grid) pl->grid = v; }
  }
  // the previous launch's status word (copied behind that launch): known once its event has fired
  if (pl->status_pending && hipEventQuery(pl->status_ev) == hipSuccess) {
    pl->status_pending = false;
    if (*pl->host_status != BWD_OK) {
      const unsigned st = *pl->host_status;
      *pl->host_status = BWD_OK;
      if (pl->plan.status) OSD_HIP(hipMemsetAsync(pl->plan.status, 0, 4, s));      // sticky on the device until reported
      return status_error(st);
    }
  }
  (void)hipGetLastError();

  bool unchanged = pl->dev && pl->rows == rows_ && pl->slabs == slabs && pl->slab_floats == slab_floats && pl->dg.size() == dg_.size() &&
                   pl->wg.size() == wg_.size();
  for (size_t i = 0; unchanged && i < dg_.size(); ++i) unchanged = same(pl->dg[i], dg_[i]);
  for (size_t i = 0; unchanged && i < wg_.size(); ++i) unchanged = same(pl->wg[i], wg_[i]);
  if (!unchanged) {
    const int nrb = (int)((rows_ + BWD_RB - 1) / BWD_RB);
    // ---- dgrad queue ----
    std::vector<BwdOp> ops;
    std::vector<BwdUnit> dq, wq;
    std::vector<int> tensor_cnt(std::max(n_tensors_, 1), 0);
    for (const BwdDgradIn& d : dg_) {
      const int BF = 64;
      const int nft = (d.g.F + BF - 1) / BF;
      if (d.sig >= 0) tensor_cnt[d.sig] = nft;
    }
    for (size_t oi = 0; oi < dg_.size(); ++oi) {
      const BwdDgradIn& d = dg_[oi];
      BwdOp op{};
      op.g = d.g; op.e = d.e;
      ops.push_back(op);
      const int BF = 64, BP = d.gw == 64 ? 128 : 64;
      const int type = d.gw == 0 ? BU_DG_PLAIN : d.gw == 32 ? (d.drop ? BU_DG_GN32D : BU_DG_GN32) : (d.drop ? BU_DG_GN64D : BU_DG_GN64);
      for (int p0 = 0; p0 < d.g.P; p0 += BP)
        for (int f0 = 0; f0 < d.g.F; f0 += BF) {
          BwdUnit u{};
          u.type = type; u.op = (int)oi; u.f0 = f0; u.p0 = p0;
          const int rb0 = p0 / BWD_RB;
          const int nb = (std::min<int>(BP, d.g.P - p0) + BWD_RB - 1) / BWD_RB;
          const int deps[2] = {d.dep0, d.dep1};
          for (int i = 0; i < 2; ++i)
            if (deps[i] >= 0) { u.dep_ctr[i] = deps[i] * nrb + rb0; u.dep_n[i] = nb; u.dep_cnt[i] = tensor_cnt[deps[i]]; }
          if (d.sig >= 0) { u.sig_ctr = d.sig * nrb + rb0; u.sig_n = nb; }
          dq.push_back(u);
        }
    }
    // ---- wgrad queue: the items of wgrad_group.h, tensor by tensor in the order the chain finalises their gz ----
    std::vector<WgItem> items;
    std::vector<WgReduce> reds;
    long total = 0;
    for (const BwdWgradIn& wi : wg_) total += (long)((wi.w.kin + 127) / 128) * ((wi.w.nout + 127) / 128) * (wi.w.rows / WG_BK);
    static const int target_items = [] { const char* e = getenv("OSD_BWD_WG_ITEMS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
    const long per_item = std::max<long>(8, (total + target_items - 1) / target_items);
    int64_t slab_off = 0;
    for (const BwdWgradIn& wi : wg_) {
      const WgPending& w = wi.w;
      const int ksteps = (int)(w.rows / WG_BK);
      int n_slices = (int)std::max<long>(1, (ksteps + per_item / 2) / per_item);
      const int64_t numel = (int64_t)w.nout * w.kin;
      while (n_slices > 1 && slab_off + (int64_t)n_slices * numel > slab_floats) --n_slices;
      int per = (ksteps + n_slices - 1) / n_slices;
      per = (per + 1) / 2 * 2;                       // whole 64-row dependency blocks (two K steps of 32 rows)
      n_slices = (ksteps + per - 1) / per;
      float* slab = slabs + slab_off;
      if (n_slices > 1) {
        reds.push_back({w.dw, w.lddw, slab, numel, w.nout, w.kin, n_slices});
        slab_off += (int64_t)n_slices * ((numel + 3) / 4 * 4);
      }
      // slice-major: the items of the rows that become final first come first
      for (int sl = 0; sl < n_slices; ++sl)
        for (int p0 = 0; p0 < w.nout; p0 += 128)
          for (int f0 = 0; f0 < w.kin; f0 += 128) {
            WgItem it{};
            it.A = w.x; it.lda = w.ldx; it.B = w.gz; it.ldb = w.ldg; it.F = w.kin; it.P = w.nout; it.f0 = f0; it.p0 = p0;
            it.k0 = sl * per * WG_BK; it.k1 = std::min<int>((sl + 1) * per, ksteps) * WG_BK;
            if (n_slices > 1) { it.out = slab + (int64_t)sl * numel; it.ldo = w.kin; }
            else { it.out = w.dw; it.ldo = w.lddw; }
            if (f0 == 0) { it.bias[0] = w.bias[0]; it.bias[1] = w.bias[1]; it.bias[2] = w.bias[2]; }
            BwdUnit u{};
            u.type = BU_WG; u.op = (int)items.size();
            if (wi.dep >= 0) {
              const int rb0 = it.k0 / BWD_RB, rb1 = (it.k1 + BWD_RB - 1) / BWD_RB;
              u.dep_ctr[0] = wi.dep * nrb + rb0; u.dep_n[0] = rb1 - rb0; u.dep_cnt[0] = tensor_cnt[wi.dep];
            }
            items.push_back(it);
            wq.push_back(u);
          }
    }
    // ---- device image ----
    const size_t o_ops = 0, o_dq = o_ops + up256(ops.size() * sizeof(BwdOp)), o_wq = o_dq + up256(dq.size() * sizeof(BwdUnit));
    const size_t o_items = o_wq + up256(wq.size() * sizeof(BwdUnit)), o_reds = o_items + up256(items.size() * sizeof(WgItem));
    const size_t o_ctr = o_reds + up256(reds.size() * sizeof(WgReduce));
    const size_t ctr_bytes = up256(((size_t)std::max(n_tensors_, 1) * nrb + 4) * 4);
    const size_t o_status = o_ctr + ctr_bytes;
    const size_t need = o_status + 256;
    OSD_HIP(hipStreamSynchronize(s));              // rare (first step, or batch / tensors changed): the old image may still be in use
    if (pl->cap < need) {
      if (pl->dev) OSD_HIP(hipFree(pl->dev));
      pl->dev = nullptr; pl->cap = 0;
      if (hipMalloc((void**)&pl->dev, need) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %zu bytes failed", need); return OSD_ENOMEM; }
      pl->cap = need;
    }
    auto put = [&](size_t off, const void* src, size_t bytes) -> hipError_t {
      return bytes ? hipMemcpy(pl->dev + off, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    OSD_HIP(put(o_ops, ops.data(), ops.size() * sizeof(BwdOp)));
    OSD_HIP(put(o_dq, dq.data(), dq.size() * sizeof(BwdUnit)));
    OSD_HIP(put(o_wq, wq.data(), wq.size() * sizeof(BwdUnit)));
    OSD_HIP(put(o_items, items.data(), items.size() * sizeof(WgItem)));
    OSD_HIP(put(o_reds, reds.data(), reds.size() * sizeof(WgReduce)));
    BwdPlan p{};
    p.ops = reinterpret_cast<const BwdOp*>(pl->dev + o_ops);
    p.dq = reinterpret_cast<const BwdUnit*>(pl->dev + o_dq);
    p.wq = reinterpret_cast<const BwdUnit*>(pl->dev + o_wq);
    p.items = reinterpret_cast<const WgItem*>(pl->dev + o_items);
    p.n_dq = (int)dq.size(); p.n_wq = (int)wq.size();
    p.ctl = reinterpret_cast<unsigned*>(pl->dev + o_ctr);
    p.ctr = p.ctl + 4;
    p.status = reinterpret_cast<unsigned*>(pl->dev + o_status);
    OSD_HIP(hipMemset(p.status, 0, 256));
    pl->plan = p;
    pl->d_reds = reinterpret_cast<WgReduce*>(pl->dev + o_reds);
    pl->n_reds = (int)reds.size();
    pl->ctr_bytes = ctr_bytes;
    pl->dg = dg_; pl->wg = wg_; pl->rows = rows_; pl->slabs = slabs; pl->slab_floats = slab_floats;
  }
  BwdPlan p = pl->plan;
  p.spin_budget = h->bwd_spin_budget;
  static const int xflags = [] { const char* e = getenv("OSD_BWD_FLAGS"); return e ? atoi(e) : 0; }();
  p.flags = xflags;
  // diagnostic: OSD_BWD_STAMPS=<file> dumps per-workgroup cycle counters of the last launch at handle destruction
  static const char* stamp_path = getenv("OSD_BWD_STAMPS");
  if (stamp_path && !pl->stamps) OSD_HIP(hipMalloc((void**)&pl->stamps, (size_t)pl->grid * 24 * 8));
  p.stamps = pl->stamps;
  OSD_HIP(hipMemsetAsync(p.ctl, 0, pl->ctr_bytes, s));
  const int units = p.n_dq + p.n_wq;
  const int grid = std::max(1, std::min(pl->grid, units));
  hipLaunchKernelGGL(bwd_persist_kernel, dim3((unsigned)grid), dim3(NTHREADS), BWD_LDS_BYTES, s, p, seed, row_offset);
  OSD_HIP(hipGetLastError());
  if (pl->n_reds > 0) {
    hipLaunchKernelGGL(wgrad_group_reduce, dim3(64, (unsigned)pl->n_reds), dim3(256), 0, s, pl->d_reds);
    OSD_HIP(hipGetLastError());
  }
  OSD_HIP(hipMemcpyAsync(pl->host_status, p.status, 4, hipMemcpyDeviceToHost, s));
  OSD_HIP(hipEventRecord(pl->status_ev, s));
  pl->status_pending = true;
  return OSD_OK;
}

}  // namespace osd
