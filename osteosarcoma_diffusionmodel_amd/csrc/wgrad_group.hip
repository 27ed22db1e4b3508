// wgrad_group.hip -- host side of the grouped weight-gradient launch (wgrad_group.h): eligibility, the work-item list
// (tensor x 128 x 128 tile x row range), slabs for split row ranges, upload (only when the list changed), launch.
#include <string.h>
#include <algorithm>
#include "wgrad_group.h"
#include "handle.h"
#include "kernels_train.h"

namespace osd {

// A workgroup runs items blockIdx.x, blockIdx.x + gridDim.x, ...: one item each when the grid covers the list, or a walk
// over it when the host caps the grid (to leave CU slots to a concurrent stream).
// MODE 1: 160 VGPRs and 48 KB of LDS -- up to three workgroups per CU
template <int MODE>
__global__ __launch_bounds__(NTHREADS, MODE == 1 ? 3 : 2) void wgrad_group_kernel(const WgItem* __restrict__ items, int n_items) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const WgItem it = items[item];            // by value: the DMA asm statements clobber "memory"
    if constexpr (MODE == 1) wgrad_item_bf3(it, reinterpret_cast<uint4*>(smem));
    else wgrad_item(it, smem);
  }
}

// out[p][f] = sum over slices (fixed order) of the dense slabs; one grid row per tensor
__global__ void wgrad_group_reduce(const WgReduce* __restrict__ items) {
  const WgReduce& r = items[blockIdx.y];
  const int c4n = r.F >> 2;
  const long long total = (long long)r.P * c4n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i / c4n);
    const int c = 4 * (int)(i - (long long)p * c4n);
    const float* s = r.slab + (size_t)p * r.F + c;
    float4 acc = *reinterpret_cast<const float4*>(s);
    for (int k = 1; k < r.n_slices; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(s + (size_t)k * r.stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(r.out + (size_t)p * r.ldo + c) = acc;
  }
}

static bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool wgrad_group_ok(const WgPending& w) {
  return w.rows >= WG_BK && w.rows % WG_BK == 0 && w.kin >= 4 && w.kin % 4 == 0 && w.nout >= 4 && w.nout % 4 == 0 && w.ldx % 4 == 0 &&
         w.ldg % 4 == 0 && w.lddw % 4 == 0 && al16p(w.x) && al16p(w.gz) && al16p(w.dw);
}

struct WgPlanDev {
  std::vector<WgItem> items;
  std::vector<WgReduce> reds;
  WgItem* d_items = nullptr; size_t cap_items = 0;
  WgReduce* d_reds = nullptr; size_t cap_reds = 0;
  std::vector<GnColItem> cols;
  GnColItem* d_cols = nullptr; size_t cap_cols = 0;
};

static bool g_wg_attr[16] = {};

void wgrad_group_free(osd_handle* h) {
  for (void* p : h->wg_plans) {
    WgPlanDev* pl = static_cast<WgPlanDev*>(p);
    hipError_t e = hipSuccess;
    if (pl->d_items) e = hipFree(pl->d_items);
    if (pl->d_reds) e = hipFree(pl->d_reds);
    if (pl->d_cols) e = hipFree(pl->d_cols);
    (void)e;
    delete pl;
  }
  h->wg_plans.clear();
}

// field-wise (the structs have padding bytes, so memcmp would report a change on every step and force the re-upload)
static bool same(const WgItem& a, const WgItem& b) {
  return a.A == b.A && a.lda == b.lda && a.B == b.B && a.ldb == b.ldb && a.F == b.F && a.P == b.P && a.f0 == b.f0 && a.p0 == b.p0 &&
         a.k0 == b.k0 && a.k1 == b.k1 && a.out == b.out && a.ldo == b.ldo && a.bias[0] == b.bias[0] && a.bias[1] == b.bias[1] && a.bias[2] == b.bias[2];
}
static bool same(const WgReduce& a, const WgReduce& b) {
  return a.out == b.out && a.ldo == b.ldo && a.slab == b.slab && a.stride == b.stride && a.P == b.P && a.F == b.F && a.n_slices == b.n_slices;
}

static bool same(const GnColItem& a, const GnColItem& b) {
  return a.gy == b.gy && a.ldy == b.ldy && a.z == b.z && a.ldz == b.ldz && a.stats == b.stats && a.C == b.C && a.gw == b.gw && a.rows == b.rows &&
         a.dgamma == b.dgamma && a.dbeta == b.dbeta;
}

template <class V, class D>
static int upload(hipStream_t s, const V& fresh, V& kept, D** dev, size_t* cap) {
  typedef typename V::value_type E;
  bool unchanged = fresh.size() == kept.size() && *dev;
  for (size_t i = 0; unchanged && i < fresh.size(); ++i) unchanged = same(fresh[i], kept[i]);
  if (unchanged) return OSD_OK;
  OSD_HIP(hipStreamSynchronize(s));            // rare (first step, or the batch / tensors changed): the old list may still be in use
  if (*cap < fresh.size()) {
    if (*dev) OSD_HIP(hipFree(*dev));
    *dev = nullptr; *cap = 0;
    const size_t want = std::max<size_t>(fresh.size(), 64);
    if (hipMalloc((void**)dev, want * sizeof(E)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    *cap = want;
  }
  kept = fresh;
  if (!kept.empty()) OSD_HIP(hipMemcpyAsync(*dev, kept.data(), kept.size() * sizeof(E), hipMemcpyHostToDevice, s));
  return OSD_OK;
}

// Launch every pending weight gradient as one grouped GEMM (+ one slab reduction) on stream s.
// max_grid > 0: at most that many workgroups walk the list (one slot per CU stays free for the kernels of another stream)
int wgrad_group_flush(osd_handle* h, hipStream_t s, int plan_index, const std::vector<WgPending>& pend, float* slabs, int64_t slab_floats,
                      int max_grid) {
  if (pend.empty()) return OSD_OK;
  while ((int)h->wg_plans.size() <= plan_index) h->wg_plans.push_back(new WgPlanDev());
  WgPlanDev* pl = static_cast<WgPlanDev*>(h->wg_plans[plan_index]);
  const int dev = h->cfg.device;
  if (dev >= 0 && dev < 16 && !g_wg_attr[dev]) {
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS_BYTES));
    g_wg_attr[dev] = true;
  }
  // row range per item: about two workgroups per CU over the whole list, never fewer than 8 K steps per item
  long total = 0;
  for (const WgPending& w : pend) total += (long)((w.kin + 127) / 128) * ((w.nout + 127) / 128) * (w.rows / WG_BK);
  static const int target_items = [] { const char* e = getenv("OSD_WGRAD_ITEMS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  const long per_item = std::max<long>(8, (total + target_items - 1) / target_items);
  std::vector<WgItem> items;
  std::vector<WgReduce> reds;
  int64_t slab_off = 0;
  for (const WgPending& w : pend) {
    const int ksteps = (int)(w.rows / WG_BK);
    int n_slices = (int)std::max<long>(1, (ksteps + per_item / 2) / per_item);
    const int64_t numel = (int64_t)w.nout * w.kin;
    while (n_slices > 1 && slab_off + (int64_t)n_slices * numel > slab_floats) --n_slices;
    const int per = (ksteps + n_slices - 1) / n_slices;
    n_slices = (ksteps + per - 1) / per;
    float* slab = slabs + slab_off;
    if (n_slices > 1) {
      reds.push_back({w.dw, w.lddw, slab, numel, w.nout, w.kin, n_slices});
      slab_off += (int64_t)n_slices * ((numel + 3) / 4 * 4);
    }
    for (int p0 = 0; p0 < w.nout; p0 += 128)
      for (int f0 = 0; f0 < w.kin; f0 += 128)
        for (int sl = 0; sl < n_slices; ++sl) {
          WgItem it{};
          it.A = w.x; it.lda = w.ldx; it.B = w.gz; it.ldb = w.ldg; it.F = w.kin; it.P = w.nout; it.f0 = f0; it.p0 = p0;
          it.k0 = sl * per * WG_BK; it.k1 = std::min<int>((sl + 1) * per, ksteps) * WG_BK;
          if (n_slices > 1) { it.out = slab + (int64_t)sl * numel; it.ldo = w.kin; }
          else { it.out = w.dw; it.ldo = w.lddw; }
          if (f0 == 0) { it.bias[0] = w.bias[0]; it.bias[1] = w.bias[1]; it.bias[2] = w.bias[2]; }
          items.push_back(it);
        }
  }
  // longest items first: the tail of the launch is then made of short ones.  (Tried: an XCD-affine order -- the tiles of one
  // (tensor, row range), which read the same rows of x and gz, on workgroup indices congruent mod 8 so that they share an L2: 204 vs
  // 205 us; the launch is not waiting for its operands.)
  std::stable_sort(items.begin(), items.end(), [](const WgItem& a, const WgItem& b) { return (a.k1 - a.k0) > (b.k1 - b.k0); });
  OSD_TRY(upload(s, items, pl->items, &pl->d_items, &pl->cap_items));
  OSD_TRY(upload(s, reds, pl->reds, &pl->d_reds, &pl->cap_reds));
  const int n_items = (int)pl->items.size();
  const int grid = max_grid > 0 ? std::min(n_items, max_grid) : n_items;
  // precision 1 (bf16x3 split, gemm_bf3.h): the same items with both operands split into bf16 planes as they are staged
  if (h->precision == 1) hipLaunchKernelGGL(wgrad_group_kernel<1>, dim3((unsigned)grid), dim3(NTHREADS), WG3_LDS_BYTES, s, pl->d_items, n_items);
  else hipLaunchKernelGGL(wgrad_group_kernel<0>, dim3((unsigned)grid), dim3(NTHREADS), WG_LDS_BYTES, s, pl->d_items, n_items);
  OSD_HIP(hipGetLastError());
  if (!pl->reds.empty()) {
    hipLaunchKernelGGL(wgrad_group_reduce, dim3(64, (unsigned)pl->reds.size()), dim3(256), 0, s, pl->d_reds);
    OSD_HIP(hipGetLastError());
  }
  return OSD_OK;
}

// The GroupNorm affine gradients of the layers whose backward ran inside a dgrad epilogue since the last flush: one launch.
int gn_colsums_flush(osd_handle* h, hipStream_t s, int plan_index, const std::vector<GnColItem>& cols) {
  if (cols.empty()) return OSD_OK;
  while ((int)h->wg_plans.size() <= plan_index) h->wg_plans.push_back(new WgPlanDev());
  WgPlanDev* pl = static_cast<WgPlanDev*>(h->wg_plans[plan_index]);
  OSD_TRY(upload(s, cols, pl->cols, &pl->d_cols, &pl->cap_cols));
  int64_t max_rows = 0;
  for (const GnColItem& c : cols) max_rows = std::max(max_rows, c.rows);
  OSD_HIP(launch_gn_colsums(s, pl->d_cols, (int)pl->cols.size(), max_rows));
  return OSD_OK;
}

}  // namespace osd
