// chain_panel.hip -- host side of the LDS-resident reverse-chain kernel (chain_panel.h): fragment-ordered weight copies, the
// per-layer panel layout, eligibility, launch.  Shares the sync words, the conditioning buffers, the status protocol and the
// failure handling with chain.hip.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "chain_panel.h"
#include "handle.h"
#include "kernels.h"
#include "fwd.h"

namespace osd {

// dst[((fbg * K8 + i) * 64 + lane) * 4 + e] = W[32 fbg + (lane & 31)][8 i + 4 (lane >> 5) + e], zero beyond (F, K)
__global__ void k_pack_fragments(const float* __restrict__ w, int ldw, int F, int K, int nfbg, int K8, float* __restrict__ dst) {
  const long long total = (long long)nfbg * K8 * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long blk = i >> 6;
    const int i8 = (int)(blk % K8), fbg = (int)(blk / K8);
    const int f = 32 * fbg + (lane & 31), k = 8 * i8 + 4 * (lane >> 5);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < F) {
      const float* r = w + (size_t)f * ldw + k;
      if (k < K) v.x = r[0];
      if (k + 1 < K) v.y = r[1];
      if (k + 2 < K) v.z = r[2];
      if (k + 3 < K) v.w = r[3];
    }
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}

hipError_t launch_pack_fragments(hipStream_t s, const float* w, int ldw, int F, int K, int nfbg, int K8, float* dst) {
  const long long total = (long long)nfbg * K8 * 64;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_pack_fragments, dim3(grid), dim3(256), 0, s, w, ldw, F, K, nfbg, K8, dst);
  return hipGetLastError();
}

struct PanelPlan {
  bool ok = false;
  int n_layers = 0;
  PanelLayer L[PC_MAX_LAYERS];          // pointers unset
  int64_t wpk_off[PC_MAX_LAYERS];       // float offsets of the packed weights
  int nfbg[PC_MAX_LAYERS];              // packed feature blocks per layer
  int64_t wpk_floats = 0;
  int64_t ws_stride = 0;
  int cp_base = 0, xp_base = 0;
};

static int in_k8(int D) {
  int k8 = (D + 63) / 64 * 8;                       // whole groups of 8 blocks
  const int r = k8 % (PC_CHUNK / 8);
  if (r != 0 && r < PC_N8_MIN) k8 += PC_N8_MIN - r;      // the last chunk runs at least two groups
  return k8;
}

// Lays the layers out in the 64-row panel region; ok = false when the architecture does not fit (it then runs on chain.h).
static PanelPlan make_plan(const Arch& a, int Dk /* columns of the chain state = K of input_proj = F of output_proj */) {
  PanelPlan p;
  if (!chain_supported(a)) return p;
  if (a.H0 != 256) return p;                                   // input_proj: 4 waves x 64 features; cond_proj's tile fits one chunk buffer
  if ((int)a.layers.size() + 2 > PC_MAX_LAYERS) return p;
  if (a.block_out[a.n_blocks - 1] != 256) return p;            // output_proj's panel + the posterior transposers
  if (Dk < 512 || Dk % 4) return p;
  int nl = 0;
  int64_t woff = 0;
  auto add_w = [&](int l, int F, int K8, bool post) {
    p.nfbg[l] = post ? (F + 127) / 128 * 4 : (F + 31) / 32;
    p.wpk_off[l] = woff;
    woff += (int64_t)p.nfbg[l] * K8 * 256;
  };
  const int nchunk = (in_k8(Dk) * 8 + PC_CHUNK - 1) / PC_CHUNK;
  {
    PanelLayer& L = p.L[nl];
    L = PanelLayer{};
    L.K8 = in_k8(Dk); L.F = a.H0; L.kind = CK_INPUT; L.nseg = 1; L.seg[0] = {0, L.K8, -1}; L.seg[1] = {0, 0, -1};
    L.in_base = 0; L.in_ld = PC_CHUNK + 4;
    L.out_base = ((nchunk - 1) & 1) * PC_HALF; L.out_ld = a.H0 + 4; L.out_col = 0; L.spill = -1;
    add_w(nl, L.F, L.K8, false);
    ++nl;
  }
  p.cp_base = (nchunk & 1) * PC_HALF;
  // running activation
  int cur_base = p.L[0].out_base, cur_ld = p.L[0].out_ld, cur_col = 0, cur_w = a.H0;
  // block outputs the decoder needs again: kept in panel columns [256, 512) or spilled
  struct Skip { int width = 0; bool kept = false; int spill = -1; int consumer = -1; };
  std::vector<Skip> skip(a.n_blocks);
  for (int b = 0; b < a.n_blocks; ++b)
    if (a.layers[2 * b].K2 > 0) {
      const int sb = a.n_enc - 1 - (b - a.n_enc - 1);
      if (sb < 0 || sb >= a.n_blocks) return p;
      skip[sb].width = a.block_out[sb];
      skip[sb].consumer = b;
      if (a.layers[2 * b].K2 != skip[sb].width) return p;
    }
  int64_t ws_off = 0;
  int kept_until = -1;                  // block index whose first layer consumes the skip kept in columns [256, 512), -1 none
  for (int b = 0; b < a.n_blocks; ++b) {
    // may this block's output stay in the panel?  256 wide, nothing else kept, and every layer up to its consumer is 256 -> 256
    bool keep = false;
    if (skip[b].consumer >= 0 && skip[b].width == 256 && kept_until < 0) {
      keep = true;
      for (int bb = b + 1; bb < skip[b].consumer; ++bb)
        if (a.block_out[bb] != 256 || a.layers[2 * bb].K1 + a.layers[2 * bb].K2 != 256) keep = false;
      if (a.layers[2 * skip[b].consumer].K1 != 256) keep = false;      // [current | skip] must be the columns [0, 256) | [256, 512)
    }
    for (int half = 0; half < 2; ++half) {
      const LayerDesc& ld = a.layers[2 * b + half];
      PanelLayer& L = p.L[nl];
      L = PanelLayer{};
      const int K = ld.K1 + ld.K2;
      if (ld.N != 256 && ld.N != 512) return p;
      if (ld.K1 != cur_w || ld.K1 % 32 || ld.K2 % 32) return p;
      L.K8 = K / 8; L.F = ld.N; L.kind = ld.gw == 64 ? CK_GN64 : CK_GN32;
      if (ld.gw != ld.N / 8) return p;
      L.in_base = cur_base; L.in_ld = cur_ld;
      L.nseg = 1; L.seg[0] = {cur_col, ld.K1 / 8, -1}; L.seg[1] = {0, 0, -1};
      if (ld.K2 > 0) {
        const int sb = a.n_enc - 1 - (b - a.n_enc - 1);
        L.nseg = 2;
        if (skip[sb].kept) {
          if (cur_col != 0 || cur_ld != 516 || ld.K1 != 256) return p;
          L.seg[1] = {256, ld.K2 / 8, -1};
          kept_until = -1;
        } else {
          if (kept_until >= 0) return p;                       // the reload overwrites columns [0, K2)
          if (cur_col != 0 || ld.K2 + 4 > cur_ld) return p;    // the reloaded panel uses the running panel's geometry
          L.seg[1] = {0, ld.K2 / 8, skip[sb].spill};
        }
      }
      for (int s = 0; s < L.nseg; ++s)
        if (L.seg[s].n8 < PC_N8_MIN || L.seg[s].n8 % 8) return p;
      // output: the common panel [64][516] at the region start, except (a) the kept skip goes to columns [256, 512) and (b) the
      // last layer writes the compact panel output_proj reads beside the posterior transposers
      const bool last = (b == a.n_blocks - 1 && half == 1);
      L.out_base = 0; L.out_ld = last ? ld.N + 4 : 516; L.out_col = 0; L.spill = -1;
      if (kept_until >= 0 && (ld.N != 256 || K != 256)) return p;
      if (half == 1 && skip[b].consumer >= 0) {
        if (keep) { L.out_col = 256; skip[b].kept = true; kept_until = skip[b].consumer; }
        else {
          if (ld.N != 512) return p;                           // the reload code moves 4 waves x 128 features
          L.spill = (int)ws_off; skip[b].spill = (int)ws_off; ws_off += (int64_t)PC_BP * ld.N;
        }
      }
      add_w(nl, L.F, L.K8, false);
      cur_base = L.out_base; cur_ld = L.out_ld; cur_col = L.out_col; cur_w = ld.N;
      ++nl;
    }
  }
  if (kept_until >= 0) return p;
  {
    PanelLayer& L = p.L[nl];
    L = PanelLayer{};
    if (cur_w != 256 || cur_col != 0 || cur_base != 0 || cur_ld != 260) return p;
    L.K8 = cur_w / 8; L.F = Dk; L.kind = CK_POST; L.nseg = 1; L.seg[0] = {0, L.K8, -1}; L.seg[1] = {0, 0, -1};
    L.in_base = 0; L.in_ld = cur_ld; L.spill = -1;
    add_w(nl, L.F, L.K8, true);
    p.xp_base = PC_BP * cur_ld;
    if (p.xp_base + PC_NW * 2048 > PC_REGION) return p;
    ++nl;
  }
  p.n_layers = nl;
  p.wpk_floats = woff;
  p.ws_stride = std::max<int64_t>(ws_off, 64);
  p.ok = true;
  return p;
}

static int state_cols(const osd_handle* h) { return h->w_out_packed ? h->Dp : h->arch.D; }

bool panel_chain_supported(const osd_handle* h) { return make_plan(h->arch, state_cols(h)).ok; }

// Fragment-ordered copies of every weight the chain reads (10.7 MB at the BASELINE shape): made by the first chain after anything
// that may have changed the parameters (refresh_derived clears panel_wpk_valid).
int panel_chain_pack(osd_handle* h, hipStream_t s) {
  const Arch& a = h->arch;
  const PanelPlan p = make_plan(a, state_cols(h));
  if (!p.ok) return OSD_OK;
  if (h->panel_wpk_floats < p.wpk_floats) {
    if (h->panel_wpk) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(h->panel_wpk)); h->panel_wpk = nullptr; h->panel_wpk_floats = 0; }
    if (hipMalloc((void**)&h->panel_wpk, (size_t)p.wpk_floats * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->panel_wpk_floats = p.wpk_floats;
  }
  const bool padded = h->w_out_packed != nullptr;
  for (int l = 0; l < p.n_layers; ++l) {
    const float* w; int ldw, F, K;
    if (l == 0) { w = h->w_in_packed; ldw = h->w_in_ld; F = a.H0; K = h->w_in_ld; }       // zero beyond D already
    else if (l == p.n_layers - 1) {
      const int hl = a.block_out[a.n_blocks - 1];
      w = padded ? h->w_out_packed : h->params[a.pm.out_w]; ldw = hl; F = state_cols(h); K = hl;
    } else {
      const LayerDesc& ld = a.layers[l - 1];
      w = h->params[ld.w]; ldw = ld.K1 + ld.K2; F = ld.N; K = ld.K1 + ld.K2;
    }
    const long long total = (long long)p.nfbg[l] * p.L[l].K8 * 64;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pack_fragments, dim3(grid), dim3(256), 0, s, w, ldw, F, K, p.nfbg[l], p.L[l].K8, h->panel_wpk + p.wpk_off[l]);
    OSD_HIP(hipGetLastError());
  }
  h->panel_wpk_valid = true;
  return OSD_OK;
}

struct PanelDev { int occ = 0; int cus = 0; bool ready = false; };
static PanelDev g_panel_dev[16];

static int panel_device_limits(int device, int* max_grid) {
  if (device < 0 || device >= 16) { set_error("device %d out of range", device); return OSD_EINVAL; }
  PanelDev& d = g_panel_dev[device];
  if (!d.ready) {
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_chain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
#ifdef OSD_DIAG
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_chain_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_BYTES));
#endif
    int occ = 0;
    OSD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, panel_chain_kernel<false>, PC_THREADS, PC_LDS_BYTES));
    hipDeviceProp_t prop;
    OSD_HIP(hipGetDeviceProperties(&prop, device));
    d.occ = occ < 1 ? occ : 1;
    d.cus = prop.multiProcessorCount;
    d.ready = true;
  }
  *max_grid = d.occ * d.cus;
  return OSD_OK;
}

int panel_chain_slots(osd_handle* h) {
  int g = 0;
  return panel_device_limits(h->cfg.device, &g) == OSD_OK ? g : 0;
}

int panel_chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
                    float* x_out, float* mut_mask_out) {
  const Arch& a = h->arch;
  const int T = a.T, H0 = a.H0;
  hipStream_t s = h->stream;
  const bool padded = h->w_out_packed != nullptr;
  if (padded && noises) { set_error("internal: injected draws with D %% 4 != 0 run on the per-layer kernels"); return OSD_EUNSUPPORTED; }
  const int D = state_cols(h);
  PanelPlan p = make_plan(a, D);
  if (!p.ok) { set_error("internal: the LDS-resident chain is not available for this model"); return OSD_EUNSUPPORTED; }
  if (!h->panel_wpk_valid) OSD_TRY(panel_chain_pack(h, s));        // the parameters may have changed since the last chain
  int max_grid = 0;
  OSD_TRY(panel_device_limits(h->cfg.device, &max_grid));
  if (max_grid < 1) { set_error("the LDS-resident chain kernel does not fit this device"); return OSD_EUNSUPPORTED; }
  const int n_tiles = (int)((n + PC_BP - 1) / PC_BP);
  int grid = std::min(n_tiles, max_grid);
  if (h->chain_grid > 0) grid = (int)std::min<int64_t>(std::min(h->chain_grid, max_grid), (int64_t)n_tiles * T);

  PanelArgs pa{};
  pa.ws_stride = p.ws_stride;
  OSD_TRY(chain_ensure_buf(&h->chain_ws, &h->chain_ws_floats, (int64_t)max_grid * p.ws_stride, s));
  pa.ws = h->chain_ws;

  // conditioning for all rows, hoisted (as chain.hip), cproj padded to whole 64-row tiles
  const int64_t rows_pad = (int64_t)n_tiles * PC_BP;
  auto up64 = [](int64_t v) { return (v + 63) / 64 * 64; };
  const int64_t c_off_ce2 = up64(n * 64), c_off_cp = c_off_ce2 + up64(n * 64);
  OSD_TRY(chain_ensure_buf(&h->chain_cond, &h->chain_cond_floats, c_off_cp + up64(rows_pad * H0), s));
  FwdWs cw;
  cw.ce1 = h->chain_cond; cw.ce2 = h->chain_cond + c_off_ce2; cw.cproj = h->chain_cond + c_off_cp;
  OSD_TRY(run_cond(h, s, cond, n, cw));
  if (rows_pad > n) OSD_HIP(hipMemsetAsync(cw.cproj + n * H0, 0, (size_t)(rows_pad - n) * H0 * 4, s));

  float* xs = padded ? nullptr : x_out;
  if (padded) {
    OSD_TRY(chain_ensure_buf(&h->chain_xpad, &h->chain_xpad_floats, n * (int64_t)D, s));
    xs = h->chain_xpad;
    OSD_HIP(hipMemsetAsync(xs, 0, (size_t)n * D * 4, s));
  }
  if (x_T) OSD_HIP(launch_copy2d(s, x_T, a.D, xs, D, n, a.D));
  else OSD_HIP(launch_fill_randn(s, xs, D, n, a.D, seed, (uint32_t)row_offset, (uint32_t)T, TAG_POSTERIOR));

  OSD_TRY(chain_ensure_sync(h, n_tiles, s));
  pa.status = h->chain_sync;
  pa.queue = h->chain_sync + 1;
  pa.progress = h->chain_sync + 4 + 2048;
  pa.stamps = h->chain_stamps;
  pa.spin_budget = h->chain_spin_budget;

  const ParamMap& pm = a.pm;
  for (int l = 0; l < p.n_layers; ++l) {
    PanelLayer& L = p.L[l];
    L.wpk = h->panel_wpk + p.wpk_off[l];
    if (l == 0) { L.bias = h->params[pm.in_b]; }
    else if (l == p.n_layers - 1) { L.bias = padded ? h->b_out_packed : h->params[pm.out_b]; }
    else {
      const LayerDesc& ld = a.layers[l - 1];
      L.bias = h->params[ld.b]; L.gamma = h->params[ld.gamma]; L.beta = h->params[ld.beta];
    }
    pa.L[l] = L;
  }
  pa.n_layers = p.n_layers;
  pa.x = xs; pa.ldx = D; pa.D = D; pa.n = (int)n; pa.n_tiles = n_tiles;
  pa.cproj = cw.cproj; pa.ldc = H0; pa.temb = h->d_temb; pa.ldt = H0; pa.coef = h->d_coef;
  pa.z = noises; pa.ldzz = D; pa.z_step_stride = (long long)n * D; pa.z_t_first = T - 1;
  pa.seed = seed; pa.row_offset = (uint32_t)row_offset;
  pa.mut_mask = mut_mask_out; pa.mutation_dim = h->cfg.mutation_dim;
  pa.cp_base = p.cp_base; pa.xp_base = p.xp_base;

  const int seg = h->chain_steps_per_launch > 0 ? h->chain_steps_per_launch : T;
  const int n_launch = (T + seg - 1) / seg;
  OSD_HIP(hipStreamSynchronize(s));
  if (h->panel_args_cap < n_launch) {
    if (h->panel_args_dev) { OSD_HIP(hipFree(h->panel_args_dev)); h->panel_args_dev = nullptr; }
    free(h->panel_args_host);
    h->panel_args_cap = 0;
    h->panel_args_host = malloc((size_t)n_launch * sizeof(PanelArgs));
    if (!h->panel_args_host) { set_error("out of host memory"); return OSD_ENOMEM; }
    if (hipMalloc(&h->panel_args_dev, (size_t)n_launch * sizeof(PanelArgs)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->panel_args_cap = n_launch;
  }
  PanelArgs* const host_args = static_cast<PanelArgs*>(h->panel_args_host);
  int launch = 0;
  for (int done = 0; done < T; done += seg) {
    pa.t_first = T - 1 - done;
    pa.n_steps = std::min(seg, T - done);
    pa.base_done = (unsigned)done;
    if (done > 0) OSD_HIP(hipMemsetAsync(pa.queue, 0, 4, s));
    host_args[launch] = pa;
    const PanelArgs* dargs = static_cast<const PanelArgs*>(h->panel_args_dev) + launch;
    OSD_HIP(hipMemcpyAsync(const_cast<PanelArgs*>(dargs), &host_args[launch], sizeof(PanelArgs), hipMemcpyHostToDevice, s));
    ++launch;
#ifdef OSD_DIAG
    if (pa.stamps) hipLaunchKernelGGL(panel_chain_kernel<true>, dim3(grid), dim3(PC_THREADS), PC_LDS_BYTES, s, dargs);
    else
#endif
    hipLaunchKernelGGL(panel_chain_kernel<false>, dim3(grid), dim3(PC_THREADS), PC_LDS_BYTES, s, dargs);
    OSD_HIP(hipGetLastError());
  }
  if (padded) OSD_HIP(launch_copy2d(s, xs, D, x_out, a.D, n, a.D));
  h->chain_pending = true;
  {
    // a unit (64 rows through every layer) alone on its CU: ~0.5 TFLOP/s
    double flop_row = 0;
    for (int l = 0; l < p.n_layers; ++l) flop_row += 2.0 * p.L[l].K8 * 8 * p.L[l].F;
    const double unit_ms = 64.0 * flop_row / 0.5e12 * 1e3;
    double rounds = (double)(((int64_t)n_tiles * T + grid - 1) / grid);
    if (grid >= n_tiles) rounds = std::max(rounds, (double)T);
    h->chain_expected_ms = rounds * unit_ms;
  }
  return OSD_OK;
}

void panel_chain_free(osd_handle* h) {
  hipError_t e = hipSuccess;
  if (h->panel_wpk) e = hipFree(h->panel_wpk);
  if (h->panel_args_dev) e = hipFree(h->panel_args_dev);
  (void)e;
  free(h->panel_args_host);
  h->panel_wpk = nullptr; h->panel_wpk_floats = 0; h->panel_wpk_valid = false;
  h->panel_args_dev = nullptr; h->panel_args_host = nullptr; h->panel_args_cap = 0;
}

}  // namespace osd
