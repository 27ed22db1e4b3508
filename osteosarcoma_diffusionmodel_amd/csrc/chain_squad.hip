// chain_squad.hip -- host side of the small-batch reverse-chain kernel (chain_squad.h): eligibility, fragment-ordered weight
// copies, the per-panel buffers, launch.  Shares the status word, the conditioning buffers and the failure handling with chain.hip.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "chain_squad.h"
#include "handle.h"
#include "kernels.h"
#include "fwd.h"

namespace osd {

struct SquadPlan {
  bool ok = false;
  int n_layers = 0;
  SquadLayer L[SQ_MAX_LAYERS];          // pointers unset
  int64_t wpk_off[SQ_MAX_LAYERS];       // float offsets of the packed trunk weights
  int64_t in_off = 0, out_off = 0, bias_off = 0, wpk_floats = 0;
  int T32 = 0, h0_out = 0, last_in = 0, K8_out = 0;
  int64_t act_floats = 0;               // per panel
};

// Architectures the squad decomposition covers: 8 GroupNorm groups of 32 or 64 features (block widths 256 / 512), input_proj to
// 256 features (8 feature blocks: two per wave, one per workgroup in the reduce phase), output_proj from 256, every K a
// multiple of 128 (a wave's quarter is whole 8-k blocks of SQ_DEPTH-friendly length).
static SquadPlan make_plan(const Arch& a) {
  SquadPlan p;
  if (!chain_supported(a)) return p;
  if (a.H0 != 256) return p;
  if ((int)a.layers.size() > SQ_MAX_LAYERS) return p;
  if (a.block_out[a.n_blocks - 1] != 256) return p;
  p.T32 = (a.D + 31) / 32;
  if (p.T32 < SQ_S) return p;                                  // every workgroup owns at least one state tile
  int64_t woff = 0;
  int off = 0;
  std::vector<int> buf_of_layer(a.layers.size());
  p.h0_out = off; off += SQ_RP * a.H0;
  int cur = p.h0_out, cur_w = a.H0;
  for (int b = 0; b < a.n_blocks; ++b)
    for (int half = 0; half < 2; ++half) {
      const int li = 2 * b + half;
      const LayerDesc& ld = a.layers[li];
      SquadLayer& L = p.L[li];
      L = SquadLayer{};
      const int K = ld.K1 + ld.K2;
      if (ld.N != 256 && ld.N != 512) return p;
      if (ld.gw != ld.N / 8) return p;
      if (ld.K1 != cur_w || K % 128 || ld.K1 % 8 || ld.K2 % 8) return p;
      L.K8 = K / 8; L.F = ld.N; L.in0 = cur; L.n8_0 = ld.K1 / 8; L.in1 = -1;
      if (ld.K2 > 0) {
        const int sb = a.n_enc - 1 - (b - a.n_enc - 1);
        if (sb < 0 || sb >= b || a.block_out[sb] != ld.K2) return p;
        L.in1 = buf_of_layer[2 * sb + 1];
      }
      L.out = off; off += SQ_RP * ld.N;
      buf_of_layer[li] = L.out;
      p.wpk_off[li] = woff;
      woff += (int64_t)(ld.N / 32) * L.K8 * 256;
      cur = L.out; cur_w = ld.N;
    }
  p.n_layers = (int)a.layers.size();
  p.last_in = cur;
  p.K8_out = cur_w / 8;
  if (p.K8_out != 32) return p;                                // the operand of output_proj: 32 KB of LDS beside the K-split partials
  p.in_off = woff; woff += (int64_t)(a.H0 / 32) * (4 * p.T32) * 256;
  p.out_off = woff; woff += (int64_t)p.T32 * p.K8_out * 256;
  p.bias_off = woff; woff += (int64_t)p.T32 * 32;
  p.wpk_floats = woff;
  if (woff * 4 >= (int64_t)1 << 31) return p;                  // byte offsets into the packed weights are ints
  p.act_floats = (off + 63) / 64 * 64;
  p.ok = true;
  return p;
}

bool squad_chain_supported(const osd_handle* h) { return make_plan(h->arch).ok; }

struct SquadDev { int occ[3] = {0, 0, 0}; int cus = 0; int lds = 0; bool ready = false; };      // occupancies at `lds` bytes of dynamic LDS
static SquadDev g_squad_dev[16];

template <int WPC>
static int squad_occ(int* occ, int lds) {
  OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(squad_chain_kernel<WPC>), hipFuncAttributeMaxDynamicSharedMemorySize, sq_lds_bytes(SQ_MAX_LAYERS)));
#ifdef OSD_DIAG
  OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(squad_chain_kernel<WPC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, sq_lds_bytes(SQ_MAX_LAYERS)));
#endif
  OSD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, squad_chain_kernel<WPC>, SQ_THREADS, lds));
  if (*occ > WPC) *occ = WPC;
  return OSD_OK;
}

static int squad_device(int device, int lds, SquadDev** out) {
  if (device < 0 || device >= 16) { set_error("device %d out of range", device); return OSD_EINVAL; }
  SquadDev& d = g_squad_dev[device];
  if (!d.ready || d.lds != lds) {
    OSD_TRY(squad_occ<1>(&d.occ[0], lds));
    OSD_TRY(squad_occ<2>(&d.occ[1], lds));
    OSD_TRY(squad_occ<3>(&d.occ[2], lds));
    d.lds = lds;
    hipDeviceProp_t prop;
    OSD_HIP(hipGetDeviceProperties(&prop, device));
    d.cus = prop.multiProcessorCount;
    d.ready = true;
  }
  *out = &d;
  return OSD_OK;
}

// Workgroups per CU (1..3) that make every squad of an n-row chain resident at once; 0 = the batch is too large for this kernel.
static int squad_wpc(osd_handle* h, int64_t n) {
  SquadDev* d = nullptr;
  if (squad_device(h->cfg.device, sq_lds_bytes((int)h->arch.layers.size()), &d) != OSD_OK) { (void)hipGetLastError(); return 0; }
  const int64_t wgs = (n + SQ_RP - 1) / SQ_RP * SQ_S;
  for (int w = 1; w <= 3; ++w)
    if (d->occ[w - 1] >= w && wgs <= (int64_t)w * d->cus) return w;
  return 0;
}

// auto: chains whose squads all fit on the chip at once (3 072 rows on 256 CUs)
bool squad_window(osd_handle* h, int64_t n) {
  if (h->chain_variant != 0 && h->chain_variant != 3) return false;
  if (!squad_chain_supported(h)) return false;
  return squad_wpc(h, n) > 0;
}

static int squad_pack(osd_handle* h, hipStream_t s, const SquadPlan& p) {
  const Arch& a = h->arch;
  if (h->squad_wpk_floats < p.wpk_floats) {
    if (h->squad_wpk) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(h->squad_wpk)); h->squad_wpk = nullptr; h->squad_wpk_floats = 0; }
    if (hipMalloc((void**)&h->squad_wpk, (size_t)p.wpk_floats * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->squad_wpk_floats = p.wpk_floats;
  }
  for (int l = 0; l < p.n_layers; ++l) {
    const LayerDesc& ld = a.layers[l];
    OSD_HIP(launch_pack_fragments(s, h->params[ld.w], ld.K1 + ld.K2, ld.N, ld.K1 + ld.K2, ld.N / 32, p.L[l].K8, h->squad_wpk + p.wpk_off[l]));
  }
  // input_proj: the unpadded parameter, zero beyond D
  OSD_HIP(launch_pack_fragments(s, h->params[a.pm.in_w], a.D, a.H0, a.D, a.H0 / 32, 4 * p.T32, h->squad_wpk + p.in_off));
  const int hl = a.block_out[a.n_blocks - 1];
  OSD_HIP(launch_pack_fragments(s, h->params[a.pm.out_w], hl, a.D, hl, p.T32, p.K8_out, h->squad_wpk + p.out_off));
  OSD_HIP(hipMemsetAsync(h->squad_wpk + p.bias_off, 0, (size_t)p.T32 * 32 * 4, s));
  OSD_HIP(hipMemcpyAsync(h->squad_wpk + p.bias_off, h->params[a.pm.out_b], (size_t)a.D * 4, hipMemcpyDeviceToDevice, s));
  h->squad_wpk_valid = true;
  return OSD_OK;
}

int squad_chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
                    float* x_out, float* mut_mask_out) {
  const Arch& a = h->arch;
  const int T = a.T, H0 = a.H0, D = a.D;
  hipStream_t s = h->stream;
  SquadPlan p = make_plan(a);
  if (!p.ok) { set_error("internal: the squad chain is not available for this model"); return OSD_EUNSUPPORTED; }
  const int wpc = squad_wpc(h, n);
  if (wpc < 1) { set_error("internal: %lld rows are more than the squad chain keeps resident", (long long)n); return OSD_EUNSUPPORTED; }
  if (!h->squad_wpk_valid) OSD_TRY(squad_pack(h, s, p));
  const int n_panels = (int)((n + SQ_RP - 1) / SQ_RP);

  // conditioning for all rows, hoisted (as chain.hip)
  auto up64 = [](int64_t v) { return (v + 63) / 64 * 64; };
  const int64_t c_off_ce2 = up64(n * 64), c_off_cp = c_off_ce2 + up64(n * 64);
  OSD_TRY(chain_ensure_buf(&h->chain_cond, &h->chain_cond_floats, c_off_cp + up64(n * H0), s));
  FwdWs cw;
  cw.ce1 = h->chain_cond; cw.ce2 = h->chain_cond + c_off_ce2; cw.cproj = h->chain_cond + c_off_cp;
  OSD_TRY(run_cond(h, s, cond, n, cw));

  // the chain state lives in the caller's rows between launches (any D: the kernel reads and writes it element-wise)
  if (x_T) { if (x_T != x_out) OSD_HIP(launch_copy2d(s, x_T, D, x_out, D, n, D)); }
  else OSD_HIP(launch_fill_randn(s, x_out, D, n, D, seed, (uint32_t)row_offset, (uint32_t)T, TAG_POSTERIOR));

  SquadArgs sa{};
  const int64_t xs_stride = (int64_t)p.T32 * 4 * 256;
  const int64_t slab_stride = (int64_t)SQ_S * H0 * SQ_RP;
  OSD_TRY(chain_ensure_buf(&h->chain_ws, &h->chain_ws_floats, (int64_t)n_panels * (xs_stride + slab_stride + p.act_floats), s));
  sa.xs = h->chain_ws; sa.xs_stride = xs_stride;
  sa.slab = sa.xs + (int64_t)n_panels * xs_stride; sa.slab_stride = slab_stride;
  sa.act = sa.slab + (int64_t)n_panels * slab_stride; sa.act_stride = p.act_floats;

  OSD_TRY(chain_ensure_sync(h, (int64_t)n_panels * 16, s));
  sa.status = h->chain_sync;
  sa.bar = h->chain_sync + 4 + 2048;
  sa.spin_budget = h->chain_spin_budget;
  sa.stamps = h->chain_stamps;

  for (int l = 0; l < p.n_layers; ++l) {
    SquadLayer& L = p.L[l];
    const LayerDesc& ld = a.layers[l];
    L.w_off = (int)p.wpk_off[l];
    L.bias = h->params[ld.b]; L.gamma = h->params[ld.gamma]; L.beta = h->params[ld.beta];
    sa.L[l] = L;
  }
  sa.n_layers = p.n_layers;
  sa.wpk = h->squad_wpk; sa.wpk_floats = p.wpk_floats;
  sa.in_off = (int)p.in_off; sa.bias_in = h->params[a.pm.in_b]; sa.H0 = H0;
  sa.out_off = (int)p.out_off; sa.bias_out = h->squad_wpk + p.bias_off;
  sa.T32 = p.T32; sa.h0_out = p.h0_out; sa.last_in = p.last_in; sa.K8_out = p.K8_out;
  sa.x = x_out; sa.ldx = D; sa.D = D; sa.n = (int)n;
  sa.cproj = cw.cproj; sa.ldc = H0; sa.temb = h->d_temb; sa.ldt = H0; sa.coef = h->d_coef;
  sa.z = noises; sa.ldzz = D; sa.z_step_stride = (long long)n * D; sa.z_t_first = T - 1;
  sa.seed = seed; sa.row_offset = (uint32_t)row_offset;
  sa.mut_mask = mut_mask_out; sa.mutation_dim = h->cfg.mutation_dim;

  const int seg = h->chain_steps_per_launch > 0 ? h->chain_steps_per_launch : T;
  const int n_launch = (T + seg - 1) / seg;
  OSD_HIP(hipStreamSynchronize(s));
  if (h->squad_args_cap < n_launch) {
    if (h->squad_args_dev) { OSD_HIP(hipFree(h->squad_args_dev)); h->squad_args_dev = nullptr; }
    free(h->squad_args_host);
    h->squad_args_cap = 0;
    h->squad_args_host = malloc((size_t)n_launch * sizeof(SquadArgs));
    if (!h->squad_args_host) { set_error("out of host memory"); return OSD_ENOMEM; }
    if (hipMalloc(&h->squad_args_dev, (size_t)n_launch * sizeof(SquadArgs)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->squad_args_cap = n_launch;
  }
  SquadArgs* const host_args = static_cast<SquadArgs*>(h->squad_args_host);
  const int grid = n_panels * SQ_S;
  const int lds = sq_lds_bytes(p.n_layers);
  int launch = 0;
  for (int done = 0; done < T; done += seg) {
    sa.t_first = T - 1 - done;
    sa.n_steps = std::min(seg, T - done);
    if (done > 0) OSD_HIP(hipMemsetAsync(sa.bar, 0, (size_t)n_panels * 16 * 4, s));      // a launch counts its barriers from zero
    host_args[launch] = sa;
    const SquadArgs* dargs = static_cast<const SquadArgs*>(h->squad_args_dev) + launch;
    OSD_HIP(hipMemcpyAsync(const_cast<SquadArgs*>(dargs), &host_args[launch], sizeof(SquadArgs), hipMemcpyHostToDevice, s));
    ++launch;
#ifdef OSD_DIAG
    if (sa.stamps && wpc == 1) hipLaunchKernelGGL((squad_chain_kernel<1, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (sa.stamps && wpc == 2) hipLaunchKernelGGL((squad_chain_kernel<2, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (sa.stamps) hipLaunchKernelGGL((squad_chain_kernel<3, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else
#endif
    if (wpc == 1) hipLaunchKernelGGL(squad_chain_kernel<1>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (wpc == 2) hipLaunchKernelGGL(squad_chain_kernel<2>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else hipLaunchKernelGGL(squad_chain_kernel<3>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    OSD_HIP(hipGetLastError());
  }
  h->chain_pending = true;
  h->chain_expected_ms = (double)T * 0.5 * wpc;       // measured: 0.1-0.2 ms per step; generous (chain.hip multiplies by 10 and adds 2 s)
  return OSD_OK;
}

void squad_chain_free(osd_handle* h) {
  hipError_t e = hipSuccess;
  if (h->squad_wpk) e = hipFree(h->squad_wpk);
  if (h->squad_args_dev) e = hipFree(h->squad_args_dev);
  (void)e;
  free(h->squad_args_host);
  h->squad_wpk = nullptr; h->squad_wpk_floats = 0; h->squad_wpk_valid = false;
  h->squad_args_dev = nullptr; h->squad_args_host = nullptr; h->squad_args_cap = 0;
}

}  // namespace osd
