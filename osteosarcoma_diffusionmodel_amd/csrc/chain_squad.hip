// chain_squad.hip -- host side of the small-batch reverse-chain kernel (chain_squad.h): eligibility, fragment-ordered weight
// copies, the per-panel buffers, launch.  Shares the status word, the conditioning buffers and the failure handling with chain.hip.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "chain_squad.h"
#include "chain_squad16.h"
#include "train_squad.h"
#include "train_squad_bwd.h"
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
// rp = patients per panel: 32 (chain_squad.h: 8-k blocks, 32-feature state tiles) or 16 (chain_squad16.h: 16-k blocks, 16-feature tiles)
static SquadPlan make_plan(const Arch& a, int rp = SQ_RP) {
  SquadPlan p;
  const int kblk = rp == 32 ? 8 : 16, tile = rp == 32 ? 32 : 16;
  if (!chain_supported(a)) return p;
  if (a.H0 != 256) return p;
  if ((int)a.layers.size() > SQ_MAX_LAYERS) return p;
  if (a.block_out[a.n_blocks - 1] != 256) return p;
  p.T32 = (a.D + tile - 1) / tile;
  if ((a.D + 31) / 32 < SQ_S) return p;                        // every workgroup owns at least one state tile (of either size)
  int64_t woff = 0;
  int off = 0;
  std::vector<int> buf_of_layer(a.layers.size());
  p.h0_out = off; off += rp * a.H0;
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
      if (ld.K1 != cur_w || K % 128 || ld.K1 % 16 || ld.K2 % 16) return p;
      L.K8 = K / kblk; L.F = ld.N; L.in0 = cur; L.n8_0 = ld.K1 / kblk; L.in1 = -1;
      if (ld.K2 > 0) {
        const int sb = a.n_enc - 1 - (b - a.n_enc - 1);
        if (sb < 0 || sb >= b || a.block_out[sb] != ld.K2) return p;
        L.in1 = buf_of_layer[2 * sb + 1];
      }
      L.out = off; off += rp * ld.N;
      buf_of_layer[li] = L.out;
      p.wpk_off[li] = woff;
      woff += (int64_t)ld.N * K;
      cur = L.out; cur_w = ld.N;
    }
  p.n_layers = (int)a.layers.size();
  p.last_in = cur;
  p.K8_out = cur_w / kblk;
  if (cur_w != 256) return p;                                  // the operand of output_proj in LDS beside the K-split partials
  p.in_off = woff; woff += (int64_t)a.H0 * p.T32 * tile;
  p.out_off = woff; woff += (int64_t)p.T32 * tile * cur_w;
  p.bias_off = woff; woff += (int64_t)p.T32 * tile;
  p.wpk_floats = woff;
  if (woff * 4 >= (int64_t)1 << 31) return p;                  // byte offsets into the packed weights are ints
  p.act_floats = (off + 63) / 64 * 64;
  p.ok = true;
  return p;
}

bool squad_chain_supported(const osd_handle* h) { return make_plan(h->arch).ok; }

struct SquadDev { int occ[3] = {0, 0, 0}; int occ16 = 0; int cus = 0; int lds = 0; bool ready = false; };      // occupancies at `lds` bytes of dynamic LDS
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
    {
      const int lds16 = lds - (SQ_STAGE_FLOATS - SQ16_STAGE_FLOATS) * 4;
      OSD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d.occ16, squad16_chain_kernel<false>, SQ_THREADS, lds16));
      if (d.occ16 > 2) d.occ16 = 2;
    }
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

// 16-patient panels (chain_squad16.h): when they put two workgroups of different squads on every CU at most -- up to 1 024 rows on
// 256 CUs -- where the 32-patient squads are one workgroup per CU or less
static bool squad_use16(osd_handle* h, int64_t n) {
  if (h->squad_panel == 32) return false;                  // osd_set_option("squad_panel"): 0 auto, 16 wherever it fits, 32 never
  SquadDev* d = nullptr;
  if (squad_device(h->cfg.device, sq_lds_bytes((int)h->arch.layers.size()), &d) != OSD_OK) { (void)hipGetLastError(); return false; }
  if (d->occ16 < 2) return false;
  const int64_t wgs16 = (n + SQ16_RP - 1) / SQ16_RP * SQ_S, wgs32 = (n + SQ_RP - 1) / SQ_RP * SQ_S;
  if (wgs16 > 2 * (int64_t)d->cus) return false;
  return h->squad_panel == 16 || wgs32 <= (int64_t)d->cus;
}

// auto: chains whose squads all fit on the chip at once (3 072 rows on 256 CUs)
bool squad_window(osd_handle* h, int64_t n) {
  if (h->chain_variant != 0 && h->chain_variant != 3) return false;
  if (!squad_chain_supported(h)) return false;
  return squad_wpc(h, n) > 0;
}

// dst[((fb * K16 + i) * 64 + lane) * 4 + e] = W[16 fb + (lane & 15)][16 i + 4 (lane >> 4) + e], zero beyond (F, K): chain_squad16.h's order
__global__ void k_pack_fragments16(const float* __restrict__ w, int ldw, int F, int K, int nfb, int K16, float* __restrict__ dst) {
  const long long total = (long long)nfb * K16 * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long blk = i >> 6;
    const int i16 = (int)(blk % K16), fb = (int)(blk / K16);
    const int f = 16 * fb + (lane & 15), k = 16 * i16 + 4 * (lane >> 4);
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

static hipError_t pack_any(hipStream_t s, int rp, const float* w, int ldw, int F, int K, int nfb, int KB, float* dst) {
  if (rp == 32) return launch_pack_fragments(s, w, ldw, F, K, nfb, KB, dst);
  const long long total = (long long)nfb * KB * 64;
  const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_pack_fragments16, dim3(grid), dim3(256), 0, s, w, ldw, F, K, nfb, KB, dst);
  return hipGetLastError();
}

// slot 0: the 32-patient kernel's copies, slot 1: the 16-patient kernel's
static int squad_pack(osd_handle* h, hipStream_t s, const SquadPlan& p, int rp) {
  const Arch& a = h->arch;
  const int slot = rp == 32 ? 0 : 1, tile = rp;
  float*& buf = h->squad_wpk[slot];
  if (h->squad_wpk_floats[slot] < p.wpk_floats) {
    if (buf) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(buf)); buf = nullptr; h->squad_wpk_floats[slot] = 0; }
    if (hipMalloc((void**)&buf, (size_t)p.wpk_floats * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->squad_wpk_floats[slot] = p.wpk_floats;
  }
  for (int l = 0; l < p.n_layers; ++l) {
    const LayerDesc& ld = a.layers[l];
    OSD_HIP(pack_any(s, rp, h->params[ld.w], ld.K1 + ld.K2, ld.N, ld.K1 + ld.K2, ld.N / tile, p.L[l].K8, buf + p.wpk_off[l]));
  }
  // input_proj: the unpadded parameter, zero beyond D
  OSD_HIP(pack_any(s, rp, h->params[a.pm.in_w], a.D, a.H0, a.D, a.H0 / tile, rp == 32 ? 4 * p.T32 : p.T32, buf + p.in_off));
  const int hl = a.block_out[a.n_blocks - 1];
  OSD_HIP(pack_any(s, rp, h->params[a.pm.out_w], hl, a.D, hl, p.T32, p.K8_out, buf + p.out_off));
  OSD_HIP(hipMemsetAsync(buf + p.bias_off, 0, (size_t)p.T32 * tile * 4, s));
  OSD_HIP(hipMemcpyAsync(buf + p.bias_off, h->params[a.pm.out_b], (size_t)a.D * 4, hipMemcpyDeviceToDevice, s));
  h->squad_wpk_valid[slot] = true;
  return OSD_OK;
}

int squad_chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
                    float* x_out, float* mut_mask_out) {
  const Arch& a = h->arch;
  const int T = a.T, H0 = a.H0, D = a.D;
  hipStream_t s = h->stream;
  const int wpc = squad_wpc(h, n);
  if (wpc < 1) { set_error("internal: %lld rows are more than the squad chain keeps resident", (long long)n); return OSD_EUNSUPPORTED; }
  const int rp = squad_use16(h, n) ? SQ16_RP : SQ_RP;
  const int slot = rp == 32 ? 0 : 1;
  SquadPlan p = make_plan(a, rp);
  if (!p.ok) { set_error("internal: the squad chain is not available for this model"); return OSD_EUNSUPPORTED; }
  if (!h->squad_wpk_valid[slot]) OSD_TRY(squad_pack(h, s, p, rp));
  const int n_panels = (int)((n + rp - 1) / rp);

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
  const int64_t xs_stride = (int64_t)p.T32 * rp * rp;              // tiles of rp features x rp patients
  const int64_t slab_stride = (int64_t)SQ_S * H0 * rp;
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
  sa.wpk = h->squad_wpk[slot]; sa.wpk_floats = p.wpk_floats;
  sa.in_off = (int)p.in_off; sa.bias_in = h->params[a.pm.in_b]; sa.H0 = H0;
  sa.out_off = (int)p.out_off; sa.bias_out = h->squad_wpk[slot] + p.bias_off;
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
  const int lds = rp == 32 ? sq_lds_bytes(p.n_layers) : sq16_lds_bytes(p.n_layers);
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
    if (rp == 16) hipLaunchKernelGGL(squad16_chain_kernel<false>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (sa.stamps && wpc == 1) hipLaunchKernelGGL((squad_chain_kernel<1, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (sa.stamps && wpc == 2) hipLaunchKernelGGL((squad_chain_kernel<2, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (sa.stamps) hipLaunchKernelGGL((squad_chain_kernel<3, true>), dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else
#endif
    if (rp == 16) hipLaunchKernelGGL(squad16_chain_kernel<false>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (wpc == 1) hipLaunchKernelGGL(squad_chain_kernel<1>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else if (wpc == 2) hipLaunchKernelGGL(squad_chain_kernel<2>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    else hipLaunchKernelGGL(squad_chain_kernel<3>, dim3(grid), dim3(SQ_THREADS), lds, s, dargs);
    OSD_HIP(hipGetLastError());
  }
  h->chain_pending = true;
  h->last_squad_rp = rp;
  h->chain_expected_ms = (double)T * 0.5 * wpc;       // measured: 0.1-0.2 ms per step; generous (chain.hip multiplies by 10 and adds 2 s)
  return OSD_OK;
}

// ---- the training forward trunk as one launch of squads (train_squad.h) ----------------------------------------------------------
// every layer's weight into fragment order in ONE launch (blockIdx.y = layer): the parameters change every step
struct PackMulti { const float* w[SQ_MAX_LAYERS]; int F[SQ_MAX_LAYERS], K[SQ_MAX_LAYERS]; long long off[SQ_MAX_LAYERS]; };
__device__ __forceinline__ void pack_multi_item(const PackMulti& pm, float* __restrict__ dst, int l) {
  const float* __restrict__ w = pm.w[l];
  const int F = pm.F[l], K = pm.K[l], K8 = K / 8;
  const long long total = (long long)(F / 32) * K8 * 64;
  float4* const out = reinterpret_cast<float4*>(dst + pm.off[l]);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long blk = i >> 6;
    const int i8 = (int)(blk % K8), fb = (int)(blk / K8);
    out[i] = *reinterpret_cast<const float4*>(w + (size_t)(32 * fb + (lane & 31)) * K + 8 * i8 + 4 * (lane >> 5));
  }
}
__global__ void k_pack_fragments_multi(PackMulti pm, float* __restrict__ dst) { pack_multi_item(pm, dst, blockIdx.y); }

// the backward squads' copies (train_squad_bwd.h): fragment order of W^T for a block of columns,
// dst[((fb * K8 + i) * 64 + lane) * 4 + e] = W[8 i + 4 (lane >> 5) + e][c0 + 32 fb + (lane & 31)]
struct PackMultiT { const float* w[2 * SQ_MAX_LAYERS]; int ldw[2 * SQ_MAX_LAYERS], c0[2 * SQ_MAX_LAYERS], F[2 * SQ_MAX_LAYERS], K[2 * SQ_MAX_LAYERS]; long long off[2 * SQ_MAX_LAYERS]; };
__device__ __forceinline__ void pack_multi_t_item(const PackMultiT& pm, float* __restrict__ dst, int l) {
  const float* __restrict__ w = pm.w[l];
  const int ldw = pm.ldw[l], c0 = pm.c0[l], F = pm.F[l], K8 = pm.K[l] / 8;
  const long long total = (long long)(F / 32) * K8 * 64;
  float4* const out = reinterpret_cast<float4*>(dst + pm.off[l]);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long blk = i >> 6;
    const int i8 = (int)(blk % K8), fb = (int)(blk / K8);
    const float* r = w + (size_t)(8 * i8 + 4 * (lane >> 5)) * ldw + c0 + 32 * fb + (lane & 31);
    out[i] = make_float4(r[0], r[ldw], r[2 * (size_t)ldw], r[3 * (size_t)ldw]);
  }
}
__global__ void k_pack_fragments_multi_t(PackMultiT pm, float* __restrict__ dst) { pack_multi_t_item(pm, dst, blockIdx.y); }
// both sets in one launch (a training step that will run the backward squads too): blockIdx.y < nl = a forward layer, else a backward item
__global__ void k_pack_fragments_both(PackMulti pm, float* __restrict__ dst, int nl, PackMultiT pt, float* __restrict__ dst_t) {
  if ((int)blockIdx.y < nl) pack_multi_item(pm, dst, blockIdx.y);
  else pack_multi_t_item(pt, dst_t, (int)blockIdx.y - nl);
}
// The backward items in the order train_squad_backward walks its phases (their offsets are the phases' w_off / skip_w_off).
static int bwd_pack_list(const osd_handle* h, PackMultiT* pm, long long* floats) {
  const Arch& a = h->arch;
  int npk = 0;
  long long woff = 0;
  auto pack = [&](const float* w, int ldw, int c0, int F, int K) {
    pm->w[npk] = w; pm->ldw[npk] = ldw; pm->c0[npk] = c0; pm->F[npk] = F; pm->K[npk] = K; pm->off[npk] = woff; ++npk;
    woff += (long long)F * K;
  };
  for (int b = a.n_blocks - 1; b >= 0; --b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    const int C = l1.N, Kt = l1.K1 + l1.K2;
    pack(h->params[l2.w], C, 0, C, C);
    pack(h->params[l1.w], Kt, 0, l1.K1, C);
    if (b > 0 && l1.K2 > 0) pack(h->params[l1.w], Kt, l1.K1, l1.K2, C);
  }
  if (floats) *floats = woff;
  return npk;
}

// floats of unit-order activations per 32-patient sub-panel (0 = the model is outside the squad decomposition) and of the
// fragment-ordered trunk weights
int64_t train_squad_act_floats(const Arch& a, int64_t* wpk_floats) {
  const SquadPlan p = make_plan(a, SQ_RP);
  if (wpk_floats) *wpk_floats = p.ok ? p.in_off : 0;
  return p.ok ? p.act_floats : 0;
}

// Worth it from a few thousand rows on (64-patient panels: 2 048 rows = half the CUs); smaller batches keep the per-layer launches.
bool train_squad_ok(const osd_handle* h, int64_t n) {
  if (h->train_squad == 0 || n < 2048 || !make_plan(h->arch, SQ_RP).ok) return false;
  for (const LayerDesc& ld : h->arch.layers)                      // 16-byte fragment loads straight from the parameters
    if (reinterpret_cast<uintptr_t>(h->params[ld.w]) & 15) return false;
  return true;
}

int train_squad_forward(osd_handle* h, hipStream_t s, const FwdWs& ws, const TrunkIn& in, float* act_units, float* wpk, unsigned* bar_and_status,
                        int64_t panels, float* loss_poison, float* wpk_t) {
  const Arch& a = h->arch;
  const SquadPlan p = make_plan(a, SQ_RP);
  if (!p.ok) { set_error("internal: the squad forward is not available for this model"); return OSD_EUNSUPPORTED; }
  static bool attr_done = false;
  if (!attr_done) {
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(train_squad_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ts_lds_bytes(SQ_MAX_LAYERS)));
    attr_done = true;
  }
  TrainSquadArgs ta{};
  PackMulti pm{};
  const bool drop = in.train && h->cfg.dropout_p > 0.f;
  for (int b = 0; b < a.n_blocks; ++b)
    for (int half = 0; half < 2; ++half) {
      const int li = 2 * b + half;
      const LayerDesc& ld = a.layers[li];
      TrainSquadLayer& L = ta.L[li];
      L.w_off = (int)p.wpk_off[li]; L.K = ld.K1 + ld.K2; L.F = ld.N;
      pm.w[li] = h->params[ld.w]; pm.F[li] = ld.N; pm.K[li] = ld.K1 + ld.K2; pm.off[li] = p.wpk_off[li];
      L.in0 = p.L[li].in0; L.n8_0 = p.L[li].n8_0; L.in1 = p.L[li].in1; L.out = p.L[li].out;
      L.bias = h->params[ld.b]; L.gamma = h->params[ld.gamma]; L.beta = h->params[ld.beta];
      L.y = half == 0 ? ws.mid[b] : ws.out[b]; L.ldy = ld.N;
      L.z = in.save ? (half == 0 ? ws.z1[b] : ws.z2[b]) : nullptr;
      L.stats = in.save ? (half == 0 ? ws.st1[b] : ws.st2[b]) : nullptr;
      L.drop_mode = (half == 0 && drop) ? (in.masks ? 1 : 2) : 0;
      L.mask = (half == 0 && drop && in.masks) ? in.masks[b] : nullptr; L.ldm = ld.N;
      L.tag = TAG_DROPOUT + (uint32_t)b;
    }
  ta.n_layers = p.n_layers;
  ta.wpk = wpk; ta.wpk_floats = p.in_off;            // the trunk weights come first in the plan's order
  // wpk_t: this step's backward will run as squads too (train_squad_backward) -- its transposed copies ride in the same launch
  h->sq_wpk_t_fresh = false;
  if (wpk_t) {
    PackMultiT pt{};
    const int npk = bwd_pack_list(h, &pt, nullptr);
    hipLaunchKernelGGL(k_pack_fragments_both, dim3(128, (unsigned)(p.n_layers + npk)), dim3(256), 0, s, pm, wpk, p.n_layers, pt, wpk_t);
    OSD_HIP(hipGetLastError());
    h->sq_wpk_t_fresh = true;
  } else {
    hipLaunchKernelGGL(k_pack_fragments_multi, dim3(128, (unsigned)p.n_layers), dim3(256), 0, s, pm, wpk);
    OSD_HIP(hipGetLastError());
  }
  ta.h0 = ws.h0; ta.ldh = a.H0; ta.h0_out = p.h0_out;
  ta.n = (int)in.n;
  ta.act = act_units; ta.act_stride = p.act_floats;
  ta.bar = bar_and_status; ta.status = bar_and_status + panels * 16;
  ta.loss_poison = loss_poison;
  ta.spin_budget = std::min<unsigned long long>(h->chain_spin_budget, 20000000ull);      // <= 0.2 s: a training step has no fallback to wait for
  ta.keep_scale = (float)(1.0 / (1.0 - (double)h->cfg.dropout_p)); ta.p_drop = h->cfg.dropout_p;
  ta.seed = in.seed; ta.row_offset = in.row_offset; ta.step = in.drop_step;
  ta.stamps = h->chain_stamps;                       // diagnostic builds (osd_dbg_chain_stamps), else null
  hipLaunchKernelGGL(train_squad_fwd_kernel, dim3((unsigned)(panels * SQ_S)), dim3(SQ_THREADS), ts_lds_bytes(p.n_layers), s, ta);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

// ---- the backward pass's dgrad chain as one launch of squads (train_squad_bwd.h) --------------------------------------------------
int64_t train_squad_bwd_wpk_floats(const Arch& a) {
  int64_t f = 0;
  for (const LayerDesc& ld : a.layers) f += (int64_t)ld.N * (ld.K1 + ld.K2);
  return f;
}

int train_squad_backward(osd_handle* h, hipStream_t s, const FwdWs& f, const TrainSquadBwdBufs& B, int64_t n, float* gact_units, float* wpk,
                         unsigned* bar_and_status, int64_t panels, float* loss_poison) {
  const Arch& a = h->arch;
  const SquadPlan p = make_plan(a, SQ_RP);
  if (!p.ok) { set_error("internal: the squad backward is not available for this model"); return OSD_EUNSUPPORTED; }
  static bool attr_done = false;
  if (!attr_done) {
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(train_squad_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ts_lds_bytes(SQ_MAX_LAYERS)));
    attr_done = true;
  }
  TrainSquadBwdArgs ta{};
  PackMultiT pm{};
  int np = 0, npk = 0;
  long long woff = 0;
  auto pack = [&](const float* w, int ldw, int c0, int F, int K) -> int {
    pm.w[npk] = w; pm.ldw[npk] = ldw; pm.c0[npk] = c0; pm.F[npk] = F; pm.K[npk] = K; pm.off[npk] = woff; ++npk;
    const int off = (int)woff;
    woff += (long long)F * K;
    return off;
  };
  for (int b = a.n_blocks - 1; b >= 0; --b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    const int C = l1.N, Kt = l1.K1 + l1.K2;
    {   // through the block's second Linear into its first layer (dropout sits behind that one)
      TrainSquadBwdPhase& P = ta.P[np++];
      P = TrainSquadBwdPhase{};
      P.w_off = pack(h->params[l2.w], C, 0, C, C); P.K = C; P.F = C;
      P.in = p.L[2 * b + 1].out; P.out = p.L[2 * b].out; P.prm = 2 * b;
      P.z = f.z1[b]; P.stats = f.st1[b]; P.gy = B.g_mid[b]; P.gz = B.g_z1[b]; P.accumulate = 0;
      P.drop_mode = B.drop ? (B.masks ? 1 : 2) : 0; P.mask = (B.drop && B.masks) ? B.masks[b] : nullptr; P.ldm = C; P.tag = TAG_DROPOUT + (uint32_t)b;
    }
    if (b == 0) {                 // into h0: no GroupNorm below -- a plain phase
      TrainSquadBwdPhase& P = ta.P[np++];
      P = TrainSquadBwdPhase{};
      P.w_off = pack(h->params[l1.w], Kt, 0, l1.K1, C); P.K = C; P.F = l1.K1;
      P.in = p.L[0].out; P.out = p.h0_out; P.prm = 0; P.plain = 1; P.gz = B.g_h0;
      break;
    }
    {   // through the block's first Linear into the layer that produced its main input; the skip columns ride along
      const LayerDesc& lp = a.layers[2 * (b - 1) + 1];
      TrainSquadBwdPhase& P = ta.P[np++];
      P = TrainSquadBwdPhase{};
      P.w_off = pack(h->params[l1.w], Kt, 0, l1.K1, C); P.K = C; P.F = l1.K1;
      P.in = p.L[2 * b].out; P.out = p.L[2 * (b - 1) + 1].out; P.prm = 2 * (b - 1) + 1;
      P.z = f.z2[b - 1]; P.stats = f.st2[b - 1]; P.gy = B.g_out[b - 1]; P.gz = B.g_z2[b - 1];
      P.accumulate = (b - 1 < a.n_enc) ? 1 : 0;
      P.drop_mode = 0;
      if (l1.K2 > 0) {
        const int skip_block = a.n_enc - 1 - (b - a.n_enc - 1);
        P.skip_w_off = pack(h->params[l1.w], Kt, l1.K1, l1.K2, C); P.skip_F = l1.K2; P.skip_out = B.g_out[skip_block];
      }
      (void)lp;
    }
  }
  ta.n_phases = np;
  ta.n_layers = p.n_layers;
  for (int li = 0; li < p.n_layers; ++li) {
    const LayerDesc& ld = a.layers[li];
    ta.gamma[li] = h->params[ld.gamma]; ta.beta[li] = h->params[ld.beta]; ta.width[li] = ld.N;
  }
  ta.wpk = wpk; ta.wpk_floats = woff;
  const int last = a.n_blocks - 1;
  ta.gz_top = B.g_z2[last]; ta.top_F = a.block_out[last]; ta.top_out = p.L[2 * last + 1].out;
  ta.n = (int)n;
  ta.act = gact_units; ta.act_stride = p.act_floats;
  ta.bar = bar_and_status; ta.status = bar_and_status + panels * 16;
  ta.loss_poison = loss_poison;
  ta.spin_budget = std::min<unsigned long long>(h->chain_spin_budget, 20000000ull);
  ta.keep_scale = (float)(1.0 / (1.0 - (double)h->cfg.dropout_p)); ta.p_drop = h->cfg.dropout_p;
  ta.seed = B.seed; ta.row_offset = B.row_offset; ta.step = 0;
  if (!h->sq_wpk_t_fresh) {      // else: packed by this step's forward launch (train_squad_forward), same list, same offsets
    hipLaunchKernelGGL(k_pack_fragments_multi_t, dim3(64, (unsigned)npk), dim3(256), 0, s, pm, wpk);
    OSD_HIP(hipGetLastError());
  }
  h->sq_wpk_t_fresh = false;
  hipLaunchKernelGGL(train_squad_bwd_kernel, dim3((unsigned)(panels * SQ_S)), dim3(SQ_THREADS), ts_lds_bytes(p.n_layers), s, ta);
  OSD_HIP(hipGetLastError());
  return OSD_OK;
}

void squad_chain_free(osd_handle* h) {
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2; ++i) {
    if (h->squad_wpk[i]) e = hipFree(h->squad_wpk[i]);
    h->squad_wpk[i] = nullptr; h->squad_wpk_floats[i] = 0; h->squad_wpk_valid[i] = false;
  }
  if (h->squad_args_dev) e = hipFree(h->squad_args_dev);
  (void)e;
  free(h->squad_args_host);
  h->squad_args_dev = nullptr; h->squad_args_host = nullptr; h->squad_args_cap = 0;
}

}  // namespace osd
