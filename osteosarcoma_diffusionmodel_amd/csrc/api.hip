// api.hip -- C ABI of libosdiff.so (include/osdiff.h): handle management, the denoiser
// forward pass, q_sample / p_sample / the hipGraph-replayed reverse chain.
// Training entry points live in train.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <new>
#include "handle.h"
#include "kernels.h"
#include "fwd.h"
#include "launch.h"
#include "split.h"

namespace osd {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

std::vector<KernelReg>& kernel_registry() {
  static std::vector<KernelReg> r;
  return r;
}
int persist_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("OSD_PERSIST");
    mode = e ? atoi(e) : 0;       // measured: with two chunks in flight the one-tile-per-workgroup grid is faster
  }
  return mode;
}
hipError_t prepare_kernels() {
  for (const KernelReg& k : kernel_registry()) {
    hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

int build_arch(const osd_config& c, Arch* a) {
  if (c.mutation_dim < 0 || c.expression_dim < 0 || c.pathway_dim < 0 || c.condition_dim <= 0) { set_error("bad feature dims"); return OSD_EINVAL; }
  a->D = c.mutation_dim + c.expression_dim + c.pathway_dim;
  if (a->D <= 0) { set_error("data_dim must be positive"); return OSD_EINVAL; }
  if (c.n_hidden < 1 || c.n_hidden > OSD_MAX_HIDDEN) { set_error("n_hidden must be in [1,%d]", OSD_MAX_HIDDEN); return OSD_EINVAL; }
  if (c.num_steps < 1) { set_error("num_steps must be >= 1"); return OSD_EINVAL; }
  if (c.time_dim < 4 || (c.time_dim & 1)) { set_error("time_dim (latent_dim) must be even and >= 4"); return OSD_EINVAL; }
  // models/diffusion.py:285 hard-wires the condition embedding width to 64 while cond_proj
  // expects latent_dim // 2 inputs (:292): only latent_dim 128/129 constructs a working model.
  if (c.time_dim / 2 != 64) { set_error("latent_dim // 2 must equal 64 (reference ConditionalEmbedding width)"); return OSD_EINVAL; }
  if (!(c.dropout_p >= 0.f && c.dropout_p < 1.f)) { set_error("dropout must be in [0,1)"); return OSD_EINVAL; }
  a->cond_dim = c.condition_dim;
  a->time_dim = c.time_dim;
  a->cond_width = 64;
  a->T = c.num_steps;
  a->hidden.assign(c.hidden_dims, c.hidden_dims + c.n_hidden);
  for (int hdim : a->hidden) {
    if (hdim <= 0 || hdim % 8) { set_error("hidden dims must be positive multiples of 8 (GroupNorm(8, C))"); return OSD_EINVAL; }
    if (!gn_width_supported(hdim / 8)) { set_error("hidden dim %d: group width %d is outside the fused kernels (power of two in [4,128])", hdim, hdim / 8); return OSD_EUNSUPPORTED; }
  }
  const int L = c.n_hidden;
  a->H0 = a->hidden[0];
  a->n_enc = L - 1;
  a->n_blocks = 2 * (L - 1) + 1;
  a->layers.clear();
  a->block_out.clear();
  ParamMap& pm = a->pm;
  pm.numel.clear();
  int idx = 0;
  auto P = [&](int64_t n) { pm.numel.push_back(n); return idx++; };
  pm.ce0_w = P((int64_t)64 * c.condition_dim); pm.ce0_b = P(64);
  pm.ce2_w = P(64 * 64); pm.ce2_b = P(64);
  pm.in_w = P((int64_t)a->H0 * a->D); pm.in_b = P(a->H0);
  pm.cp_w = P((int64_t)a->H0 * (c.time_dim / 2)); pm.cp_b = P(a->H0);
  pm.tp_w = P((int64_t)a->H0 * c.time_dim); pm.tp_b = P(a->H0);
  int bi = 0;
  auto block = [&](int k1, int k2, int n) {
    LayerDesc l1{k1, k2, n, 0, 0, 0, 0, n / 8, bi, 0};
    l1.w = P((int64_t)n * (k1 + k2)); l1.b = P(n); l1.gamma = P(n); l1.beta = P(n);
    LayerDesc l2{n, 0, n, 0, 0, 0, 0, n / 8, bi, 1};
    l2.w = P((int64_t)n * n); l2.b = P(n); l2.gamma = P(n); l2.beta = P(n);
    a->layers.push_back(l1); a->layers.push_back(l2);
    a->block_out.push_back(n);
    ++bi;
  };
  int cin = a->hidden[0];
  for (int i = 1; i < L; ++i) { block(cin, 0, a->hidden[i]); cin = a->hidden[i]; }
  block(cin, 0, cin);
  int cur = a->hidden[L - 1];
  for (int i = L - 2; i >= 0; --i) { block(cur, a->hidden[i + 1], a->hidden[i]); cur = a->hidden[i]; }
  pm.out_w = P((int64_t)a->D * cur); pm.out_b = P(a->D);
  pm.n_params = idx;
  int64_t per = 64 + 64 + 2 * (int64_t)a->H0;
  for (int n : a->block_out) per += 2 * (int64_t)n;
  a->act_floats_per_row = per;
  return OSD_OK;
}

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int ensure_arena(Slot* s, int64_t floats) {
  if (s->arena_floats >= floats) return OSD_OK;
  if (s->arena) { hipError_t e = hipFree(s->arena); (void)e; s->arena = nullptr; s->arena_floats = 0; }
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)floats * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %lld bytes failed", (long long)floats * 4); return OSD_ENOMEM; }
  s->arena = (float*)p;
  s->arena_floats = floats;
  return OSD_OK;
}

// Carve forward activations for n rows out of `base`; returns floats used.
int64_t carve_fwd(const Arch& a, float* base, int64_t n, bool train, FwdWs* ws) {
  int64_t off = 0;
  auto take = [&](int64_t floats) { float* p = base ? base + off : nullptr; off += align_up(floats, 64); return p; };
  ws->ce1 = take(n * 64); ws->ce2 = take(n * 64);
  ws->cproj = take(n * a.H0); ws->h0 = take(n * a.H0);
  ws->mid.resize(a.n_blocks); ws->out.resize(a.n_blocks);
  ws->z1.assign(a.n_blocks, nullptr); ws->z2.assign(a.n_blocks, nullptr);
  ws->st1.assign(a.n_blocks, nullptr); ws->st2.assign(a.n_blocks, nullptr);
  for (int b = 0; b < a.n_blocks; ++b) {
    const int64_t c = a.block_out[b];
    ws->mid[b] = take(n * c); ws->out[b] = take(n * c);
    if (train) {
      ws->z1[b] = take(n * c); ws->z2[b] = take(n * c);
      ws->st1[b] = take(n * 16); ws->st2[b] = take(n * 16);
    }
  }
  return off;
}

// ConditionalEmbedding + cond_proj (models/diffusion.py:101-105, 226): cproj[n][H0]
int run_cond(osd_handle* h, hipStream_t s, const float* cond, int64_t n, const FwdWs& ws) {
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  GemmArgs g{};
  g.A = h->params[pm.ce0_w]; g.lda = a.cond_dim; g.B0 = cond; g.ldb0 = a.cond_dim; g.K0 = a.cond_dim;
  g.F = 64; g.P = (int)n; g.K = a.cond_dim;
  OSD_HIP(launch_linear(s, g, true, true, h->params[pm.ce0_b], ws.ce1, 64, true, false));
  g.A = h->params[pm.ce2_w]; g.lda = 64; g.B0 = ws.ce1; g.ldb0 = 64; g.K0 = 64; g.K = 64;
  OSD_HIP(launch_linear(s, g, true, true, h->params[pm.ce2_b], ws.ce2, 64, false, false));
  g.A = h->params[pm.cp_w]; g.lda = 64; g.B0 = ws.ce2; g.ldb0 = 64; g.K0 = 64; g.K = 64; g.F = a.H0;
  OSD_HIP(launch_linear(s, g, true, true, h->params[pm.cp_b], ws.cproj, a.H0, false, false));
  return OSD_OK;
}

static int prof_mark(osd_handle* h, hipStream_t s) {
  if (h->prof_events) {
    if (h->prof_i >= (int)h->prof_events->size()) { set_error("profile event overflow"); return OSD_EINVAL; }
    OSD_HIP(hipEventRecord((*h->prof_events)[h->prof_i++], s));
  }
  return OSD_OK;
}

// input_proj + blocks (models/diffusion.py:229-251); result in ws.out[n_blocks-1].
int run_trunk(osd_handle* h, hipStream_t s, const FwdWs& ws, const TrunkIn& in) {
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  const int n = (int)in.n;
  {
    GemmArgs g{};
    const int kx = in.kx > 0 ? in.kx : a.D;
    g.A = h->w_in_packed; g.lda = h->w_in_ld; g.B0 = in.x; g.ldb0 = in.ldx; g.K0 = kx;
    g.F = a.H0; g.P = n; g.K = kx;
    if (in.a_unpacked) { g.A = h->params[pm.in_w]; g.lda = a.D; g.a_kmax = a.D; }
    g.ksplit = in.ksplit ? 1 : 0;
    EpiInput::Args ea{h->params[pm.in_b], h->d_temb, a.H0, in.t_index, in.t_dev, in.t_imm, ws.cproj, a.H0, ws.h0, a.H0};
    bool done = false;
    if (in.in_slices > 1 && in.in_slabs) {
      const hipError_t e = launch_input_splitk(s, g, ea, in.in_slabs, in.in_slices);
      if (e == hipSuccess) done = true;
      else if (e != hipErrorInvalidValue) OSD_HIP(e);
      else (void)hipGetLastError();
    }
    if (!done) {
      hipError_t e = launch_input(s, g, ea, true);
      if (e == hipErrorInvalidValue && in.a_unpacked) {
        // the clamped-weight path needs 16-byte aligned parameter pointers (launch.h: glds_ok / fast_ok), which the caller of
        // osd_load_weights does not owe us: pack the padded copy after all and read that (x already has zero pad columns)
        (void)hipGetLastError();
        OSD_HIP(launch_copy2d(s, h->params[pm.in_w], a.D, h->w_in_packed, h->w_in_ld, a.H0, a.D));
        g.A = h->w_in_packed; g.lda = h->w_in_ld; g.a_kmax = 0;
        e = launch_input(s, g, ea, true);
      }
      OSD_HIP(e);
    }
    OSD_TRY(prof_mark(h, s));
  }
  if (in.input_only) return OSD_OK;
  const float* cur = ws.h0;
  int cur_w = a.H0;
  for (int b = 0; b < a.n_blocks; ++b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    GemmArgs g{};
    g.A = h->params[l1.w]; g.lda = l1.K1 + l1.K2;
    g.B0 = cur; g.ldb0 = cur_w; g.K0 = l1.K1;
    if (l1.K2 > 0) {
      const int skip_block = a.n_enc - 1 - (b - a.n_enc - 1);   // LIFO: decoder j pops encoder n_enc-1-j
      g.B1 = ws.out[skip_block]; g.ldb1 = a.block_out[skip_block];
    }
    g.F = l1.N; g.P = n; g.K = l1.K1 + l1.K2; g.ksplit = in.ksplit ? 1 : 0;
    GnArgs ga{};
    ga.bias = h->params[l1.b]; ga.gamma = h->params[l1.gamma]; ga.beta = h->params[l1.beta];
    ga.out = ws.mid[b]; ga.ldo = l1.N;
    ga.z_out = in.save ? ws.z1[b] : nullptr; ga.ldz = l1.N; ga.stats = in.save ? ws.st1[b] : nullptr;
    const bool drop = in.train && h->cfg.dropout_p > 0.f;
    // small batches: a deep layer is a few dozen tiles of 16-32 sequential K steps -- K in slices over workgroups + a reduce kernel
    auto gn_split = [&](const GemmArgs& gg, const GnArgs& aa) -> int {      // 1 = launched, 0 = not applicable, < 0 = error
      if (in.gn_slices < 2 || !in.gn_slabs || in.save || gg.K < 512) return 0;
      const hipError_t e = launch_gn_silu_splitk(s, gg, aa, in.gn_slabs, in.gn_slices);
      if (e == hipSuccess) return 1;
      (void)hipGetLastError();
      if (e == hipErrorInvalidValue) return 0;
      set_error("launch_gn_silu_splitk failed: %s", hipGetErrorString(e));
      return OSD_EHIP;
    };
    if (drop) {
      ga.drop_mode = in.masks ? 1 : 2;
      ga.mask = in.masks ? in.masks[b] : nullptr; ga.ldm = l1.N;
      ga.keep_scale = (float)(1.0 / (1.0 - (double)h->cfg.dropout_p)); ga.p_drop = h->cfg.dropout_p;
      ga.seed = in.seed; ga.row_offset = in.row_offset; ga.step = in.drop_step; ga.tag = TAG_DROPOUT + (uint32_t)b;
      ga.step_dev = in.drop_step_dev;
      OSD_HIP(launch_gn_silu_drop(s, g, l1.gw, ga));
    } else {
      const int sp = gn_split(g, ga);
      if (sp < 0) return sp;
      if (!sp) OSD_HIP(launch_gn_silu(s, g, l1.gw, ga));
    }
    OSD_TRY(prof_mark(h, s));
    GemmArgs g2{};
    g2.A = h->params[l2.w]; g2.lda = l2.K1; g2.B0 = ws.mid[b]; g2.ldb0 = l1.N; g2.K0 = l2.K1;
    g2.F = l2.N; g2.P = n; g2.K = l2.K1; g2.ksplit = in.ksplit ? 1 : 0;
    GnArgs gb{};
    gb.bias = h->params[l2.b]; gb.gamma = h->params[l2.gamma]; gb.beta = h->params[l2.beta];
    gb.out = ws.out[b]; gb.ldo = l2.N;
    gb.z_out = in.save ? ws.z2[b] : nullptr; gb.ldz = l2.N; gb.stats = in.save ? ws.st2[b] : nullptr;
    {
      const int sp = gn_split(g2, gb);
      if (sp < 0) return sp;
      if (!sp) OSD_HIP(launch_gn_silu(s, g2, l2.gw, gb));
    }
    OSD_TRY(prof_mark(h, s));
    cur = ws.out[b];
    cur_w = l2.N;
  }
  return OSD_OK;
}

// Derived copies that follow the current parameters: the t_emb table time_proj(TimeEmbedding(t/T))
// (models/diffusion.py:222-223; all rows of a sampling step share t and training rows gather their
// t, so the Linear runs T times, not B) and the zero-padded input_proj.weight.
int refresh_derived(osd_handle* h, hipStream_t s, bool pack_in_w) {
  const Arch& a = h->arch;
  GemmArgs g{};
  g.A = h->params[a.pm.tp_w]; g.lda = a.time_dim; g.B0 = h->d_time_emb; g.ldb0 = a.time_dim; g.K0 = a.time_dim;
  g.F = a.H0; g.P = a.T; g.K = a.time_dim;
  OSD_HIP(launch_linear(s, g, true, true, h->params[a.pm.tp_b], h->d_temb, a.H0, false, false));
  h->panel_wpk_valid = false;           // the LDS-resident chain repacks its fragment-ordered copies before its next run
  h->squad_wpk_valid[0] = h->squad_wpk_valid[1] = false;      // ... and the squad chains theirs
  h->sq_wpk_t_fresh = false;
  h->split_valid = false;               // ... and the bf16x3 engine its weight planes
  if (!pack_in_w) {                     // a training step that reads input_proj.weight directly: the packed copies go stale and are
    h->w_packed_stale = true;           // refreshed by the next entry point that reads them (ensure_packed) or osd_load_weights
    return OSD_OK;
  }
  h->w_packed_stale = false;
  OSD_HIP(launch_copy2d(s, h->params[a.pm.in_w], a.D, h->w_in_packed, h->w_in_ld, a.H0, a.D));   // pad columns stay zero
  if (h->w_out_packed) {                       // D % 4 != 0: rows [D, Dp) stay zero
    const size_t hl = (size_t)a.block_out[a.n_blocks - 1];
    OSD_HIP(hipMemcpyAsync(h->w_out_packed, h->params[a.pm.out_w], (size_t)a.D * hl * 4, hipMemcpyDeviceToDevice, s));
    OSD_HIP(hipMemcpyAsync(h->b_out_packed, h->params[a.pm.out_b], (size_t)a.D * 4, hipMemcpyDeviceToDevice, s));
  }
  return OSD_OK;
}

// The packed input_proj / output_proj copies follow the parameters lazily: a training step that does not read them skips the
// repack (refresh_derived above), and a C-ABI caller may sample right after such a step without another osd_load_weights.
int ensure_packed(osd_handle* h, hipStream_t s) {
  if (!h->w_packed_stale) return OSD_OK;
  return refresh_derived(h, s, true);
}

// padded: the epilogue works on the padded chain state (Dp columns; the packed weight has zero rows for the pad columns)
GemmArgs output_proj_args(osd_handle* h, const FwdWs& ws, int64_t n, bool padded) {
  const Arch& a = h->arch;
  const int last = a.n_blocks - 1;
  GemmArgs g{};
  g.A = padded ? h->w_out_packed : h->params[a.pm.out_w]; g.lda = a.block_out[last];
  g.B0 = ws.out[last]; g.ldb0 = a.block_out[last]; g.K0 = a.block_out[last];
  g.F = padded ? h->Dp : a.D; g.P = (int)n; g.K = a.block_out[last];
  return g;
}

int check_ready(osd_handle* h) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  if (!h->have_schedule) { set_error("osd_set_schedule has not been called"); return OSD_ESTATE; }
  if (!h->have_weights) { set_error("osd_load_weights has not been called"); return OSD_ESTATE; }
  return OSD_OK;
}

int check_rows(int64_t n) {
  if (n < 0 || n > 0x7fffffff / 2) { set_error("row count %lld out of range", (long long)n); return OSD_EINVAL; }
  return OSD_OK;
}

// Global row ids address the Philox stream as 32-bit counters: a shard whose ids would wrap would repeat another
// shard's draws, so it is rejected instead of truncated.
int check_row_offset(int64_t row_offset, int64_t n) {
  if (row_offset < 0 || row_offset + n > ((int64_t)1 << 32)) {
    set_error("row_offset %lld + %lld rows is outside the 32-bit global row id space [0, 2^32)", (long long)row_offset, (long long)n);
    return OSD_EINVAL;
  }
  return OSD_OK;
}

// Caller-supplied per-row timestep indices, clamped into [0, T) in a handle-owned buffer (see k_clamp_int).
int sanitize_t(osd_handle* h, hipStream_t s, const int32_t* t_index, int64_t n, const int** out) {
  if (!t_index) { *out = nullptr; return OSD_OK; }
  if (h->t_san_cap < n) {
    if (h->t_san) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(h->t_san)); h->t_san = nullptr; h->t_san_cap = 0; }
    const int64_t cap = (n + 1023) / 1024 * 1024;
    if (hipMalloc((void**)&h->t_san, (size_t)cap * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %lld bytes failed", (long long)cap * 4); return OSD_ENOMEM; }
    h->t_san_cap = cap;
  }
  OSD_HIP(launch_clamp_int(s, t_index, n, 0, h->arch.T - 1, h->t_san));
  *out = h->t_san;
  return OSD_OK;
}

}  // namespace osd

using namespace osd;

extern "C" {

int osd_version(void) { return OSD_VERSION; }
const char* osd_last_error(void) { return g_err; }

int osd_num_params(const osd_config* cfg) {
  if (!cfg) return 0;
  Arch a;
  if (build_arch(*cfg, &a) != OSD_OK) return 0;
  return a.pm.n_params;
}

int64_t osd_param_numel(const osd_config* cfg, int i) {
  if (!cfg) return -1;
  Arch a;
  if (build_arch(*cfg, &a) != OSD_OK) return -1;
  if (i < 0 || i >= a.pm.n_params) return -1;
  return a.pm.numel[i];
}

static int create_device_state(osd_handle* h) {
  const Arch& a = h->arch;
  const int T = a.T;
  OSD_HIP(hipMalloc((void**)&h->d_sqrt_ac, (size_t)T * 4));
  OSD_HIP(hipMalloc((void**)&h->d_sqrt_1m, (size_t)T * 4));
  OSD_HIP(hipMalloc((void**)&h->d_coef, (size_t)T * 4 * 4));
  const size_t t_rows = (size_t)(T + 31) / 32 * 32;     // zero rows up to whole K steps of the grouped weight-gradient kernel (time_proj.weight)
  OSD_HIP(hipMalloc((void**)&h->d_time_emb, t_rows * a.time_dim * 4));
  OSD_HIP(hipMemset(h->d_time_emb, 0, t_rows * a.time_dim * 4));
  OSD_HIP(hipMalloc((void**)&h->d_temb, (size_t)T * a.H0 * 4));
  h->w_in_ld = (a.D + BK - 1) / BK * BK;
  OSD_HIP(hipMalloc((void**)&h->w_in_packed, (size_t)a.H0 * h->w_in_ld * 4));
  OSD_HIP(hipMemset(h->w_in_packed, 0, (size_t)a.H0 * h->w_in_ld * 4));
  h->Dp = (a.D + 3) / 4 * 4;
  if (h->Dp != a.D) {
    const size_t hl = (size_t)a.block_out[a.n_blocks - 1];
    OSD_HIP(hipMalloc((void**)&h->w_out_packed, (size_t)h->Dp * hl * 4));
    OSD_HIP(hipMemset(h->w_out_packed, 0, (size_t)h->Dp * hl * 4));
    OSD_HIP(hipMalloc((void**)&h->b_out_packed, (size_t)h->Dp * 4));
    OSD_HIP(hipMemset(h->b_out_packed, 0, (size_t)h->Dp * 4));
  }
  OSD_HIP(hipMalloc((void**)&h->main.t_dev, 64));
  OSD_HIP(hipMalloc((void**)&h->loss_dev, 64));
  OSD_HIP(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  return OSD_OK;
}

int osd_create(const osd_config* cfg, osd_handle** out) {
  if (!cfg || !out) { set_error("null argument"); return OSD_EINVAL; }
  *out = nullptr;
  Arch a;
  OSD_TRY(build_arch(*cfg, &a));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available: libosdiff has no CPU fallback"); return OSD_EHIP; }
  if (cfg->device < 0 || cfg->device >= ndev) { set_error("device %d out of range (%d devices)", cfg->device, ndev); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(cfg->device));
  OSD_HIP(prepare_kernels());
  osd_handle* h = new (std::nothrow) osd_handle();
  if (!h) { set_error("out of host memory"); return OSD_ENOMEM; }
  h->cfg = *cfg;
  h->arch = a;
  if (const char* e = getenv("OSD_GROUPED_WGRAD")) h->grouped_wgrad = atoi(e) != 0;      // A/B knobs, see osd_set_option
  if (const char* e = getenv("OSD_WGRAD_MID_FLUSH")) h->wgrad_mid_flush = atoi(e) != 0;
  if (const char* e = getenv("OSD_FUSED_GN_BWD")) h->fused_gn_bwd = atoi(e) != 0;
  if (const char* e = getenv("OSD_TRAIN_INPUT_SPLITK")) h->train_input_splitk = atoi(e);
  if (const char* e = getenv("OSD_DUAL_DGRAD")) h->dual_dgrad = atoi(e) != 0;
  if (const char* e = getenv("OSD_TRAIN_KSPLIT")) h->train_ksplit = atoi(e) != 0;
  if (const char* e = getenv("OSD_COND_BWD_FUSED")) h->cond_bwd_fused = atoi(e) != 0;
  if (const char* e = getenv("OSD_TWO_STREAM_BWD")) h->two_stream_bwd = atoi(e) != 0;
  const int rc = create_device_state(h);
  if (rc != OSD_OK) {                 // nothing of a half-built handle survives (osd_destroy frees what was allocated)
    osd_destroy(h);
    (void)hipGetLastError();          // the failed call's sticky error must not surface in a later launch check
    return rc;
  }
  *out = h;
  return OSD_OK;
}

static void free_slot(Slot& s, bool own_stream) {
  hipError_t e;
  if (s.exec) { e = hipGraphExecDestroy(s.exec); }
  if (s.graph) { e = hipGraphDestroy(s.graph); }
  if (s.arena) { e = hipFree(s.arena); }
  if (s.t_dev) { e = hipFree(s.t_dev); }
  if (s.done) { e = hipEventDestroy(s.done); }
  if (own_stream && s.stream) { e = hipStreamDestroy(s.stream); }
  (void)e;
  s = Slot();
}

int osd_destroy(osd_handle* h) {
  if (!h) return OSD_OK;
  hipError_t e = hipSetDevice(h->cfg.device);
  e = hipDeviceSynchronize();
  for (auto& s : h->slots) free_slot(s, true);
  free_slot(h->main, false);
  float* bufs[] = {h->w_in_packed, h->w_out_packed, h->b_out_packed, h->chain_xpad, h->d_sqrt_ac, h->d_sqrt_1m, h->d_coef, h->d_time_emb, h->d_temb, h->train_arena, h->loss_dev};
  for (float* p : bufs) if (p) e = hipFree(p);
  if (h->normsq_dev) e = hipFree(h->normsq_dev);
  if (h->parts_dev) e = hipFree(h->parts_dev);
  if (h->t_san) e = hipFree(h->t_san);
  for (hipEvent_t ev : h->ev_pool) e = hipEventDestroy(ev);
  if (h->wgrad_stream) e = hipStreamDestroy(h->wgrad_stream);
  cons_free_plan(&h->cons);
  chain_free(h);
  split_free(h);
  wgrad_group_free(h);
  if (h->fork_ev) e = hipEventDestroy(h->fork_ev);
  (void)e;
  delete h;
  return OSD_OK;
}

int osd_set_stream(osd_handle* h, void* hip_stream) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  h->stream = (hipStream_t)hip_stream;
  return OSD_OK;
}

int osd_set_option(osd_handle* h, const char* name, int64_t value) {
  if (!h || !name) { set_error("null argument"); return OSD_EINVAL; }
  if (!strcmp(name, "chunk_rows")) {
    if (value < 1) { set_error("chunk_rows must be >= 1"); return OSD_EINVAL; }
    h->chunk_rows = value;
    return OSD_OK;
  }
  if (!strcmp(name, "n_streams")) {
    if (value < 1 || value > 8) { set_error("n_streams must be in [1,8]"); return OSD_EINVAL; }
    h->n_streams = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "sampler")) {                 // 0 auto, 1 persistent chain kernel where supported, 2 per-layer kernels
    if (value < 0 || value > 2) { set_error("sampler must be 0 (auto), 1 (chain kernel) or 2 (per-layer kernels)"); return OSD_EINVAL; }
    h->sampler = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "train_squad")) {             // from 2 048 rows on: 2 (default) the trunk of a training forward pass AND the dgrad chain each as one launch of squads (train_squad.h, train_squad_bwd.h; the backward one in single-process steps), 1 the forward only, 0 per-layer launches
    if (value < 0 || value > 2) { set_error("train_squad must be 0 (per-layer launches), 1 (forward trunk as squads) or 2 (forward and the dgrad chain)"); return OSD_EINVAL; }
    h->train_squad = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "squad_panel")) {             // the squad chain's panel: 0 auto, 16 (chain_squad16.h) wherever its squads fit two per CU, 32 (chain_squad.h)
    if (value != 0 && value != 16 && value != 32) { set_error("squad_panel must be 0 (auto), 16 or 32"); return OSD_EINVAL; }
    h->squad_panel = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_variant")) {           // 0 auto, 1 workspace chain (chain.h), 2 LDS-resident chain (chain_panel.h) where the architecture fits
    if (value < 0 || value > 3) { set_error("chain_variant must be 0 (auto), 1 (workspace chain), 2 (LDS-resident chain) or 3 (squad chain)"); return OSD_EINVAL; }
    h->chain_variant = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "precision")) {               // 0 fp32 MFMA (default), 1 bf16x3 split: fp32 accuracy on the bf16 matrix pipe (gemm_bf3.h)
    if (value < 0 || value > 1) { set_error("precision must be 0 (fp32) or 1 (bf16x3 split)"); return OSD_EINVAL; }
    if (value == 1 && !split_supported(h->arch)) { set_error("precision 1 (bf16x3 split) covers trunks of width 256 / 512 only"); return OSD_EUNSUPPORTED; }
    h->precision = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_grid")) {
    if (value < 0 || value > 65536) { set_error("chain_grid must be in [0,65536]"); return OSD_EINVAL; }
    h->chain_grid = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "input_splitk")) {            // per-layer sampling engine: 0 off (default), -1 auto (batches with < 128 input_proj tiles), n slices
    if (value < -1 || value > 64) { set_error("input_splitk must be in [-1,64]"); return OSD_EINVAL; }
    h->input_splitk = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "train_ksplit")) {            // 1: the training forward's GEMMs use two wave groups per workgroup (gemm_glds.h, NG = 2)
    if (value < 0 || value > 1) { set_error("train_ksplit must be 0 or 1"); return OSD_EINVAL; }
    h->train_ksplit = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "dual_dgrad")) {
    if (value < 0 || value > 1) { set_error("dual_dgrad must be 0 or 1"); return OSD_EINVAL; }
    h->dual_dgrad = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "cond_bwd_fused")) {
    if (value < 0 || value > 1) { set_error("cond_bwd_fused must be 0 or 1"); return OSD_EINVAL; }
    h->cond_bwd_fused = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "train_input_splitk")) {
    if (value < 0 || value > 16) { set_error("train_input_splitk must be in [0,16]"); return OSD_EINVAL; }
    h->train_input_splitk = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_spin_budget")) {       // s_memrealtime ticks (100 MHz) a dependency wait inside the chain kernel may take
    if (value < 0) { set_error("chain_spin_budget must be >= 0"); return OSD_EINVAL; }
    h->chain_spin_budget = (unsigned long long)value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_wall_budget_ms")) {    // host-side budget of a synchronous chain; 0 = 10 x the estimated run time + 2 s
    if (value < 0) { set_error("chain_wall_budget_ms must be >= 0"); return OSD_EINVAL; }
    h->chain_wall_budget_ms = value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_steps_per_launch")) {
    if (value < 0) { set_error("chain_steps_per_launch must be >= 0"); return OSD_EINVAL; }
    h->chain_steps_per_launch = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "chain_stagger")) {
    if (value < 0 || value > 100000000) { set_error("chain_stagger must be in [0,1e8] cycles"); return OSD_EINVAL; }
    h->chain_stagger = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "grouped_wgrad")) {           // 1 (default): the weight gradients of a backward pass in two grouped launches
    if (value < 0 || value > 1) { set_error("grouped_wgrad must be 0 or 1"); return OSD_EINVAL; }
    h->grouped_wgrad = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "fused_gn_bwd")) {
    if (value < 0 || value > 1) { set_error("fused_gn_bwd must be 0 or 1"); return OSD_EINVAL; }
    h->fused_gn_bwd = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "wgrad_mid_flush")) {         // 1: the decoder + bottleneck weight gradients are launched (one workgroup per CU)
    if (value < 0 || value > 1) { set_error("wgrad_mid_flush must be 0 or 1"); return OSD_EINVAL; }    // while the encoder half of backward runs
    h->wgrad_mid_flush = (int)value;
    return OSD_OK;
  }
  if (!strcmp(name, "train_streams")) {
    if (value < 1 || value > 2) { set_error("train_streams must be 1 or 2"); return OSD_EINVAL; }
    h->two_stream_bwd = value == 2;
    return OSD_OK;
  }
  set_error("unknown option '%s'", name);
  return OSD_EINVAL;
}

int osd_get_option(osd_handle* h, const char* name, int64_t* value) {
  if (!h || !name || !value) { set_error("null argument"); return OSD_EINVAL; }
  struct { const char* n; int64_t v; } tab[] = {
      {"chunk_rows", h->chunk_rows}, {"n_streams", h->n_streams}, {"sampler", h->sampler}, {"chain_grid", h->chain_grid},
      {"chain_steps_per_launch", h->chain_steps_per_launch}, {"chain_stagger", h->chain_stagger},
      {"chain_spin_budget", (int64_t)h->chain_spin_budget}, {"chain_wall_budget_ms", h->chain_wall_budget_ms},
      {"grouped_wgrad", h->grouped_wgrad}, {"fused_gn_bwd", h->fused_gn_bwd}, {"wgrad_mid_flush", h->wgrad_mid_flush},
      {"input_splitk", h->input_splitk}, {"train_streams", h->two_stream_bwd ? 2 : 1}, 
      // read-only counters
      {"precision", h->precision}, {"last_precision", h->last_precision}, {"split_supported", split_supported(h->arch) ? 1 : 0},
      {"chain_fallbacks", h->chain_fallbacks}, {"last_engine", h->last_engine},
      {"chain_variant", h->chain_variant}, {"last_chain_variant", h->last_chain_variant}, {"panel_chain_supported", panel_chain_supported(h) ? 1 : 0},
      {"squad_chain_supported", squad_chain_supported(h) ? 1 : 0}, {"last_squad_panel", h->last_squad_rp}, {"squad_panel", h->squad_panel}, {"train_squad", h->train_squad}, {"cond_bwd_fused", h->cond_bwd_fused}};
  for (const auto& e : tab)
    if (!strcmp(name, e.n)) { *value = e.v; return OSD_OK; }
  set_error("unknown option '%s'", name);
  return OSD_EINVAL;
}

int osd_set_schedule(osd_handle* h, const float* sqrt_ac, const float* sqrt_1m_ac, const float* post_coef, const float* time_emb) {
  if (!h || !sqrt_ac || !sqrt_1m_ac || !post_coef || !time_emb) { set_error("null argument"); return OSD_EINVAL; }
  const Arch& a = h->arch;
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_HIP(hipMemcpy(h->d_sqrt_ac, sqrt_ac, (size_t)a.T * 4, hipMemcpyHostToDevice));
  OSD_HIP(hipMemcpy(h->d_sqrt_1m, sqrt_1m_ac, (size_t)a.T * 4, hipMemcpyHostToDevice));
  {
    // fold the reference's six per-step scalars into x' = A*x + B*eps + C*z (see EpiPosterior)
    std::vector<float> abc((size_t)a.T * 4, 0.f);
    for (int t = 0; t < a.T; ++t) {
      const double c0 = post_coef[6 * t], c1 = post_coef[6 * t + 1], c2 = post_coef[6 * t + 2], c3 = post_coef[6 * t + 3],
                   c4 = post_coef[6 * t + 4], c5 = post_coef[6 * t + 5];
      if (t > 0) {
        abc[4 * t] = (float)(c4 / c3 + c2 / (c1 * c3));
        abc[4 * t + 1] = (float)(-c0 * c2 / (c1 * c3));
        abc[4 * t + 2] = (float)c5;
      } else {
        abc[0] = (float)(1.0 / c1);
        abc[1] = (float)(-c0 / c1);
        abc[2] = 0.f;
      }
    }
    OSD_HIP(hipMemcpy(h->d_coef, abc.data(), abc.size() * 4, hipMemcpyHostToDevice));
  }
  OSD_HIP(hipMemcpy(h->d_time_emb, time_emb, (size_t)a.T * a.time_dim * 4, hipMemcpyHostToDevice));
  h->have_schedule = true;
  return OSD_OK;
}

int osd_load_weights(osd_handle* h, const float* const* params, int n) {
  if (!h || !params) { set_error("null argument"); return OSD_EINVAL; }
  const Arch& a = h->arch;
  if (n != a.pm.n_params) { set_error("expected %d parameter tensors, got %d", a.pm.n_params, n); return OSD_EINVAL; }
  if (!h->have_schedule) { set_error("osd_set_schedule must precede osd_load_weights"); return OSD_ESTATE; }
  for (int i = 0; i < n; ++i)
    if (!params[i]) { set_error("parameter %d is null", i); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  h->params.assign(params, params + n);
  OSD_TRY(refresh_derived(h, h->stream));
  h->have_weights = true;
  return OSD_OK;
}

int osd_denoiser_forward(osd_handle* h, const float* x, const int32_t* t_index, int32_t t_all, const float* cond, int64_t n,
                         float* eps, int flags, const float* const* masks, uint64_t seed) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!x || !cond || !eps) { set_error("null tensor"); return OSD_EINVAL; }
  const Arch& a = h->arch;
  if (!t_index && (t_all < 0 || t_all >= a.T)) { set_error("t=%d outside [0,%d)", t_all, a.T); return OSD_EINVAL; }
  if (n == 0) return OSD_OK;
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  OSD_TRY(ensure_packed(h, s));
  const int* t_idx = nullptr;
  OSD_TRY(sanitize_t(h, s, t_index, n, &t_idx));
  if (h->precision == 1 && !(flags & OSD_F_TRAIN_MODE) && !masks) {      // eval mode on the bf16 matrix pipe; dropout stays fp32
    OSD_TRY(split_denoiser_forward(h, x, t_idx, t_all, cond, n, eps));
    if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
    return OSD_OK;
  }
  h->last_precision = 0;
  FwdWs ws;
  const int64_t need = carve_fwd(a, nullptr, n, false, &ws);
  OSD_TRY(ensure_arena(&h->main, need));
  carve_fwd(a, h->main.arena, n, false, &ws);
  OSD_TRY(run_cond(h, s, cond, n, ws));
  TrunkIn in{};
  in.x = x; in.ldx = a.D; in.n = n; in.t_index = t_idx; in.t_imm = t_all;
  in.train = (flags & OSD_F_TRAIN_MODE) != 0; in.masks = masks; in.seed = seed;
  OSD_TRY(run_trunk(h, s, ws, in));
  GemmArgs g = output_proj_args(h, ws, n);
  OSD_HIP(launch_linear(s, g, true, true, h->params[a.pm.out_b], eps, a.D, false, false));
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_q_sample(osd_handle* h, const float* x0, const int32_t* t_index, const float* noise_in, int64_t n, uint64_t seed,
                 int64_t row_offset, float* x_t, float* noise_out) {
  if (!h || !h->have_schedule) { set_error("schedule not set"); return OSD_ESTATE; }
  OSD_TRY(check_rows(n));
  if (!x0 || !t_index || !x_t) { set_error("null tensor"); return OSD_EINVAL; }
  if (!noise_in && !noise_out) { set_error("noise_out is required when noise is generated"); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  OSD_HIP(hipSetDevice(h->cfg.device));
  const int* t_idx = nullptr;
  OSD_TRY(sanitize_t(h, h->stream, t_index, n, &t_idx));
  OSD_HIP(launch_q_sample(h->stream, x0, t_idx, h->d_sqrt_ac, h->d_sqrt_1m, noise_in, n, h->arch.D, seed, (uint32_t)row_offset, x_t, noise_out));
  return OSD_OK;
}

int osd_p_sample_step(osd_handle* h, const float* x_t, int32_t t, const float* cond, const float* z, int64_t n, uint64_t seed,
                      int64_t row_offset, float* x_out, int flags) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!x_t || !cond || !x_out) { set_error("null tensor"); return OSD_EINVAL; }
  const Arch& a = h->arch;
  if (t < 0 || t >= a.T) { set_error("t=%d outside [0,%d)", t, a.T); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  if (n == 0) return OSD_OK;
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_TRY(ensure_packed(h, h->stream));
  if (h->precision == 1 && !((flags & OSD_F_TRAIN_MODE) && h->cfg.dropout_p > 0.f)) {
    OSD_TRY(split_p_sample_step(h, x_t, t, cond, z, n, seed, row_offset, x_out));
    if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(h->stream));
    return OSD_OK;
  }
  h->last_precision = 0;
  FwdWs ws;
  const int64_t need = carve_fwd(a, nullptr, n, false, &ws);
  OSD_TRY(ensure_arena(&h->main, need));
  carve_fwd(a, h->main.arena, n, false, &ws);
  hipStream_t s = h->stream;
  OSD_TRY(run_cond(h, s, cond, n, ws));      // recomputed every step, as models/diffusion.py:395 does
  TrunkIn in{};
  in.x = x_t; in.ldx = a.D; in.n = n; in.t_imm = t;
  in.train = (flags & OSD_F_TRAIN_MODE) != 0; in.seed = seed; in.row_offset = (uint32_t)row_offset; in.drop_step = (uint32_t)t;
  OSD_TRY(run_trunk(h, s, ws, in));
  GemmArgs g = output_proj_args(h, ws, n);
  EpiPosterior::Args ea{};
  ea.bias = h->params[a.pm.out_b]; ea.xin = x_t; ea.ldx = a.D; ea.xout = x_out; ea.ldo = a.D; ea.coef = h->d_coef;
  ea.t_dev = nullptr; ea.t_imm = t; ea.z = z; ea.ldzz = a.D; ea.z_step_stride = 0; ea.t_first = t;
  ea.seed = seed; ea.row_offset = (uint32_t)row_offset; ea.mut_mask = nullptr; ea.mutation_dim = 0;
  OSD_HIP(launch_posterior(s, g, ea));
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

// Drop the slot's previous graph once everything replayed from it has finished.
static int release_graph(Slot& sl) {
  if (!sl.exec && !sl.graph) return OSD_OK;
  OSD_HIP(hipStreamSynchronize(sl.stream));
  if (sl.exec) OSD_HIP(hipGraphExecDestroy(sl.exec));
  if (sl.graph) OSD_HIP(hipGraphDestroy(sl.graph));
  sl.exec = nullptr;
  sl.graph = nullptr;
  return OSD_OK;
}

// One chunk of the reverse chain on one slot: rows [r0, r0+m).
static int chain_chunk(osd_handle* h, Slot& sl, const float* cond, int64_t n_total, int64_t r0, int64_t m, const float* x_T,
                       const float* noises, uint64_t seed, int64_t row_offset, float* x_out, float* mut_mask_out, int flags) {
  const Arch& a = h->arch;
  const int D = a.D, T = a.T;
  hipStream_t s = sl.stream;
  OSD_TRY(release_graph(sl));
  FwdWs ws;
  const int64_t need = carve_fwd(a, nullptr, m, false, &ws);
  // D % 4 != 0 with device-generated draws: the state of the chunk lives in a padded buffer behind the activations (rows of Dp
  // floats, pad columns zero at the start) and is copied to the caller's rows at the end; injected draws ([T-1][n][D], rows not
  // 16-byte aligned) keep the guarded kernels on the caller's tensor
  const bool padded = h->w_out_packed != nullptr && !noises;
  const int ldx = padded ? h->Dp : D;
  const int64_t need_pad = (need + 63) / 64 * 64;
  // small batches: input_proj split-K (k_fused.hip) -- few output tiles, each a long sequential K loop
  int in_slices = 0;
  static const int64_t splitk_target = [] { const char* e = getenv("OSD_INPUT_SPLITK_TARGET"); return e ? atol(e) : 768L; }();
  {
    const int64_t tiles = (int64_t)((a.H0 + 63) / 64) * ((m + 63) / 64);
    if (h->input_splitk > 0 && !h->splitk_suspended) in_slices = h->input_splitk;
    else if (h->input_splitk < 0 && !h->splitk_suspended && ldx >= 1024 && tiles < 384)      // fewer tiles than 1.5 per CU
      in_slices = (int)std::min<int64_t>(16, std::max<int64_t>(2, (splitk_target + tiles / 2) / tiles));
    in_slices = std::min(in_slices, ldx / 128);
    if (in_slices < 2) in_slices = 0;
  }
  const int64_t x_floats = padded ? (m * (int64_t)ldx + 63) / 64 * 64 : 0;
  // ... and the deep Linear+GroupNorm layers likewise (same switch: the small-batch mode): slices so that a 512-wide layer has ~640
  // workgroups (64 x 64 tiles)
  int gn_slices = 0, max_c = a.H0;
  for (int c : a.block_out) max_c = std::max(max_c, c);
  if (in_slices > 0 && !(flags & OSD_F_TRAIN_MODE)) {
    const int64_t tiles = (int64_t)((max_c + 63) / 64) * ((m + 63) / 64);
    gn_slices = (int)std::min<int64_t>(4, (640 + tiles / 2) / tiles);
    if (gn_slices < 4) gn_slices = 0;      // measured (dims 62 / 5054 / 26): 999 rows 257 -> 238 us per step with 4 slices; 3000 rows 351 -> 385 us with 2
  }
  const int64_t slab_floats = std::max<int64_t>((int64_t)in_slices * m * a.H0, gn_slices ? (int64_t)(gn_slices + 1) * m * max_c : 0);
  OSD_TRY(ensure_arena(&sl, need_pad + x_floats + slab_floats));
  carve_fwd(a, sl.arena, m, false, &ws);
  float* x = padded ? sl.arena + need_pad : x_out + r0 * D;       // else the chain state lives in the output rows
  float* in_slabs = in_slices ? sl.arena + need_pad + x_floats : nullptr;
  if (padded) OSD_HIP(hipMemsetAsync(x, 0, (size_t)m * ldx * 4, s));
  const uint32_t roff = (uint32_t)(row_offset + r0);
  const bool train = (flags & OSD_F_TRAIN_MODE) != 0;
  // conditioning is loop-invariant in eval mode (no dropout inside the embedding MLP): hoisted
  OSD_TRY(run_cond(h, s, cond + r0 * a.cond_dim, m, ws));
  if (x_T) OSD_HIP(launch_copy2d(s, x_T + r0 * D, D, x, ldx, m, D));
  else OSD_HIP(launch_fill_randn(s, x, ldx, m, D, seed, roff, (uint32_t)T, TAG_POSTERIOR));
  OSD_HIP(launch_set_int(s, sl.t_dev, T - 1));

  auto enqueue_step = [&](void) -> int {
    TrunkIn in{};
    in.x = x; in.ldx = ldx; in.kx = ldx; in.n = m; in.t_dev = sl.t_dev; in.in_slabs = in_slabs; in.in_slices = in_slices;
    in.gn_slabs = in_slabs; in.gn_slices = gn_slices;
    in.ksplit = in_slices > 1;             // the small-batch mode already trades bit-equality with the chain kernel for latency: long-K layers on two wave groups
    in.train = train; in.seed = seed; in.row_offset = roff; in.drop_step_dev = sl.t_dev;
    OSD_TRY(run_trunk(h, s, ws, in));
    GemmArgs g = output_proj_args(h, ws, m, padded);
    EpiPosterior::Args ea{};
    ea.bias = padded ? h->b_out_packed : h->params[a.pm.out_b]; ea.xin = x; ea.ldx = ldx; ea.xout = x; ea.ldo = ldx; ea.coef = h->d_coef;
    ea.t_dev = sl.t_dev; ea.t_imm = 0;
    ea.z = noises ? noises + r0 * D : nullptr; ea.ldzz = D; ea.z_step_stride = (long long)n_total * D; ea.t_first = T - 1;
    ea.seed = seed; ea.row_offset = roff;
    ea.mut_mask = mut_mask_out ? mut_mask_out + r0 * h->cfg.mutation_dim : nullptr; ea.mutation_dim = h->cfg.mutation_dim;
    OSD_HIP(launch_posterior(s, g, ea));
    OSD_HIP(launch_add_int(s, sl.t_dev, -1));
    return OSD_OK;
  };

  if (flags & OSD_F_GRAPH) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    OSD_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step();
    hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc != OSD_OK) { if (graph) { hipError_t e = hipGraphDestroy(graph); (void)e; } return rc; }
    OSD_HIP(ce);
    OSD_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    // the exec object must outlive its launches: the slot keeps it until its next use
    sl.graph = graph;
    sl.exec = exec;
    for (int it = 0; it < T; ++it) OSD_HIP(hipGraphLaunch(exec, s));
  } else {
    for (int it = 0; it < T; ++it) OSD_TRY(enqueue_step());
  }
  if (padded) OSD_HIP(launch_copy2d(s, x, ldx, x_out + r0 * D, D, m, D));
  return OSD_OK;
}

int osd_sample_chain(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed,
                     int64_t row_offset, float* x_out, float* mut_mask_out, int flags) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!cond || !x_out) { set_error("null tensor"); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  if (n == 0) return OSD_OK;
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_TRY(chain_check_status(h));            // a previous chain-kernel run that gave up is reported here at the latest
  OSD_TRY(ensure_packed(h, h->stream));
  // bf16x3 split precision: eval-mode chains on the per-layer launches of split.hip (dropout inside the chain stays fp32)
  const bool split = h->precision == 1 && !((flags & OSD_F_TRAIN_MODE) && h->cfg.dropout_p > 0.f);
  h->last_precision = split ? 1 : 0;
  if (split) OSD_TRY(split_prepare(h, h->stream));
  h->last_engine = split ? 0 : chain_pick_engine(h, n, flags);
  if (h->last_engine == 1 && noises && h->w_out_packed && !chain_uses_squad(h, n)) h->last_engine = 0;      // injected draws at D % 4 != 0: guarded per-layer kernels (the squad chain reads any layout)
  bool fell_back = false;
  if (h->last_engine == 1) {
    OSD_TRY(chain_run(h, cond, n, x_T, noises, seed, row_offset, x_out, mut_mask_out));
    if (!(flags & OSD_F_SYNC)) return OSD_OK;       // asynchronous: a chain that gives up is reported by the next call on this handle
    int gave_up = 0;
    OSD_TRY(chain_finish(h, &gave_up));
    if (!gave_up) return OSD_OK;
    // models/diffusion.py:427-449 cannot fail: the chain is re-run on the per-layer kernels, which compute the same bits from
    // the same x_T / seed (the chain state lives in x_out, so an aliased x_T is gone)
    if (x_T == x_out) {
      set_error("the reverse-chain kernel gave up and x_T aliases x_out: nothing left to re-run the chain from");
      return OSD_EHIP;
    }
    ++h->chain_fallbacks;
    h->last_engine = 0;
    fell_back = true;
    h->splitk_suspended = true;          // the re-run must produce the chain kernel's bits: single-pass input_proj
  }
  // equal chunks (rounded up to whole 128-row tiles) of at most chunk_rows rows
  int64_t n_chunks = (n + h->chunk_rows - 1) / h->chunk_rows;
  int64_t chunk = ((n + n_chunks - 1) / n_chunks + 127) / 128 * 128;
  if (chunk > n) chunk = n;
  n_chunks = (n + chunk - 1) / chunk;
  const int n_slots = (int)std::min<int64_t>(h->n_streams, n_chunks);
  while ((int)h->slots.size() < n_slots) {
    Slot sl;
    OSD_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    OSD_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    OSD_HIP(hipMalloc((void**)&sl.t_dev, 64));
    h->slots.push_back(sl);
  }
  // fork: slot streams wait for everything already queued on the caller's stream
  OSD_HIP(hipEventRecord(h->fork_ev, h->stream));
  for (int i = 0; i < n_slots; ++i) OSD_HIP(hipStreamWaitEvent(h->slots[i].stream, h->fork_ev, 0));
  int rc = OSD_OK;
  for (int64_t c = 0; c < n_chunks && rc == OSD_OK; ++c) {
    const int64_t r0 = c * chunk;
    const int64_t m = std::min<int64_t>(chunk, n - r0);
    if (split) rc = split_chain_chunk(h, h->slots[c % n_slots], cond, n, r0, m, x_T, noises, seed, row_offset, x_out, mut_mask_out, flags);
    else rc = chain_chunk(h, h->slots[c % n_slots], cond, n, r0, m, x_T, noises, seed, row_offset, x_out, mut_mask_out, flags);
  }
  // join
  for (int i = 0; i < n_slots; ++i) {
    hipError_t e = hipEventRecord(h->slots[i].done, h->slots[i].stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, h->slots[i].done, 0);
    if (e != hipSuccess && rc == OSD_OK) { set_error("join failed: %s", hipGetErrorString(e)); rc = OSD_EHIP; }
  }
  h->splitk_suspended = false;
  if (rc != OSD_OK) return rc;
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(h->stream));
  if (fell_back)       // a warning, not an error: osd_last_error() tells what happened, osd_get_option("chain_fallbacks") counts
    set_error("warning: the reverse-chain kernel gave up in a dependency wait; the chain was re-run on the per-layer kernels (%s)",
              h->last_chain_variant == 3 ? "results agree with the squad chain's to fp32 rounding" : "same results");
  return OSD_OK;
}

int osd_sample_engine(osd_handle* h, int64_t n, int flags) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  if (n < 0) return h->last_engine;
  if (h->precision == 1 && !((flags & OSD_F_TRAIN_MODE) && h->cfg.dropout_p > 0.f)) return 0;     // bf16x3: per-layer launches (split.hip)
  return chain_pick_engine(h, n, flags);
}

int osd_mixup(osd_handle* h, const float* data, const float* cond, const float* surv, const int64_t* perm, double lam, int64_t n,
              float* data_out, float* cond_out, float* surv_out) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  OSD_TRY(check_rows(n));
  if (!perm) { set_error("null perm"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_HIP(launch_mixup3(h->stream, data_out ? data : nullptr, cond_out ? cond : nullptr, surv_out ? surv : nullptr, perm, lam, n, h->arch.D,
                        h->arch.cond_dim, data_out, cond_out, surv_out));
  return OSD_OK;
}

// Per-launch timing of one reverse step with HIP events on the handle's stream (bench.py's
// roofline leg): launch i of the step (0 = input_proj, 1.. = the block halves in execution
// order, last = output_proj + posterior) averaged over `reps` eager steps at t = T/2.
int osd_profile_step(osd_handle* h, const float* cond, int64_t n, int reps, float* ms_out, double* flop_out, int max_entries,
                     int* n_entries) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!cond || !ms_out || !flop_out || !n_entries || n <= 0 || reps <= 0) { set_error("bad argument"); return OSD_EINVAL; }
  const Arch& a = h->arch;
  const int n_launch = 2 + 2 * a.n_blocks;
  if (max_entries < n_launch) { set_error("need room for %d entries", n_launch); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  OSD_TRY(ensure_packed(h, s));
  FwdWs ws;
  const int64_t fwd = carve_fwd(a, nullptr, n, false, &ws);
  const int64_t need = fwd + align_up(n * a.D, 64);
  OSD_TRY(ensure_arena(&h->main, need));
  carve_fwd(a, h->main.arena, n, false, &ws);
  float* x = h->main.arena + fwd;
  OSD_TRY(run_cond(h, s, cond, n, ws));
  OSD_HIP(launch_fill_randn(s, x, a.D, n, a.D, 1, 0, (uint32_t)a.T, TAG_POSTERIOR));
  std::vector<hipEvent_t> evs((size_t)n_launch + 1);
  for (auto& e : evs) OSD_HIP(hipEventCreate(&e));
  std::vector<double> acc((size_t)n_launch, 0.0);
  int rc = OSD_OK;
  for (int r = 0; r < reps + 1 && rc == OSD_OK; ++r) {       // first pass warms up, untimed
    h->prof_events = &evs;
    h->prof_i = 0;
    hipError_t e = hipEventRecord(evs[h->prof_i++], s);
    TrunkIn in{};
    in.x = x; in.ldx = a.D; in.n = n; in.t_imm = a.T / 2;
    if (e == hipSuccess) rc = run_trunk(h, s, ws, in);
    if (rc == OSD_OK && e == hipSuccess) {
      GemmArgs g = output_proj_args(h, ws, n);
      EpiPosterior::Args ea{};
      ea.bias = h->params[a.pm.out_b]; ea.xin = x; ea.ldx = a.D; ea.xout = x; ea.ldo = a.D; ea.coef = h->d_coef;
      ea.t_imm = a.T / 2; ea.ldzz = a.D; ea.seed = 1; ea.t_first = a.T / 2;
      e = launch_posterior(s, g, ea);
      if (e == hipSuccess) e = hipEventRecord(evs[h->prof_i++], s);
    }
    h->prof_events = nullptr;
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("profile step failed: %s", hipGetErrorString(e)); rc = OSD_EHIP; break; }
    if (r == 0) continue;
    for (int i = 0; i < n_launch; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, evs[i], evs[i + 1]) == hipSuccess) acc[i] += ms;
    }
  }
  h->prof_events = nullptr;
  for (auto& e : evs) { hipError_t x2 = hipEventDestroy(e); (void)x2; }
  if (rc != OSD_OK) return rc;
  for (int i = 0; i < n_launch; ++i) ms_out[i] = (float)(acc[i] / reps);
  flop_out[0] = 2.0 * (double)n * a.D * a.H0;
  for (int i = 0; i < 2 * a.n_blocks; ++i) {
    const LayerDesc& l = a.layers[i];
    flop_out[1 + i] = 2.0 * (double)n * (l.K1 + l.K2) * l.N;
  }
  flop_out[n_launch - 1] = 2.0 * (double)n * a.block_out[a.n_blocks - 1] * a.D;
  *n_entries = n_launch;
  return OSD_OK;
}

// ---- building blocks for the parity tests -----------------------------------------
int osd_op_linear(osd_handle* h, const float* x, const float* w, const float* b, int64_t n, int K, int N, int silu, float* y) {
  if (!h || !x || !w || !y) { set_error("null argument"); return OSD_EINVAL; }
  OSD_TRY(check_rows(n));
  OSD_HIP(hipSetDevice(h->cfg.device));
  if (h->precision == 1 && !silu && n > 0 && K > 0 && N > 0) return split_op_linear(h, x, w, b, n, K, N, y);
  GemmArgs g{};
  g.A = w; g.lda = K; g.B0 = x; g.ldb0 = K; g.K0 = K; g.F = N; g.P = (int)n; g.K = K;
  OSD_HIP(launch_linear(h->stream, g, true, true, b, y, N, silu != 0, false));
  return OSD_OK;
}

int osd_op_linear_gn_silu(osd_handle* h, const float* x, int K1, const float* x2, int K2, const float* w, const float* b,
                          const float* gamma, const float* beta, int64_t n, int N, float* y) {
  if (!h || !x || !w || !b || !gamma || !beta || !y) { set_error("null argument"); return OSD_EINVAL; }
  OSD_TRY(check_rows(n));
  if (N % 8 || !gn_width_supported(N / 8)) { set_error("unsupported GroupNorm width %d", N / 8); return OSD_EUNSUPPORTED; }
  if (K2 > 0 && (!x2 || K1 % 4)) { set_error("second panel needs x2 and K1 %% 4 == 0"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  GemmArgs g{};
  g.A = w; g.lda = K1 + K2; g.B0 = x; g.ldb0 = K1; g.K0 = K1; g.B1 = x2; g.ldb1 = K2; g.F = N; g.P = (int)n; g.K = K1 + K2;
  GnArgs ga{};
  ga.bias = b; ga.gamma = gamma; ga.beta = beta; ga.out = y; ga.ldo = N;
  OSD_HIP(launch_gn_silu(h->stream, g, N / 8, ga));
  return OSD_OK;
}

int osd_op_gemm(osd_handle* h, const float* A, int lda, int a_kc, const float* B, int ldb, int b_kc, int F, int P, int K, float* C,
                int ldc, int accumulate) {
  if (!h || !A || !B || !C) { set_error("null argument"); return OSD_EINVAL; }
  if (F <= 0 || P <= 0 || K <= 0) { set_error("bad extents"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B0 = B; g.ldb0 = ldb; g.K0 = K; g.F = F; g.P = P; g.K = K;
  hipError_t e = launch_linear(h->stream, g, a_kc != 0, b_kc != 0, nullptr, C, ldc, false, accumulate != 0);
  if (e == hipErrorInvalidValue) { set_error("layout/accumulate combination not instantiated"); return OSD_EUNSUPPORTED; }
  OSD_HIP(e);
  return OSD_OK;
}

int osd_op_randn(osd_handle* h, float* out, int64_t rows, int cols, uint64_t seed, int64_t row_offset, uint32_t step, uint32_t kind) {
  if (!h || !out) { set_error("null argument"); return OSD_EINVAL; }
  OSD_TRY(check_rows(rows));
  OSD_HIP(hipSetDevice(h->cfg.device));
  const uint32_t tag = kind == 0 ? (uint32_t)TAG_POSTERIOR : kind == 1 ? (uint32_t)TAG_QNOISE : (uint32_t)TAG_USER + (kind & 0xffu);
  OSD_HIP(launch_fill_randn(h->stream, out, cols, rows, cols, seed, (uint32_t)row_offset, step, tag));
  return OSD_OK;
}

}  // extern "C"
