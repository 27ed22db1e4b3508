// train.hip -- training entry points of the C ABI (include/osdiff.h):
// fused forward + backward of the eps-prediction MSE loss, mixup lives in api.hip,
// clip_grad_norm_ + AdamW over flat buffers.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "handle.h"
#include "kernels.h"
#include "kernels_train.h"
#include "fwd.h"

namespace osd {

static int64_t align_up64(int64_t v) { return (v + 63) / 64 * 64; }
static int t_pad(int T) { return (T + 31) / 32 * 32; }


struct TrainWs {
  FwdWs f;
  int xld;           // row stride of x_t: D, or roundup(D, 32) with zero pad columns (input_proj then reads the unpacked weight)
  float *x_t, *noise, *d_out, *u0;
  float* cond_mix;   // [n][cond_dim]: the batch's condition rows when they come from a resident dataset (osd_train_batch_source)
  float* x0_mix;     // [n][D]: the batch's data rows, only carved when the constraint losses read them
  int* t_idx;
  float *g_h0, *g_ce2, *g_ce1, *g_u, *g_temb;
  std::vector<float*> g_out, g_z2, g_mid, g_z1;
  double* normsq;
  float* partials;   // gn backward column partials
  float* slabs;      // split-K wgrad partial tiles
  int64_t slab_floats;
  float* sq_act;     // train_squad.h: unit-order activations, per 32-patient sub-panel (null: model outside the squad decomposition)
  float* sq_wpk;     // ... this step's fragment-ordered trunk weights
  float* sq_gact;    // train_squad_bwd.h: unit-order gradients, same geometry as sq_act
  float* sq_wpk_t;   // ... this step's fragment-ordered transposed weights
  unsigned* sq_bar2; // ... its barrier counters + status word
  unsigned* sq_bar;  // ... its barrier counters [panels][16] + the status word
  int64_t sq_panels;
  // constraint losses (only carved when configured)
  float *pred, *g_x0;
  ConsWs cw;
};

static int x_t_stride(const Arch& a, const ConsPlan* cp) { return (cp || a.D % 4) ? a.D : (a.D + 31) / 32 * 32; }

static int64_t carve_train(const Arch& a, float* base, int64_t n, const ConsPlan* cp, TrainWs* w) {
  int64_t off = 0;
  auto take = [&](int64_t floats) { float* p = base ? base + off : nullptr; off += align_up64(floats); return p; };
  float* fbase = base;
  const int64_t fwd = carve_fwd(a, fbase, n, true, &w->f);
  off = align_up64(fwd);
  w->xld = x_t_stride(a, cp);
  w->x_t = take(n * (int64_t)w->xld); w->noise = take(n * a.D); w->d_out = take(n * a.D);
  w->u0 = take(n * 64);
  w->cond_mix = take(n * (int64_t)a.cond_dim);
  w->t_idx = (int*)take(n);
  w->g_h0 = take(n * a.H0); w->g_ce2 = take(n * 64); w->g_ce1 = take(n * 64); w->g_u = take(n * 64);
  w->g_temb = take((int64_t)t_pad(a.T) * a.H0 + COND_BWD_PART_FLOATS);      // + k_cond_bwd's partial copies (zeroed with the table)
  w->g_out.resize(a.n_blocks); w->g_z2.resize(a.n_blocks); w->g_mid.resize(a.n_blocks); w->g_z1.resize(a.n_blocks);
  for (int b = 0; b < a.n_blocks; ++b) {
    const int64_t c = a.block_out[b];
    w->g_out[b] = take(n * c); w->g_z2[b] = take(n * c); w->g_mid[b] = take(n * c); w->g_z1[b] = take(n * c);
  }
  w->normsq = (double*)take(16);
  int cmax = 0;
  for (int c : a.block_out) cmax = c > cmax ? c : cmax;
  w->partials = take((int64_t)GN_BWD_MAX_BLOCKS * 3 * cmax);
  w->slab_floats = 16 * 1024 * 1024;      // 64 MB of split-K slabs
  w->slabs = take(w->slab_floats);
  {
    int64_t wf = 0;
    const int64_t af = train_squad_act_floats(a, &wf);
    w->sq_panels = (n + 63) / 64;
    w->sq_act = af ? take(2 * w->sq_panels * af) : nullptr;
    w->sq_wpk = af ? take(wf) : nullptr;
    w->sq_bar = af ? (unsigned*)take(w->sq_panels * 16 + 16) : nullptr;      // barrier counters | status word
    w->sq_gact = af ? take(2 * w->sq_panels * af) : nullptr;
    w->sq_wpk_t = af ? take(train_squad_bwd_wpk_floats(a)) : nullptr;
    w->sq_bar2 = af ? (unsigned*)take(w->sq_panels * 16 + 16) : nullptr;
  }
  w->pred = w->g_x0 = w->x0_mix = nullptr;
  if (cp) {
    w->pred = take(n * a.D); w->g_x0 = take(n * a.D); w->x0_mix = take(n * a.D);
    const int64_t bytes = cons_carve(*cp, n, a.D, nullptr, &w->cw);
    char* cbase = (char*)take((bytes + 3) / 4);
    cons_carve(*cp, n, a.D, cbase, &w->cw);
  }
  return off;
}

// out[p][f] = sum_k A(f,k) B(p,k) helpers for the two backward GEMM shapes
// dW[n_out][k_in] = sum_m gz[m][n_out] * x[m][k_in].  The reduction runs over the batch, the output is only
// n_out x k_in: split the batch over blockIdx.y so that ~1024 workgroups exist, each writing its partial
// tile to a slab, then sum the slabs in a fixed order (deterministic; no float atomics).
static bool small_wgrad_ok(int kin, int nout, int lddw) { return kin <= 8 && nout * kin <= 256 && lddw == kin; }

// `dbias_small`: taken by the small path only (it adds sum_m gz[m][n] into it); every other path leaves the bias to the caller
static hipError_t wgrad(hipStream_t s, const TrainWs& w, const float* x, int ldx, int kin, const float* gz, int ldg, int nout, int64_t rows, float* dw, int lddw,
                        float* dbias_small = nullptr) {
  if (small_wgrad_ok(kin, nout, lddw)) return launch_small_wgrad(s, x, kin, gz, ldg, nout, rows, dw, dbias_small);
  GemmArgs g{};
  g.A = x; g.lda = ldx; g.B0 = gz; g.ldb0 = ldg; g.K0 = (int)rows; g.F = kin; g.P = nout; g.K = (int)rows;
  const long tiles = (long)((kin + 63) / 64) * ((nout + 63) / 64);
  static const int target = [] { const char* e = getenv("OSD_WGRAD_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  int slices = (int)((target + tiles - 1) / tiles);
  const int max_slices = (int)((rows + 127) / 128);
  if (slices > max_slices) slices = max_slices;
  const int64_t numel = (int64_t)nout * kin;
  while (slices > 1 && (int64_t)slices * numel > w.slab_floats) --slices;
  if (slices <= 1 || kin % 4 || lddw % 4 || (reinterpret_cast<uintptr_t>(dw) & 15)) return launch_linear(s, g, false, false, nullptr, dw, lddw, false, false);
  g.kchunk = (int)(((rows + slices - 1) / slices + 31) / 32 * 32);
  const int ns = (int)((rows + g.kchunk - 1) / g.kchunk);
  hipError_t e = launch_wgrad_splitk(s, g, w.slabs, kin, numel);
  if (e != hipSuccess) return e;
  return launch_slab_reduce(s, w.slabs, ns, nout, kin, numel, dw, lddw);
}
static hipError_t dgrad(hipStream_t s, const float* w, int ldw, int kin, const float* gz, int ldg, int nout, int64_t rows, float* dx, int lddx, bool accumulate) {
  // dX[m][k_in] (+)= sum_n gz[m][n] * W[n][k_in]
  GemmArgs g{};
  g.A = w; g.lda = ldw; g.B0 = gz; g.ldb0 = ldg; g.K0 = nout; g.F = kin; g.P = (int)rows; g.K = nout;
  return launch_linear(s, g, false, true, nullptr, dx, lddx, false, accumulate);
}


// sizes (and on growth re-allocates) the training arena for n rows and carves it
static int ensure_train_ws(osd_handle* h, hipStream_t s, int64_t n, const ConsPlan* cp, TrainWs* w) {
  const Arch& a = h->arch;
  const int64_t need = carve_train(a, nullptr, n, cp, w);
  if (h->train_arena_floats < need) {
    if (h->train_arena) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(h->train_arena)); h->train_arena = nullptr; h->train_arena_floats = 0; }
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)need * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %lld bytes failed", (long long)need * 4); return OSD_ENOMEM; }
    h->train_arena = (float*)p;
    h->train_arena_floats = need;
  }
  carve_train(a, h->train_arena, n, cp, w);
  return OSD_OK;
}

// ConditionalEmbedding + cond_proj with the pre-activation kept for backward (models/diffusion.py:101-105, 226)
// (Tried: the whole conditioning branch + the time_proj table as ONE VALU kernel instead of k_cond_mlp_fwd + two tile GEMMs of
// 7-10 us each: 49 us -- one 87 KB-LDS workgroup per CU, 1.25 rounds, no latency hiding.  Dropped.)
static int cond_embed_fwd(osd_handle* h, hipStream_t s, const float* cond, int64_t n, TrainWs& w) {
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  // the two 64-wide layers and the SiLU between them in one launch (k_cond_mlp_fwd), cond_proj on the tile GEMM
  OSD_HIP(launch_cond_mlp_fwd(s, cond, a.cond_dim, h->params[pm.ce0_w], h->params[pm.ce0_b], h->params[pm.ce2_w], h->params[pm.ce2_b], n,
                              w.u0, w.f.ce1, w.f.ce2));
  GemmArgs g{};
  g.A = h->params[pm.cp_w]; g.lda = 64; g.B0 = w.f.ce2; g.ldb0 = 64; g.K0 = 64; g.K = 64; g.P = (int)n; g.F = a.H0;
  OSD_HIP(launch_linear(s, g, true, true, h->params[pm.cp_b], w.f.cproj, a.H0, false, false));
  return OSD_OK;
}

// zeroes what the backward accumulates atomically (time-embedding table gradient, the small bias gradients)
static void add_backward_zeros(const Arch& a, const TrainWs& w, float* const* grads, ZeroList* zl) {
  const ParamMap& pm = a.pm;
  auto add = [&](float* p, int64_t c) { zl->ptr[zl->n] = p; zl->count[zl->n] = c; ++zl->n; };
  add(w.g_temb, (int64_t)t_pad(a.T) * a.H0 + COND_BWD_PART_FLOATS);
  const int small[] = {pm.ce0_b, pm.ce2_b, pm.in_b, pm.cp_b, pm.tp_b, pm.out_b};
  for (int i : small) add(grads[i], pm.numel[i]);
  if (small_wgrad_ok(a.cond_dim, 64, a.cond_dim)) add(grads[pm.ce0_w], pm.numel[pm.ce0_w]);     // k_small_wgrad adds into it
  for (const LayerDesc& l : a.layers) {          // GroupNorm backward adds its per-block column sums atomically
    add(grads[l.b], pm.numel[l.b]); add(grads[l.gamma], pm.numel[l.gamma]); add(grads[l.beta], pm.numel[l.beta]);
  }
}


// the handle's side stream (memory-bound leaves of the backward pass: GroupNorm affine gradients, small weight gradients).
// Default priority: a LOW-priority stream makes the HIP runtime open a low-priority hardware queue, and streams created later
// in the process -- e.g. the sampling slots of a model built after a training run, the reference's `--steps train generate` --
// were seen to land on it: their launch-bound hipGraph replays then ran 3.3x slower (bench.py reference_workload: 3700 -> 1130
// patients/s).  The leaves no longer need the low priority anyway: the grouped weight-gradient launch runs on the main stream.
static int side_stream(osd_handle* h, hipStream_t* out) {
  if (!h->wgrad_stream) OSD_HIP(hipStreamCreateWithFlags(&h->wgrad_stream, hipStreamNonBlocking));
  *out = h->wgrad_stream;
  return OSD_OK;
}

// The backward pass from dL/d eps_hat (d_out [n][D]) to every parameter gradient (and optionally dL/dx_t), over the
// activations a training-mode forward left in W.
static int backward_from(osd_handle* h, hipStream_t s, TrainWs& W, const float* x_t, int x_ld, const int* t_idx, const float* cond, int64_t n,
                         const float* d_out, bool train, const float* const* masks, uint64_t seed, uint32_t roff, float* const* grads,
                         float* dx_t, void* const* events, float* loss_poison = nullptr) {
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  const int D = a.D;
  const bool drop = train && h->cfg.dropout_p > 0.f;
  const int last = a.n_blocks - 1;
  const int Hl = a.block_out[last];
  // ---- backward ----
  // Two streams: the chain  GroupNorm/SiLU backward -> dgrad -> next layer  is the critical path and stays on the
  // handle's stream; every weight/bias gradient (wgrad, split-K slab sums, column sums) is a leaf and goes to a
  // lower-priority side stream that fills the CUs the small dgrad launches leave idle.  fork() orders the side
  // stream behind what the main stream has produced so far; the side stream owns the slab workspace.
  hipStream_t s2 = s;
  if (h->two_stream_bwd) OSD_TRY(side_stream(h, &s2));
  size_t ev_used = 0;
  auto next_event = [&](hipEvent_t* out) -> int {
    if (ev_used == h->ev_pool.size()) {
      hipEvent_t e;
      OSD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->ev_pool.push_back(e);
    }
    *out = h->ev_pool[ev_used++];
    return OSD_OK;
  };
  auto fork = [&]() -> int {          // side stream waits for everything enqueued on the main stream so far
    if (s2 == s) return OSD_OK;
    hipEvent_t e;
    OSD_TRY(next_event(&e));
    OSD_HIP(hipEventRecord(e, s));
    OSD_HIP(hipStreamWaitEvent(s2, e, 0));
    return OSD_OK;
  };
  // Weight gradients are leaves: they are collected and launched as ONE grouped GEMM per flush point (after the decoder +
  // bottleneck half of the backward pass, and at its end) instead of ~17 launches of a few tiles each.  A gradient bucket is
  // final once the flush that carries its weight gradients has been enqueued, so bucket events are recorded there.
  std::vector<WgPending> pend;
  std::vector<WgPending>* grp = h->grouped_wgrad ? &pend : nullptr;
  int ev = 0, ev_closed = 0, n_flush = 0;
  bool s2_slabs_busy = false;          // an immediate split-K weight gradient on the side stream may still be using W.slabs
  auto record = [&]() -> int {        // bucket complete up to the pending weight gradients
    ++ev_closed;
    if (!grp) { if (events) OSD_HIP(hipEventRecord((hipEvent_t)events[ev], s2)); ++ev; }
    return OSD_OK;
  };
  // a weight gradient: deferred to the next grouped launch when eligible, else launched now on the side stream (which then
  // has to see what the main stream produced: each fork costs the main stream a few microseconds, so only then)
  // b0..b2: bias gradients equal to the column sums of gz (the Linear's own bias and tensors that share it); they ride along
  // with the grouped launch, with the small kernel, or -- immediate GEMM path -- take a column-sum launch
  // (Round 3 also built the whole trunk backward as ONE persistent launch -- dgrad tiles and weight-gradient items as work units
  // ordered by dependency counters.  Parity-green and slower, 800 vs 539 us: a wave streaming fp32 MFMAs starves the co-resident
  // wave's VALU epilogue, so the dgrad chain stretched.  Removed in round 4; the stamps and the verdict are profiles/r03_bwd_persist.md.)
  auto wg = [&](const float* x, int ldx, int kin, const float* gz, int ldg, int nout, int64_t rows, float* dw, int lddw,
                float* b0 = nullptr, float* b1 = nullptr, float* b2 = nullptr) -> int {
    const WgPending wp{x, ldx, kin, gz, ldg, nout, rows, dw, lddw, {b0, b1, b2}};
    if (grp && kin >= 16 && wgrad_group_ok(wp)) { pend.push_back(wp); return OSD_OK; }
    const bool small = small_wgrad_ok(kin, nout, lddw);
    if (small && !events) {            // a 5 us kernel whose inputs are on the main stream: run it there (no fork, no event)
      OSD_HIP(wgrad(s, W, x, ldx, kin, gz, ldg, nout, rows, dw, lddw, b0));
      return OSD_OK;
    }
    OSD_TRY(fork());
    OSD_HIP(wgrad(s2, W, x, ldx, kin, gz, ldg, nout, rows, dw, lddw, b0));
    if (!small && s2 != s) s2_slabs_busy = true;
    if (b0 && !small) OSD_HIP(launch_colsum(s2, gz, ldg, rows, nout, b0));
    if (b0 && small && (b1 || b2)) { set_error("internal: shared bias on the small weight-gradient path"); return OSD_EINVAL; }
    if (b0 && !small)
      for (float* bx : {b1, b2})
        if (bx) OSD_HIP(hipMemcpyAsync(bx, b0, (size_t)nout * 4, hipMemcpyDeviceToDevice, s2));
    return OSD_OK;
  };
  // mid-pass flush (data parallel: the decoder half's buckets can go to the wire early): grouped weight gradients on the side
  // stream, one workgroup per CU walking the list so that the other slot of every CU stays with the dgrad chain of the main
  // stream (a full-width launch starved it: a 16 us dgrad took 104 us)
  hipEvent_t mid_done = nullptr;       // the side stream is through with the slab workspace
  auto flush_mid = [&]() -> int {
    if (!grp) return OSD_OK;
    OSD_TRY(fork());                  // the side stream sees every gz produced so far
    static const int mid_cap = [] { const char* e = getenv("OSD_WGRAD_MID_CAP"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    OSD_TRY(wgrad_group_flush(h, s2, n_flush++, pend, W.slabs, W.slab_floats, s2 != s ? mid_cap : 0));
    pend.clear();
    for (; ev < ev_closed; ++ev)
      if (events) OSD_HIP(hipEventRecord((hipEvent_t)events[ev], s2));
    if (s2 != s) { OSD_TRY(next_event(&mid_done)); OSD_HIP(hipEventRecord(mid_done, s2)); }
    return OSD_OK;
  };
  // GroupNorm backward: inside the epilogue of the dgrad that produces the layer's upstream gradient (group widths 32 / 64), or
  // -- option off, other widths -- as its own pass between the GEMMs
  bool fuse = h->fused_gn_bwd != 0;
  fuse = fuse && grp != nullptr;       // the fused path's bias / affine gradients ride with the grouped launches
  for (const LayerDesc& l : a.layers) fuse = fuse && dgrad_gnbwd_supported(l.gw);
  // single-GPU steps: the dgrad chain between the first launch (output_proj) and the last (into h0) as one launch of squads
  const bool squad_bwd = fuse && !events && !h->wgrad_mid_flush && W.sq_gact && loss_poison && h->train_squad >= 2 && train_squad_ok(h, n);
  const float keep_scale = (float)(1.0 / (1.0 - (double)h->cfg.dropout_p));
  std::vector<GnColItem> cols;
  // d gamma / d beta of the layers whose backward ran in a dgrad epilogue: memory-bound leaves, one launch per call.  They
  // go to the side stream as soon as the block loop is through (beside the last small GEMMs of the main stream), not next to
  // the grouped weight-gradient launch, whose 512 workgroups would keep them off the CUs until it ends.
  int n_cols_flush = 0;
  auto side_leaves = [&](hipStream_t st) -> int {
    if (!cols.empty()) {
      OSD_TRY(gn_colsums_flush(h, st, 8 + n_cols_flush++, cols));      // plan slots 8.. hold the column-sum lists
      cols.clear();
    }
    return OSD_OK;
  };
  auto flush_all = [&](bool mid) -> int {
    if (mid) {
      // the events flush_mid() records behind the weight gradients must cover the affine gradients too: those go first
      if (s2 != s && !cols.empty()) OSD_TRY(fork());
      OSD_TRY(side_leaves(s2));
      return flush_mid();
    }
    // End of the pass.  The grouped weight-gradient GEMM (the long pole, ~200 us at batch 4096) stays on the MAIN stream: no
    // stream hop in front of it or between it and the optimizer.  The memory-bound leaves run beside it on the side stream and
    // are long done when the main stream joins.
    if (s2 != s) {
      if (!cols.empty()) { OSD_TRY(fork()); OSD_TRY(side_leaves(s2)); }
      if (mid_done) OSD_HIP(hipStreamWaitEvent(s, mid_done, 0));     // slab workspace handed back by the mid-pass flush
      if (s2_slabs_busy && grp && !pend.empty()) {                    // ... and by immediate split-K weight gradients (e.g. the
        hipEvent_t e;                                                 // ConditionalEmbedding's first Linear at cond_dim 8 or 12)
        OSD_TRY(next_event(&e));
        OSD_HIP(hipEventRecord(e, s2));
        OSD_HIP(hipStreamWaitEvent(s, e, 0));
        s2_slabs_busy = false;
      }
    } else {
      OSD_TRY(side_leaves(s));
    }
    if (grp) {
      OSD_TRY(wgrad_group_flush(h, s, n_flush++, pend, W.slabs, W.slab_floats, 0));
      pend.clear();
    }
    if (s2 != s) {
      hipEvent_t e;
      OSD_TRY(next_event(&e));
      OSD_HIP(hipEventRecord(e, s2));
      OSD_HIP(hipStreamWaitEvent(s, e, 0));
    }
    for (; ev < ev_closed; ++ev)
      if (events) OSD_HIP(hipEventRecord((hipEvent_t)events[ev], s));
    return OSD_OK;
  };
  // dgrad whose epilogue is the GroupNorm+SiLU(+dropout) backward of `ln` (z / stats of that layer): writes dL/dz and dL/dy
  auto dgrad_fused = [&](const float* w, int ldw, int kin, const float* gz_next, int ldg, int nout, const LayerDesc& ln, const float* z,
                         const float* stats, float* gz_out, float* gy_buf, bool accumulate, bool with_drop, int blk,
                         const float* w_skip = nullptr, int kin_skip = 0, float* out_skip = nullptr, bool* skip_done = nullptr,
                         bool launch = true) -> int {
    if (!launch) {      // the squad launch (train_squad_bwd.h) has produced gz_out / gy_buf (and the skip share): only the column sums are left to queue
      if (skip_done && w_skip) *skip_done = true;
      cols.push_back({gy_buf, kin, z, kin, stats, kin, ln.gw, n, grads[ln.gamma], grads[ln.beta]});
      return OSD_OK;
    }
    GemmArgs g{};
    g.A = w; g.lda = ldw; g.B0 = gz_next; g.ldb0 = ldg; g.K0 = nout; g.F = kin; g.P = (int)n; g.K = nout;
    g.ksplit = h->train_ksplit != 0;       // launch.h: two wave groups where a launch has ~one tile per CU and >= 32 K tiles (the first dgrad)
    GnBwdEpi e{};
    e.z = z; e.ldz = kin; e.stats = stats; e.gamma = h->params[ln.gamma]; e.beta = h->params[ln.beta];
    e.gz = gz_out; e.ldg = kin; e.gy = gy_buf; e.ldy = kin; e.accumulate = accumulate ? 1 : 0;
    e.drop_mode = with_drop ? (masks ? 1 : 2) : 0;
    e.mask = (with_drop && masks) ? masks[blk] : nullptr; e.ldm = kin; e.keep_scale = keep_scale; e.p_drop = h->cfg.dropout_p;
    e.seed = seed; e.row_offset = roff; e.step = 0; e.tag = TAG_DROPOUT + (uint32_t)blk;
    {
      bool launched = false;
      if (w_skip && h->dual_dgrad) {
        // the skip connection's share of the same gz (plain dX = gz W_skip) rides in the same launch
        GemmArgs g2{};
        g2.A = w_skip; g2.lda = ldw; g2.B0 = gz_next; g2.ldb0 = ldg; g2.K0 = nout; g2.F = kin_skip; g2.P = (int)n; g2.K = nout;
        const hipError_t de = launch_dgrad_gnbwd_dual(s, g, ln.gw, e, g2, out_skip, kin_skip);
        if (de == hipSuccess) { launched = true; if (skip_done) *skip_done = true; }
        else if (de != hipErrorInvalidValue) OSD_HIP(de);
        else (void)hipGetLastError();
      }
      if (!launched) OSD_HIP(launch_dgrad_gnbwd(s, g, ln.gw, e));
    }
    cols.push_back({gy_buf, kin, z, kin, stats, kin, ln.gw, n, grads[ln.gamma], grads[ln.beta]});
    return OSD_OK;
  };
  // plain dgrad dX = gz W (no epilogue)
  auto dgrad_plain = [&](const float* w, int ldw, int kin, const float* gz, int ldg, int nout, float* dx, int lddx) -> int {
    OSD_HIP(dgrad(s, w, ldw, kin, gz, ldg, nout, n, dx, lddx, false));
    return OSD_OK;
  };
  // output_proj
  OSD_TRY(wg(W.f.out[last], Hl, Hl, d_out, D, D, n, grads[pm.out_w], Hl, grads[pm.out_b]));
  OSD_TRY(record());
  if (fuse) {
    const LayerDesc& lz = a.layers[2 * last + 1];
    OSD_TRY(dgrad_fused(h->params[pm.out_w], Hl, Hl, d_out, D, D, lz, W.f.z2[last], W.f.st2[last], W.g_z2[last], W.g_out[last], false, false, last));
    if (squad_bwd) {      // every dgrad between this one and the last (into h0) in one launch of squads
      TrainSquadBwdBufs B{W.g_out.data(), W.g_z2.data(), W.g_mid.data(), W.g_z1.data(), W.g_h0, masks, drop, seed, roff};
      OSD_TRY(train_squad_backward(h, s, W.f, B, n, W.sq_gact, W.sq_wpk_t, W.sq_bar2, W.sq_panels, loss_poison));
    }
  } else {
    OSD_HIP(dgrad(s, h->params[pm.out_w], Hl, Hl, d_out, D, D, n, W.g_out[last], Hl, false));
  }

  for (int b = a.n_blocks - 1; b >= 0; --b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    const int C = l1.N;
    const int Kt = l1.K1 + l1.K2;
    const float* xin = (b == 0) ? W.f.h0 : W.f.out[b - 1];
    int skip_block = -1;
    if (l1.K2 > 0) skip_block = a.n_enc - 1 - (b - a.n_enc - 1);
    float* gdst = (b == 0) ? W.g_h0 : W.g_out[b - 1];
    const bool acc = (b >= 1) && (b - 1 < a.n_enc);      // encoder outputs already hold their skip gradient
    if (fuse) {
      // dL/dz of the second half is in g_z2[b] (left by the dgrad above it); bias gradients ride with the weight gradients
      bool skip_done = false;
      OSD_TRY(wg(W.f.mid[b], C, C, W.g_z2[b], C, C, n, grads[l2.w], C, grads[l2.b]));
      OSD_TRY(dgrad_fused(h->params[l2.w], C, C, W.g_z2[b], C, C, l1, W.f.z1[b], W.f.st1[b], W.g_z1[b], W.g_mid[b], false, drop, b,
                          nullptr, 0, nullptr, nullptr, !squad_bwd));
      OSD_TRY(wg(xin, l1.K1, l1.K1, W.g_z1[b], C, C, n, grads[l1.w], Kt, grads[l1.b]));
      if (l1.K2 > 0) OSD_TRY(wg(W.f.out[skip_block], l1.K2, l1.K2, W.g_z1[b], C, C, n, grads[l1.w] + l1.K1, Kt));
      OSD_TRY(record());
      if (b == a.n_enc && (h->wgrad_mid_flush || events)) OSD_TRY(flush_all(true));
      if (b == 0) {
        if (!squad_bwd) OSD_TRY(dgrad_plain(h->params[l1.w], Kt, l1.K1, W.g_z1[b], C, C, gdst, l1.K1));
      } else {
        const LayerDesc& lp = a.layers[2 * (b - 1) + 1];      // the layer that produced this block's main input
        // an encoder output already holds its skip gradient (written by the decoder block that popped it): second dependency
        OSD_TRY(dgrad_fused(h->params[l1.w], Kt, l1.K1, W.g_z1[b], C, C, lp, W.f.z2[b - 1], W.f.st2[b - 1], W.g_z2[b - 1], W.g_out[b - 1], acc,
                            false, b - 1, l1.K2 > 0 ? h->params[l1.w] + l1.K1 : nullptr, l1.K2, l1.K2 > 0 ? W.g_out[skip_block] : nullptr, &skip_done,
                            !squad_bwd));
      }
      if (l1.K2 > 0 && !skip_done) OSD_TRY(dgrad_plain(h->params[l1.w] + l1.K1, Kt, l1.K2, W.g_z1[b], C, C, W.g_out[skip_block], l1.K2));
      continue;
    }
    // second half: GroupNorm+SiLU backward, wgrad, dgrad
    GnBwdArgs ga{};
    ga.g = W.g_out[b]; ga.z = W.f.z2[b]; ga.stats = W.f.st2[b]; ga.gamma = h->params[l2.gamma]; ga.beta = h->params[l2.beta];
    ga.gz = W.g_z2[b]; ga.dgamma = grads[l2.gamma]; ga.dbeta = grads[l2.beta]; ga.dbias = grads[l2.b];
    ga.rows = n; ga.C = C; ga.drop_mode = 0; ga.partials = W.partials; ga.atomic_cols = 0;      // fixed-order partial reduce: deterministic
    OSD_HIP(launch_gn_silu_bwd(s, l2.gw, ga));
    OSD_TRY(wg(W.f.mid[b], C, C, W.g_z2[b], C, C, n, grads[l2.w], C));
    OSD_HIP(dgrad(s, h->params[l2.w], C, C, W.g_z2[b], C, C, n, W.g_mid[b], C, false));
    // first half (dropout sits behind it)
    GnBwdArgs gb{};
    gb.g = W.g_mid[b]; gb.z = W.f.z1[b]; gb.stats = W.f.st1[b]; gb.gamma = h->params[l1.gamma]; gb.beta = h->params[l1.beta];
    gb.gz = W.g_z1[b]; gb.dgamma = grads[l1.gamma]; gb.dbeta = grads[l1.beta]; gb.dbias = grads[l1.b];
    gb.rows = n; gb.C = C; gb.partials = W.partials; gb.atomic_cols = 0;
    gb.drop_mode = drop ? (masks ? 1 : 2) : 0;
    gb.mask = (drop && masks) ? masks[b] : nullptr; gb.keep_scale = keep_scale; gb.p_drop = h->cfg.dropout_p;
    gb.seed = seed; gb.row_offset = roff; gb.step = 0; gb.tag = TAG_DROPOUT + (uint32_t)b;
    OSD_HIP(launch_gn_silu_bwd(s, l1.gw, gb));
    OSD_TRY(wg(xin, l1.K1, l1.K1, W.g_z1[b], C, C, n, grads[l1.w], Kt));
    if (l1.K2 > 0) OSD_TRY(wg(W.f.out[skip_block], l1.K2, l1.K2, W.g_z1[b], C, C, n, grads[l1.w] + l1.K1, Kt));
    OSD_TRY(record());
    if (b == a.n_enc && (h->wgrad_mid_flush || events)) OSD_TRY(flush_all(true));    // decoder blocks + bottleneck done: first half of the weight gradients
    // dgrad into the producer of the main input
    OSD_HIP(dgrad(s, h->params[l1.w], Kt, l1.K1, W.g_z1[b], C, C, n, gdst, l1.K1, acc));
    if (l1.K2 > 0) OSD_HIP(dgrad(s, h->params[l1.w] + l1.K1, Kt, l1.K2, W.g_z1[b], C, C, n, W.g_out[skip_block], l1.K2, false));
  }
  // Data parallel (bucket events requested): the encoder blocks' weight gradients go out NOW, in a grouped launch of their own, so
  // that their buckets' events fire one launch before the end of the pass -- what stays behind the last launch, and so cannot
  // overlap with any compute of this step, is the final bucket alone (input_proj + the conditioning branch: 2.1 MB of the 10.66 MB
  // instead of 5.2 MB).  Two grouped launches of ~250 items each fill the machine less well than one of 500 (the work-item list
  // is re-cut to the launch, wgrad_group.hip), so a single process keeps the one launch.
  if (events) OSD_TRY(flush_all(false));
  // input_proj, time_proj, cond_proj, ConditionalEmbedding  (h0 = x W^T + b + t_emb[t] + c_proj)
  // h0 = x W^T + b_in + (t_emb W_t^T + b_t)[t] + (c W_c^T + b_c): the three biases share one gradient, the column sums of g_h0
  // (Tried: the conditioning branch's backward -- five dependent launches of 5-13 us -- and the affine-gradient column sums on the
  // side stream BESIDE the grouped weight-gradient launch instead of in front of it.  The grouped launch's older waves starve
  // them: k_gn_colsums took 202 us instead of 44 and the side chain ended after the main one -- 1042 vs 988 us per step.  Dropped.)
  if (s2 != s && !cols.empty()) { OSD_TRY(fork()); OSD_TRY(side_leaves(s2)); }      // every GroupNorm layer's gy / z is final
  if (dx_t) OSD_HIP(dgrad(s, h->params[pm.in_w], D, D, W.g_h0, a.H0, a.H0, n, dx_t, D, false));
  OSD_TRY(wg(x_t, x_ld, D, W.g_h0, a.H0, a.H0, n, grads[pm.in_w], D, grads[pm.in_b], grads[pm.cp_b], grads[pm.tp_b]));
  OSD_TRY(wg(W.f.ce2, 64, 64, W.g_h0, a.H0, a.H0, n, grads[pm.cp_w], 64));
  // the branch below h0 (scatter into the time table, cond_proj's and the second embedding Linear's dgrads, SiLU backward): one launch
  const bool cond_fused = h->cond_bwd_fused && cond_bwd_ok(a.H0, W.g_h0, W.u0, W.g_ce2, W.g_u);
  // ... and, where the first embedding Linear has at most four inputs (k_small_wgrad's case), its weight gradient rides along
  const bool ce0_fused = cond_fused && a.cond_dim <= 4 && small_wgrad_ok(a.cond_dim, 64, a.cond_dim);
  if (cond_fused) {
    OSD_HIP(launch_cond_bwd(s, W.g_h0, a.H0, t_idx, W.g_temb, h->params[pm.cp_w], h->params[pm.ce2_w], W.u0, n, W.g_ce2, W.g_u,
                            ce0_fused ? cond : nullptr, a.cond_dim, W.g_temb + (int64_t)t_pad(a.T) * a.H0, grads[pm.ce0_w], grads[pm.ce0_b]));
  } else {
    OSD_HIP(launch_scatter_rows(s, W.g_h0, t_idx, n, a.H0, W.g_temb));
    OSD_HIP(dgrad(s, h->params[pm.cp_w], 64, 64, W.g_h0, a.H0, a.H0, n, W.g_ce2, 64, false));
  }
  // both tables carry zero rows up to a multiple of 32 (whole K steps of the grouped kernel): they add nothing
  OSD_TRY(wg(h->d_time_emb, a.time_dim, a.time_dim, W.g_temb, a.H0, a.H0, t_pad(a.T), grads[pm.tp_w], a.time_dim));
  OSD_TRY(wg(W.f.ce1, 64, 64, W.g_ce2, 64, 64, n, grads[pm.ce2_w], 64, grads[pm.ce2_b]));
  if (!cond_fused) {
    OSD_HIP(dgrad(s, h->params[pm.ce2_w], 64, 64, W.g_ce2, 64, 64, n, W.g_ce1, 64, false));
    OSD_HIP(launch_silu_bwd(s, W.u0, W.g_ce1, W.g_u, n * 64));
  }
  if (!ce0_fused) OSD_TRY(wg(cond, a.cond_dim, a.cond_dim, W.g_u, 64, 64, n, grads[pm.ce0_w], a.cond_dim, grads[pm.ce0_b]));
  OSD_TRY(record());
  OSD_TRY(flush_all(false));          // ends with the side stream joined: the caller's stream owns every result again
  return OSD_OK;
}

}  // namespace osd

using namespace osd;

extern "C" {

int osd_grad_buckets(const osd_config* cfg, int32_t* first, int32_t* last, int max_buckets) {
  if (!cfg) return 0;
  Arch a;
  if (build_arch(*cfg, &a) != OSD_OK) return 0;
  // backward finalises: output_proj, then the blocks last-to-first, then everything before the blocks
  std::vector<std::pair<int, int>> bk;
  bk.push_back({a.pm.out_w, a.pm.out_b});
  for (int b = a.n_blocks - 1; b >= 0; --b) bk.push_back({a.layers[2 * b].w, a.layers[2 * b + 1].beta});
  bk.push_back({0, a.pm.tp_b});
  const int n = (int)bk.size();
  if (first && last)
    for (int i = 0; i < n && i < max_buckets; ++i) { first[i] = bk[i].first; last[i] = bk[i].second; }
  return n;
}

int osd_train_loss_fwd_bwd(osd_handle* h, const float* x0, const float* cond, int64_t n, const int32_t* t_index, const float* noise,
                           const float* const* masks, uint64_t seed, int64_t row_offset, int flags, float* loss_out,
                           float* const* grads, double loss_scale, void* const* events, int n_events) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  const bool from_src = h && h->have_batch_src;          // one-shot: consumed (or dropped) by this call
  if (h) h->have_batch_src = false;
  if ((!from_src && (!x0 || !cond)) || !loss_out) { set_error("null tensor"); return OSD_EINVAL; }
  if (n == 0) { set_error("empty batch"); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  const Arch& a = h->arch;
  const ParamMap& pm = a.pm;
  const int n_buckets = a.n_blocks + 2;
  if (events && n_events != n_buckets) { set_error("expected %d events (osd_grad_buckets), got %d", n_buckets, n_events); return OSD_EINVAL; }
  if (grads)
    for (int i = 0; i < pm.n_params; ++i)
      if (!grads[i]) { set_error("grads[%d] is null", i); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  const bool train = (flags & OSD_F_TRAIN_MODE) != 0;
  const int D = a.D;
  const uint32_t roff = (uint32_t)row_offset;

  const bool use_pw = h->cons.n_pathways > 0 && h->w_pathway != 0.0;
  const bool use_me = h->cons.n_a > 0 && h->w_mutexpr != 0.0;
  const ConsPlan* cp = (use_pw || use_me) ? &h->cons : nullptr;
  if (cp && n < 2) { set_error("the constraint losses need at least 2 rows"); return OSD_EINVAL; }
  if (!h->parts_dev) OSD_HIP(hipMalloc((void**)&h->parts_dev, 64));
  TrainWs w;
  OSD_TRY(ensure_train_ws(h, s, n, cp, &w));
  h->saved_rows = -1;                    // the workspace no longer matches an osd_denoiser_forward_train call

  // ---- everything that is accumulated atomically must start at zero: zeroed by the q_sample kernel's own grid ----
  ZeroList zl{};
  {
    auto add = [&](float* p, int64_t c) { zl.ptr[zl.n] = p; zl.count[zl.n] = c; ++zl.n; };
    add(loss_out, 1);
    if (w.sq_bar && train_squad_ok(h, n)) {      // the squads' barrier counters + status words (forward, backward)
      add(reinterpret_cast<float*>(w.sq_bar), w.sq_panels * 16 + 16);
      add(reinterpret_cast<float*>(w.sq_bar2), w.sq_panels * 16 + 16);
    }
    if (cp) add(h->parts_dev, 3);
    if (cp && grads) add(w.g_x0, n * (int64_t)D);
    if (grads) add_backward_zeros(a, w, grads, &zl);
    if (zl.n > 128) { set_error("too many parameter tensors"); return OSD_EUNSUPPORTED; }
  }

  // ---- forward (models/diffusion.py:361-377) ----
  // x_t rows are padded to whole K steps with zeros when nothing else reads them with the dense stride: input_proj then takes
  // input_proj.weight as it is (clamped at D) and the per-step packed copy of that weight is not made
  const bool split_in = h->train_input_splitk > 1 && (int64_t)h->train_input_splitk * n * a.H0 <= w.slab_floats;
  const bool unpacked = w.xld > D && !split_in;
  // the t_emb table (and, unless input_proj reads the weight itself, its padded copy) follow the current parameters
  OSD_TRY(refresh_derived(h, s, !unpacked));
  const int* t_idx = nullptr;
  OSD_TRY(sanitize_t(h, s, t_index, n, &t_idx));
  // t ~ randint(0, T) is drawn inside q_sample (one launch less) and kept in w.t_idx for the layers that gather by it
  int* t_draw = nullptr;
  if (!t_idx) { t_draw = w.t_idx; t_idx = w.t_idx; }
  if (from_src) {
    // rows gathered from the resident dataset, mixed up and noised in one pass; conditions land in the workspace
    OSD_HIP(launch_q_sample_src(s, h->batch_src, t_draw ? nullptr : t_idx, h->d_sqrt_ac, h->d_sqrt_1m, noise, n, D, a.cond_dim, seed, roff, w.x_t,
                                w.noise, t_draw, a.T, w.cond_mix, cp ? w.x0_mix : nullptr, w.xld, &zl));
    cond = w.cond_mix;
    x0 = cp ? w.x0_mix : nullptr;
  } else {
    OSD_HIP(launch_q_sample(s, x0, t_draw ? nullptr : t_idx, h->d_sqrt_ac, h->d_sqrt_1m, noise, n, D, seed, roff, w.x_t, w.noise, t_draw, a.T, w.xld, &zl));
  }
  const float* eps_true = noise ? noise : w.noise;
  OSD_TRY(cond_embed_fwd(h, s, cond, n, w));
  TrainWs& W = w;
  TrunkIn in{};
  in.x = W.x_t; in.ldx = W.xld; in.kx = unpacked ? W.xld : D; in.a_unpacked = unpacked; in.ksplit = h->train_ksplit != 0;
  in.n = n; in.t_index = t_idx; in.train = train; in.save = grads != nullptr;
  in.masks = masks; in.seed = seed; in.row_offset = roff; in.drop_step = 0;
  // input_proj at the training batch: 256 output tiles of 63 sequential K steps, one workgroup per CU -- optionally K in slices
  // over more workgroups (k_fused.hip: partial tiles to slabs, then sum + epilogue); the slab workspace is idle during forward
  if (split_in) { in.in_slabs = W.slabs; in.in_slices = h->train_input_splitk; }
  if (W.sq_act && train_squad_ok(h, n)) {
    in.input_only = true;
    OSD_TRY(run_trunk(h, s, W.f, in));
    // the step's backward as squads as well (backward_from's condition): both weight repacks in the forward's launch
    const bool squad_bwd_next = grads && h->fused_gn_bwd && !events && !h->wgrad_mid_flush && W.sq_gact && h->train_squad >= 2;
    OSD_TRY(train_squad_forward(h, s, W.f, in, W.sq_act, W.sq_wpk, W.sq_bar, W.sq_panels, loss_out, squad_bwd_next ? W.sq_wpk_t : nullptr));
  } else {
    OSD_TRY(run_trunk(h, s, W.f, in));
  }
  {
    GemmArgs g = output_proj_args(h, W.f, n);
    EpiMse::Args ea{};
    ea.bias = h->params[pm.out_b]; ea.noise = eps_true; ea.ldn = D;
    ea.dout = grads ? W.d_out : nullptr; ea.ldd = D; ea.pred = cp ? W.pred : nullptr; ea.ldp = D; ea.loss = loss_out;
    ea.inv_count = (float)(1.0 / ((double)n * (double)D));
    ea.gscale = (float)(2.0 * (double)loss_scale / ((double)n * (double)D));
    // precision = 1: output_proj + MSE on the bf16 matrix pipe, operands split where they are staged (gemm_b3t.h)
    hipError_t me = h->precision == 1 ? launch_mse_b3t(s, g, ea) : hipErrorInvalidValue;
    if (me == hipErrorInvalidValue) { (void)hipGetLastError(); me = launch_mse(s, g, ea); }
    OSD_HIP(me);
  }
  if (cp) {
    OSD_HIP(hipMemcpyAsync(h->parts_dev, loss_out, 4, hipMemcpyDeviceToDevice, s));
    // constraint terms on x0_hat (models/diffusion.py:405) against the batch's x0; their gradient joins dL/d eps_hat
    OSD_HIP(launch_x0hat(s, W.x_t, t_idx, h->d_sqrt_ac, h->d_sqrt_1m, n, D, W.pred));
    OSD_HIP(hipMemsetAsync(W.cw.acc, 0, (size_t)W.cw.acc_doubles * 8, s));
    OSD_HIP(cons_moments(s, W.pred, D, n, D, W.cw.acc, W.cw.mi_r));
    float* gx = grads ? W.g_x0 : nullptr;
    if (use_pw)
      OSD_HIP(cons_pathway(s, *cp, W.cw, W.pred, D, n, D, (float)h->w_pathway, (float)(h->w_pathway * loss_scale), loss_out, h->parts_dev + 1, gx));
    if (use_me) {
      OSD_HIP(cons_moments(s, x0, D, n, D, W.cw.acc + 2 * (int64_t)D, W.cw.mi_t));
      OSD_HIP(cons_mutexpr(s, *cp, W.cw, W.pred, x0, D, n, D, (float)h->w_mutexpr, (float)(h->w_mutexpr * loss_scale), loss_out, h->parts_dev + 2, gx));
    }
    if (grads) OSD_HIP(launch_x0hat_bwd(s, W.g_x0, t_idx, h->d_sqrt_ac, h->d_sqrt_1m, n, D, W.d_out));
  }
  if (!grads) return OSD_OK;

  OSD_TRY(backward_from(h, s, W, W.x_t, W.xld, t_idx, cond, n, W.d_out, train, masks, seed, roff, grads, nullptr, events, loss_out));
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_train_batch_source(osd_handle* h, const float* data, int64_t ld_data, const float* cond, int64_t ld_cond, const int64_t* idx_a,
                           const int64_t* idx_b, double lam) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  if (!data || !cond) { set_error("null dataset tensor"); return OSD_EINVAL; }
  if (ld_data < h->arch.D || ld_cond < h->arch.cond_dim) { set_error("dataset row strides %lld / %lld are smaller than the model's dims", (long long)ld_data, (long long)ld_cond); return OSD_EINVAL; }
  if (!(lam >= 0.0 && lam <= 1.0)) { set_error("lam must be in [0,1]"); return OSD_EINVAL; }
  BatchSrc b{};
  b.data = data; b.ldd = ld_data; b.cond = cond; b.ldc = ld_cond; b.idx_a = idx_a; b.idx_b = idx_b;
  // python: lam and (1 - lam) are float64 scalars; torch multiplies an fp32 tensor by each as fp32 (launch_mixup)
  b.lam = (float)lam; b.oml = (float)(1.0 - lam);
  h->batch_src = b;
  h->have_batch_src = true;
  return OSD_OK;
}

int osd_denoiser_forward_train(osd_handle* h, const float* x_t, const int32_t* t_index, const float* cond, int64_t n,
                               const float* const* masks, uint64_t seed, int64_t row_offset, int flags, float* eps_out) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!x_t || !t_index || !cond || !eps_out || n == 0) { set_error("null tensor or empty batch"); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  const Arch& a = h->arch;
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  TrainWs W;
  OSD_TRY(ensure_train_ws(h, s, n, nullptr, &W));
  h->saved_rows = -1;
  const int* t_idx = nullptr;
  OSD_TRY(sanitize_t(h, s, t_index, n, &t_idx));
  OSD_TRY(refresh_derived(h, s));
  OSD_TRY(cond_embed_fwd(h, s, cond, n, W));
  TrunkIn in{};
  in.x = x_t; in.ldx = a.D; in.n = n; in.t_index = t_idx; in.train = (flags & OSD_F_TRAIN_MODE) != 0; in.save = true;
  in.masks = masks; in.seed = seed; in.row_offset = (uint32_t)row_offset; in.drop_step = 0;
  OSD_TRY(run_trunk(h, s, W.f, in));
  GemmArgs g = output_proj_args(h, W.f, n);
  OSD_HIP(launch_linear(s, g, true, true, h->params[a.pm.out_b], eps_out, a.D, false, false));
  h->saved_rows = n;
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_denoiser_backward(osd_handle* h, const float* x_t, const int32_t* t_index, const float* cond, int64_t n, const float* dout,
                          const float* const* masks, uint64_t seed, int64_t row_offset, int flags, float* const* grads, float* dx_t,
                          void* const* events, int n_events) {
  OSD_TRY(check_ready(h));
  OSD_TRY(check_rows(n));
  if (!x_t || !t_index || !cond || !dout || !grads || n == 0) { set_error("null tensor or empty batch"); return OSD_EINVAL; }
  if (h->saved_rows != n) { set_error("osd_denoiser_backward needs the activations of an osd_denoiser_forward_train call on the same %lld rows", (long long)n); return OSD_ESTATE; }
  const Arch& a = h->arch;
  const int n_buckets = a.n_blocks + 2;
  if (events && n_events != n_buckets) { set_error("expected %d events (osd_grad_buckets), got %d", n_buckets, n_events); return OSD_EINVAL; }
  for (int i = 0; i < a.pm.n_params; ++i)
    if (!grads[i]) { set_error("grads[%d] is null", i); return OSD_EINVAL; }
  OSD_TRY(check_row_offset(row_offset, n));
  OSD_HIP(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  TrainWs W;
  carve_train(a, h->train_arena, n, nullptr, &W);       // same carving as the forward call: pointers to its activations
  const int* t_idx = nullptr;
  OSD_TRY(sanitize_t(h, s, t_index, n, &t_idx));
  ZeroList zl{};
  add_backward_zeros(a, W, grads, &zl);
  OSD_HIP(launch_zero_many(s, zl));
  OSD_TRY(backward_from(h, s, W, x_t, a.D, t_idx, cond, n, dout, (flags & OSD_F_TRAIN_MODE) != 0, masks, seed, (uint32_t)row_offset, grads, dx_t, events));
  if (flags & OSD_F_SYNC) OSD_HIP(hipStreamSynchronize(s));
  return OSD_OK;
}

int osd_set_constraints(osd_handle* h, const osd_constraints* c) {
  if (!h) { set_error("null handle"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  if (h->stream) OSD_HIP(hipStreamSynchronize(h->stream));
  ConsPlan fresh;
  if (c) {
    if (c->pathway_weight < 0 || c->mutexpr_weight < 0) { set_error("constraint weights must be >= 0"); return OSD_EINVAL; }
    OSD_TRY(cons_build_plan(c->pathway_offsets, c->pathway_members, c->n_pathways, c->cols_a, c->n_a, c->cols_b, c->n_b, h->arch.D, &fresh));
  }
  cons_free_plan(&h->cons);
  h->cons = fresh;
  h->w_pathway = c ? c->pathway_weight : 0.0;
  h->w_mutexpr = c ? c->mutexpr_weight : 0.0;
  return OSD_OK;
}

int osd_get_loss_parts(osd_handle* h, float* parts_host3) {
  if (!h || !parts_host3) { set_error("null argument"); return OSD_EINVAL; }
  if (!h->parts_dev || !(h->cons.n_pathways > 0 || h->cons.n_a > 0)) { set_error("no constraint losses are configured on this handle"); return OSD_ESTATE; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  OSD_HIP(hipMemcpyAsync(parts_host3, h->parts_dev, 12, hipMemcpyDeviceToHost, h->stream));
  OSD_HIP(hipStreamSynchronize(h->stream));
  return OSD_OK;
}

static int clip_adamw(hipStream_t stream, double* norm_ws, float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                      double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm, int64_t step, float* grad_norm_out) {
  AdamArgs a{};
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.decay = (float)(1.0 - lr * weight_decay);
  a.one_minus_b1 = (float)(1.0 - beta1);
  a.b2 = (float)beta2;
  a.one_minus_b2 = (float)(1.0 - beta2);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  a.neg_step_size = (float)(-(lr / bc1));
  a.max_norm = (float)max_norm;
  OSD_HIP(launch_clip_adamw(stream, param, grad, exp_avg, exp_avg_sq, numel, a, norm_ws, grad_norm_out));
  return OSD_OK;
}

int osd_clip_adamw_step(osd_handle* h, float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, double lr,
                        double beta1, double beta2, double eps, double weight_decay, double max_norm, int64_t step, float* grad_norm_out) {
  if (!h || !param || !grad || !exp_avg || !exp_avg_sq) { set_error("null argument"); return OSD_EINVAL; }
  if (numel <= 0 || step < 1) { set_error("numel and step must be positive"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(h->cfg.device));
  if (!h->normsq_dev) OSD_HIP(hipMalloc((void**)&h->normsq_dev, 256 * sizeof(double)));
  // the handle's own optimizer is about to change parameters it may hold derived copies of (t_emb table, packed input_proj /
  // output_proj, chain and bf16x3 weight copies): the next forward / sampling entry point refreshes them (api.hip: ensure_packed)
  h->w_packed_stale = true;
  return clip_adamw(h->stream, h->normsq_dev, param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, max_norm, step,
                    grad_norm_out);
}

int osd_nn_clip_adamw_step(void* stream, int device, double* normsq_ws, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                           int64_t numel, double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                           int64_t step, float* grad_norm_out) {
  if (!normsq_ws || !param || !grad || !exp_avg || !exp_avg_sq) { set_error("null argument"); return OSD_EINVAL; }
  if (numel <= 0 || step < 1) { set_error("numel and step must be positive"); return OSD_EINVAL; }
  OSD_HIP(hipSetDevice(device));
  return clip_adamw((hipStream_t)stream, normsq_ws, param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, max_norm, step,
                    grad_norm_out);
}

}  // extern "C"
