// chain.hip -- host side of the persistent reverse-chain kernel (chain.h): eligibility, workspace, launch, status.
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <vector>
#include "chain.h"
#include "handle.h"
#include "kernels.h"
#include "fwd.h"

namespace osd {

static int64_t up64(int64_t v) { return (v + 63) / 64 * 64; }

// The chain kernel covers the 128 x 128 tile with GroupNorm groups of 32 or 64 channels (block widths 256 / 512, the
// BASELINE shape and its neighbours) in eval mode; everything else runs on the per-layer kernels.
bool chain_supported(const Arch& a) {
  if (a.H0 % 4) return false;              // D % 4 != 0 runs on the padded chain state (handle.h: Dp)
  for (int c : a.block_out)
    if (c != 256 && c != 512) return false;
  if (a.H0 != 256 && a.H0 != 512) return false;
  if ((int)a.layers.size() + 2 > CHAIN_MAX_LAYERS) return false;
  for (const LayerDesc& l : a.layers)
    if (l.K1 % BK || (l.K1 + l.K2) % BK) return false;        // panel switch on a K-step boundary, no K tail (block weights are not padded)
  return true;
}

struct ChainDev { int occ = 0; int cus = 0; bool ready = false; };
static ChainDev g_chain_dev[16];

static int chain_device_limits(int device, int* max_grid) {
  if (device < 0 || device >= 16) { set_error("device %d out of range", device); return OSD_EINVAL; }
  ChainDev& d = g_chain_dev[device];
  if (!d.ready) {
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_BYTES));
#ifdef OSD_DIAG
    OSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_BYTES));
#endif
    int occ = 0;
    OSD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, chain_kernel<false>, NTHREADS, CHAIN_LDS_BYTES));
    hipDeviceProp_t prop;
    OSD_HIP(hipGetDeviceProperties(&prop, device));
    d.occ = occ < CHAIN_WPS ? occ : CHAIN_WPS;      // two 64 KB tiles per CU by design
    d.cus = prop.multiProcessorCount;
    d.ready = true;
  }
  *max_grid = d.occ * d.cus;
  return OSD_OK;
}

// Which chain kernel.  Measured on MI355X (tools/probes/engine_sizes.py, D = 2000, T = 200, M patient-steps/s):
//   rows      per-layer   workspace chain   LDS-resident chain
//    8 192      17.8           5.0               11.4
//   10 240      13.9           6.3               14.3
//   12 288      16.2           7.6               17.1
//   16 384      20.1           9.9               22.5
//   20 480      16.4          12.3               22.6
//   28 672      19.6          17.1               22.3
//   32 768      21.7          19.5               22.4
//   49 152      21.9          19.7               22.4
//   65 536      22.2          22.5               22.4
//   81 920      22.3          23.0               22.4
//   98 304      22.5          23.0               22.4
// The LDS-resident kernel (chain_panel.h) runs at 22.4 M from 16 384 rows on -- 64-row units keep every CU busy and its queue has
// no step boundary -- where the per-layer kernels are fast only at whole rounds of their 128 x 128 tiles (multiples of 16 384
// rows) and the workspace chain needs a 128-row tile for each of its 512 slots; from there on the workspace chain leads by
// 1-2.5 % (two independent workgroups per CU hide each other's epilogues).  auto: workspace chain from
// 512 tiles on, LDS-resident chain from 5/8 of the CUs' worth of units (10 240 rows), per-layer kernels below.
static bool panel_window(osd_handle* h, int64_t n) {
  if (!panel_chain_supported(h)) return false;
  const int slots = panel_chain_slots(h);
  if (slots < 1) return false;
  const int64_t units = (n + 63) / 64;
  return units * 8 >= (int64_t)slots * 5;
}

static bool chain_use_panel(osd_handle* h, int64_t n) {
  if (h->chain_variant == 1) return false;
  if (h->chain_variant == 2) return panel_chain_supported(h);
  int max_grid = 0;
  if (chain_device_limits(h->cfg.device, &max_grid) != OSD_OK) return false;
  if ((n + ChainTile::BP - 1) / ChainTile::BP >= (int64_t)max_grid) return false;       // the workspace chain has a tile per slot
  return panel_window(h, n);
}

// The squad chain: asked for by name, or auto's choice for a small batch in the small-batch mode (osd_set_option("input_splitk",
// != 0): the switch that already trades "results do not depend on the batch size, bit for bit" for latency; the library default
// keeps that invariance, utils/generate.py's front end turns the mode on).  An explicit sampler = chain keeps the kernels that
// are bit-identical to the per-layer engine.
bool chain_uses_squad(osd_handle* h, int64_t n) {
  if (!squad_window(h, n)) return false;
  return h->chain_variant == 3 || (h->sampler == 0 && h->input_splitk != 0 && !h->splitk_suspended);
}

// 0 = per-layer kernels (eager or hipGraph), 1 = persistent chain kernel
int chain_pick_engine(osd_handle* h, int64_t n, int flags) {
  if (h->sampler == 2) return 0;
  if ((flags & OSD_F_TRAIN_MODE) && h->cfg.dropout_p > 0.f) return 0;       // dropout inside the chain: per-layer kernels
  if (!chain_supported(h->arch)) return 0;
  if (h->sampler == 1) return 1;
  int max_grid = 0;
  if (chain_device_limits(h->cfg.device, &max_grid) != OSD_OK || max_grid < 2) return 0;
  const int64_t n_tiles = (n + ChainTile::BP - 1) / ChainTile::BP;
  // every slot gets a tile: with fewer tiles than resident workgroups the workspace chain idles CUs (384 tiles: 19.4 M
  // patient-steps/s against 21.8 M for the per-layer engine, which also tiles the features; 512 tiles: 22.2 vs 22.1; beyond
  // that the chain kernel leads -- tools/probes/engine_crossover.py); below that, the LDS-resident chain where it leads
  if (n_tiles >= (int64_t)max_grid) return 1;
  if (h->chain_variant != 1 && h->chain_variant != 3 && panel_window(h, n)) return 1;
  // small batches (the reference's own generation sizes) in the small-batch mode: every 32-patient squad of the chain resident at once
  if (chain_uses_squad(h, n)) return 1;
  return 0;
}

int chain_ensure_buf(float** p, int64_t* cap, int64_t floats, hipStream_t s) {
  if (*cap >= floats) return OSD_OK;
  if (*p) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
  void* q = nullptr;
  if (hipMalloc(&q, (size_t)floats * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %lld bytes failed", (long long)floats * 4); return OSD_ENOMEM; }
  *p = (float*)q;
  *cap = floats;
  return OSD_OK;
}

// sync words: [status, queue, pad x2 | cu arrivals x2048 | progress x n_tiles], zeroed before every chain
int chain_ensure_sync(osd_handle* h, int64_t n_tiles, hipStream_t s) {
  const int64_t words = 4 + 2048 + ((n_tiles + 3) / 4) * 4;
  if (h->chain_sync_words < words) {
    if (h->chain_sync) { OSD_HIP(hipStreamSynchronize(s)); OSD_HIP(hipFree(h->chain_sync)); h->chain_sync = nullptr; h->chain_sync_words = 0; }
    if (hipMalloc((void**)&h->chain_sync, (size_t)words * 4) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->chain_sync_words = words;
  }
  OSD_HIP(hipMemsetAsync(h->chain_sync, 0, (size_t)words * 4, s));
  return OSD_OK;
}

// Waits for the chain launched last and reads its status word into *st.  The wait is a hipStreamQuery poll with a wall-clock
// budget (10 x the estimated run time + 2 s, or osd_set_option("chain_wall_budget_ms")): the in-kernel spin budget cannot see a
// workgroup that is stuck at a barrier, and a blind hipStreamSynchronize would then block the caller forever.  On expiry the
// host raises CHAIN_ABORT in the status word through a second stream -- every dependency wait and every unit boundary checks
// that word, so a kernel whose workgroups are merely waiting drains within microseconds -- and gives the device a grace period;
// OSD_EHIP only if the kernel still does not end (the stream is then unusable and the message says so).
static int chain_wait_status(osd_handle* h, unsigned* st) {
  *st = CHAIN_OK;
  if (!h->chain_pending || !h->chain_sync) return OSD_OK;
  hipStream_t s = h->stream;
  const double budget_ms = h->chain_wall_budget_ms > 0 ? (double)h->chain_wall_budget_ms : 10.0 * h->chain_expected_ms + 2000.0;
  auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; };
  auto poll = [&](double limit_ms) -> hipError_t {
    const double t0 = now_ms();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipErrorNotReady) return e;
      const double el = now_ms() - t0;
      if (el > limit_ms) return hipErrorNotReady;
      if (el > 2.0) usleep(el > 100.0 ? 1000 : 100);      // short chains: spin; long ones: sleep between polls
    }
  };
  hipError_t e = poll(budget_ms);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    if (!h->abort_stream) OSD_HIP(hipStreamCreateWithFlags(&h->abort_stream, hipStreamNonBlocking));
    static const unsigned abort_word = CHAIN_ABORT;
    OSD_HIP(hipMemcpyAsync(h->chain_sync, &abort_word, 4, hipMemcpyHostToDevice, h->abort_stream));
    OSD_HIP(hipStreamSynchronize(h->abort_stream));
    e = poll(std::max(5000.0, 0.25 * budget_ms));
    if (e == hipErrorNotReady) {
      (void)hipGetLastError();
      set_error("the reverse-chain kernel did not finish within %.0f ms and did not react to the abort flag (a workgroup is stuck outside a "
                "dependency wait); the handle's stream is blocked -- destroy the process's HIP context", budget_ms);
      return OSD_EHIP;
    }
  }
  OSD_HIP(e);
  OSD_HIP(hipMemcpyAsync(st, h->chain_sync, 4, hipMemcpyDeviceToHost, s));
  OSD_HIP(hipStreamSynchronize(s));
  h->chain_pending = false;
  return OSD_OK;
}

// Blocks until the previous chain's status word is known; OSD_EHIP if that chain gave up in a dependency wait.
int chain_check_status(osd_handle* h) {
  unsigned st = CHAIN_OK;
  OSD_TRY(chain_wait_status(h, &st));
  if (st != CHAIN_OK) {
    set_error("the reverse-chain kernel gave up waiting for a row tile of an earlier step (status %u: %s): its results are invalid; "
              "call osd_sample_chain with OSD_F_SYNC to have such a chain re-run on the per-layer kernels, or osd_set_option(\"sampler\", 2)",
              st, st == CHAIN_ABORT ? "host wall-clock budget" : "in-kernel spin budget");
    return OSD_EHIP;
  }
  return OSD_OK;
}

// Synchronous variant for osd_sample_chain(OSD_F_SYNC): *gave_up = 1 when the chain's results are invalid (the caller re-runs
// the chain on the per-layer kernels), errors only for HIP failures and a device that does not come back.
int chain_finish(osd_handle* h, int* gave_up) {
  unsigned st = CHAIN_OK;
  OSD_TRY(chain_wait_status(h, &st));
  *gave_up = st != CHAIN_OK;
  return OSD_OK;
}

int chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
              float* x_out, float* mut_mask_out) {
  const Arch& a = h->arch;
  const int T = a.T, H0 = a.H0;
  hipStream_t s = h->stream;
  OSD_TRY(chain_check_status(h));
  if (chain_uses_squad(h, n)) {
    h->last_chain_variant = 3;
    return squad_chain_run(h, cond, n, x_T, noises, seed, row_offset, x_out, mut_mask_out);
  }
  if (chain_use_panel(h, n)) {
    h->last_chain_variant = 2;
    return panel_chain_run(h, cond, n, x_T, noises, seed, row_offset, x_out, mut_mask_out);
  }
  h->last_chain_variant = 1;
  // D % 4 != 0: the kernel works on an internal copy of the state with rows of Dp = roundup(D, 4) floats (pad columns start at
  // zero, meet zero weights in input_proj and get zero eps from the packed output_proj) and the result is copied out at the end
  const bool padded = h->w_out_packed != nullptr;
  if (padded && noises) { set_error("internal: injected draws with D %% 4 != 0 run on the per-layer kernels"); return OSD_EUNSUPPORTED; }
  const int D = padded ? h->Dp : a.D;
  float* const x_state = padded ? nullptr : x_out;
  int max_grid = 0;
  OSD_TRY(chain_device_limits(h->cfg.device, &max_grid));
  if (max_grid < 1) { set_error("the chain kernel does not fit this device"); return OSD_EUNSUPPORTED; }
  const int BP = ChainTile::BP;
  const int n_tiles = (int)((n + BP - 1) / BP);
  // chain_grid > 0 sets the workgroup count (tests): below the tile count it forces every hand-off across workgroups, above it
  // the surplus workgroups start on later steps of a tile and wait for the earlier ones
  int grid = std::min(n_tiles, max_grid);
  if (h->chain_grid > 0) grid = (int)std::min<int64_t>(std::min(h->chain_grid, max_grid), (int64_t)n_tiles * T);

  // ---- per-slot activation workspace.  Buffers: h0, and (mid, out) of every block, each [128][C]; a buffer is live from the
  // layer that writes it to the last layer that reads it (block outputs of the encoder live on until their decoder block pops
  // them), and dead buffers are reused first-fit: 0.9 MB per slot instead of 1.9 MB at the BASELINE shape, which is what the
  // 512 slots keep cycling through L2 / Infinity Cache ----
  ChainArgs ca{};
  const int nbuf = 1 + 2 * a.n_blocks;                 // 0: h0, 1 + 2b: mid[b], 2 + 2b: out[b]
  std::vector<int> width(nbuf), def(nbuf), last(nbuf);
  width[0] = H0; def[0] = 0; last[0] = 1;              // layer index: 0 input_proj, 1 + 2b / 2 + 2b the halves of block b, 1 + 2 n_blocks output_proj
  for (int b = 0; b < a.n_blocks; ++b) {
    width[1 + 2 * b] = a.block_out[b]; def[1 + 2 * b] = 1 + 2 * b; last[1 + 2 * b] = 2 + 2 * b;
    width[2 + 2 * b] = a.block_out[b]; def[2 + 2 * b] = 2 + 2 * b; last[2 + 2 * b] = 3 + 2 * b;     // next block's first half, or output_proj
    if (a.layers[2 * b].K2 > 0) {
      const int skip_block = a.n_enc - 1 - (b - a.n_enc - 1);
      last[2 + 2 * skip_block] = std::max(last[2 + 2 * skip_block], 1 + 2 * b);
    }
  }
  std::vector<int64_t> boff(nbuf, -1);
  int64_t off = 0;
  {
    struct Seg { int64_t off, len; int free_from; };    // free_from: first layer index that may overwrite it
    std::vector<Seg> segs;
    for (int i = 0; i < nbuf; ++i) {                     // buffers are defined in increasing layer order
      const int64_t need = up64((int64_t)BP * width[i]);
      int pick = -1;
      for (int sgi = 0; sgi < (int)segs.size(); ++sgi)
        if (segs[sgi].free_from <= def[i] && segs[sgi].len >= need && (pick < 0 || segs[sgi].len < segs[pick].len)) pick = sgi;
      if (pick < 0) { segs.push_back({off, need, 0}); pick = (int)segs.size() - 1; off += need; }
      boff[i] = segs[pick].off;
      // a layer reads its inputs while it writes its output: the segment is reusable by layers AFTER the last reader
      segs[pick].free_from = last[i] + 1;
    }
  }
  const int o_h0 = (int)boff[0];
  std::vector<int> o_mid(a.n_blocks), o_out(a.n_blocks);
  for (int b = 0; b < a.n_blocks; ++b) { o_mid[b] = (int)boff[1 + 2 * b]; o_out[b] = (int)boff[2 + 2 * b]; }
  ca.ws_stride = off;
  OSD_TRY(chain_ensure_buf(&h->chain_ws, &h->chain_ws_floats, (int64_t)max_grid * off, s));
  ca.ws = h->chain_ws;

  // ---- conditioning for all rows, hoisted out of the chain (loop-invariant in eval mode): ce1, ce2, cproj padded to whole tiles ----
  const int64_t rows_pad = (int64_t)n_tiles * BP;
  const int64_t c_off_ce2 = up64(n * 64), c_off_cp = c_off_ce2 + up64(n * 64);
  OSD_TRY(chain_ensure_buf(&h->chain_cond, &h->chain_cond_floats, c_off_cp + up64(rows_pad * H0), s));
  FwdWs cw;
  cw.ce1 = h->chain_cond; cw.ce2 = h->chain_cond + c_off_ce2; cw.cproj = h->chain_cond + c_off_cp;
  OSD_TRY(run_cond(h, s, cond, n, cw));
  if (rows_pad > n) OSD_HIP(hipMemsetAsync(cw.cproj + n * H0, 0, (size_t)(rows_pad - n) * H0 * 4, s));

  // ---- x_T ----
  float* xs = x_state;
  if (padded) {
    OSD_TRY(chain_ensure_buf(&h->chain_xpad, &h->chain_xpad_floats, n * (int64_t)D, s));
    xs = h->chain_xpad;
    OSD_HIP(hipMemsetAsync(xs, 0, (size_t)n * D * 4, s));
  }
  if (x_T) OSD_HIP(launch_copy2d(s, x_T, a.D, xs, D, n, a.D));
  else OSD_HIP(launch_fill_randn(s, xs, D, n, a.D, seed, (uint32_t)row_offset, (uint32_t)T, TAG_POSTERIOR));

  OSD_TRY(chain_ensure_sync(h, n_tiles, s));
  ca.status = h->chain_sync;
  ca.queue = h->chain_sync + 1;
  ca.cu_arrivals = h->chain_stagger > 0 ? h->chain_sync + 4 : nullptr;
  ca.progress = h->chain_sync + 4 + 2048;
  ca.stagger = h->chain_stagger;
  ca.stamps = h->chain_stamps;
  ca.spin_budget = h->chain_spin_budget;    // default 5 s of s_memrealtime ticks: a unit takes milliseconds

  // ---- layer table ----
  const ParamMap& pm = a.pm;
  int nl = 0;
  {
    ChainLayer& L = ca.L[nl++];
    L.A = h->w_in_packed; L.lda = h->w_in_ld; L.K = D; L.K0 = D; L.F = H0;
    L.in0 = -1; L.ld0 = D; L.in1 = 0; L.ld1 = 0; L.out = o_h0; L.ldo = H0; L.kind = CK_INPUT;
    L.bias = h->params[pm.in_b]; L.gamma = nullptr; L.beta = nullptr;
  }
  int cur = o_h0, cur_w = H0;
  for (int b = 0; b < a.n_blocks; ++b) {
    const LayerDesc& l1 = a.layers[2 * b];
    const LayerDesc& l2 = a.layers[2 * b + 1];
    ChainLayer& A1 = ca.L[nl++];
    A1.A = h->params[l1.w]; A1.lda = l1.K1 + l1.K2; A1.K = l1.K1 + l1.K2; A1.K0 = l1.K2 > 0 ? l1.K1 : l1.K1 + l1.K2; A1.F = l1.N;
    A1.in0 = cur; A1.ld0 = cur_w; A1.in1 = 0; A1.ld1 = 0;
    if (l1.K2 > 0) {
      const int skip_block = a.n_enc - 1 - (b - a.n_enc - 1);      // LIFO: decoder j pops encoder n_enc-1-j
      A1.in1 = o_out[skip_block]; A1.ld1 = a.block_out[skip_block];
    }
    A1.out = o_mid[b]; A1.ldo = l1.N; A1.kind = l1.gw == 64 ? CK_GN64 : CK_GN32;
    A1.bias = h->params[l1.b]; A1.gamma = h->params[l1.gamma]; A1.beta = h->params[l1.beta];
    ChainLayer& A2 = ca.L[nl++];
    A2.A = h->params[l2.w]; A2.lda = l2.K1; A2.K = l2.K1; A2.K0 = l2.K1; A2.F = l2.N;
    A2.in0 = o_mid[b]; A2.ld0 = l1.N; A2.in1 = 0; A2.ld1 = 0;
    A2.out = o_out[b]; A2.ldo = l2.N; A2.kind = l2.gw == 64 ? CK_GN64 : CK_GN32;
    A2.bias = h->params[l2.b]; A2.gamma = h->params[l2.gamma]; A2.beta = h->params[l2.beta];
    cur = o_out[b]; cur_w = l2.N;
  }
  {
    ChainLayer& L = ca.L[nl++];
    L.A = padded ? h->w_out_packed : h->params[pm.out_w]; L.lda = cur_w; L.K = cur_w; L.K0 = cur_w; L.F = D;
    L.in0 = cur; L.ld0 = cur_w; L.in1 = 0; L.ld1 = 0; L.out = 0; L.ldo = 0; L.kind = CK_POST;
    L.bias = padded ? h->b_out_packed : h->params[pm.out_b]; L.gamma = nullptr; L.beta = nullptr;
  }
  ca.n_layers = nl;
  ca.x = xs; ca.D = D; ca.n = (int)n; ca.n_tiles = n_tiles;
  ca.cproj = cw.cproj; ca.ldc = H0; ca.temb = h->d_temb; ca.ldt = H0; ca.coef = h->d_coef;
  ca.z = noises; ca.ldzz = D; ca.z_step_stride = (long long)n * D; ca.z_t_first = T - 1;
  ca.seed = seed; ca.row_offset = (uint32_t)row_offset;
  ca.mut_mask = mut_mask_out; ca.mutation_dim = h->cfg.mutation_dim;

  // ---- launches: the whole chain in one, or segments of chain_steps_per_launch steps (progress carries over) ----
  const int seg = h->chain_steps_per_launch > 0 ? h->chain_steps_per_launch : T;
  const int n_launch = (T + seg - 1) / seg;
  OSD_HIP(hipStreamSynchronize(s));        // the host copies are about to be rewritten: earlier uploads must have been consumed
  if (h->chain_args_cap < n_launch) {
    if (h->chain_args_dev) { OSD_HIP(hipFree(h->chain_args_dev)); h->chain_args_dev = nullptr; }
    free(h->chain_args_host);
    h->chain_args_cap = 0;
    h->chain_args_host = malloc((size_t)n_launch * sizeof(ChainArgs));
    if (!h->chain_args_host) { set_error("out of host memory"); return OSD_ENOMEM; }
    if (hipMalloc(&h->chain_args_dev, (size_t)n_launch * sizeof(ChainArgs)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return OSD_ENOMEM; }
    h->chain_args_cap = n_launch;
  }
  ChainArgs* const host_args = static_cast<ChainArgs*>(h->chain_args_host);
  int launch = 0;
  for (int done = 0; done < T; done += seg) {
    ca.t_first = T - 1 - done;
    ca.n_steps = std::min(seg, T - done);
    ca.base_done = (unsigned)done;
    if (done > 0) {                       // per-launch words: the unit queue and the CU arrival counters (status and progress carry over)
      OSD_HIP(hipMemsetAsync(ca.queue, 0, 4, s));
      if (ca.cu_arrivals) OSD_HIP(hipMemsetAsync(ca.cu_arrivals, 0, 2048 * 4, s));
    }
    // the argument block of this launch: host copy kept alive in the handle, device copy read by the kernel
    host_args[launch] = ca;
    const ChainArgs* dargs = static_cast<const ChainArgs*>(h->chain_args_dev) + launch;
    OSD_HIP(hipMemcpyAsync(const_cast<ChainArgs*>(dargs), &host_args[launch], sizeof(ChainArgs), hipMemcpyHostToDevice, s));
    ++launch;
#ifdef OSD_DIAG
    if (ca.stamps) hipLaunchKernelGGL(chain_kernel<true>, dim3(grid), dim3(NTHREADS), CHAIN_LDS_BYTES, s, dargs);
    else
#endif
    hipLaunchKernelGGL(chain_kernel<false>, dim3(grid), dim3(NTHREADS), CHAIN_LDS_BYTES, s, dargs);
    OSD_HIP(hipGetLastError());
  }
  if (padded) OSD_HIP(launch_copy2d(s, xs, D, x_out, a.D, n, a.D));
  h->chain_pending = true;
  // run-time estimate for the host's wall-clock budget: a unit (128 rows through every layer) runs at ~0.24 TFLOP/s per
  // resident workgroup when two share a CU (2.8 ms at the BASELINE shape)
  {
    double flop_row = 0;
    for (int l = 0; l < nl; ++l) flop_row += 2.0 * ca.L[l].K * ca.L[l].F;
    const double unit_ms = 128.0 * flop_row / 0.237e12 * 1e3;
    double rounds = (double)(((int64_t)n_tiles * T + grid - 1) / grid);
    if (grid >= n_tiles) rounds = std::max(rounds, (double)T);        // the steps of a tile are serial
    h->chain_expected_ms = rounds * unit_ms;
  }
  return OSD_OK;
}

void chain_free(osd_handle* h) {
  hipError_t e = hipSuccess;
  if (h->chain_ws) e = hipFree(h->chain_ws);
  if (h->chain_cond) e = hipFree(h->chain_cond);
  if (h->chain_sync) e = hipFree(h->chain_sync);
  if (h->chain_args_dev) e = hipFree(h->chain_args_dev);
  if (h->abort_stream) { e = hipStreamDestroy(h->abort_stream); h->abort_stream = nullptr; }
  panel_chain_free(h);
  squad_chain_free(h);
  h->chain_args_dev = nullptr;
  free(h->chain_args_host);
  h->chain_args_host = nullptr;
  h->chain_args_cap = 0;
  (void)e;
  h->chain_ws = h->chain_cond = nullptr;
  h->chain_sync = nullptr;
}

}  // namespace osd
