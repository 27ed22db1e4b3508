// handle.h -- host-side state behind osd_handle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/osdiff.h"
#include "constraints.h"
#include "batch_src.h"

namespace osd {

void set_error(const char* fmt, ...);

#define OSD_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::osd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* reported: do not let it resurface in a later launch check */            \
      return OSD_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

#define OSD_TRY(call)               \
  do {                              \
    int r__ = (call);               \
    if (r__ != OSD_OK) return r__;  \
  } while (0)

// One Linear of the denoiser trunk (with or without GroupNorm+SiLU behind it).
struct LayerDesc {
  int K1, K2;          // input panel widths (K2 > 0: concat-free decoder input)
  int N;               // output width
  int w, b;            // parameter indices
  int gamma, beta;     // -1 when no GroupNorm
  int gw;              // channels per group (N/8) when GroupNorm
  int block;           // block index (execution order) or -1
  int half;            // 0 first Linear of the block (dropout behind it), 1 second
};

// Parameter indices in named_parameters() order.
struct ParamMap {
  int ce0_w, ce0_b, ce2_w, ce2_b;
  int in_w, in_b, cp_w, cp_b, tp_w, tp_b;
  int out_w, out_b;
  int n_params;
  std::vector<int64_t> numel;
};

struct Arch {
  int D, H0, cond_dim, time_dim, cond_width, T;
  int n_blocks, n_enc;
  std::vector<int> hidden;
  std::vector<LayerDesc> layers;   // 2 per block, execution order
  std::vector<int> block_out;      // output width of block i
  ParamMap pm;
  int64_t act_floats_per_row;      // forward workspace per row
};

int build_arch(const osd_config& cfg, Arch* a);

// Forward activations of one row chunk (all device pointers into one arena).
struct FwdWs {
  float* ce1;     // [n][64]   SiLU(Linear(cond))
  float* ce2;     // [n][64]   condition embedding
  float* cproj;   // [n][H0]
  float* h0;      // [n][H0]
  std::vector<float*> mid;   // per block: first-half output (post dropout)  [n][C]
  std::vector<float*> out;   // per block: block output                      [n][C]
  // training extras (null in inference)
  std::vector<float*> z1, z2;        // pre-norm activations of both halves
  std::vector<float*> st1, st2;      // (mean, rstd) [n][8][2]
};

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  float* arena = nullptr;
  int64_t arena_floats = 0;
  int* t_dev = nullptr;
  hipGraph_t graph = nullptr;        // graph of the last captured reverse step (kept alive
  hipGraphExec_t exec = nullptr;     // until its replays have drained)
};

}  // namespace osd

struct osd_handle {
  osd_config cfg;
  osd::Arch arch;
  hipStream_t stream = nullptr;
  std::vector<const float*> params;
  bool have_schedule = false, have_weights = false;
  float* w_in_packed = nullptr;      // input_proj.weight zero-padded to [H0][roundup(D,32)] for the LDS-DMA kernel
  int w_in_ld = 0;
  bool w_packed_stale = false;      // a training step skipped the repack of w_in_packed / w_out_packed: refreshed lazily (api.hip: ensure_packed)
  // D % 4 != 0 (e.g. the reference's real dims 62 + 5054 + 26 = 5142): the reverse chain keeps its state in an internal buffer
  // whose rows are padded to Dp = roundup(D, 4) floats, so that every operand is 16-byte aligned and the LDS-DMA / FAST tile code
  // and the chain kernel apply; the pad columns carry finite values that only ever meet zero weights
  int Dp = 0;
  int train_ksplit = 1;              // osd_set_option("train_ksplit", 0|1): two wave groups per workgroup in the training forward's GEMMs (gemm_glds.h)
  int dual_dgrad = 1;                // osd_set_option("dual_dgrad", 0|1): a decoder block's two input dgrads in one launch (k_gnbwd.hip)
  int cond_bwd_fused = 1;            // osd_set_option("cond_bwd_fused", 0|1): the conditioning branch's backward below h0 as one launch (k_cond_bwd, k_train.hip)
  bool sq_wpk_t_fresh = false;       // the backward squads' transposed weight copies were packed by this step's forward launch (chain_squad.hip)
  int train_input_splitk = 0;        // osd_set_option("train_input_splitk"): K slices of input_proj in the training forward (0 / 1 = single pass)
  bool splitk_suspended = false;     // a chain-kernel fallback re-run in progress: no split-K (bit-identical to the chain kernel)
  int input_splitk = 0;              // osd_set_option("input_splitk"): 0 off (default: a row's result does not depend on how rows are chunked / sharded),
                                     // -1 auto (chunks with < 128 input_proj tiles), n = slices
  float* w_out_packed = nullptr;     // [Dp][H_last]: output_proj.weight + zero rows for the pad columns (allocated iff D % 4)
  float* b_out_packed = nullptr;     // [Dp]
  float* chain_xpad = nullptr; int64_t chain_xpad_floats = 0;     // padded chain state of the chain kernel [n][Dp]
  float *d_sqrt_ac = nullptr, *d_sqrt_1m = nullptr, *d_coef = nullptr, *d_time_emb = nullptr, *d_temb = nullptr;
  int64_t chunk_rows = 65536;
  int n_streams = 2;
  std::vector<osd::Slot> slots;
  osd::Slot main;            // workspace used by the single-stream entry points
  hipEvent_t fork_ev = nullptr;
  // training workspace
  float* train_arena = nullptr;
  int64_t train_arena_floats = 0;
  float* loss_dev = nullptr;
  int* t_san = nullptr;              // clamped copy of a caller-supplied t_index (sanitize_t)
  int64_t t_san_cap = 0;
  double* normsq_dev = nullptr;      // 256 per-workgroup partials of the gradient norm (osd_clip_adamw_step)
  // weight-gradient side stream of the backward pass and its fork/join events
  hipStream_t wgrad_stream = nullptr;
  std::vector<hipEvent_t> ev_pool;
  int two_stream_bwd = 1;            // osd_set_option("train_streams", 1|2)
  osd::BatchSrc batch_src{}; bool have_batch_src = false;      // osd_train_batch_source: one-shot source of the next training call's rows
  int64_t saved_rows = -1;           // rows of the last osd_denoiser_forward_train whose activations are still in the arena
  // constraint losses (osd_set_constraints); parts_dev = (mse, L_pc, L_me) of the last training call
  osd::ConsPlan cons;
  double w_pathway = 0.0, w_mutexpr = 0.0;
  float* parts_dev = nullptr;
  std::vector<void*> wg_plans;       // grouped weight-gradient launches (wgrad_group.hip): one cached work list per flush point
  int fused_gn_bwd = 1;              // osd_set_option("fused_gn_bwd", 0|1): GroupNorm backward inside the dgrad epilogue (group widths 32 / 64)
  int wgrad_mid_flush = 0;           // osd_set_option("wgrad_mid_flush", 0|1): also launch the decoder-half weight gradients mid-pass
  int grouped_wgrad = 1;             // osd_set_option("grouped_wgrad", 0|1)
  // persistent reverse-chain kernel (chain.h / chain.hip)
  int sampler = 0;                   // osd_set_option("sampler"): 0 auto, 1 chain kernel whenever the architecture allows, 2 per-layer kernels
  int chain_grid = 0;                // 0 = min(row tiles, resident slots); > 0 caps the workgroup count (tests: force cross-workgroup hand-offs)
  int chain_steps_per_launch = 0;    // 0 = the whole chain in one launch
  int chain_stagger = 30000;         // shader cycles between the starts of the two workgroups of a CU (0 = off)
  float* chain_ws = nullptr; int64_t chain_ws_floats = 0;
  float* chain_cond = nullptr; int64_t chain_cond_floats = 0;
  unsigned* chain_sync = nullptr; int64_t chain_sync_words = 0;
  void* chain_args_dev = nullptr;    // device copies of the launches' argument blocks (ChainArgs, chain.h)
  void* chain_args_host = nullptr;   // host copies of the same (kept alive while their uploads may be pending)
  int chain_args_cap = 0;
  bool chain_pending = false;        // a chain was launched whose status word has not been read yet
  unsigned long long chain_spin_budget = 500000000ull;   // osd_set_option("chain_spin_budget"): s_memrealtime ticks (100 MHz) a dependency wait may take (5 s)
  int64_t chain_wall_budget_ms = 0;  // osd_set_option("chain_wall_budget_ms"): host-side budget of a synchronous chain; 0 = 10 x the expected run time + 2 s
  double chain_expected_ms = 0.0;    // run-time estimate of the chain launched last (chain_run)
  int64_t chain_fallbacks = 0;       // chains that gave up and were re-run on the per-layer kernels (osd_get_option)
  hipStream_t abort_stream = nullptr;   // carries the host's abort flag to a chain kernel that overran its wall-clock budget
  unsigned long long* chain_stamps = nullptr;   // diagnostic builds (csrc/diag): device buffer of 8 counters per workgroup, else null
  int last_engine = 0;               // engine of the most recent osd_sample_chain (0 per-layer, 1 chain kernel)
  // LDS-resident variant of the chain kernel (chain_panel.h / chain_panel.hip)
  int chain_variant = 0;             // osd_set_option("chain_variant"): 0 auto (chain.hip: chain_use_panel / squad_window), 1 workspace chain (chain.h),
                                     // 2 LDS-resident chain where the architecture fits (else 1), 3 squad chain (chain_squad.h) where model and batch fit (else auto's choice without it)
  int last_chain_variant = 0;        // variant the most recent chain-kernel run used (osd_get_option)
  float* panel_wpk = nullptr; int64_t panel_wpk_floats = 0;   // fragment-ordered copies of the weights
  bool panel_wpk_valid = false;      // false after anything that may have changed the parameters: repacked by the next chain
  void* panel_args_dev = nullptr; void* panel_args_host = nullptr; int panel_args_cap = 0;
  // small-batch variant (chain_squad.h / chain_squad.hip)
  float* squad_wpk[2] = {nullptr, nullptr}; int64_t squad_wpk_floats[2] = {0, 0};      // fragment-ordered weights: [0] 32-patient panels, [1] 16-patient panels
  bool squad_wpk_valid[2] = {false, false};
  int last_squad_rp = 0;             // patients per panel of the squad chain that ran last (osd_get_option "last_squad_panel")
  int train_squad = 2;               // osd_set_option("train_squad"): the training forward trunk as one launch of squads (train_squad.h) from 2 048 rows on
  int squad_panel = 0;               // osd_set_option("squad_panel"): 0 auto (16-patient panels up to one 32-patient workgroup per CU), 16, 32
  void* squad_args_dev = nullptr; void* squad_args_host = nullptr; int squad_args_cap = 0;
  // bf16x3 split precision (gemm_bf3.h / split.hip)
  int precision = 0;                 // osd_set_option("precision"): 0 fp32 MFMA (default; the reference's arithmetic), 1 bf16x3 split on the bf16 matrix pipe
                                     // (fp32 accuracy, eval-mode sampling / forward of 256 / 512 wide trunks; everything else stays fp32)
  int last_precision = 0;            // precision the most recent forward / p_sample / sample call computed in (osd_get_option)
  void* split_plan = nullptr;        // weight planes (split.hip)
  bool split_valid = false;          // false after anything that may have changed the parameters: repacked by the next split-precision call
  // osd_profile_step: when non-null, run_trunk records prof_events[prof_i++] after every launch
  std::vector<hipEvent_t>* prof_events = nullptr;
  int prof_i = 0;
};
