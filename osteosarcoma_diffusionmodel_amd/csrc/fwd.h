// fwd.h -- forward-pass plumbing shared by api.hip and train.hip.
#pragma once
#include <vector>
#include "handle.h"
#include "gemm.h"

namespace osd {

struct TrunkIn {
  const float* x; int ldx; int64_t n;
  float* in_slabs; int in_slices; // > 1: input_proj split-K over that many slices (k_fused.hip), slabs = in_slices x n x H0 floats
  float* gn_slabs; int gn_slices; // > 1: the >= 512-deep Linear+GroupNorm layers split K over workgroups (k_fused.hip: launch_gn_silu_splitk); slabs =
                                 // (gn_slices + 1) x n x max width floats.  Eval-mode sampling of small batches only (another fp32 summation order)
  bool ksplit;                   // training-sized batches: ask for the two-wave-group GEMM variant (gemm_glds.h, NG = 2) in every layer
  bool a_unpacked;               // input_proj reads input_proj.weight itself (clamped at D, GemmArgs::a_kmax) instead of the packed copy;
                                 // x must then be zero in the columns [D, kx) (training: the library's own x_t buffer)
  int kx;                        // K extent of input_proj: 0 = D; Dp when x is the padded chain state (handle.h)
  const int* t_index;            // per-row t (training) or null
  const int* t_dev; int t_imm;   // shared t: device counter (sampling chain) or immediate
  bool input_only;               // stop after input_proj (the blocks run elsewhere: train_squad.h)
  bool train;                    // dropout active
  bool save;                     // keep pre-norm activations + GroupNorm statistics for backward
  const float* const* masks;     // injected keep-masks per block, or null -> Philox
  uint64_t seed; uint32_t row_offset; uint32_t drop_step; const int* drop_step_dev;
};

int ensure_arena(Slot* s, int64_t floats);
int64_t carve_fwd(const Arch& a, float* base, int64_t n, bool train, FwdWs* ws);
int run_cond(osd_handle* h, hipStream_t s, const float* cond, int64_t n, const FwdWs& ws);
int run_trunk(osd_handle* h, hipStream_t s, const FwdWs& ws, const TrunkIn& in);
int refresh_derived(osd_handle* h, hipStream_t s, bool pack_in_w = true);
int ensure_packed(osd_handle* h, hipStream_t s);
GemmArgs output_proj_args(osd_handle* h, const FwdWs& ws, int64_t n, bool padded = false);
int check_ready(osd_handle* h);
int check_rows(int64_t n);
// chain.hip
bool chain_supported(const Arch& a);
int chain_pick_engine(osd_handle* h, int64_t n, int flags);
bool chain_uses_squad(osd_handle* h, int64_t n);
int chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
              float* x_out, float* mut_mask_out);
int chain_check_status(osd_handle* h);
int chain_finish(osd_handle* h, int* gave_up);
void chain_free(osd_handle* h);
int chain_ensure_buf(float** p, int64_t* cap, int64_t floats, hipStream_t s);
int chain_ensure_sync(osd_handle* h, int64_t n_tiles, hipStream_t s);
// chain_panel.hip
bool panel_chain_supported(const osd_handle* h);
int panel_chain_slots(osd_handle* h);
int panel_chain_pack(osd_handle* h, hipStream_t s);
int panel_chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
                    float* x_out, float* mut_mask_out);
void panel_chain_free(osd_handle* h);
hipError_t launch_pack_fragments(hipStream_t s, const float* w, int ldw, int F, int K, int nfbg, int K8, float* dst);
// chain_squad.hip
bool squad_chain_supported(const osd_handle* h);
bool squad_window(osd_handle* h, int64_t n);
int squad_chain_run(osd_handle* h, const float* cond, int64_t n, const float* x_T, const float* noises, uint64_t seed, int64_t row_offset,
                    float* x_out, float* mut_mask_out);
void squad_chain_free(osd_handle* h);
// train_squad.h (host side in chain_squad.hip): the training forward trunk as one launch of squads
int64_t train_squad_act_floats(const Arch& a, int64_t* wpk_floats);
bool train_squad_ok(const osd_handle* h, int64_t n);
// train_squad_bwd.h: the dgrad chain (single-GPU steps, fused GroupNorm backward) as one launch of squads
struct TrainSquadBwdBufs {
  float* const* g_out; float* const* g_z2; float* const* g_mid; float* const* g_z1;      // per block, row-major (train.hip: TrainWs)
  float* g_h0;                   // dL/dh0 [n][H0]
  const float* const* masks; bool drop; uint64_t seed; uint32_t row_offset;
};
int64_t train_squad_bwd_wpk_floats(const Arch& a);
int train_squad_backward(osd_handle* h, hipStream_t s, const FwdWs& f, const TrainSquadBwdBufs& B, int64_t n, float* gact_units, float* wpk,
                         unsigned* bar_and_status, int64_t panels, float* loss_poison);
int train_squad_forward(osd_handle* h, hipStream_t s, const FwdWs& ws, const TrunkIn& in, float* act_units, float* wpk, unsigned* bar_and_status,
                        int64_t panels, float* loss_poison, float* wpk_t = nullptr);
// wgrad_group.hip
struct WgPending;
int wgrad_group_flush(osd_handle* h, hipStream_t s, int plan_index, const std::vector<WgPending>& pend, float* slabs, int64_t slab_floats,
                      int max_grid);
struct GnColItem;
int gn_colsums_flush(osd_handle* h, hipStream_t s, int plan_index, const std::vector<GnColItem>& cols);
void wgrad_group_free(osd_handle* h);
int check_row_offset(int64_t row_offset, int64_t n);
int sanitize_t(osd_handle* h, hipStream_t s, const int32_t* t_index, int64_t n, const int** out);

}  // namespace osd
