/*
 * osdiff.h -- C ABI of libosdiff.so, the MI355X (gfx950) implementation of the
 * diffusion hot path of rare-resilience-ai/Osteosarcoma_DiffusionModel.
 *
 * The reference has no FFI boundary of its own: its hot path is the Python object
 * API of models/diffusion.py, utils/train.py and utils/generate.py.  Each entry
 * point below names the reference function (file:line, relative to the reference
 * tree) whose arithmetic it replaces; the Python shim in
 * osteosarcoma_diffusionmodel_amd/ keeps the reference's class and method names
 * and calls these through ctypes.
 *
 * Conventions
 *   - every function returns 0 (OSD_OK) or a negative OSD_E* code; no exceptions,
 *     no aborts.  osd_last_error() returns the message of the last failure on the
 *     calling thread.
 *   - all tensors are row-major contiguous fp32 unless stated; "dev" pointers are
 *     device memory of the handle's device, "host" pointers are host memory.
 *   - the caller owns every tensor; the library borrows pointers for the duration
 *     of a call, except the parameter pointers given to osd_load_weights(), which
 *     are borrowed until the next osd_load_weights() / osd_destroy().
 *   - all work is enqueued on the handle's stream (osd_set_stream) and is
 *     asynchronous unless stated; calls on one handle are not re-entrant.
 *   - no CPU fallback exists: without a HIP device every compute call fails.
 *   - row_offset (global id of the call's first row) addresses the Philox stream with
 *     32-bit row counters: row_offset < 0 or row_offset + n > 2^32 is OSD_EINVAL.
 *   - t_index entries must lie in [0, T): the reference's buffer gather raises
 *     IndexError otherwise (models/diffusion.py:337) and so does the Python shim; the
 *     library itself cannot see device values without a sync, so it clamps a
 *     caller-supplied t_index into [0, T) (no out-of-bounds table read can occur).
 */
#ifndef OSDIFF_H
#define OSDIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSD_VERSION 100 /* 0.1.0 */

#define OSD_OK            0
#define OSD_EINVAL       -1 /* bad argument / shape (Python: ValueError)        */
#define OSD_ENOMEM       -2 /* allocation failed                                 */
#define OSD_EHIP         -3 /* HIP runtime error (Python: RuntimeError)          */
#define OSD_ESTATE       -4 /* call order: schedule / weights not loaded         */
#define OSD_EUNSUPPORTED -5 /* architecture outside what the kernels cover       */

#define OSD_MAX_HIDDEN 8

/* flags */
#define OSD_F_GRAPH      1 /* replay the reverse step from a captured hipGraph   */
#define OSD_F_TRAIN_MODE 2 /* dropout active (model.train())                     */
#define OSD_F_SYNC       4 /* synchronise the handle's stream before returning   */

typedef struct osd_handle osd_handle;

/* Constructor arguments of BiologyAwareDiffusionModel (models/diffusion.py:264-301). */
typedef struct osd_config {
  int32_t mutation_dim;
  int32_t expression_dim;
  int32_t pathway_dim;
  int32_t condition_dim;
  int32_t time_dim;                    /* config.model.latent_dim; cond width is time_dim/2 and must be 64 */
  int32_t n_hidden;                    /* len(config.model.hidden_dims)                    */
  int32_t hidden_dims[OSD_MAX_HIDDEN]; /* each divisible by 8 (GroupNorm(8, C))            */
  int32_t num_steps;                   /* config.model.diffusion.num_steps (T)             */
  float   dropout_p;                   /* config.model.gnn.dropout (models/diffusion.py:294)*/
  int32_t device;                      /* HIP device ordinal                                */
} osd_config;

int         osd_version(void);
const char *osd_last_error(void);

/* Number of trainable parameter tensors for this architecture (52 at n_hidden == 3),
 * in the order of BiologyAwareDiffusionModel.named_parameters(); 0 on bad config. */
int osd_num_params(const osd_config *cfg);
/* Element count of parameter i (same order); -1 on bad index. */
int64_t osd_param_numel(const osd_config *cfg, int i);

/* models/diffusion.py:264-310 (construction).  Allocates schedule tables and streams. */
int osd_create(const osd_config *cfg, osd_handle **out);
int osd_destroy(osd_handle *h);

/* Bind the HIP stream (hipStream_t, e.g. torch.cuda.current_stream().cuda_stream). */
int osd_set_stream(osd_handle *h, void *hip_stream);

/* Tunables: "sampler" (0 auto, 1 persistent chain kernel wherever the architecture allows it, 2 per-layer kernels),
 * "chunk_rows" / "n_streams" (per-layer path: rows per sampling chunk, chunks in flight), "chain_grid" (cap on the chain
 * kernel's workgroups), "chain_steps_per_launch" (0 = the whole chain in one launch), "chain_stagger" (shader cycles
 * between the starts of the two workgroups of a CU), "train_streams" (1 | 2: weight-gradient leaves on a side stream),
 * "grouped_wgrad" (1: every weight gradient of a backward pass in one grouped launch), "wgrad_mid_flush" (1: the decoder
 * half's weight gradients already mid-pass; always on under data parallel), "fused_gn_bwd" (1: GroupNorm backward inside
 * the dgrad epilogue), "chain_variant" (which chain kernel: 1 the workspace chain, 128-row tiles whose activations
 * pass through a private workspace -- the faster one from 65 536 rows on; 2 the LDS-resident chain, 64 patients per workgroup
 * with every activation in LDS, bit-identical, for architectures whose panels fit -- hidden_dims[0] = 256 = the last block's
 * width --, others fall back to 1; it runs at its full rate from 16 384 rows on; 3 the squad chain, the small-batch kernel: eight
 * workgroups per 32 patients for the whole chain, for models it decomposes ("squad_chain_supported") and batches whose squads are all
 * resident -- 3 072 rows on 256 CUs --, others run what auto would; it agrees with the other engines to fp32 rounding, not bitwise
 * ("squad_panel": its patients per panel, 0 auto = 16 up to 1 024 rows and 32 above, or 16 / 32; "last_squad_panel" reads back);
 * 0 auto: 1 from 65 536 rows on, 2 from 10 240 rows on, 3 for resident batches when "input_splitk" != 0), "dual_dgrad" / "train_ksplit" /
 * "train_input_splitk" (training-step experiments), "train_squad" (from 2 048 rows on: 2, the default, runs the ten
 * Linear+GroupNorm+SiLU layers of a training forward pass as one launch of squads, csrc/train_squad.h, and the dgrad chain of the backward
 * pass as another, csrc/train_squad_bwd.h -- the backward one in single-process steps only: bucket events or a mid-pass flush keep the
 * per-layer dgrads --; 1: the forward only; 0: per-layer launches), "cond_bwd_fused" (1, the default: the conditioning branch's backward
 * below h0 -- time-table scatter, two 64-wide dgrads, SiLU backward, the first embedding Linear's weight gradient -- as one launch,
 * k_cond_bwd in csrc/k_train.hip, for hidden_dims[0] <= 256 and a multiple of 32; 0: five launches),
 * "input_splitk" (the small-batch mode of sampling: 0 off --
 * the default: a row's result does not depend on the batch it is in, bit for bit --, -1 auto, n > 0 slices: small batches run
 * input_proj and the deep layers split over K, or, where chain_variant 3 applies, the squad chain; another fp32 summation order,
 * chain tolerance against the default).  Auto sampler: the chain kernel when the batch has at least as many 128-row tiles as the device
 * holds resident workgroups (65 536 rows on an MI355X) or falls into the LDS-resident kernel's window (above), and the model is
 * in eval mode; else the per-layer kernels. */
int osd_set_option(osd_handle *h, const char *name, int64_t value);

/* Reads an option back, or one of the read-only counters "chain_fallbacks" (chains that gave up -- see osd_sample_chain --
 * and were re-run on the per-layer kernels), "last_engine" (0 per-layer kernels, 1 chain kernel), "last_chain_variant" (1 | 2 | 3,
 * the chain kernel that ran last), "panel_chain_supported" / "squad_chain_supported" (1 when "chain_variant" 2 / 3 applies to this
 * model).  Further options:
 * "chain_spin_budget" (ticks of the 100 MHz s_memrealtime counter a dependency wait inside the chain kernel may take, default
 * 5 s), "chain_wall_budget_ms" (host-side budget of a synchronous chain; 0 = 10 x the estimated run time + 2 s). */
int osd_get_option(osd_handle *h, const char *name, int64_t *value);

/* Schedule + time-embedding tables, computed by the host with the reference's own
 * fp32 expressions so they are bit-identical (models/diffusion.py:299-326, 131-137,
 * 401-419).  All host pointers:
 *   sqrt_ac[T], sqrt_1m_ac[T]   buffers used by q_sample (:337-338)
 *   post_coef[T*6]              per-step scalars of p_sample (:401-419), layout of
 *                               oracle/diffusion_oracle.py:posterior_coefficients
 *   time_emb[T*time_dim]        TimeEmbedding(t/T) rows for t = 0..T-1 (:131-137) */
int osd_set_schedule(osd_handle *h, const float *sqrt_ac, const float *sqrt_1m_ac,
                     const float *post_coef, const float *time_emb);

/* nn.Module parameters (models/diffusion.py:283-295): n == osd_num_params() device
 * pointers in named_parameters() order.  Borrowed; derived tables (time_proj applied
 * to the time-embedding table, packed / plane copies of weights) are recomputed on the
 * handle's stream.  Call again after any in-place parameter update the library cannot
 * see (an update through osd_clip_adamw_step on this handle, or a training call that
 * skipped a repack, is seen: the next forward / sampling entry point refreshes the
 * derived copies itself). */
int osd_load_weights(osd_handle *h, const float *const *params, int n);

/* DiffusionUNet.forward in eval/train mode on n rows (models/diffusion.py:210-256)
 * including ConditionalEmbedding (:101-114):  eps[n][D].
 *   t_index  dev int32[n] per-row timestep index, or NULL -> every row uses t_all
 *   masks    train mode only: dev float 0/1 keep-masks, one [n][C] per block in
 *            execution order (n_blocks pointers, host array), or NULL -> Philox(seed) */
int osd_denoiser_forward(osd_handle *h, const float *x, const int32_t *t_index, int32_t t_all,
                         const float *cond, int64_t n, float *eps, int flags,
                         const float *const *masks, uint64_t seed);

/* q_sample (models/diffusion.py:328-342): x_t = sqrt_ac[t]*x0 + sqrt_1m_ac[t]*noise.
 * noise_in NULL -> Philox(seed, row_offset) normals, written to noise_out. */
int osd_q_sample(osd_handle *h, const float *x0, const int32_t *t_index, const float *noise_in,
                 int64_t n, uint64_t seed, int64_t row_offset, float *x_t, float *noise_out);

/* p_sample (models/diffusion.py:382-425): one reverse step at python-int t.
 * z NULL -> Philox(seed,row_offset,t).  In-place (x_out == x_t) is allowed. */
int osd_p_sample_step(osd_handle *h, const float *x_t, int32_t t, const float *cond, const float *z,
                      int64_t n, uint64_t seed, int64_t row_offset, float *x_out, int flags);

/* sample (models/diffusion.py:427-449) + binarisation of utils/generate.py:135.
 *   x_T     dev [n][D] start noise or NULL -> Philox
 *   noises  dev [T-1][n][D] in draw order (t = T-1 .. 1) or NULL -> Philox
 *   x_out   dev [n][D]
 *   mut_mask_out  dev float [n][mutation_dim] = (x_out[:, :mutation_dim] > 0.5) or NULL
 * Rows are split into chunks that run the whole T-step chain independently on the
 * handle's internal streams; row_offset makes Philox draws independent of sharding. */
int osd_sample_chain(osd_handle *h, const float *cond, int64_t n, const float *x_T,
                     const float *noises, uint64_t seed, int64_t row_offset, float *x_out,
                     float *mut_mask_out, int flags);
/* Two engines run osd_sample_chain with identical results: the per-layer kernels (12 launches per step, replayed from a
 * hipGraph under OSD_F_GRAPH) and, for >= ~50 000 rows of an architecture with 256/512-wide blocks in eval mode, ONE
 * persistent kernel that carries each 128-row tile through all layers and all T steps (csrc/chain.h).  Returns the
 * engine a call with these n / flags would use (0 per-layer, 1 chain kernel); n < 0: the engine of the last call.  The
 * chain kernel bounds every inter-workgroup wait ("chain_spin_budget"), and a synchronous call bounds the kernel itself
 * ("chain_wall_budget_ms": hipStreamQuery poll, then an abort flag the waits observe).  If the chain gives up its results are
 * invalid: under OSD_F_SYNC the SAME call re-runs the chain on the per-layer kernels from x_T / seed (bit-identical results;
 * returns OSD_OK, osd_last_error() holds a warning, "chain_fallbacks" counts, osd_sample_engine(h, -1, 0) then reports 0) --
 * models/diffusion.py:427-449 cannot fail; without OSD_F_SYNC the next osd_sample_chain on the handle returns OSD_EHIP.
 * OSD_EHIP from a synchronous call means the kernel did not even react to the abort flag (device hung). */
int osd_sample_engine(osd_handle *h, int64_t n, int flags);

/* Device-resident epoch path (utils/train.py:204-250 hands every batch over from host memory; here the dataset of
 * OsteosarcomaDataset (utils/train.py:22-82) stays in HBM).  The NEXT osd_train_loss_fwd_bwd on this handle takes its n rows
 * from the dataset instead of its x0 / cond arguments (pass NULL there):
 *   row i  = lam * data[idx_a[i]] + (1 - lam) * data[idx_b[i]]      MixupAugmentation, utils/train.py:117-119, the two
 *            products rounded separately as torch evaluates them; conditions likewise
 *   idx_b NULL: no mixup;  idx_a NULL: rows 0 .. n-1.  idx_a / idx_b: dev int64[n].  One-shot: consumed by that call.
 * Gather, mixup, the draw of t and q_sample run as ONE pass over the batch (SURVEY a5 + a11). */
int osd_train_batch_source(osd_handle *h, const float *data, int64_t ld_data, const float *cond, int64_t ld_cond,
                           const int64_t *idx_a, const int64_t *idx_b, double lam);

/* Training forward+backward (models/diffusion.py:344-380 + loss.backward(),
 * utils/train.py:236-239): loss (dev float[1]) and gradients of all parameters.
 *   t_index   dev int32[n] or NULL -> Philox randint
 *   noise     dev [n][D] or NULL -> Philox
 *   masks     keep-masks as in osd_denoiser_forward, or NULL -> Philox / none in eval
 *   grads     n_params device pointers (host array); overwritten (not accumulated)
 *   loss_scale  multiplies the gradients (1.0 for plain backward)
 *   events/n_events  optional hipEvent_t array recorded on the handle's stream as
 *             each gradient bucket (see osd_grad_bucket) becomes final, for
 *             overlapping the RCCL all-reduce with the rest of backward */
int osd_train_loss_fwd_bwd(osd_handle *h, const float *x0, const float *cond, int64_t n,
                           const int32_t *t_index, const float *noise, const float *const *masks,
                           uint64_t seed, int64_t row_offset, int flags, float *loss_out,
                           float *const *grads, double loss_scale, void *const *events, int n_events);

/* The same network split at the loss, for losses other than the eps-MSE (config.yaml:47 names l1 / huber):
 * osd_denoiser_forward_train is osd_denoiser_forward with per-row t_index that keeps the activations in the
 * handle's training workspace (valid until the next training call on the handle); osd_denoiser_backward takes an
 * arbitrary upstream gradient dout[n][D] = dL/d eps and writes every parameter gradient (overwritten) and, when
 * dx_t != NULL, dL/dx_t [n][D].  x_t, t_index, cond, masks / seed / row_offset and flags must be the forward call's. */
int osd_denoiser_forward_train(osd_handle *h, const float *x_t, const int32_t *t_index, const float *cond,
                               int64_t n, const float *const *masks, uint64_t seed, int64_t row_offset,
                               int flags, float *eps_out);
int osd_denoiser_backward(osd_handle *h, const float *x_t, const int32_t *t_index, const float *cond, int64_t n,
                          const float *dout, const float *const *masks, uint64_t seed, int64_t row_offset,
                          int flags, float *const *grads, float *dx_t, void *const *events, int n_events);

/* Gradient buckets in the order backward finalises them: bucket b covers parameters
 * [first, last] (indices in named_parameters() order).  Returns the bucket count. */
int osd_grad_buckets(const osd_config *cfg, int32_t *first, int32_t *last, int max_buckets);

/* ---- data-parallel gradient exchange over RCCL / xGMI ------------------------------
 * The reference is single-process; the slot these fill is between loss.backward() and clip_grad_norm_
 * (utils/train.py:239-244).  One communicator per process (one process per GPU).  RCCL is bound at run time
 * (dlopen): OSD_EUNSUPPORTED when no librccl.so can be found.
 *   osd_comm_unique_id   rank 0 draws the 128-byte rendezvous id (ncclGetUniqueId); the caller ships it to the
 *                        other ranks (the Python shim broadcasts it through torch.distributed's store)
 *   osd_comm_create      collective over all ranks (ncclCommInitRank) on HIP device `device`
 *   osd_allreduce_grads_begin  SUM all-reduce of flat_grad[start[b], end[b]) (element offsets) for b = 0..n_buckets-1
 *                        on the communicator's stream; bucket b waits for events[b] (the hipEvent_t array handed
 *                        to osd_train_loss_fwd_bwd), or -- events == NULL -- for everything queued on the
 *                        handle's stream.  Gradients are pre-scaled by 1/world through loss_scale.
 *   osd_allreduce_grads_end    the handle's stream waits for the collectives (then clip + AdamW may run) */
#define OSD_COMM_ID_BYTES 128
typedef struct osd_comm osd_comm;
int osd_comm_unique_id(void *id_out128);
int osd_comm_create(const void *id128, int rank, int world, int device, osd_comm **out);
int osd_comm_destroy(osd_comm *c);
int osd_allreduce_grads_begin(osd_handle *h, osd_comm *c, float *flat_grad, const int64_t *start,
                              const int64_t *end, void *const *events, int n_buckets);
int osd_allreduce_grads_end(osd_handle *h, osd_comm *c);

/* MixupAugmentation.__call__ (utils/train.py:108-120): out = lam*v + (1-lam)*v[perm]
 * for data[n][D], conditions[n][cond_dim], survival[n]; perm dev int64[n]. */
int osd_mixup(osd_handle *h, const float *data, const float *cond, const float *surv,
              const int64_t *perm, double lam, int64_t n, float *data_out, float *cond_out,
              float *surv_out);

/* clip_grad_norm_(max_norm) + AdamW.step (utils/train.py:242-244, 169-173) over flat
 * contiguous buffers of `numel` floats.  step is the 1-based count after increment.
 * grad_norm_out (dev float[1], may be NULL) receives the pre-clip global L2 norm.
 * max_norm <= 0 disables clipping.  Hyper-parameters are python floats (doubles): derived
 * scalars such as 1 - beta2 are formed in double and rounded to fp32 once, as torch does. */
int osd_clip_adamw_step(osd_handle *h, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                        int64_t numel, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double max_norm, int64_t step, float *grad_norm_out);

/* The same step without a model handle (any nn.Module's flat buffers, e.g. the cVAE): normsq_ws is
 * caller-owned device scratch of 256 doubles (no initial state needed: the global gradient norm is formed from one partial
 * sum per workgroup, added up in a fixed order -- deterministic, and independent of any previous call). */
int osd_nn_clip_adamw_step(void *stream, int device, double *normsq_ws, float *param, float *grad,
                           float *exp_avg, float *exp_avg_sq, int64_t numel, double lr, double beta1,
                           double beta2, double eps, double weight_decay, double max_norm, int64_t step,
                           float *grad_norm_out);

/* Measurement aid for bench.py: per-launch HIP-event timing of one reverse step on n rows
 * (eager launches on the handle's stream, averaged over reps after one warm-up pass).
 * Entry 0 = input_proj, 1..2*n_blocks = the Linear+GroupNorm+SiLU halves in execution order,
 * last = output_proj+posterior.  flop_out = algorithmic GEMM FLOPs of each launch. */
int osd_profile_step(osd_handle *h, const float *cond, int64_t n, int reps, float *ms_out,
                     double *flop_out, int max_entries, int *n_entries);

/* ---- validation metrics on the device (utils/validation.py) ----------------
 * Stream/device based (no model handle), synchronous: results are host scalars. */
/* compute_mmd (utils/validation.py:273-298): RBF kernel, gamma <= 0 -> 1/D, means over all pairs
 * including the diagonal, sqrt(max(XX + YY - 2 XY, 0)).  X dev [n][D], Y dev [m][D]. */
int osd_val_mmd(void *stream, int device, const float *X, int64_t n, const float *Y, int64_t m, int D,
                double gamma, double *mmd_out);
/* One block of the above for row-sharded data (multi-GPU validation): sum_out = sum_{i<n, j<m} exp(-gamma |a_i - b_j|^2). */
int osd_val_rbf_sum(void *stream, int device, const float *A, int64_t n, const float *B, int64_t m, int D,
                    double gamma, double *sum_out);
/* scipy.stats.ks_2samp as used at utils/validation.py:238-245, for features 0..nf-1 of real dev [n1][ld]
 * and synth dev [n2][ld]: exact integer extremes of cnt(real<=v)*n2 - cnt(synth<=v)*n1 over all sample
 * points v; the statistic is max(dmax, -dmin, 0) / (n1*n2) (p-values follow on the host). */
int osd_val_ks_extremes(void *stream, int device, const float *real, int64_t n1, const float *synth,
                        int64_t n2, int ld, int nf, int64_t *dmax_out, int64_t *dmin_out);
/* Mean of the strict upper triangle of the Pearson matrix of data[:, cols] (utils/validation.py:156-161);
 * cols: host array of g column indices, 2 <= g <= 512. */
int osd_val_mean_offdiag_corr(void *stream, int device, const float *data, int64_t rows, int ld,
                              const int32_t *cols_host, int g, double *out);
/* Column sums (double) of data[:, :cols] -- the mutation frequencies of utils/validation.py:45-46 are sums / rows. */
int osd_val_column_sums(void *stream, int device, const float *x, int64_t rows, int ld, int cols,
                        double *sums_host);
/* Raw Gram matrix of up to 64 selected columns: gram_host[i*g + j] = sum_r x[r][c_i] * x[r][c_j].  For 0/1 mutation
 * columns these are the exact joint counts behind the 2x2 contingency tables of utils/validation.py:98-111 and
 * the both-mutated counts of :75-78. */
int osd_val_gram(void *stream, int device, const float *x, int64_t rows, int ld, const int32_t *cols_host,
                 int g, double *gram_host);
/* The two passes of osd_val_mean_offdiag_corr as partial sums over a row shard (all-reduce between them):
 * per selected column sum and sum of squares; then S = sum_rows (sum_g (x - mu_g) * isd_g)^2. */
int osd_val_col_moments(void *stream, int device, const float *data, int64_t rows, int ld,
                        const int32_t *cols_host, int g, double *sum_host, double *sumsq_host);
int osd_val_rowz_sq(void *stream, int device, const float *data, int64_t rows, int ld,
                    const int32_t *cols_host, int g, const double *mu_host, const double *isd_host,
                    double *S_host);
/* Pearson correlation of two strided device columns (Series.corr at utils/validation.py:205); the five raw
 * sums (sum a, sum b, sum a^2, sum b^2, sum ab) of a row shard. */
int osd_val_pearson_sums(void *stream, int device, const float *a, int lda, const float *b, int ldb,
                         int64_t rows, double *out5_host);
int osd_val_pearson(void *stream, int device, const float *a, int lda, const float *b, int ldb,
                    int64_t rows, double *out);

/* ---- biological constraint losses (north_star) ------------------------------
 * The reference declares them at models/cvae.py:262-302 as stubs that return 0.0 (and the diffusion
 * model has none), so nothing is added unless osd_set_constraints() configures them:
 *   pathway coherence    L_pc = mean_P (1 - c_P), c_P = mean off-diagonal Pearson correlation over the
 *                        batch rows of the member columns of pathway P (utils/validation.py:144-173);
 *                        pathways with fewer than 2 members are skipped
 *   mutation-expression  L_me = mean_{i in A, j in B} (corr_recon(i,j) - corr_true(i,j))^2
 *                        ("MSE on correlation matrices", models/cvae.py:296-297), |A|, |B| <= 64
 * Columns are indices into the D-wide feature vector; constant columns count as correlation 0. */
typedef struct osd_constraints {
  const int32_t *pathway_offsets;   /* host int32[n_pathways + 1], CSR over pathway_members */
  const int32_t *pathway_members;   /* host int32[offsets[n_pathways]] */
  int32_t n_pathways;               /* 0 disables the pathway term */
  double pathway_weight;            /* config.yaml:58 pathway_coherence_weight */
  const int32_t *cols_a;            /* host int32[n_a], e.g. mutation columns */
  const int32_t *cols_b;            /* host int32[n_b], e.g. pathway-score or expression columns */
  int32_t n_a, n_b;                 /* 0 disables the mutation-expression term */
  double mutexpr_weight;            /* config.yaml:59 mutation_expression_weight */
} osd_constraints;
/* Configures (c == NULL: clears) the terms osd_train_loss_fwd_bwd adds to the eps-MSE, evaluated on
 * x0_hat = (x_t - sqrt(1-ac_t) eps_hat) / sqrt(ac_t) (models/diffusion.py:405) against x0 of the batch:
 * loss = mse + pathway_weight * L_pc + mutexpr_weight * L_me, gradients flow into eps_hat. */
int osd_set_constraints(osd_handle *h, const osd_constraints *c);
/* (mse, L_pc, L_me) of the last osd_train_loss_fwd_bwd call; synchronises the handle's stream. */
int osd_get_loss_parts(osd_handle *h, float *parts_host3);
/* Stand-alone ops (stream/device based, synchronous): loss_out (dev float[1]) += weight * L and, when
 * dx != NULL, dx (dev [rows][ld]) += weight * dL/dx.  x, x_recon, x_true: dev [rows][ld], cols <= ld. */
int osd_loss_pathway_coherence(void *stream, int device, const float *x, int64_t rows, int ld, int cols,
                               const int32_t *offsets_host, const int32_t *members_host, int n_pathways,
                               double weight, float *loss_out, float *dx);
int osd_loss_mutation_expression(void *stream, int device, const float *x_recon, const float *x_true,
                                 int64_t rows, int ld, int cols, const int32_t *cols_a_host, int n_a,
                                 const int32_t *cols_b_host, int n_b, double weight, float *loss_out,
                                 float *dx);

/* ---- layer ops behind the cVAE mirror (models/cvae.py) -----------------------
 * Stream/device based, asynchronous (nothing synchronises), every tensor a device pointer. */
/* torch.cat([x1, x2], -1) -> nn.Linear (models/cvae.py:54-55, 97-98): y[n][N] = [x1|x2] w[N][K1+K2]^T + b.
 * K2 == 0: plain Linear (x2 ignored). */
int osd_nn_linear(void *stream, int device, const float *x1, int K1, const float *x2, int K2,
                  const float *w, const float *b, int64_t n, int N, float *y);
/* Its backward: dw[N][K1+K2] = gy^T [x1|x2] (overwritten), db[N] = column sums of gy (NULL: skipped),
 * dx1[n][K1] = gy w[:, :K1] (NULL: skipped; the x2 panel -- the conditions -- gets no gradient). */
int osd_nn_linear_bwd(void *stream, int device, const float *x1, int K1, const float *x2, int K2,
                      const float *w, const float *gy, int64_t n, int N, float *dx1, float *dw, float *db);
/* nn.BatchNorm1d -> nn.ReLU -> nn.Dropout (models/cvae.py:29-32, 79-82) on z[n][C].
 *   training != 0: batch statistics (biased variance), running_mean/var updated with `momentum`
 *                  (unbiased variance), dropout from `mask` (dev float 0/1 keep-mask [n][C]) or, when
 *                  mask == NULL, Philox(seed, tag); n >= 2 required as in torch
 *   training == 0: running statistics, no dropout
 *   use_bn == 0:   ReLU -> Dropout only (the survival head, models/cvae.py:251-252)
 * save_mean / save_invstd (dev float[C]) receive the statistics the backward needs. */
int osd_nn_bn_relu_dropout(void *stream, int device, const float *z, int64_t n, int C, const float *gamma,
                           const float *beta, float *running_mean, float *running_var, double momentum,
                           double eps, int training, int use_bn, double p_drop, const float *mask,
                           uint64_t seed, uint32_t tag, float *y, float *save_mean, float *save_invstd);
int osd_nn_bn_relu_dropout_bwd(void *stream, int device, const float *gy, const float *z, int64_t n, int C,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd, int training, int use_bn, double p_drop,
                               const float *mask, uint64_t seed, uint32_t tag, float *dz, float *dgamma,
                               float *dbeta);
/* reparameterize (models/cvae.py:152-156): z = mu + eps * exp(0.5 * logvar); eps_in NULL -> Philox(seed)
 * normals, written to eps_out when it is not NULL. */
int osd_nn_reparameterize(void *stream, int device, const float *mu, const float *logvar, const float *eps_in,
                          uint64_t seed, int64_t n, int Lz, float *z, float *eps_out);
/* Backward of reparameterize: d_logvar = 0.5 * gz * (z - mu)   (d_mu is gz itself). */
int osd_nn_reparameterize_bwd(void *stream, int device, const float *gz, const float *mu, const float *z,
                              int64_t count, float *d_logvar);
/* VAE loss (models/cvae.py:178-181): parts3 (dev float[3]) = (recon + kl, recon, kl) with
 * recon = sum (x_recon - x)^2 / n, kl = -0.5 sum(1 + logvar - mu^2 - exp(logvar)) / n, and their
 * gradients d_recon[n][D], d_mu[n][Lz], d_logvar[n][Lz] (each may be NULL). */
int osd_nn_vae_loss(void *stream, int device, const float *x_recon, const float *x, const float *mu,
                    const float *logvar, int64_t n, int D, int Lz, float *parts3, float *d_recon,
                    float *d_mu, float *d_logvar);
/* One tensor of MixupAugmentation.__call__ (utils/train.py:108-120) without a model handle:
 * out[rows][cols] = lam * v + (1 - lam) * v[perm]. */
int osd_nn_mixup(void *stream, int device, const float *v, const int64_t *perm, double lam, int64_t rows,
                 int cols, float *out);
/* The three tensors of one MixupAugmentation call (data [rows][data_cols], conditions [rows][cond_cols], survival [rows]) in
 * ONE launch; a NULL output skips that tensor. */
int osd_nn_mixup3(void *stream, int device, const float *data, const float *cond, const float *surv,
                  const int64_t *perm, double lam, int64_t rows, int data_cols, int cond_cols,
                  float *data_out, float *cond_out, float *surv_out);
/* F.mse_loss(a, b) over `count` elements (models/cvae.py:323): loss_out dev float[1], da (may be NULL). */
int osd_nn_mse(void *stream, int device, const float *a, const float *b, int64_t count, float *loss_out,
               float *da);

/* ---- building blocks, exported for the parity tests ------------------------ */
/* y[n][N] = act(x[n][K] @ w[N][K]^T + b), act = identity (silu=0) or SiLU. */
int osd_op_linear(osd_handle *h, const float *x, const float *w, const float *b, int64_t n, int K,
                  int N, int silu, float *y);
/* Linear -> GroupNorm(8) -> SiLU, one half of _make_block (models/diffusion.py:200-203).
 * x2/K2: optional second K panel (concat-free decoder input, :250). */
int osd_op_linear_gn_silu(osd_handle *h, const float *x, int K1, const float *x2, int K2,
                          const float *w, const float *b, const float *gamma, const float *beta,
                          int64_t n, int N, float *y);
/* C[p][f] (+)= sum_k A(f,k) B(p,k), C row-major [P][F] with leading dimension ldc;
 * a_kc/b_kc: operand stored [row][k] (1) or [k][row] (0).  Forward Linear is (1,1),
 * dgrad (0,1), wgrad (0,0). */
int osd_op_gemm(osd_handle *h, const float *A, int lda, int a_kc, const float *B, int ldb, int b_kc,
                int F, int P, int K, float *C, int ldc, int accumulate);
/* out[rows][cols] standard normals from the library's Philox stream, addressed by
 * (seed, row_offset + row, col/4, step, kind): kind 0 = the z of p_sample at t == step
 * (step == T is x_T), 1 = the eps of q_sample (step 0), >= 2 = user streams. */
int osd_op_randn(osd_handle *h, float *out, int64_t rows, int cols, uint64_t seed,
                 int64_t row_offset, uint32_t step, uint32_t kind);

#ifdef __cplusplus
}
#endif
#endif /* OSDIFF_H */
