/* pvae.h -- C ABI of libpvae_gfx950.so: the MI355X-native hot path of the PhysicsVAE
 * supervised training loop (world model + conditional VAE).
 *
 * The reference has no FFI on this path: it is pure Python calling stock PyTorch ops
 * (SURVEY.md 8b).  The entry points below are therefore the boundary a maintainer would
 * bind from the reference's own Python (ctypes stub in INTEGRATION.md); each one names
 * the reference code it replaces.  `tpv` = train_physics_vae.py, `tm` = torch_models.py,
 * `rmt` = rllib_model_torch.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every call returns 0 on success, <0 on error; pvae_last_error() gives the text
 *     (thread-local).  No call synchronises the host with the device.
 *   - the CALLER owns every device buffer (parameters, Adam moments, gradients, dataset,
 *     workspace) and passes the hipStream_t (as void*) to launch on.  The library owns
 *     only the opaque pvae_ctx (layout tables + bound pointers).  One ctx per process per
 *     GPU; a ctx is not thread-safe.
 *   - all arithmetic is fp32 (v_mfma_f32_16x16x4_f32 for the contractions).
 *
 * Parameter arena.  The trainable stacks live in ONE flat fp32 arena
 *   [ task encoder | motor decoder | decoder helper | learned prior | world model ]      (TE | MD | MH | PR | WM;
 *   MH and PR are empty segments unless the configuration has them -- see the PVAE_NET_* enum below)
 * each Linear stored as W[n_out_pad][ld] (row-major, ld = n_in rounded up to 64 floats,
 * n_out_pad = n_out rounded up to 64) followed by bias[n_out_pad]; pad entries are zero
 * and stay zero.  The checkpoint tensor `<net>._model.<i>._model.0.weight` of shape
 * [n_out, n_in] (rmt:234-283) is the strided view W[:n_out, :n_in] of that block, so the
 * reference's state_dict layout is the source of truth and no packing pass exists.
 * Gradients and the two Adam moments use arenas of the same layout.
 */
#ifndef PVAE_H
#define PVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVAE_ABI_VERSION 12

typedef struct pvae_ctx pvae_ctx;

/* PVAE_NET_PR: the learned prior mean `_latent_prior` (rmt:627-635), present only with
 * PVAE_PRIOR_STATE_MEAN.  PVAE_NET_MH: the motor decoder's helper `_motor_decoder_helper` (rmt:670-680, 833-835),
 * present only with pvae_config.mh_depth > 0: a second stack on the decoder's input whose tanh output, scaled by
 * mh_range, is added to the action; the reference's supervised loss sees that term inside a_hat, so the joint phase
 * trains the helper with the decoder (pvae_step_params.adam_t[PVAE_NET_MH] == 0: frozen for this step).  With
 * lookahead > 1 (tpv:367-428) the helper runs in every unrolled step and the WORLD phase trains it too -- the state the
 * world model continues from is its own prediction under the helped action (tpv:417-421) -- as upstream, whose trainer
 * never freezes it.
 * Arena order is TE | MD | MH | PR | WM, so that the stacks trained together in the joint phase form one
 * contiguous segment. */
enum { PVAE_NET_TE = 0, PVAE_NET_MD = 1, PVAE_NET_WM = 2, PVAE_NET_PR = 3, PVAE_NET_MH = 4, PVAE_NUM_NETS = 5 };
/* latent_prior_type (rmt:614-635, 795-819; tpv:384-409).  Only the first runs upstream; the other two
 * follow the specification in oracle/refpath.py (the reference sketches them and crashes):
 *   ZERO_MEAN   "normal_zero_mean_one_std"  KL(N(mu,s^2) || N(0,1))
 *   STATE_MEAN  "normal_state_mean_one_std" KL(N(mu,s^2) || N(mu_p(s_body),1)), mu_p = PR stack (Db -> Z)
 *   HYPERSPHERE "hypersphere_uniform"       encoder emits Z values, z = e/|e|, loss_kl = mean <z, n/|n|>
 *   NONE        latent_prior_type = False   encoder emits Z values that ARE the code: no sampling, no KL term
 *                                           (rmt:622-623, 815-816; a mode the reference runs: pinned by a capture)
 * The non-default kinds need lookahead == 1. */
enum { PVAE_PRIOR_ZERO_MEAN = 0, PVAE_PRIOR_STATE_MEAN = 1, PVAE_PRIOR_HYPERSPHERE = 2, PVAE_PRIOR_NONE = 3 };
enum { PVAE_PHASE_WORLD = 0, PVAE_PHASE_JOINT = 1 };
/* loss_fn of the three reconstruction terms (get_loss_fn tm:97-107; trainer key "loss", tpv:257) */
enum { PVAE_LOSS_MSE = 0, PVAE_LOSS_L1 = 1 };

/* hidden-layer activations (pvae_config.act_kind / layer_act; get_activation_fn rmt:30-46).  "swish" is not
 * offered: upstream it is ray's Swish module with a learnable beta (a parameter outside the five checkpoint
 * files' Linear tensors), and it needs the pre-activation in the backward pass.  PVAE_ACT_LINEAR: no
 * activation after a hidden layer (rmt:32-33 "linear" / None), per-layer use only. */
enum { PVAE_ACT_RELU = 0, PVAE_ACT_TANH = 1, PVAE_ACT_SIGMOID = 2, PVAE_ACT_ELU = 3, PVAE_ACT_LINEAR = 4 };
#define PVAE_MAX_HIDDEN 15 /* hidden layers per stack */

/* flags for pvae_forward_backward */
enum {
    PVAE_FLAG_FUSED_ADAM = 1, /* apply Adam inside the backward launches (1 GPU): in the weight-
                               * gradient epilogue, or -- when a gradient arena is bound -- deferred by
                               * one launch to extra workgroups, the arena serving as scratch (its
                               * contents are unspecified afterwards); same arithmetic either way  */
    PVAE_FLAG_NO_BACKWARD = 2 /* forward + losses only (tm:147-156 test loop, parity probes) */
};

/* Architecture.  Mirrors the dict keys of tpv:247-286 / gen_layers tpv:180-192:
 * TE: 2*Db -> te_width x te_depth -> 2*Z (Z with PVAE_PRIOR_HYPERSPHERE) ; MD: Db+Z -> ... -> Da ;
 * WM: Db+Da -> ... -> Db ; PR (PVAE_PRIOR_STATE_MEAN): Db -> pr_width x pr_depth -> Z.
 * act_kind after every hidden layer (ReLU unless set), linear output layer. */
typedef struct pvae_config {
    int32_t dim_body;   /* Db */
    int32_t dim_action; /* Da */
    int32_t latent;     /* Z  */
    int32_t te_width, te_depth;
    int32_t md_width, md_depth;
    int32_t wm_width, wm_depth;
    int32_t max_batch;  /* largest minibatch (rows) this ctx will be asked to process */
    int32_t lookahead;  /* L >= 1: steps unrolled through the world model per sample (tpv:277,
                           367-428); sizes the workspace (L blocks per panel) */
    int32_t prior_kind; /* PVAE_PRIOR_* (0 = the reference's working default)                 */
    int32_t pr_width, pr_depth; /* learned prior stack (PVAE_PRIOR_STATE_MEAN only; else ignored) */
    int32_t act_kind;   /* PVAE_ACT_*: hidden activation of every stack = the trainer's "act_fn" (tpv:262;
                           get_activation_fn rmt:30-46).  0 = relu, so a zeroed field keeps the default */
    /* Stacks gen_layers cannot emit but FC accepts (rmt:234-270: any list of fc layers, each with its own
     * hidden_size and activation; reached through custom_model_config's *_layers, rmt:462-510).  Indexed
     * [PVAE_NET_*][hidden layer]; zeroed = the uniform stack described by <net>_width / act_kind. */
    int32_t layer_width[PVAE_NUM_NETS][16]; /* > 0: width of that hidden layer instead of <net>_width    */
    int32_t layer_act[PVAE_NUM_NETS][16];   /* > 0: 1 + PVAE_ACT_* of that hidden layer instead of act_kind */
    /* `task_encoder_inputs` / `motor_decoder_inputs` (rmt:470, 485; 607-613, 646-653, 776-783, 822-829): which of
     * (body, task) the encoder reads -- s_t, s_{t+1} -- and which of (body, task) the decoder reads -- s_t, z.
     * PVAE_INPUT_* bits; 0 = both (a zeroed field keeps the default).  A subset is a COLUMN WINDOW of the same
     * first-layer weight block: the block keeps its full-width layout [body | task], the columns outside the window
     * are structural zeros (never touched by initialisation or checkpoints, and kept exactly zero by training: the
     * staged input panels hold zeros there, so their weight gradient is exactly zero); pvae_layer_info reports the
     * window (n_in, col0), so a checkpoint tensor has the reference's shape. */
    int32_t te_inputs, md_inputs;
    /* motor_decoder_helper_enable / _layers / _range (rmt:490-498): mh_depth hidden layers (0: no helper) of mh_width
     * (layer_width / layer_act[PVAE_NET_MH] per layer), an output layer of dim_action values ending in tanh, input =
     * the decoder's (md_inputs applies); needs mh_range > 0 (asserted upstream, rmt:672-673). */
    int32_t mh_width, mh_depth;
    float mh_range;
} pvae_config;
#define PVAE_INPUT_BODY 1
#define PVAE_INPUT_TASK 2

typedef struct pvae_layer_info {
    int32_t net;       /* PVAE_NET_*                                    */
    int32_t index;     /* position inside the stack (0 = first Linear)  */
    int32_t n_in;      /* checkpoint shape is [n_out, n_in]             */
    int32_t n_out;
    int32_t ld;        /* row stride of W in floats (n_in padded to 64) */
    int32_t n_out_pad; /* rows allocated (n_out padded to 64)           */
    int64_t w_offset;  /* float offset of W[0][0] in the arena          */
    int64_t b_offset;  /* float offset of bias[0] in the arena          */
    int32_t act;       /* PVAE_ACT_* applied to this layer's output (PVAE_ACT_LINEAR for the output layer) */
    int32_t col0;      /* first column of the checkpoint tensor inside the block's rows: W[r][c] of the checkpoint is
                          arena[w_offset + r * ld + col0 + c] (non-zero only for a first layer on an input subset) */
} pvae_layer_info;

/* Loss weights and Adam hyper-parameters of one optimizer step.
 * tpv:331-335 (phase coefficients), tm:119-122 (Adam defaults), tm:158-159 (lr from StepLR). */
typedef struct pvae_step_params {
    float a_rec_coeff;     /* motor_decoder_a_rec_coeff (1.0)  */
    float kl_coeff;        /* vae_kl_coeff (beta)              */
    float s_rec_coeff;     /* world_model_s_rec_coeff          */
    float cycle_coeff;     /* vae_cycle_coeff (1e-3)           */
    double lr;             /* learning rate of this epoch (double: torch keeps these as   */
    double beta1, beta2, adam_eps; /* Python floats; 1 - beta and the bias corrections are   */
                           /* formed in double before rounding to fp32)               */
    int32_t adam_t[PVAE_NUM_NETS]; /* 1-based Adam step count per net for THIS update */
    int32_t global_rows;   /* rows of the global minibatch (= rows on 1 GPU); losses and
                              gradients are scaled by 1/global_rows so a sum all-reduce over
                              ranks yields the reference's batch-mean gradient */
    uint64_t rng_seed;     /* Philox key when eps == NULL      */
    uint64_t rng_offset;   /* (global step, first global row) -> counter */
    int32_t loss_kind;     /* PVAE_LOSS_MSE (nn.MSELoss, the trainer's setting) or PVAE_LOSS_L1 */
    float weight_decay;    /* torch.optim.Adam's L2 term: g <- g + weight_decay * p before the moments
                              (tm:119-122; the trainer's config holds 0.0, tpv:253) */
} pvae_step_params;

/* ---- layout queries (pure host arithmetic, no GPU needed) --------------------------- */
int pvae_abi_version(void);
const char* pvae_last_error(void);
int pvae_num_layers(const pvae_config* cfg);
int pvae_layer(const pvae_config* cfg, int i, pvae_layer_info* out);
int64_t pvae_arena_floats(const pvae_config* cfg);
/* contiguous [offset, offset+count) of one net inside the arena */
int pvae_net_segment(const pvae_config* cfg, int net, int64_t* offset, int64_t* count);
size_t pvae_workspace_bytes(const pvae_config* cfg);
/* Float offset of a workspace panel (inspection / tests).  kind: 0 = input panel of `net`
 * [Bp][ld0], 1 = its gradient, 2 = output of layer `layer` [Bp][n_out_pad], 3 = gradient wrt the
 * pre-activation of that layer, 4 = target next-state panel, 5 = target action panel,
 * 6 = eps [Bp][Z], 7 = dense copy [rows][2 Db] of the observation rows of the last pvae_infer /
 * pvae_infer_logits call with <= 4 rows (what a deferred read of that forward -- mu / logvar, the
 * world model's prediction, the value estimate -- re-uses, so the caller may recycle its buffer).
 * Bp = max_batch rounded up to 32.  Returns <0 on bad arguments.
 * (Kinds 0, 4, 5 name the FIRST of the two sets of staging panels; pvae_train_step_prefetch /
 * pvae_dp_train_step alternate between the two, so inspect these panels only around the plain
 * pvae_gather / pvae_set_batch entry points.) */
int64_t pvae_workspace_offset(const pvae_config* cfg, int kind, int net, int layer);

/* ---- context ------------------------------------------------------------------------ */
int pvae_create(const pvae_config* cfg, pvae_ctx** out);
void pvae_destroy(pvae_ctx* ctx);
/* params / grads / exp_avg / exp_avg_sq: device arenas of pvae_arena_floats() floats.
 * Replaces the per-tensor storage of nn.Linear + torch.optim.Adam state (tm:119-122). */
int pvae_bind_arenas(pvae_ctx* ctx, float* params, float* grads, float* exp_avg,
                     float* exp_avg_sq);
int pvae_bind_workspace(pvae_ctx* ctx, void* workspace, size_t bytes);
/* Demonstration set resident in HBM, de-duplicated: states[n_rows][Db], actions[n_rows][Da]
 * (fp32, dense), window_row[n_windows] = row of s_t (s_{t+1} is row+1, a_t is the same row;
 * with lookahead L the window spans rows row .. row+L, which must stay inside one episode).
 * Replaces the float64 X[N,1,2Db] / Y[N,1,Da] arrays of load_dataset_for_PhysicsVAE
 * (tpv:117-164) and DatasetBase.__getitem__ (tm:52-56). */
int pvae_bind_dataset(pvae_ctx* ctx, const float* states, const float* actions,
                      const int32_t* window_row, int64_t n_rows, int64_t n_windows);
/* Optional, after pvae_bind_dataset: `next_states` [n_rows][Db] fp32, row-aligned with `states`; row r
 * holds what follows state r in a window -- the second half of x and the state-reconstruction target
 * are then read from next_states[row] instead of states[row + 1].  This is how cond = "rel" of
 * load_dataset_for_PhysicsVAE (tpv:149-150: x = [s_t | s_{t+1} - s_t]) reaches the gather kernel: the
 * differences are formed once in float64 on the host, as the reference does.  NULL restores the
 * default; a new pvae_bind_dataset resets it. */
int pvae_bind_dataset_next(pvae_ctx* ctx, const float* next_states);
/* First layers that read the demonstration set where it lies (the torch.cat sites tpv:377, rmt:829, 842
 * as address arithmetic inside the kernels' loaders).  The training-step entry points (pvae_train_step, _prefetch,
 * pvae_dp_train_step) stage NOTHING when the step qualifies: the first layer of every stack fetches its rows of
 * `states` / `actions` itself (16-byte chunks from 4-byte-aligned rows), the z / a_hat column blocks come from where the
 * sampler / the decoder left them, and the two targets s_{t+1}, a_t are read from the set by the loss epilogues.  A step
 * qualifies at lookahead 1 with the default prior, more than 4 rows, first layers wide enough for the 32x32 / 64-row
 * tile kernels, no `next_states` array, and dataset allocations that are readable 16 bytes past their last row (checked
 * at pvae_bind_dataset with hipMemGetAddressRange: a chunk may reach 12 bytes past a row).  Everything else -- and
 * pvae_gather / pvae_set_batch + pvae_forward_backward always -- goes through the staging launch as before; both paths
 * give the same bits.  OPT-IN: pvae_set_direct(ctx, 1) (default off -- at BASELINE's 256 rows per GPU the staged step is
 * the faster one: its gather rides in the previous step's last launch); pvae_direct_active tells
 * whether a step with these arguments would take the direct path (1 / 0). */
int pvae_set_direct(pvae_ctx* ctx, int on);
/* Switches of schedule and tile geometry.  The library reads NO environment variable: what rounds 1-4 switched through
 * PVAE_* variables is set here (the Python host, physicsvae_amd/engine.py, still maps those variables onto these calls for
 * the tests and the A/B scripts).  Defaults are the production values.
 *   ctx == NULL, process-wide kernel geometry:  "ws64" "ws6464" "ws6464_rows" "pair64" "dgrad16" (0 / 1), "wgrad32" (0 / 1 / 2),
 *       "krot" "rowxcd" (0 / 1, experiments, default 0), "look_pair" "rollout_fused" (0 / 1)
 *   ctx, that context's schedule:  "pair" "defer_adam" "same_layer" "fold_sampler" (0 / 1), "direct" (= pvae_set_direct),
 *       "p2p_timeout_ms" (> 0), "p2p_selftest_flags_only" (0 / 1), "server_mailbox" (0 auto, 1 pinned host memory, 2 device)
 * Every variant gives the same bits as the default (held by tests/test_gpu_shapes.py, test_gpu_fuzz.py).  -1: unknown name. */
int pvae_set_option(pvae_ctx* ctx, const char* name, int64_t value);
int pvae_direct_active(pvae_ctx* ctx, int phase, int32_t rows, const pvae_step_params* sp, int fused);

/* ---- hot path ------------------------------------------------------------------------ */
/* Minibatch gather: windows [first_window, first_window+rows) -> network input panels.
 * Replaces DataLoader + default collate over DatasetBase (tm:166-175, 137-139). */
int pvae_gather(pvae_ctx* ctx, int64_t first_window, int32_t rows, void* stream);
/* Declare the staging panels dirty: the staged training minibatch and a minibatch gathered ahead
 * (pvae_train_step_prefetch) are forgotten, so the next training call gathers again and
 * pvae_forward_backward without a new gather fails with "staged rows" instead of running on
 * overwritten panels.  Needed by callers that replay a captured HIP graph of pvae_infer: the graph
 * writes the panels that were current when it was captured, and the host-side bookkeeping of
 * pvae_infer does not run on replay.  Host-side only; launches nothing. */
int pvae_invalidate_staging(pvae_ctx* ctx);
/* Same, from explicit device tensors x[rows][L][2*Db], y[rows][L][Da] (dense fp32, L =
 * lookahead): the compute_loss(y, x) entry of tpv:361 for callers that bring their own batch. */
int pvae_set_batch(pvae_ctx* ctx, const float* x, const float* y, int32_t rows, void* stream);

/* Forward + losses (+ backward, + optional fused Adam) over the batch staged by
 * pvae_gather/pvae_set_batch.  Replaces compute_loss (tpv:361-435), PhysicsVAE.forward
 * (rmt:742-853), loss.backward() (tm:142) and, with PVAE_FLAG_FUSED_ADAM,
 * optimizer.step() (tm:143).
 *   eps      : device [L][rows][Z] standard-normal draws for the reparameterisation
 *              (rmt:734-740; one [rows][Z] slice per unrolled step, in call order), or NULL to
 *              draw them on chip (Philox4x32-10, Box-Muller; step t uses rng_offset + t).
 *   loss_out : device float[5] = {total, loss_a, loss_kl, loss_s, loss_cyc} (this rank's
 *              share: sums over local rows / global_rows).
 * World phase, lookahead 1: only the world model runs (the reference's discarded TE/MD/VB
 * forward, tpv:378, is not algorithmically required and is skipped).
 * lookahead L > 1 (tpv:367-428): step t+1 starts from the world model's own prediction of step
 * t, so encoder, sampler, decoder and world model run for every step in BOTH phases, the
 * backward pass walks the steps last to first through all three stacks, and each layer's weight
 * gradient is one contraction over the L stacked steps.  Terms are means over the L steps. */
int pvae_forward_backward(pvae_ctx* ctx, int phase, int32_t rows, const pvae_step_params* sp,
                          const float* eps, float* loss_out, int flags, void* stream);
/* Adam over the nets in net_mask (bit n = PVAE_NET_n) using the bound grads arena.
 * Replaces torch.optim.Adam.step (tm:143) for the data-parallel path (after the gradient
 * all-reduce). */
int pvae_adam(pvae_ctx* ctx, int net_mask, const pvae_step_params* sp, void* stream);

/* Staged variant for data-parallel training (overlap of the gradient all-reduce with backward):
 *   pvae_forward_seed          forward + loss partials + gradient seeds (no backward launches)
 *   pvae_backward_stage(k)     launch k of the backward pass, k = 0 .. *num_stages-1 in order;
 *                              [*ready_offset, +*ready_count) is the slice of the GRADIENT arena
 *                              (net *ready_net) that is final after this stage (count 0: none) --
 *                              the caller may start its all-reduce immediately; loss_out (may be
 *                              NULL) is finalised by the last stage
 *   pvae_adam_segment          Adam on one such slice once its reduction has completed
 * Together they replace loss.backward() + optimizer.step() (tm:142-143) on N GPUs. */
int pvae_forward_seed(pvae_ctx* ctx, int phase, int32_t rows, const pvae_step_params* sp, const float* eps,
                      void* stream);
int pvae_backward_stage(pvae_ctx* ctx, int phase, int32_t rows, const pvae_step_params* sp, int stage,
                        float* loss_out, void* stream, int64_t* ready_offset, int64_t* ready_count,
                        int* ready_net, int* num_stages);
int pvae_adam_segment(pvae_ctx* ctx, int net, int64_t offset, int64_t count, const pvae_step_params* sp,
                      void* stream);
/* The backward pass of a (phase, sp) step as pvae_backward_stage would walk it, WITHOUT launching anything: stage k
 * finishes the slice [offset[k], +count[k]) of net[k] (count 0: none).  At most `max` entries are written; *num_stages
 * is the total.  Does not depend on the rows of the minibatch nor on what is staged -- so a rank whose shard of a ragged
 * last global minibatch is EMPTY (no launch at all on it) can issue the very sequence of collectives its peers issue
 * behind their stages (tm:142-143 on N GPUs; physicsvae_amd/torch_models.py dp_step). */
int pvae_backward_plan(pvae_ctx* ctx, int phase, const pvae_step_params* sp, int64_t* offset, int64_t* count, int* net,
                       int max, int* num_stages);

/* Data-parallel exchange issued by the library itself (RCCL over xGMI; one process per GPU).
 * Replaces what DistributedDataParallel would add around tm:142-143; the reference has no
 * multi-GPU path (tm:115-118 moves one model to one device).
 *   pvae_comm_unique_id   rank 0: 128-byte RCCL id, to be handed to every rank by the caller
 *                         (torch.distributed broadcast, a file, MPI ...)
 *   pvae_comm_init        collective: every rank joins the communicator (rank, world, id)
 *   pvae_allreduce_grads  in-place SUM all-reduce of a slice of the gradient arena, stream-ordered
 *                         on `stream` like any other launch of this library (no side stream)
 *   pvae_dp_train_step    one data-parallel optimizer step in ONE call: gather + forward +
 *                         backward (gradients scaled by 1/global_rows) + all-reduce + Adam.
 *                         Default: one reduction per stack, in line on `stream` as soon as the
 *                         stack's gradient is final (no cross-stream hand-off: one costs
 *                         ~18 us of idle GPU each way on this runtime).
 *                         With bucket_bytes > 0 (pvae_comm_config) the exchange is bucketed
 *                         and overlapped instead: a stack's gradient becomes final last layer
 *                         first, and every time whole layers worth bucket_bytes are done
 *                         that bucket is reduced and Adam-applied on the library's own
 *                         high-priority exchange stream while `stream` keeps launching the
 *                         rest of the backward pass; `stream` rejoins (event wait) before the
 *                         call returns.  Bucket boundaries depend on the layout and
 *                         bucket_bytes only -- identical on every rank.  rows may be 0 (empty
 *                         shard of a ragged last global batch);
 *                         [next_first, +next_rows) = this rank's shard of the following step
 *                         (0 rows: unknown), gathered inside this step's last launch as in
 *                         pvae_train_step_prefetch.
 * The RCCL library is resolved at run time (the copy PyTorch already loaded, else the system
 * one); a missing library is an error from these calls only. */
int pvae_comm_unique_id(void* id128);
int pvae_comm_init(pvae_ctx* ctx, int rank, int world, const void* id128);
int pvae_comm_destroy(pvae_ctx* ctx);
/* What the ctx's communicator itself reports (ncclCommUserRank / ncclCommCount): *nranks = 0 when
 * the ctx has no communicator.  bench.py prints this as `rccl_ranks`, so that the number of ranks
 * in the JSON line comes from RCCL and not from a command-line flag. */
int pvae_comm_info(pvae_ctx* ctx, int* rank, int* nranks);
/* Exchange settings of pvae_dp_train_step (same values on every rank): bucket_bytes = 0: one bucket per
 * stack, reduced in line on the caller's stream; > 0: buckets of that size, reduced and applied on the
 * library's exchange stream while the backward pass continues.  Until this is called (or
 * PVAE_DP_BUCKET_MB is set at pvae_comm_init) the library chooses per step: in line with one rank and in
 * the world phase, 6 MiB overlapped buckets in the joint phase with more than one rank (the decoder's
 * reduction then hides behind the encoder's backward pass); test_delay_us > 0 puts a spin kernel of that length in
 * front of every reduction (ordering tests).  Overlap pays when the backward work still to be
 * launched after a bucket closes exceeds the two hand-offs (long stacks, lookahead > 1, slow
 * links); with the caller on the NULL stream also set GPU_MAX_HW_QUEUES=8 before HIP starts --
 * with the default 4 hardware queues the exchange stream can share one with the NULL stream and
 * every hand-off then stalls ~250 us. */
int pvae_comm_config(pvae_ctx* ctx, int64_t bucket_bytes, int32_t test_delay_us);
/* How pvae_dp_train_step exchanges a bucket (same value on every rank, chosen before the first step
 * and not changed afterwards):
 *   PVAE_EXCHANGE_ALLREDUCE (default)  ncclAllReduce of the gradient, then every rank applies the same
 *                                      Adam update to the whole bucket (replicated moments);
 *   PVAE_EXCHANGE_SHARDED              ncclReduceScatter (rank r ends with the summed gradient of ITS
 *                                      1/N slice), Adam on that slice only (1/N of the p, g, m, v
 *                                      traffic per rank), ncclAllGather of the updated parameter
 *                                      slices.  Each rank then holds valid Adam moments for its own
 *                                      slice only (ZeRO-1 shaped).  Buckets whose length is not a
 *                                      multiple of 4*N floats fall back to the all-reduce form.
 *   PVAE_EXCHANGE_P2P                  no RCCL: the direct all-pairs exchange of SURVEY.md section 8e over
 *                                      peer-mapped arenas (pvae_p2p_export / pvae_p2p_open below).  ONE
 *                                      launch per bucket: rank r signals "my gradient is final", waits for
 *                                      every peer's signal, reads slice r of every rank's gradient arena
 *                                      directly (7 links in parallel on an 8-GPU xGMI mesh), sums in RANK
 *                                      ORDER (so every element is reduced once, by its owner, the same way:
 *                                      replicas stay bit-identical by construction), applies Adam to its
 *                                      slice and writes the updated parameters straight into every peer's
 *                                      parameter arena; the last workgroup tells the peers it is done and
 *                                      waits for theirs.  Moments as in the sharded form (ZeRO-1 shaped).
 * PVAE_DP_SHARDED=1 in the environment selects the sharded form at pvae_comm_init. */
enum { PVAE_EXCHANGE_ALLREDUCE = 0, PVAE_EXCHANGE_SHARDED = 1, PVAE_EXCHANGE_P2P = 2,
       PVAE_EXCHANGE_LOCAL = 3 /* MEASUREMENT ONLY: no exchange at all, every rank applies Adam to its own
                                  gradient (replicas diverge) -- the step time without the collective, against
                                  which bench.py reports what an exchange adds */,
       PVAE_EXCHANGE_P2P_PUSH = 4 /* PVAE_EXCHANGE_P2P with remote WRITES only: every rank pushes its contribution to
                                  slice q into owner q's staging buffer, the owner sums its slice from LOCAL memory in
                                  rank order, applies Adam and pushes the parameters (same result bit for bit; links
                                  that favour posted writes over read round trips want this form) */ };
int pvae_comm_mode(pvae_ctx* ctx, int mode);
/* Peer-mapped exchange (PVAE_EXCHANGE_P2P).  The reference has no counterpart (tm:131-161 is one process);
 * this is what `north_star` asks for on top of it.  Set-up, after pvae_bind_arenas:
 *   pvae_p2p_export   writes PVAE_P2P_BLOB_BYTES describing this rank's gradient arena, parameter arena, flag
 *                     block and staging buffer (hipIpcGetMemHandle of the allocations they live in + offsets).  The arenas
 *                     must come from hipMalloc (PyTorch's default allocator does; expandable segments do not).
 *   (caller)          all-gathers the blobs of all ranks, by any transport (torch.distributed, a file, MPI)
 *   pvae_p2p_open     collective in effect: maps every peer's three buffers (hipIpcOpenMemHandle) -- peers may
 *                     be other GPUs of the node (xGMI) or other processes on the SAME device (functional tests
 *                     on a 1-GPU box).  world <= PVAE_P2P_MAX_RANKS.
 *   pvae_p2p_status   *timeouts = waits that gave up (a peer never signalled within the time-out, default 20 s,
 *                     PVAE_P2P_TIMEOUT_MS): results are then garbage and the caller must stop.  Synchronises
 *                     `stream`.  *rank / *world as passed to pvae_p2p_open (0 ranks: not open).
 * pvae_dp_train_step then runs with or without an RCCL communicator. */
#define PVAE_P2P_BLOB_BYTES 512
#define PVAE_P2P_MAX_RANKS 8
int pvae_p2p_export(pvae_ctx* ctx, void* blob);
int pvae_p2p_open(pvae_ctx* ctx, int rank, int world, const void* blobs);
int pvae_p2p_close(pvae_ctx* ctx);
int pvae_p2p_status(pvae_ctx* ctx, int* rank, int* world, uint32_t* timeouts, void* stream);
/* Collective in effect (every rank calls it after pvae_p2p_open): (1) each rank writes a record into every peer's flag
 * block, signals, checks the records that arrived in its own block and reads its records back from the peers' --
 * remote write, remote read and flag delivery are proven (or fail within 1 s) before the first training step;
 * (2) the CACHED arenas, through the exchange's own access paths: a test region of the parameter arena and of the
 * staging buffer is first read into this device's L2s from every XCD (LDS-DMA and plain loads, what the forward
 * kernels use), the peers overwrite it with the exchange's `buffer_store ... sc0 sc1`, and a fresh dependent launch
 * re-reads it from every XCD -- a stale line is an error; a gradient line is read remotely, overwritten by its owner
 * and read again (reader-side staleness).  The regions are restored.  A failure (-22) means this machine must not
 * run the peer-mapped forms: replicas would stay bit-identical while training on stale weights.
 * PVAE_P2P_SELFTEST_FLAGS_ONLY=1 skips (2).  Synchronises `stream`. */
int pvae_p2p_selftest(pvae_ctx* ctx, void* stream);
/* Zero the count pvae_p2p_status reports (after the caller has handled the time-outs, e.g. dropped a calibration
 * candidate and restored its snapshot).  Since ABI 7 a wait that gives up also ABORTS that rank's part of the
 * exchange launch: no gradient is summed, no moment moves, nothing is pushed to the peers.  Synchronises `stream`. */
int pvae_p2p_clear_errors(pvae_ctx* ctx, void* stream);
/* One bucket through the peer-mapped exchange, stream-ordered like any launch of this library: the slice
 * [offset, offset + count) of the gradient arena (inside stack `net`, float4-aligned) is summed over the ranks
 * by its owners, Adam-applied and the updated parameters written to every rank -- what pvae_dp_train_step does
 * per bucket in PVAE_EXCHANGE_P2P mode.  Every rank must issue the same sequence of calls. */
int pvae_p2p_exchange(pvae_ctx* ctx, int net, int64_t offset, int64_t count, const pvae_step_params* sp,
                      void* stream);
/* Which ranges of the arenas THIS rank keeps valid Adam moments for, for stack `net` trained in `phase`, under the
 * current exchange mode and bucket settings: one entry per exchange bucket -- the rank's slice of it (sharded / peer-
 * mapped forms; replicated[i] = 0) or the whole bucket (all-reduce form and buckets that fall back to it;
 * replicated[i] = 1: every rank holds the same values).  What a checkpoint writer needs to assemble full moments
 * (physicsvae_amd TrainModel.gather_moments).  *n entries are written (at most `max`). */
int pvae_owned_slices(pvae_ctx* ctx, int phase, int net, int64_t* offsets, int64_t* counts, int32_t* replicated,
                      int32_t max, int32_t* n);
int pvae_allreduce_grads(pvae_ctx* ctx, int64_t offset, int64_t count, void* stream);
int pvae_dp_train_step(pvae_ctx* ctx, int phase, int64_t first_window, int32_t rows,
                       const pvae_step_params* sp, const float* eps, float* loss_out,
                       int64_t next_first, int32_t next_rows, void* stream);

/* One whole optimizer step on the bound dataset: gather + forward/backward + Adam.
 * This is the body of the `for data in self.train_loader` loop (tm:137-144). */
int pvae_train_step(pvae_ctx* ctx, int phase, int64_t first_window, int32_t rows,
                    const pvae_step_params* sp, const float* eps, float* loss_out, void* stream);

/* pvae_train_step that also gathers the NEXT minibatch (windows [next_first, +next_rows), 0 rows:
 * none): the gather rides as extra workgroups in this step's last launch and lands in a second set
 * of input panels; the following call finds its minibatch already staged and skips the gather
 * launch (falls back to a normal gather whenever what was prefetched is not what is asked for).
 * Same results as pvae_train_step, one launch less per step (lookahead 1). */
int pvae_train_step_prefetch(pvae_ctx* ctx, int phase, int64_t first_window, int32_t rows,
                             const pvae_step_params* sp, const float* eps, float* loss_out,
                             int64_t next_first, int32_t next_rows, void* stream);

/* ---- inspection (parity tests) ------------------------------------------------------- */
/* Copy a forward intermediate of the last pvae_forward_backward into dst (dense
 * [rows][width] fp32, device).  what: 0 = mu, 1 = logvar, 2 = z, 3 = a_hat (MD output),
 * 4 = s2_hat (WM output; with lookahead > 1 the prediction that feeds the next step), 5 = eps
 * actually used (PVAE_PRIOR_HYPERSPHERE: the unit prior sample u), 6 = prior mean mu_p
 * (PVAE_PRIOR_STATE_MEAN).  With PVAE_PRIOR_HYPERSPHERE 0 reads the raw encoder output e, 2 the
 * unit vector z = e/|e|, and 1 is not available.  Add 8*t to read time step t of a
 * lookahead > 1 batch. */
int pvae_read_tensor(pvae_ctx* ctx, int what, float* dst, int32_t rows, void* stream);

/* Rollout inference (rmt:742-771 at small batch): obs[rows][2*Db] -> a_hat[rows][Da]
 * (+ optional s2_hat[rows][Db], mu/logvar/z).  eps NULL + noise=0 -> z = mu
 * (latent_prior_noise False). */
int pvae_infer(pvae_ctx* ctx, const float* obs, int32_t rows, const float* eps, int noise,
               uint64_t rng_seed, uint64_t rng_offset, float* a_hat, float* s2_hat, float* z_out,
               void* stream);
/* 1 when calls of <= 4 rows take the rollout path that keeps the observation rows for deferred reads (rmt:742-771's
 * _cur_* state) and leaves a staged training minibatch alone; 0 under PVAE_ROLLOUT_FUSED=0 (read once per process by
 * the LIBRARY: callers ask here instead of reading the environment themselves). */
int pvae_rollout_is_fused(void);
/* pvae_infer with the module's output layout (rmt:742-771 + AppendLogStd rmt:160-206): the action lands in
 * logits[r * ld_logits + 0 .. Da) and, when log_std (device, [Da]) is given, log_std behind it --
 * logits = [a_hat | log_std], what PhysicsVAE.forward returns to RLlib's action distribution. */
int pvae_infer_logits(pvae_ctx* ctx, const float* obs, int32_t rows, const float* eps, int noise,
                      uint64_t rng_seed, uint64_t rng_offset, float* logits, int32_t ld_logits,
                      const float* log_std, float* s2_hat, float* z_out, void* stream);
/* ---- call-persistent rollout server (the single-launch rollout path) ------------------------
 * Replaces, for the 30 Hz control loop's forward at B = 1, PhysicsVAE.forward (rmt:742-771: task encoder ->
 * sampler rmt:734-740 -> motor decoder; callers envs/rllib_env_imitation.py:215-266) by ZERO launches per call: one
 * kernel stays resident on the 32 CUs of one XCD with the encoder's and the decoder's weights in LDS (1/32 of every
 * layer's output features per workgroup), polls a mailbox in pinned host memory, walks the layers -- every value handed
 * from a layer to the next as one tagged 8-byte word that the consumer polls in that XCD's L2, no barrier -- and writes
 * the action back to the mailbox.  All pointers below are HOST pointers; the
 * calls are plain host functions (no stream, no launch) once the server runs.
 *   pvae_rollout_server_start   plans the LDS layout, allocates the mailbox and launches the kernel on a stream of its own;
 *                               returns once it serves.  scope 0: one XCD (32 workgroups; the other seven XCDs stay free
 *                               for training launches) when 1/32 of every layer fits a CU's LDS, else the whole chip (256
 *                               workgroups, 1/256 of every layer each: 4x1024 stacks take 122 KB per CU -- kernels that
 *                               need more than the remaining LDS then wait for the server to leave); 1 / 2 force either
 *                               (PVAE_SERVER_SCOPE=xcd|chip as well).  A hand-over word costs 0.43-0.53 us inside an XCD,
 *                               0.51-0.64 us across XCDs (tools/xcd_pingpong.hip).  -24 with a message when nothing fits.  The kernel
 *                               leaves by itself after idle_timeout_ms without a request (default 100 ms; the next
 *                               pvae_rollout_server_infer brings it back) and never lives longer than lifetime_s
 *                               (default 600 s).  Arguments <= 0 keep the current / default values.  While it is
 *                               resident a DEVICE-wide synchronisation waits for it (at most the idle time-out).
 *   pvae_rollout_server_infer   obs[2*Db] -> a_hat[Da] (+ mu_logvar[2*Z], z[Z] when not NULL): the same values as
 *                               pvae_infer(rows = 1, eps = NULL, noise, rng_seed, rng_offset), bit for bit.  reload != 0:
 *                               the weights are copied from the parameter arena into LDS again first (after an
 *                               optimizer step or load_weights*, rmt:870-928).  Blocks at most timeout_ms (<= 0: 1 s).
 *                               Optimizer steps issued through THIS library (pvae_train_step*, pvae_dp_train_step,
 *                               pvae_adam*, pvae_p2p_exchange) are noticed: the next request waits for their stream and
 *                               re-reads the weights by itself.  Parameters written behind the library's back (a
 *                               framework's load_state_dict into the arena, rmt:870-928) are announced with
 *                               pvae_params_changed.
 *   pvae_rollout_server_infer_rows  the same for 1-4 observations in one request.  A server starts as the single-row
 *                               instance of the kernel; the first request with more rows swaps in the multi-row instance
 *                               (once, a few hundred microseconds; it needs LDS for four input vectors -- -24 if that no
 *                               longer fits, the single-row instance then stays); it serves until pvae_rollout_server_stop.
 *                               A model with the motor decoder's helper
 *                               (pvae_config.mh_depth > 0) is served with the helper's layers behind the decoder's.
 *   pvae_rollout_server_stop    ends the kernel (also done by pvae_destroy).
 *   pvae_rollout_server_status  *lds_bytes < 0: dealt out over the whole chip (|value| bytes per workgroup).
 *                               *serving != 0 while the kernel is resident and answering: 2 when the request block lives in
 *                               DEVICE memory that the host writes through the BAR (hipDeviceAttributeIsLargeBar; the host
 *                               pushes the observation, the kernel polls local memory), 1 when it lives in pinned host
 *                               memory that the kernel pulls from (PVAE_SERVER_MAILBOX=host forces this form). */
int pvae_rollout_server_start(pvae_ctx* ctx, double idle_timeout_ms, double lifetime_s, int scope);
int pvae_rollout_server_infer(pvae_ctx* ctx, const float* obs, int noise, uint64_t rng_seed, uint64_t rng_offset, int reload,
                              float* a_hat, float* mu_logvar, float* z, double timeout_ms);
/* pvae_rollout_server_infer for 1 <= rows <= 4 observations in ONE request (rmt:742-771 serves any batch):
 * obs[rows][2*Db] -> a_hat[rows][Da] (+ mu_logvar[rows][2*Z], z[rows][Z]); row r draws Philox row r, as pvae_infer does --
 * bit-identical to pvae_infer on the same rows.  Every weight fragment read from LDS feeds all rows. */
int pvae_rollout_server_infer_rows(pvae_ctx* ctx, const float* obs, int32_t rows, int noise, uint64_t rng_seed,
                                   uint64_t rng_offset, int reload, float* a_hat, float* mu_logvar, float* z, double timeout_ms);
/* forward_decoder at B = 1 (rmt:822-837; the "pass_through" rollout of envs/rllib_env_imitation.py:233-258, where the caller draws
 * z itself): s1_z = [s1 (Db) | z (Z)] (host) -> a_hat[Da] (host).  The encoder's layers are skipped.
 * Without a helper: the same bits as pvae_net_forward(PVAE_NET_MD) on that row.  With one (pvae_config.mh_depth > 0) the
 * reply is decoder + mh_range * helper (rmt:833-835), formed by ONE fma in the kernel (as the served full request does):
 * within an ulp of a host that adds mh_range * h to pvae_net_forward's row itself (tests/test_gpu_rollout_server.py). */
int pvae_rollout_server_decode(pvae_ctx* ctx, const float* s1_z, float* a_hat, double timeout_ms);
int pvae_rollout_server_stop(pvae_ctx* ctx);
/* The caller wrote the parameter arena itself (load_state_dict / load_weights*, rmt:870-928; a torch optimizer) with work
 * queued on `stream` (NULL: the default stream): whatever holds a copy of the parameters -- the rollout server's LDS -- refreshes it before its
 * next answer, after that stream has drained. */
int pvae_params_changed(pvae_ctx* ctx, void* stream);
/* Measurement: n requests back to back with one observation, us[i] = host observation -> host action of request i on the
 * host's steady clock, taken inside the call (a compiled host's view; tools/infer_latency.py reports it next to Python's). */
int pvae_rollout_server_selfbench(pvae_ctx* ctx, const float* obs, int noise, int32_t n, double* us);
/* Measurement: where the LAST request's time went on the device (wall-clock stamps of workgroup 0, microseconds after it saw
 * the request): us[1 + 2 l] = layer l's inputs in LDS, us[2 + 2 l] = layer l's outputs published, us[1 + 2 n_layers] =
 * completion word issued, then ONE more entry: the shader clock during the request in MHz (s_memtime ticks per microsecond
 * of the 100 MHz wall clock: a mostly-idle resident kernel lets the part clock down).  *n = entries written (at most max). */
int pvae_rollout_server_timeline(pvae_ctx* ctx, double* us, int32_t max, int32_t* n);
int pvae_rollout_server_status(pvae_ctx* ctx, int32_t* serving, uint32_t* served, int32_t* lds_bytes);

/* A stack of Linear layers on CALLER-owned dense row-major weights (W[i]: [n_out[i]][n_in[i]], row stride
 * ldw[i], no alignment needed; bias may be NULL), hidden activation PVAE_ACT_* (act_kind for every hidden
 * layer, or layer_acts[i] per hidden layer when not NULL), linear output layer -- or, with (1 + PVAE_ACT_*) << 8 added
 * to act_kind, that activation on the output layer (the motor decoder's helper ends in tanh, rmt:491-495, 672).
 * scratch: 2 * rows * (widest hidden layer) floats (device).  No context: this is the value branch of
 * the rollout model (rmt:846-853, value_fn_layers 2*Db -> 256 -> 256 -> 1), whose parameters stay plain
 * torch tensors because the supervised loss never touches them; likewise the motor decoder's helper (rmt:670-680,
 * 833-835), a residual policy for RL fine-tuning that train_physics_vae.py never builds. */
int pvae_mlp_forward(const float* x, int32_t rows, int32_t ldx, int32_t n_layers, const float* const* W,
                     const float* const* bias, const int32_t* n_in, const int32_t* n_out,
                     const int32_t* ldw, int32_t act_kind, const int32_t* layer_acts, float* scratch, float* out,
                     int32_t ld_out, void* stream);

/* One stack on its own: in[rows][n_in] (dense) -> out[rows][n_out] (dense).  The building
 * block behind forward_encoder / forward_decoder / forward_world (rmt:773-844) when a caller
 * drives the stages separately (e.g. EnvRunner pass_through: z ~ N(0,I) straight into the
 * motor decoder, envs/rllib_env_imitation.py:234-266). */
int pvae_net_forward(pvae_ctx* ctx, int net, const float* in, int32_t rows, float* out, void* stream);
/* Sampler on its own (rmt:734-740): mu_logvar[rows][2Z] -> z[rows][Z];
 * eps NULL -> Philox; noise = 0 -> z = mu. */
int pvae_reparam(pvae_ctx* ctx, const float* mu_logvar, int32_t rows, const float* eps, int noise,
                 uint64_t rng_seed, uint64_t rng_offset, float* z_out, void* stream);

/* Per-kernel timing with HIP events on the launch stream (bench.py's `roofline` object).
 * While enabled every contraction launch carries an event pair stamped by the device at the
 * kernel's own start and end (hipExtLaunchKernelGGL), i.e. the duration rocprofv3 --kernel-trace
 * reports, without the launch seam (this serialises the host a little, so it is used in a
 * separate instrumented pass, never in a timed region).
 * category: 0 = forward kernel (32x32 / 64x32 tiles), 5 = forward kernel of the narrow layers (16x16 tiles: output
 * layers, the fused-loss layer), 1 = input-gradient kernel (alone), 2 = weight-gradient(+Adam)
 * launches (single or the two trailing layers in one launch), 3 = fused input-gradient +
 * weight-gradient(+Adam) launch, 4 = the RCCL all-reduce of pvae_allreduce_grads /
 * pvae_dp_train_step (events recorded on the stream around the collective; total_flops then
 * carries the payload BYTES).
 * pvae_profile_read synchronises on the recorded events and returns the summed duration
 * (ms), launch count and ALGORITHMIC flops (2*rows*n_in*n_out on the unpadded dims). */
int pvae_profile_enable(int on);
/* Shader clock the chip sustains while every SIMD issues fp32 MFMAs back to back on `operands` (>= 8192
 * floats of representative data, e.g. the parameter arena; DVFS makes this depend on the data: zeros run at
 * the 2.4 GHz spec clock, trained weights ~10 % lower) and the fp32-MFMA peak that clock allows
 * (CUs x 256 FLOP/clk x clock).  `scratch` = 512 device floats.  Synchronises `stream` (measurement aid for
 * bench.py: context for `roofline.frac`, which is quoted against the 2.4 GHz spec peak). */
int pvae_mfma_clock_probe(const float* operands, int64_t n_operands, float* scratch, double* ghz,
                          double* tflops_peak, void* stream);
int pvae_profile_read(int category, double* total_ms, int64_t* launches, double* total_flops);

/* GEMM micro-entry for kernel-level parity/roofline probes:
 * kind 0: C[M][N] = relu?(A[M][K] * W[N][K]^T + bias)   (forward layer)
 * kind 1: C[M][K] = (dZ[M][N] * W[N][K]) (.* (mask > 0)) (input gradient)
 * kind 2: C[N][K] = dZ[M][N]^T * X[M][K]                 (weight gradient)
 * All dims multiples of 64 (M of 32), leading dims given in floats. */
int pvae_gemm_probe(int kind, const float* a, int lda, const float* b, int ldb, float* c, int ldc,
                    const float* bias_or_mask, int ld_mask, int m, int n, int k, int relu,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PVAE_H */
