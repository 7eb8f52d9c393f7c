/*
 * pmbrl.h -- C ABI of libpmbrl_hip.so: the MI355X (gfx950) implementation of the
 * MC-PILCO particle-rollout + back-prop-through-rollout hot path of
 * mcgillmrl/prob_mbrl.
 *
 * The reference has no FFI (it is pure Python on torch); the boundary it offers
 * is the Python call surface.  Each entry point below names the reference
 * function(s) whose arithmetic it replaces (paths relative to the reference
 * repository root).  The Python host layer (prob_mbrl_amd/) keeps the
 * reference's signatures and calls these through ctypes; see INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no exceptions, no Python/torch types; every pointer named *_d is
 *    a DEVICE pointer owned by the caller (a torch tensor's data_ptr()) that
 *    must stay alive until `stream` has drained; all arrays are contiguous,
 *    row-major, fp32 unless stated.
 *  - return value: 0 = ok, < 0 = usage / HIP error (message via
 *    pmbrl_last_error()).  Numerical failures inside a rollout (non-finite
 *    state, non-positive Cholesky pivot) do not fail the call: they are
 *    reported through the int32 `status_d` word (see pmbrl_rollout_fwd), which
 *    the host turns into the RuntimeError the reference's control flow relies
 *    on (utils/rollout.py:154-157, algorithms/mc_pilco.py:122-131).
 *  - calls on one plan are stream-ordered and not thread-safe; different plans
 *    may be used from different host threads.
 */
#ifndef PMBRL_H
#define PMBRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMBRL_MAX_LAYERS 8  /* Linear layers per network (hidden + output) */
#define PMBRL_MAX_ANGLE 8
#define PMBRL_MAX_TIP 8
#define PMBRL_MAX_DIM 64    /* max state / action / expanded-state width */

#define PMBRL_FLAG_MM_STATES 1   /* utils/rollout.py:121-132 */
#define PMBRL_FLAG_MM_REWARDS 2  /* utils/rollout.py:135-145 */
#define PMBRL_FLAG_INFER_NS 4    /* utils/rollout.py:6-17 (mm_resample_infer_ns_) */
#define PMBRL_FLAG_FORCE_GENERIC 16 /* do not use the latency-optimised kernel variants (tests) */
#define PMBRL_FLAG_NO_SHAPED 32 /* do not use the shape-specialised instantiations (tests) */
#define PMBRL_FLAG_POL_MASKS_PER_STEP 64  /* pol_mask_bits_d[l] is [H, B, ceil(h/16)]: a fresh dropout mask at every
                                            step (utils/rollout.py:95-98 resample_policy=True -> models/modules.py:55-58) */
#define PMBRL_FLAG_DYN_MASKS_PER_STEP 128 /* the same for the dynamics model (resample_model=True, utils/rollout.py:110-115
                                            -> models/modules.py:134-139,155-157).  Either flag selects the general
                                            kernel family (the latency-optimised one keeps the masks in LDS for the launch) */
#define PMBRL_FLAG_GMM_EXACT_NOISE_GRAD 256 /* mixture head: differentiate the noise term with the step's own noise
                                   * (default: with the LAST step's, as the reference's autograd does -- its
                                   * sampler swaps the storage of a saved tensor, models/densities.py:229-231) */
#define PMBRL_MAX_COMP 8          /* mixture components of the dynamics head */
#define PMBRL_FLAG_ZMM_PER_STEP 8 /* z_mm/z_rr are [H, B_global, .] fresh draws per step
                                     (utils/rollout.py:58-59, z=None) instead of the cyclic
                                     PEGASUS buffer of utils/rollout.py:53-57 */

/* Arithmetic of the hidden-width (K >= 32) products of the sweep kernels.  Everything else (first
 * layers, elementwise phases, moment matching, the dW GEMM, the optimiser) is fp32 / fp64 either way.
 *  F32    v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation (1/16 of the bf16 matrix rate)
 *  SPLIT  every fp32 operand as a sum of bf16 pieces on v_mfma_f32_16x16x32_bf16 with fp32
 *         accumulation: three pieces (6 MFMAs per K=32, fp32-equivalent) in the forward sweep, two
 *         (3 MFMAs) in the adjoint sweep -- prob_mbrl_amd/csrc/pmbrl_split.h.  Offered by the
 *         latency-optimised kernel family for workgroups of up to 32 rows; other plans run F32
 *         (pmbrl_plan_info reports the arithmetic in use). */
#define PMBRL_PREC_F32 0
#define PMBRL_PREC_SPLIT 1
/*  SPLIT_F16  as SPLIT, with the forward sweep on TWO fp16 pieces (22 significant bits, 3 MFMAs per K=32,
 *         4 instead of 6 bytes per weight on the weight stream -- which is what bounds the sweeps).  fp16's
 *         range applies to the forward sweep's hidden activations and weights: a value beyond +-65504
 *         becomes inf and the rollout is reported as failed at that step (status word). */
#define PMBRL_PREC_SPLIT_F16 2

#define PMBRL_REWARD_EXP 0 /* r = exp(-w (d'Qd + u'Ru)): envs/cartpole/env.py:41-86 */
#define PMBRL_REWARD_NEG 1 /* r = -w (d'Qd + u'Ru):      envs/rendezvous/env.py:32-45 */

/* One MLP as built by models/core.py:15-99 (mlp): Linear -> ReLU -> dropout ...
 * -> Linear.  dims[0] = input width, dims[n_layers] = output width.
 * keep[i] is the keep-probability the masked activation of hidden layer i is
 * DIVIDED by (BDropout, models/modules.py:61) or 1.0 (eval-mode CDropout,
 * models/modules.py:158-160, and layers without dropout). */
typedef struct pmbrl_mlp {
  int32_t n_layers;
  int32_t dims[PMBRL_MAX_LAYERS + 1];
  float keep[PMBRL_MAX_LAYERS];
} pmbrl_mlp;

/* Analytic reward, the common form of envs/<env>/env.py:*Reward.forward and
 * losses.py:67-75:  phi = expand ? [others, sin(angles), cos(angles)] : x
 * (utils/angles.py:7-42);  delta = (C phi - tip_target) / norm;
 * cost = w (delta' Q delta + u' R u);  r = exp(-cost) | -cost. */
typedef struct pmbrl_reward {
  int32_t kind;   /* PMBRL_REWARD_* */
  int32_t expand; /* 1: expand angle_dims inside the reward */
  int32_t n_angle;
  int32_t angle_dims[PMBRL_MAX_ANGLE];
  int32_t k; /* rows of C (tip coordinates) */
  float C[PMBRL_MAX_TIP * PMBRL_MAX_DIM]; /* [k, De] row-major, De = D + (expand ? n_angle : 0) */
  float tip_target[PMBRL_MAX_TIP];
  float norm;
  float w;
  float Q[PMBRL_MAX_TIP * PMBRL_MAX_TIP]; /* [k, k] */
  float R[PMBRL_MAX_DIM * PMBRL_MAX_DIM]; /* [U, U] */
} pmbrl_reward;

/* Problem shape.  B is the number of particle rows resident on THIS device;
 * rows are laid out as utils/core.py:188-190 (tile): row = particle*S + sample.
 * With moment matching, rows are grouped contiguously: mm_groups groups of
 * B/mm_groups rows (utils/rollout.py:125-129); mm_groups = 0 with an MM flag
 * set means one group of all B rows (the examples' default).
 * For multi-GPU sharding B_global / row_offset locate this shard in the global
 * batch: the cyclic noise index of utils/rollout.py:53-59 is
 * (t + row_offset + b) mod B_global. */
typedef struct pmbrl_config {
  int32_t B, D, U, H;
  int32_t B_global, row_offset;
  int32_t flags;
  int32_t mm_groups;
  float max_log_std_pol; /* models/densities.py:75: log(max_noise_std) */
  float max_log_std_dyn;
  pmbrl_mlp pol; /* dims[0] = D + n_pol_angle, dims[n] = 2U */
  pmbrl_mlp dyn; /* dims[0] = D + U + n_dyn_angle, dims[n] = 2D */
  pmbrl_reward reward;
  int32_t rows_per_wg_hint; /* 0 = choose automatically */
  int32_t precision;        /* PMBRL_PREC_*: arithmetic of the hidden-width GEMMs of the sweeps */
  /* angle_dims of Policy (models/core.py:233-234) and of the dynamics Regressor (models/core.py:173-174):
   * the network sees utils/angles.py:39-42's [other dims in order | sin(angles) | cos(angles)] of the
   * state (policy) or of [state | action] (dynamics; angle dims must be state dims).  mx / iSx of the
   * rollout arguments then have dyn.dims[0] entries, in that feature order. */
  int32_t n_pol_angle;
  int32_t pol_angle_dims[PMBRL_MAX_ANGLE];
  int32_t n_dyn_angle;
  int32_t dyn_angle_dims[PMBRL_MAX_ANGLE];
  /* GaussianMixtureDensity dynamics head (models/densities.py:151-259; examples/deep_pilco_mm.py:117-121):
   * n components, dyn.dims[n_layers] = (2D + 1) n + 1 = [n D means | n D log-stds | n logits | log-temperature],
   * means / log-stds indexed [d * n + c].  0 or 1: the DiagGaussianDensity head (dyn.dims[n_layers] = 2D).
   * Needs z_pi_d and u_cat_d, and z_dyn_d as a per-step draw (z_dyn_step_stride = B * D: the reference redraws
   * the Gaussian noise of this head at every step). */
  int32_t dyn_components;
  /* Moment-matching groups whose rows are spread over several ranks (e.g. mm_groups=None of
   * examples/deep_pilco_mm.py:31 on a sharded run: ONE Gaussian over the particles of all GPUs, utils/rollout.py:20-29).
   * mm_span_rows: rows of one group over all ranks (0: groups are local to this device, the fields above say it
   * all); mm_span_offset: position of this rank's B / mm_groups rows inside each group -- the cyclic noise row of
   * local row i of group g is then (t + g * mm_span_rows + mm_span_offset + i) mod B_global, row_offset is not used
   * by the moment matching.  Per step the ranks exchange the groups' sufficient statistics (fp64; forward: one
   * in-place sum of ranks x groups x (D^2 + 3 D + 1) values, adjoint: groups x (D^2 + D)) through the
   * collective attached with pmbrl_plan_set_comm / pmbrl_plan_set_collective; the sweeps then run as one launch
   * per step.  Not offered together with grad_states. */
  int32_t mm_span_rows, mm_span_offset;
  int32_t mm_span_ranks, mm_span_rank; /* ranks a group is spread over and this rank's index among them */
} pmbrl_config;

/* Environment switches read by pmbrl_plan_create (and by nothing else in the library).  They select between code paths
 * that compute the same results -- the tests use them to hold a non-default path against the default one, the profiles to
 * time alternatives -- and are not needed for normal use: what they override is chosen from the configuration.
 *   PMBRL_FORCE_F32=1       exact-fp32 MFMA sweeps whatever pmbrl_config.precision says
 *   PMBRL_MM_PARTS=n        moment-matching groups split over n workgroups where that can be done (1: whole groups)
 *   PMBRL_LDS_TILES=0       no LDS-resident weight tiles (pmbrl_fast.h, lds_tile_s): the shape-specialised 16-row
 *                           split-precision instances count on them, so this launches the generic instances
 *   PMBRL_MM_XCH=0          split groups exchange their rows and meet at a flag barrier (the generic kernel instances)
 *                           instead of exchanging fp64 sums as data-tagged granules (tests; A/B timing)
 *   PMBRL_MM_MODE2=1        groups that span workgroups: separate moment-matching kernels between per-step launches
 *                           instead of the in-sweep form (mm_mode 3)
 *   PMBRL_MM_PERSTEP=1      mm_mode 3 as one launch per step instead of one launch with a device-wide barrier per step
 *   PMBRL_MM_NO_SPAN1=1     a group beyond the CU count keeps the per-step prologue form instead of the one-rank span form
 *   PMBRL_MM_NO_WIDE=1      states wider than 6: the one-wave moment-matching routines instead of the LDS-staged
 *                           multi-wave kernels (pmbrl_mm_wide.h)
 *   PMBRL_DW_F32=1          dW GEMM on fp32 MFMA on a split-precision plan;  PMBRL_DW_SPLIT=1: the opposite
 *   PMBRL_DW_NO_WIDE=1      no LDS-staged tile kernel for layers >= PMBRL_DW_WIDE_MIN (default 128) wide
 *   PMBRL_DW_NO_LAYER13=1   no whole-layer-per-workgroup kernel for a <= 13 x 13-tile layer
 *   PMBRL_DW_PIPE=off | n0,n1,...   dW GEMM behind the adjoint sweep on a second stream: never / these step ranges
 * Read by the Python layer: PMBRL_PRECISION (default arithmetic: f32 | split | split_f16), PMBRL_LIB_PATH (the shared
 * library to load), PMBRL_TORCH_ALLREDUCE=1 (gradient all-reduce through torch.distributed instead of the C ABI). */

typedef struct pmbrl_plan pmbrl_plan;

/* plan->info indices for pmbrl_plan_info */
enum {
  PMBRL_INFO_ROWS_PER_WG = 0,
  PMBRL_INFO_N_WG = 1,
  PMBRL_INFO_ROW_TILES = 2,
  PMBRL_INFO_LDS_BYTES = 3,
  PMBRL_INFO_N_POL_PARAMS = 4,
  PMBRL_INFO_N_DYN_PARAMS = 5,
  PMBRL_INFO_DW_SPLITS = 6,
  PMBRL_INFO_PRECISION = 13, /* PMBRL_PREC_* actually in use */
  PMBRL_INFO_DW_PIPE = 14,   /* launches the adjoint sweep is cut into so that the dW GEMM runs behind it (1: no) */
  PMBRL_INFO_MM_PARTS = 15,  /* workgroups a moment-matching group is split over (in-kernel moment matching; 1: whole groups) */
  PMBRL_INFO_REG = 16,       /* 1: the plain whole-horizon sweeps of this plan run on the register-resident family (pmbrl_reg.h) */
  PMBRL_INFO_REPLAY = 17,    /* 1: repeated calls of this plan are replayed as hipGraphs (pmbrl_plan_set_replay) */
  PMBRL_INFO_INPLACE = 18,   /* general family: 0 two activation buffers, 1 in-place layers (64-row workgroups), 2 the same on the 512-wide layers of pmbrl_wide.h */
  PMBRL_INFO_REG_FWD_CALLS = 19, /* forward sweeps of this plan the register-resident family has served so far (a graph replay counts once, when recorded) */
  PMBRL_INFO_REG_BWD_CALLS = 20, /* ... adjoint sweeps */
  PMBRL_INFO_COUNT = 21
};

const char* pmbrl_last_error(void);
int pmbrl_version(void);
/* 16 hex digits: hash of the kernel sources the library was built from.  Hardware-counter measurements under
 * profiles/ record it; bench.py reports such a measurement only if it was taken on the build it is running. */
const char* pmbrl_build_id(void);

/* hipGraph replay of library calls (the H-step loop as one captured graph).  Queue any sequence of library
 * calls on `stream` (not the default stream) between begin and end -- a forward call, the loss, the adjoint call with its
 * optimiser step: a whole iteration -- and replay it with ONE launch.  Pays where a call is many launches (one per
 * step: moment-matching groups beyond a workgroup, states wider than 6, the pipelined adjoint: C5 with moment matching is
 * 400+ launches per iteration); the sweeps of the cart-pole shapes are one launch each already.  Every buffer the calls
 * were given must stay where it was; the status word and the optimiser's step counter are read and written on the
 * device, so every replay is a new iteration.  Not capturable: a host-side collective (pmbrl_plan_set_collective) and the
 * peer-to-peer all-reduce (pmbrl_p2p_allreduce_*: returns -4 on a recording stream). */
typedef struct pmbrl_graph pmbrl_graph;
int pmbrl_graph_capture_begin(void* stream);
int pmbrl_graph_capture_end(void* stream, pmbrl_graph** graph_out);
int pmbrl_graph_launch(pmbrl_graph* graph, void* stream);
int pmbrl_graph_num_nodes(pmbrl_graph* graph, int64_t* n_out);
void pmbrl_graph_destroy(pmbrl_graph* graph);

/* Validates the shape, chooses the tiling and sizes the workspace. */
int pmbrl_plan_create(const pmbrl_config* cfg, int device, pmbrl_plan** out);
void pmbrl_plan_destroy(pmbrl_plan* plan);
/* Bytes of scratch the caller must provide (one device allocation, 256-byte
 * aligned) to every fwd/bwd call on this plan: fragment-packed weights, the
 * activation / gradient stashes, bit masks, dW partial sums. */
size_t pmbrl_plan_workspace_bytes(const pmbrl_plan* plan);
int pmbrl_plan_info(const pmbrl_plan* plan, int32_t* info /* [PMBRL_INFO_COUNT] */);

/* {0,1} float mask [B, h] -> bit rows [B, ceil(h/16)] uint16 (little-endian bit
 * order).  Replaces the per-step fp32 mask multiply of models/modules.py:61,160
 * by a one-off packing at resample() time (models/modules.py:40-44,95-118). */
int pmbrl_pack_mask(void* stream, const float* mask_d, int32_t B, int32_t h,
                    int32_t src_ld, uint16_t* bits_d);

/* Dropout masks drawn ON THE DEVICE, as the bit rows the sweeps read (same layout as pmbrl_pack_mask): one launch
 * for all `rows` = H x B rows of a rollout that resamples its masks at every step (utils/rollout.py:95-98,110-115:
 * resample_policy / resample_model) or of a resample() between iterations (algorithms/mc_pilco.py:99-105, pegasus=False).
 *   kind 0  BDropout.update_noise (models/modules.py:40-44):   bit = u < keep[j]
 *   kind 1  CDropout in eval mode (models/modules.py:95-118):  probs = sigmoid((logit_p[j] + log((u + 1e-7) /
 *           (1 - (u - 1e-7)))) / temp),  bit = v < probs      (the hard sample of torch.bernoulli(probs))
 * u, v: 24-bit uniforms of Philox4x32-10 keyed by `seed`, counter (row, unit, stream, offset) -- a draw is a
 * function of (seed, offset, row, unit) only; advance `offset` between draws under one seed.  param_d: `param_len`
 * (1 or h) keep probabilities / logits.  u_d / v_d non-NULL: uniforms [rows][h] to use instead (replaying recorded
 * draws).  aux_*: for rows [aux_row0, aux_row0 + aux_rows) the float uniforms / hard samples / probabilities are
 * written too ([aux_rows][h] each, any may be NULL) -- a module keeps its last draw as state. */
int pmbrl_draw_masks(void* stream, int32_t kind, uint64_t seed, uint64_t offset, const float* param_d,
                     int32_t param_len, float temp, int32_t rows, int32_t h, const float* u_d, const float* v_d,
                     uint16_t* bits_d, int32_t aux_row0, int32_t aux_rows, float* u_out_d, float* hard_out_d,
                     float* probs_out_d);

/* Network + noise inputs shared by forward and backward. */
typedef struct pmbrl_inputs {
  const float* x0_d;          /* [B, D] */
  const float* pol_params_d;  /* flat, torch parameter order: W0[out,in], b0, W1, b1, ... */
  const float* dyn_params_d;  /* same for the dynamics MLP */
  const float* mx_d;          /* [dyn.dims[0]] (= D+U without angle dims)  models/core.py:141-145 */
  const float* iSx_d;         /* [D+U] */
  const float* my_d;          /* [D] */
  const float* Sy_d;          /* [D] */
  const float* pol_scale_d;   /* [U]  models/core.py:201-206 */
  const float* pol_bias_d;    /* [U] */
  const uint16_t* pol_mask_bits_d[PMBRL_MAX_LAYERS]; /* per hidden layer, from pmbrl_pack_mask */
  const uint16_t* dyn_mask_bits_d[PMBRL_MAX_LAYERS];
  const float* z_pol_d;       /* [B, U]  models/densities.py:78,111-119 */
  const float* z_dyn_d;       /* [B, D] */
  int64_t z_pol_step_stride;  /* elements between steps; 0 = the same frozen z at every step
                                 (resample_noise=False), B*U = a fresh [H,B,U] draw per step */
  int64_t z_dyn_step_stride;
  const float* z_mm_d;        /* [>= B_global, D] or NULL  algorithms/mc_pilco.py:57-62 */
  const float* z_rr_d;        /* [>= B_global, 1] or NULL */
  /* mixture head only (NULL otherwise) */
  const float* z_pi_d;        /* [B, n] frozen Gumbel noise (models/densities.py:166-167,213-216) */
  const float* u_cat_d;       /* [H, B] uniforms in [0, 1): the component of (t, b) is drawn by inverse CDF over the
                                 tempered softmax (the reference draws it from torch's generator at every step,
                                 models/densities.py:221-222) */
} pmbrl_inputs;

/* utils/rollout.py:62-163 (rollout) fused over all H steps, including
 * models/core.py:221-248 (Policy.forward), models/core.py:265-303
 * (DynamicsModel.forward), models/densities.py:87-121, the reward module and
 * utils/rollout.py:20-29 (mm_resample_).
 * Outputs: states [H+1,B,D], actions [H,B,U], rewards [H,B,1].
 * status_d: device int32[2].  [0] is set to 0x7fffffff on entry; on a numerical
 * failure at step t (non-finite state/reward, non-positive pivot) atomically
 * min'ed with t (= number of valid steps before the failure).  [1] is the adjoint
 * sweep's failure flag (pmbrl_rollout_bwd) and is CLEARED by this call: a caller
 * that only ever evaluates forward must still hand over two words. */
int pmbrl_rollout_fwd(pmbrl_plan* plan, void* stream, void* workspace_d,
                      const pmbrl_inputs* in, float* states_d, float* actions_d,
                      float* rewards_d, int32_t* status_d);

/* The adjoint of the above = what loss.backward() computes in
 * algorithms/mc_pilco.py:190-197 for loss = sum_{t,b} grad_rewards[t,b] *
 * rewards[t,b] (+ sum grad_states[t,b,:] . states[t,b,:] when grad_states_d
 * [H+1,B,D] is given -- terminal value bootstrap, algorithms/mc_pilco.py:136-140;
 * + sum grad_actions[t,b,:] . actions[t,b,:] when grad_actions_d [H,B,U] is given).
 * Must follow a pmbrl_rollout_fwd on the same plan/workspace/inputs.
 * Outputs: grad_pol_flat_d [n_pol_params] (overwritten), optional grad_x0_d
 * [B,D], optional action_grad_norms_d [H,B] (||dL/da_t|| per row, the
 * prioritised-replay hook of algorithms/mc_pilco.py:156-188).
 * status_d (optional, device int32[2]): status_d[0] is the word pmbrl_rollout_fwd
 * wrote; the adjoint is taken over the first n = min(H, status_d[0]) steps only --
 * the truncated horizon the reference continues with after a late failure
 * (utils/rollout.py:154-157): grad_rewards / grad_actions of steps >= n are ignored,
 * the terminal state gradient is grad_states[n], action_grad_norms rows >= n are
 * left untouched.  Read on the device: no host round trip between the sweeps.
 * status_d[1] is set non-zero if the sweep itself failed (a barrier or a statistics
 * exchange of a moment-matching group spanning workgroups timed out); it is CLEARED BY
 * pmbrl_rollout_fwd (the forward call's status array must be this one: int32[2]) --
 * a second adjoint call behind a failed one still reports the failure.
 * NULL: the full horizon. */
int pmbrl_rollout_bwd(pmbrl_plan* plan, void* stream, void* workspace_d,
                      const pmbrl_inputs* in, const float* states_d,
                      const float* actions_d, const float* rewards_d,
                      const float* grad_rewards_d, const float* grad_states_d,
                      const float* grad_actions_d, float* grad_pol_flat_d,
                      float* grad_x0_d, float* action_grad_norms_d, int32_t* status_d);

/* out[0] = sum_i a[i] * w[i]  (the discounted-return loss of
 * algorithms/mc_pilco.py:134-144,190 given w = dL/dr). Deterministic. */
int pmbrl_weighted_sum(void* stream, const float* a_d, const float* w_d,
                       int64_t n, float* out_d);
/* The same over the first min(n_steps, *status_d) blocks of n_per_step entries: the loss of
 * a truncated horizon (status_d = the word pmbrl_rollout_fwd wrote, read on the device). */
int pmbrl_weighted_sum_steps(void* stream, const float* a_d, const float* w_d,
                             int64_t n_per_step, int32_t n_steps, const int32_t* status_d,
                             float* out_d);

/* torch.nn.utils.clip_grad_norm_ (algorithms/mc_pilco.py:209-210) fused with
 * torch.optim.Adam.step (examples/deep_pilco_mm.py:166; no weight decay, no
 * amsgrad).  max_norm <= 0 disables clipping.  norm_out_d (optional) receives
 * the pre-clip global L2 norm.  grads are left scaled by the clip coefficient,
 * like the reference. */
int pmbrl_clip_adam(void* stream, float* params_d, float* grads_d,
                    float* exp_avg_d, float* exp_avg_sq_d, int64_t n,
                    int64_t step, double lr, double beta1, double beta2, double eps,
                    double max_norm, float* norm_out_d);

/* The same step, decided on the device: taken only if the rollout that produced
 * the gradient completed (*status_d >= expect, status_d = the word
 * pmbrl_rollout_fwd wrote; expect = H to require the whole horizon, or
 * min(H, 6) to continue on a truncated horizon of more than 5 steps like
 * utils/rollout.py:154-157), in which case the device-side step
 * counter *step_d is advanced first and its bias corrections are used;
 * otherwise parameters, moments and counter are left untouched.  This is the
 * reference's "RuntimeError -> skip the optimiser step" (algorithms/
 * mc_pilco.py:122-131) without a host round trip per iteration: the host may
 * read the status word one iteration late.  status_d is the two-word status of
 * pmbrl_rollout_fwd / _bwd: the step is also skipped when status_d[1] != 0 (the adjoint
 * sweep reported a failure of its own -- a barrier between workgroups timed out). */
int pmbrl_clip_adam_guarded(void* stream, float* params_d, float* grads_d,
                            float* exp_avg_d, float* exp_avg_sq_d, int64_t n,
                            int64_t* step_d, double lr, double beta1, double beta2,
                            double eps, double max_norm, float* norm_out_d,
                            const int32_t* status_d, int32_t expect);

/* ---- fused iteration tail (one process; with a gradient all-reduce in between use the separate calls) -------------
 * pmbrl_plan_set_loss: dL/dr weights [H][B] (device, resident; algorithms/mc_pilco.py:134-144: +-gamma_t / B) -- every
 * pmbrl_rollout_fwd on the plan then also leaves sum_{t < valid, b} w r in *loss_out_d (the reduction of
 * pmbrl_weighted_sum_steps, queued by the forward call itself); NULL switches it off.
 * pmbrl_rollout_bwd_adam: pmbrl_rollout_bwd followed by clip_grad_norm_ + Adam.step as pmbrl_clip_adam_guarded defines
 * them (taken on the device only if the rollout completed `expect_steps` steps and the adjoint reported no failure), in
 * one call: the whole tail of an iteration is queued without returning to the host language.  status_d: int32[2],
 * required. */
typedef struct pmbrl_adam {
  float* params_d;        /* flat policy parameters (the vector pmbrl_inputs::pol_params_d points at) */
  float* exp_avg_d;
  float* exp_avg_sq_d;
  int64_t* step_d;        /* device-side step counter, advanced when the step is taken */
  double lr, beta1, beta2, eps, max_norm;   /* max_norm <= 0: no clipping */
  float* norm_out_d;      /* optional: the gradient norm before clipping */
  int32_t expect_steps;   /* the step is taken if the rollout completed at least this many steps (0: H) */
  float* loss_out_d;      /* optional (round 6): the loss sum_{t < valid steps, b} grad_rewards[t][b] * rewards[t][b] of THIS
                           * call's arguments, written by the optimiser launch whether or not the step is taken -- the
                           * launch pmbrl_plan_set_loss queues behind the forward call is then not needed (one launch
                           * less per iteration); NULL: not wanted */
} pmbrl_adam;
int pmbrl_plan_set_loss(pmbrl_plan* plan, const float* loss_weights_d, float* loss_out_d);
int pmbrl_rollout_bwd_adam(pmbrl_plan* plan, void* stream, void* workspace_d, const pmbrl_inputs* in,
                           const float* states_d, const float* actions_d, const float* rewards_d,
                           const float* grad_rewards_d, const float* grad_states_d, const float* grad_actions_d,
                           float* grad_pol_flat_d, float* grad_x0_d, float* action_grad_norms_d, int32_t* status_d,
                           const pmbrl_adam* opt);

/* ---- gradient all-reduce over xGMI (RCCL) ---------------------------------- */
/* The one collective of the sharded path: the sum of the flat policy gradient over the
 * ranks, in place, issued on the caller's stream right behind pmbrl_rollout_bwd -- no host round
 * trip, no second stream.  Rows are sharded by whole moment-matching groups, so nothing else crosses
 * GPUs (algorithms/mc_pilco.py has no counterpart: the reference is single-process).
 *   pmbrl_comm_unique_id: rank 0 obtains the 128-byte RCCL id and hands it to the other ranks by
 *   whatever channel the launcher has (prob_mbrl_amd.distributed broadcasts it over
 *   torch.distributed's store); pmbrl_comm_init: every rank, once; collective.
 * librccl is loaded at run time (dlopen; the copy the process already has, else the system's), so
 * the library has no link-time dependency on it and single-GPU users never load it. */
#define PMBRL_COMM_ID_BYTES 128
typedef struct pmbrl_comm pmbrl_comm;
int pmbrl_comm_unique_id(void* id_out /* host, PMBRL_COMM_ID_BYTES */);
int pmbrl_comm_init(const void* id /* host, PMBRL_COMM_ID_BYTES */, int32_t rank, int32_t nranks,
                    int32_t device, pmbrl_comm** out);
int pmbrl_allreduce_sum(pmbrl_comm* comm, void* stream, float* buf_d, int64_t n);
/* the number of ranks the COMMUNICATOR reports (ncclCommCount) -- what bench.py prints as rccl_ranks, so that a
 * scaling record proves RCCL saw N ranks rather than echoing WORLD_SIZE */
int pmbrl_comm_count(pmbrl_comm* comm, int32_t* nranks_out);
void pmbrl_comm_destroy(pmbrl_comm* comm);

/* In-place fp64 sum over the ranks of buf_d[0..n) on `stream`: the per-step statistics exchange of
 * moment-matching groups that span ranks (pmbrl_config.mm_span_rows).  pmbrl_plan_set_comm: RCCL through the
 * communicator above (no host round trip; capturable).  pmbrl_plan_set_collective: any other transport --
 * fn(ctx, stream, buf_d, n) must return 0 once the sum is ORDERED on `stream` (it may block the host); e.g. a
 * host-staged torch.distributed / MPI all-reduce, which is how the two-process tests run on one device.
 * The plan keeps the pointer, not the communicator: destroy the plan first. */
typedef int (*pmbrl_collective_fn)(void* ctx, void* stream, double* buf_d, int64_t n);
int pmbrl_plan_set_comm(pmbrl_plan* plan, pmbrl_comm* comm);
int pmbrl_plan_set_collective(pmbrl_plan* plan, pmbrl_collective_fn fn, void* ctx);

/* ---- one-shot peer-to-peer all-reduce for latency-bound messages (pmbrl_p2p.hip) -------------------------------
 * The per-step statistics of moment-matching groups spread over ranks (a few KB, 2 H + 2 times per iteration) and the
 * flat policy gradient (163 KiB) are latency-, not bandwidth-bound: a ring pays 2 (N - 1) hops.  Here every rank writes
 * its contribution into a slot of every peer's buffer (stores through IPC-mapped pointers: xGMI between GPUs), raises
 * a flag per peer, waits for its own N flags and adds the N slots in rank order -- one hop, the same bits on every rank,
 * one kernel per rank on the caller's stream.  Setup: every rank creates its object (one uncached device allocation
 * sized for messages of up to max_bytes), exports a 64-byte handle (hipIpcGetMemHandle), the launcher carries the
 * handles to all ranks by any channel, every rank opens every peer's.  Ranks may be processes on different GPUs of
 * one node, or on ONE GPU (how the tests run it).  Waits are bounded: pmbrl_p2p_error reports a peer that never
 * arrived.  pmbrl_plan_set_p2p attaches it as the statistics exchange of a plan (like pmbrl_plan_set_comm).
 * NOT capturable: pmbrl_p2p_allreduce_* returns -4 on a stream that is recording (pmbrl_graph_capture_begin): the
 * generation its flags carry is a kernel argument, a replay would read stale slots. */
#define PMBRL_P2P_HANDLE_BYTES 64
#define PMBRL_P2P_MAX_RANKS 16
typedef struct pmbrl_p2p pmbrl_p2p;
int pmbrl_p2p_create(int32_t rank, int32_t nranks, int32_t device, int64_t max_bytes, pmbrl_p2p** out);
int pmbrl_p2p_handle(pmbrl_p2p* p, void* handle_out /* host, PMBRL_P2P_HANDLE_BYTES */);
int pmbrl_p2p_open(pmbrl_p2p* p, int32_t peer, const void* handle /* host, PMBRL_P2P_HANDLE_BYTES */);
int pmbrl_p2p_allreduce_f32(pmbrl_p2p* p, void* stream, float* buf_d, int64_t n);    /* in place */
int pmbrl_p2p_allreduce_f64(pmbrl_p2p* p, void* stream, double* buf_d, int64_t n);
int pmbrl_plan_set_p2p(pmbrl_plan* plan, pmbrl_p2p* p);
int pmbrl_p2p_error(pmbrl_p2p* p, int32_t* err_out);   /* host sync; 1: a wait timed out since the last call */
void pmbrl_p2p_destroy(pmbrl_p2p* p);

/* Optional per-kernel timing for bench.py's roofline line: when enabled, the
 * library brackets its kernels with hipEvents on the caller's stream;
 * pmbrl_plan_read_timing waits for them and returns the last call's durations
 * in milliseconds (-1 = not run). */
enum {
  PMBRL_TIMER_PACK = 0,      /* weight fragment packing */
  PMBRL_TIMER_FWD = 1,       /* pm_rollout_fwd (+ external mm kernels) */
  PMBRL_TIMER_BWD = 2,       /* pm_rollout_bwd (+ external mm kernels) */
  PMBRL_TIMER_DW = 3,        /* pm_dw_kernel */
  PMBRL_TIMER_DW_REDUCE = 4, /* pm_dw_reduce */
  PMBRL_TIMER_REWARD = 5,    /* pm_reward_all_kernel (+ reward moment matching) */
  PMBRL_TIMER_COUNT = 8
};
int pmbrl_plan_set_timing(pmbrl_plan* plan, int on);

/* Replay of repeated calls.  Some forms of the sweeps are ONE LAUNCH PER STEP (moment-matching groups
 * beyond what a workgroup or a set of exchanging workgroups holds, wide states): 100-400 launches per iteration, whose
 * launch cost is then most of the iteration.  With replay on, pmbrl_rollout_fwd / pmbrl_rollout_bwd(_adam) compare their
 * arguments (stream, workspace, every pointer and scalar of the call, the queued loss) with the previous call's: the
 * second identical call is recorded as a hipGraph (nothing but launches, memsets and event
 * hand-offs is queued by these calls) and launched, every further one is ONE hipGraphLaunch.  Any difference drops the
 * graph and runs the call as usual: an optimisation loop that keeps its buffers gets the replay, a caller that
 * allocates new outputs every time never pays for a capture.  Results are those of the eager calls bit for bit (the
 * same kernels in the same order).  The recording is made on a stream of the plan's own, so the caller's may be the
 * legacy default stream.  Not used while per-kernel timing or the cycle-stamp profile is on, inside a capture of the
 * caller's own, or with a host-side collective attached.
 * Default: on for the one-launch-per-step forms, off elsewhere (a plain iteration is a dozen launches the queue
 * already hides).  on = 0: never; 1: the default; 2: every form.  Environment PMBRL_REPLAY=0|1|2 sets the default. */
int pmbrl_plan_set_replay(pmbrl_plan* plan, int on);
/* calls that went out as one graph launch so far: n_out[0] forward calls, n_out[1] adjoint calls */
int pmbrl_plan_replay_count(const pmbrl_plan* plan, int64_t* n_out /* [2] */);
int pmbrl_plan_read_timing(pmbrl_plan* plan, float* ms /* [PMBRL_TIMER_COUNT] */);

/* Debug: workgroup 0 writes shader-clock stamps [H][32] at its phase boundaries
 * (forward into fwd_d, backward sweep into bwd_d; NULL = off). */
int pmbrl_plan_set_prof(pmbrl_plan* plan, long long* fwd_d, long long* bwd_d);

/* ---- stand-alone network evaluation ------------------------------------- */
/* One Bayesian MLP with a diagonal-Gaussian head evaluated on B independent rows: what the
 * reference's Policy.forward (models/core.py:221-248) and Regressor / DynamicsModel.forward
 * (models/core.py:169-187, 265-303) compute outside a rollout (apply_controller, model
 * evaluation):
 *   xin = (x - in_shift) * in_iscale                      (both NULL: xin = x)
 *   h   = relu(xin W0^T + b0) * mask0 / keep0 ; ...       (mask NULL: no dropout on that layer)
 *   (mu, l) = split(h W_L^T + b_L) ;  l <- -softplus(-l + max_log_std) + max_log_std
 *   out_scale/out_shift given:  mu <- mu*out_scale + out_shift ;  l <- l + log(out_scale)
 *   sample = mu + z * exp(l)                              (z NULL: sample = mu)
 *   sq_scale/sq_bias given:    sample <- sq_scale * tanh(sample) + sq_bias
 * Outputs may be NULL.  mask_bits are bit rows as produced by pmbrl_pack_mask. */
typedef struct {
  int32_t B;                  /* rows */
  pmbrl_mlp net;              /* dims[0] = input width, dims[n_layers] = 2 * output width */
  float max_log_std;
} pmbrl_mlp_call;

size_t pmbrl_mlp_workspace_bytes(const pmbrl_mlp_call* call);
int pmbrl_mlp_forward(void* stream, const pmbrl_mlp_call* call, void* workspace_d,
                      const float* x_d, const float* params_flat_d,
                      const uint16_t* const* mask_bits_d /* [n_layers-1], entries may be NULL */,
                      const float* z_d /* [B][n_out] or NULL */,
                      const float* in_shift_d, const float* in_iscale_d,
                      const float* out_scale_d, const float* out_shift_d,
                      const float* sq_scale_d, const float* sq_bias_d,
                      float* sample_d /* [B][n_out] */, float* mean_d, float* log_std_d);

/* Gradient of pmbrl_mlp_forward's outputs with respect to its input rows (network constant):
 * grad_x = (d sample / d x)^T g_sample + (d mean / d x)^T g_mean + (d log_std / d x)^T g_log_std,
 * same arguments as the forward; any of the three upstream gradients may be NULL.  Used when a
 * stand-alone network sits inside a differentiable computation (the terminal value V(x_H) of
 * algorithms/mc_pilco.py:136-140). */
int pmbrl_mlp_grad_input(void* stream, const pmbrl_mlp_call* call, void* workspace_d,
                         const float* x_d, const float* params_flat_d,
                         const uint16_t* const* mask_bits_d, const float* z_d,
                         const float* in_shift_d, const float* in_iscale_d,
                         const float* out_scale_d, const float* out_shift_d,
                         const float* sq_scale_d, const float* sq_bias_d,
                         const float* g_sample_d, const float* g_mean_d, const float* g_log_std_d,
                         float* grad_x_d /* [B][n_in] */);

/* ---- BNN maximum-likelihood training of the dynamics model ---------------- */
/* Loss and gradient of one minibatch: the iteration body of the reference's
 * utils.train_regressor (utils/train_regressor.py:113-131) with the model in train() mode --
 * Regressor.forward(x, normalize=False, resample=True) (models/core.py:169-187), concrete
 * dropout with a straight-through Bernoulli sample (models/modules.py:102-118,120-160), Gaussian
 * log-likelihood (losses.py:16-37) and the dropout regulariser (models/modules.py:30-35,88-93,
 * 234-274):   loss = -mean_rows lml + reg_weight * reg / N.
 * params / grad are flat in the module's parameter order: W0, b0, [logit_p0], W1, b1, [logit_p1],
 * ..., W_L, b_L  (logit_p_l, one per unit of hidden layer l, present iff temperature[l] > 0).
 * u / bvar: the uniform noise of the concrete relaxation and the uniform variate of the Bernoulli
 * draw (hard = bvar < probs), per dropout layer a block [M][h_l], blocks concatenated in layer
 * order.  The optimiser step is pmbrl_clip_adam on the same flat vectors. */
typedef struct {
  int32_t M;                  /* minibatch rows */
  int32_t N;                  /* dataset rows (the regulariser is divided by N) */
  pmbrl_mlp net;              /* dims[0] = input width, dims[n_layers] = 2 * output width; keep unused */
  float max_log_std;
  float temperature[PMBRL_MAX_LAYERS];   /* <= 0: hidden layer l has no dropout */
  float reg_scale[PMBRL_MAX_LAYERS];     /* CDropout.regularizer_scale buffer (= 0.5 * ctor argument) */
  float drop_reg[PMBRL_MAX_LAYERS];      /* CDropout.dropout_regularizer */
  float reg_weight;
  int32_t loss_kind;                     /* 0: Gaussian negative log-likelihood (losses.py:16-37);
                                          * 1: mean squared error of the mean head against the targets
                                          *    (the critic fit of examples/deep_pilco_no_mm_with_value.py:40-44;
                                          *    the log-std half of the head is ignored);
                                          * 2: negative log-likelihood of a mixture of diagonal Gaussians
                                          *    (losses.py:40-64 over models/densities.py:173-207,
                                          *    return_samples=False): net.dims[n_layers] =
                                          *    (2 n_out + 1) n_components + 1, see pmbrl_config.dyn_components */
  int32_t n_components;                  /* loss_kind 2 only */
} pmbrl_bnn_config;
typedef struct pmbrl_bnn_plan pmbrl_bnn_plan;

int pmbrl_bnn_plan_create(const pmbrl_bnn_config* cfg, int device, pmbrl_bnn_plan** out);
void pmbrl_bnn_plan_destroy(pmbrl_bnn_plan* plan);
size_t pmbrl_bnn_plan_workspace_bytes(const pmbrl_bnn_plan* plan);
int64_t pmbrl_bnn_plan_n_params(const pmbrl_bnn_plan* plan);
int pmbrl_bnn_loss_grad(pmbrl_bnn_plan* plan, void* stream, void* workspace_d,
                        const float* Xn_d /* [N][n_in] */, const float* Yn_d /* [N][n_out] */,
                        const int32_t* idx_d /* [M] */, const float* params_flat_d,
                        const float* u_d, const float* bvar_d,
                        float* grad_flat_d, float* loss_out_d /* [3]: loss, -E[lml], reg */);

/* The same step with the options of utils/train_regressor.py:113-147:
 *  - row_weight_d [M] (optional): importance-sampling weight of every minibatch
 *    row's log-likelihood (prioritized_sampling, :117-131);
 *  - row_logprob_d [M] (optional, out): the rows' unweighted log-likelihoods,
 *    from which the caller updates its priorities;
 *  - terms: 1 = likelihood only, 2 = regulariser only (its gradient alone, for
 *    the separate SGD step of decoupled_reg, :133-147), 3 = both. */
int pmbrl_bnn_loss_grad_ex(pmbrl_bnn_plan* plan, void* stream, void* workspace_d,
                           const float* Xn_d, const float* Yn_d, const int32_t* idx_d,
                           const float* params_flat_d, const float* u_d, const float* bvar_d,
                           float* grad_flat_d, float* loss_out_d,
                           const float* row_weight_d, float* row_logprob_d, int32_t terms);

/* Whole training iterations of utils/train_regressor.py:113-131 (loss_kind 0: Gaussian likelihood + regulariser, then
 * torch.optim.Adam's step -- the reference clips nothing here) queued by ONE call, two launches an iteration:
 * forward + backward over the minibatch, then dW / db / regulariser / Adam / re-packed weights / loss in one kernel.
 * Minibatch i = rows idx_all_d[i * M .. (i + 1) * M).  Dropout noise: recorded draws u_d / bvar_d
 * ([n_steps][M * sum_h], the layout of pmbrl_bnn_loss_grad) or, when both are null, drawn inside the kernel (Philox
 * keyed by `seed`, counter = (row, unit, layer, first_step + i)).  exp_avg / exp_avg_sq / step_d: Adam's state (the
 * device-side step counter is advanced by every iteration).  loss_out_d [3]: the last iteration's loss, -E[lml],
 * reg; loss_hist_d (optional) [n_steps][3]: every iteration's.  Replaces eight launches and two torch.rand calls per
 * iteration (train_regressor.py:58-165 runs 2 000 of them per policy-search round). */
int pmbrl_bnn_train_steps(pmbrl_bnn_plan* plan, void* stream, void* workspace_d, const float* Xn_d, const float* Yn_d,
                          const int32_t* idx_all_d, int32_t n_steps, float* params_flat_d, float* exp_avg_d,
                          float* exp_avg_sq_d, int64_t* step_d, double lr, double beta1, double beta2, double eps,
                          uint64_t seed, uint64_t first_step, const float* u_d, const float* bvar_d,
                          float* loss_out_d, float* loss_hist_d);

/* ---- test hooks (used by tests/ only) ---------------------------------- */
/* y[R,O] = x[R,K] W[O,K]^T + b through the same MFMA tile routine the rollout
 * kernels use (R <= 64). */
int pmbrl_debug_linear(void* stream, const float* x_d, const float* W_d,
                       const float* b_d, int32_t R, int32_t K, int32_t O,
                       int32_t transpose_w, float* y_d, float* scratch_d);

#ifdef __cplusplus
}
#endif
#endif /* PMBRL_H */
