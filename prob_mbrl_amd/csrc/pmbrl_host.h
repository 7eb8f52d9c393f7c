// Host-side declarations shared by the translation units of libpmbrl_hip.so.  The library is built from
// several .hip files so that the kernel families compile in parallel: pmbrl.hip (C ABI, plan, small kernels,
// general kernel family), pmbrl_fast_f32.hip (latency-optimised sweeps, exact fp32 MFMA) and
// pmbrl_fast_split.hip (the same on split bf16 / fp16 operands; compiled once per precision).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "pmbrl.h"
#include "pmbrl_dev.h"
#include "pmbrl_dw.h"

int pm_fail(int code, const std::string& msg);
#define HIPCHK(expr)                                                               \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return pm_fail(-100 - (int)_e, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct NetPlan {
  int nl;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  float keep[PM_MAXL];
  size_t w_off[PM_MAXL], b_off[PM_MAXL];   // offsets (floats) in the flat parameter vector
  size_t n_params;
  // workspace offsets (bytes)
  size_t wf[PM_MAXL], wb[PM_MAXL], bias[PM_MAXL], abits[PM_MAXL];
};

#define PM_PIPE_MAX 8
struct pmbrl_plan {
  pmbrl_config cfg;
  int device;
  int RT, rows_per_wg, nwg, LD, mm_mode, G, M, fast, CA, CB;   // CA/CB: k-blocks per weight-stream stage
  int prec, LDB;   // PMBRL_PREC_* in use; split precision: leading dimension of the bf16 piece planes
  size_t lds_bytes;
  NetPlan pol, dyn;
  RewardDev* rew_d;
  int rew_k = PMBRL_MAX_TIP;   // tip residuals of the reward (which instance of the reward launch: pm_reward_kernel_for)
  int* wflag_d;        // weight-range flag of the fp16 packer (see RolloutArgs::wflag)
  int wgen;            // generation of the last forward call
  int dw_split;        // dW GEMM on split bf16 operands (pm_dw_kernel_s)
  AngleDev* ang_d;
  DwBlock* dw_blocks_d;
  DwUnit* dw_units_d;   // 256 x 128 output tiles of the wide layers (pm_dw_wide_kernel); their blocks are not in dw_blocks_d
  int n_dw_units;
  int dw_narrow[PM_MAXL]; // 1: first-layer type (delta stash 512 wide), 2: head type (input stash 512 wide): pm_dw_narrow_pre_kernel
  unsigned dw_pre_mask;   // layers whose dW GEMM reads pre-split stashes (pm_dw_wide_pre_kernel; the wide sweeps write them)
  int dw_layer13;       // >= 0: this layer (at most 13 x 13 tiles) runs whole in one workgroup per row-step range (pm_dw_layer_kernel)
  int n_dw_blocks, dw_nsplit, dw_chunks_per_split, dw_n_chunks;
  // dW GEMM behind the adjoint sweep (pmbrl_rollout_bwd): the sweep as pipe_K launches over descending step
  // ranges [pipe_lo[k], pipe_lo[k-1]); the GEMM of range k runs on pipe_stream, on the CUs the sweep leaves
  // idle, while the sweep is in range k+1.  pipe_K <= 1: off.
  int pipe_K, pipe_lo[PM_PIPE_MAX], pipe_rows, pipe_grid;
  hipStream_t pipe_stream;
  hipEvent_t pipe_ev[PM_PIPE_MAX];
  int dw_wave_first[PM_DW_NW + 1];
  // workspace offsets (bytes)
  size_t off_actT[PM_MAXL], off_gT[PM_MAXL], off_Tp, off_Td, off_xt, off_rt, off_part, off_mmfac,
      off_gxc, off_gxc2, off_gsync, off_xch, xch_bytes, off_grt, off_Jx, off_Ja, off_gmm_c, off_gmm_k, ws_bytes;
  int wlds_off, lds_last_lanes;   // LDS-resident tiles (lds_tile_s): where they go (floats), lanes stored of the last K32 block
  const float* loss_w;   // pmbrl_plan_set_loss: dL/dr [H][B]; the forward call also leaves the loss in *loss_out
  float* loss_out;

  int mm_grid;   // mm_mode 3 as one launch per sweep with a device-wide barrier per step
  // moment-matching groups spread over ranks (pmbrl_config.mm_span_rows; mm_mode 2 with the statistics exchanged
  // through `coll` between the two halves of the external moment matching, pmbrl_mmx.h)
  int span;
  size_t off_mmx_buf, off_mmx_fac, off_mmx_rfac;
  int (*coll)(void* ctx, void* stream, double* buf_d, int64_t n);
  void* coll_ctx;
  int inplace;   // general family on split operands: 64-row workgroups with in-place layers (pm_rollout_fwd<4, 2, true>)
  int mm_wide;   // mm_mode 2 through the LDS-staged kernels for 6 < D <= 32 (pmbrl_mm_wide.h)
  int mm_parts;  // mm_mode 1 with every group split over this many workgroups (RolloutArgs::mm_parts); 1: whole groups
  int xch_zeroed;          // the exchange's granules hold no tag a launch of the register-resident family could mistake for its own
  mutable unsigned xch_gen;   // ... whose tags carry this launch generation (pmbrl_reg_mm.h)
  int mm_gpb;    // ... launched in batches of this many groups (0: all at once) -- more workgroups than CUs, and the
                 // statistics exchange needs a group's workgroups resident together
  int mm_fan;    // ... more than 8: parts per collector of the two-level sum exchange (0: one level)
  size_t off_ztab;   // ... and the noise standardisation of the whole group per step (pm_mm_ztable_kernel)
  int reg;       // the register-resident family (pmbrl_reg.h) serves this plan's plain whole-horizon launches
  mutable int reg_calls[2];   // sweeps it has served: [0] forward, [1] adjoint (pmbrl_plan_info)
  int reg_mm;    // ... including the sweeps that moment-match the states of split groups (pmbrl_reg_mm.h): state width, 0 = no
  size_t off_reg_linv;   // [H][groups][D D] doubles: L^-1, forward sweep -> adjoint sweep
  size_t off_reg_pack;   // its packed weights in the workspace
  size_t off_reg_ab[2][2];   // its activity words [net][hidden layer]: [step][workgroup][wave][lane] x 32 bits (pmbrl_reg.h)
  int abits_packed;      // the last forward call left the activity bits in that form only (pm_reg_unpack_abits: -> NetPlan::abits)
  int old_pack_stale;    // the last forward call packed the register-resident family's weights only
  // Replay (pmbrl_plan_set_replay): a call whose arguments equal the previous call's is recorded as a hipGraph the second
  // time it is seen and launched as one graph from then on.  Slot 0: pmbrl_rollout_fwd, slot 1: pmbrl_rollout_bwd(_adam).
  int replay;            // 0 off, 1 the one-launch-per-step forms, 2 every form
  hipStream_t cap_stream;   // what the calls are recorded on
  struct ReplaySlot { unsigned long long key; std::vector<unsigned char> bytes; int seen, dead, aux; hipGraphExec_t exec; long long launches; } rp[2];
  int reg_bwd;   // PMBRL_REG_BWD at plan creation (0: the family's adjoint sweep is off -- debugging aid)
  // optional per-kernel timing (hipEvents on the caller's stream)
  long long* prof_fwd;
  long long* prof_bwd;
  int timing;
  hipEvent_t ev[PMBRL_TIMER_COUNT][2];
  bool ev_set[PMBRL_TIMER_COUNT];
};

struct ScopedTimer {
  pmbrl_plan* p; int slot; hipStream_t s;
  ScopedTimer(pmbrl_plan* p_, int slot_, hipStream_t s_) : p(p_), slot(slot_), s(s_) {
    if (p->timing) (void)hipEventRecord(p->ev[slot][0], s);
  }
  ~ScopedTimer() {
    if (p->timing) { (void)hipEventRecord(p->ev[slot][1], s); p->ev_set[slot] = true; }
  }
};

// the instantiated (row tiles, stage pair) combinations -- keep in sync with stages_for()
#define PM_FAST_CASES                                                                  \
  PM_FAST_CASE(1, 8, 8) PM_FAST_CASE(1, 7, 6) PM_FAST_CASE(1, 4, 4) PM_FAST_CASE(1, 2, 2) \
  PM_FAST_CASE(2, 4, 4) PM_FAST_CASE(2, 4, 3) PM_FAST_CASE(2, 2, 2) PM_FAST_CASE(4, 2, 2) PM_FAST_CASE(4, 1, 1)
// shape-specialised instantiations (pmbrl_fast.h: PfShape): RT, CA, CB, variant, D, U, LD, layers,
// 16-wide tiles per hidden layer.
// The shipped configurations: cart-pole (D=4) and double cart-pole (D=6) states, one action,
// 2 x 200 hidden units; plain / per-step / in-kernel moment matching.
#define PM_FAST_SHAPED_CASES                                  \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_LEAN, 4, 1, 216, 3, 13)      \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_EXT, 4, 1, 216, 3, 13)       \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_LEAN, 5, 1, 216, 3, 13)      \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_LEAN, 6, 1, 216, 3, 13)      \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_EXT, 5, 1, 216, 3, 13)       \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_EXT, 6, 1, 216, 3, 13)       \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MM, 5, 1, 216, 3, 13)        \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MM, 6, 1, 216, 3, 13)        \
  PM_FAST_SHAPED(2, 4, 3, PF_VAR_MM, 4, 1, 232, 3, 13)        \
  PM_FAST_SHAPED(2, 4, 3, PF_VAR_MM, 6, 1, 232, 3, 13)        \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MM, 4, 1, 216, 3, 13)        \
  PM_FAST_SHAPED(4, 1, 1, PF_VAR_MM, 6, 1, 232, 3, 13)        \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MMG, 4, 1, 216, 3, 13)       \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MMG, 5, 1, 216, 3, 13)       \
  PM_FAST_SHAPED(1, 7, 6, PF_VAR_MMG, 6, 1, 216, 3, 13)

// split-bf16 instantiations (pmbrl_split.h): stage pairs in K32 blocks
// (64-row workgroups exist with two fp16 piece planes only: PM_SPLIT_PR == 2)
#define PM_SPLIT_CASES_12 PM_SPLIT_CASE(1, 4, 3) PM_SPLIT_CASE(1, 1, 1) PM_SPLIT_CASE(2, 4, 3) PM_SPLIT_CASE(2, 1, 1)
#if defined(PM_SPLIT_PR) && PM_SPLIT_PR == 2
#define PM_SPLIT_CASES PM_SPLIT_CASES_12 PM_SPLIT_CASE(4, 4, 3)
#define PM_SPLIT_SHAPED_RT4(LDV) PM_FAST_SHAPED(4, 4, 3, PF_VAR_MM, 6, 1, LDV, 3, 13)
#else
#define PM_SPLIT_CASES PM_SPLIT_CASES_12
#define PM_SPLIT_SHAPED_RT4(LDV)
#endif
// shape-specialised split instantiations; LDV: floats per row of an activation buffer = piece planes x 240 / 2
// (three bf16 planes: 360, two fp16 planes: 240)
#define PM_SPLIT_SHAPED_CASES(LDV)                             \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_LEAN, 4, 1, LDV, 3, 13)       \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_EXT, 4, 1, LDV, 3, 13)        \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_LEAN, 5, 1, LDV, 3, 13)       \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_LEAN, 6, 1, LDV, 3, 13)       \
  PM_FAST_SHAPED(2, 4, 3, PF_VAR_MM, 4, 1, LDV, 3, 13)         \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_MM, 4, 1, LDV, 3, 13)         \
  PM_FAST_SHAPED(2, 4, 3, PF_VAR_MM, 6, 1, LDV, 3, 13)         \
  PM_FAST_SHAPED(1, 4, 3, PF_VAR_MMG, 5, 1, LDV, 3, 13)         \
  PM_SPLIT_SHAPED_RT4(LDV)

// 16-wide tiles of the hidden layers if every hidden layer of both nets has the same width, else -1
static inline int hidden_tiles(const RolloutArgs& A) {
  const int nt = A.pol.nt[1];
  for (int l = 1; l < A.pol.nl; ++l) if (A.pol.nt[l] != nt) return -1;
  for (int l = 1; l < A.dyn.nl; ++l) if (A.dyn.nt[l] != nt) return -1;
  return nt;
}
static inline int fast_variant(int RT, const RolloutArgs& A) {
  // variant: see pmbrl_fast.h (PF_VAR_*)
  const bool mm = A.mm_mode == 1 || A.mm_mode == 3;   // moment matching inside the sweep launches
  const bool ext = A.prof || A.grad_states || A.grad_actions || A.agn || A.zpol_ss != 0 || A.zdyn_ss != 0 ||
                   (A.flags & PMBRL_FLAG_MM_STATES) || A.t0 != 0 || A.t1 != A.H || A.gx_from_carry;
  const bool mmg = RT == 1 && A.mm_mode == 3 && A.mm_grid;
  return mmg ? PF_VAR_MMG : mm ? PF_VAR_MM : (ext ? PF_VAR_EXT : PF_VAR_LEAN);
}

// register-resident family (pmbrl_reg.hip)
bool pm_reg_plan_ok(const pmbrl_plan* p);
bool pm_reg_shape_ok(const pmbrl_plan* p, int prec1);
bool pm_reg_mm_shape_ok(const pmbrl_plan* p, int prec1);
int pm_reg_mm_width(const pmbrl_plan* p);
size_t pm_reg_pack_bytes();
int pm_reg_set_attr(const pmbrl_plan* p);
bool pm_reg_can_run(const pmbrl_plan* p, const RolloutArgs& A, bool fwd);
struct ZtabArgs;   // (pmbrl_mm.h: the noise table of split moment-matching groups, formed by extra workgroups of the pack launch)
void pm_reg_pack_launch(const pmbrl_plan* p, char* ws, const float* pol_params, const float* dyn_params, int* wflag, int gen,
                        hipStream_t s, int* status_reset, const ZtabArgs* ztab);
void pm_reg_unpack_abits(const pmbrl_plan* p, char* ws, hipStream_t s);
void pm_reg_launch(const pmbrl_plan* p, char* ws, const RolloutArgs& A, const float* pol_params, const float* dyn_params,
                   hipStream_t s, bool fwd);
// per-family entry points (defined in pmbrl_fast_f32.hip / pmbrl_fast_split.hip)
// *_mmg_blocks_per_cu: workgroups of the barrier-form sweep (PF_VAR_MMG) the runtime says can be resident
// per CU with this plan's LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor, min over forward / adjoint;
// 0 = no such instantiation): the one-launch form is selected only if every workgroup can be resident
int pm_fast_f32_mmg_blocks_per_cu(const pmbrl_plan* p);
int pm_fast_split1_mmg_blocks_per_cu(const pmbrl_plan* p);
int pm_fast_split2_mmg_blocks_per_cu(const pmbrl_plan* p);
// general family on split operands (pmbrl_general_split.hip)
int pm_general_split_set_attr(const pmbrl_plan* p);
void pm_general_split_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd);
int pm_fast_f32_set_attr(const pmbrl_plan* p);
void pm_fast_f32_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd);
int pm_fast_split1_set_attr(const pmbrl_plan* p);     // PMBRL_PREC_SPLIT
void pm_fast_split1_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd);
int pm_fast_split2_set_attr(const pmbrl_plan* p);     // PMBRL_PREC_SPLIT_F16
void pm_fast_split2_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd);
