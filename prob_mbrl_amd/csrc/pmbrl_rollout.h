// Fused H-step rollout kernels: one workgroup owns `rows_per_wg` particle rows
// for ALL time steps (rows are independent except inside a moment-matching
// group, and a workgroup always owns whole groups), so the whole rollout is ONE
// launch with no grid-wide synchronisation.  The particle state never leaves
// LDS between steps; HBM sees only the trajectory outputs and the stashes the
// backward sweep / dW GEMM consume.
#pragma once
#include "pmbrl_dev.h"
#include "pmbrl_mm.h"
#include "pmbrl_gsplit.h"
#include "pmbrl_wide.h"

// ---------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------
// hidden layer, forward:  h = relu(acc + b) * mask / keep     (models/modules.py:46-61,120-160)
// NP = 0: fp32 rows out (leading dimension ld floats); NP = 2: two piece planes (pmbrl_gsplit.h; ld in 16-bit
// elements, R rows per plane; F16: fp16 pieces with range check into *ovf)
template <int NP = 0, bool F16 = false, int R = 0>
struct EpiHiddenFwdT {
  const float* bias;
  const uint16_t* mask;   // [B][nt]
  uint16_t* abits;        // [B][nt] slice of step t
  float keep;
  float* lds_out;
  float* stash;           // feature-major block [nt*16][Rw] or nullptr
  int ld, Rw, row0, nvalid, nt, lane;
  int* ovf;
  // the epilogue's operands from HBM / L2 (bias, dropout bits): fetched by the split-operand GEMMs when a tile
  // group STARTS, so that the round trip hides behind the group's MFMAs
  struct Pre {
    f32x4 b;
    unsigned mw;
  };
  __device__ __forceinline__ Pre pre(int ot, int rt) const {
    const int g = lane >> 4;
    const int lrow = rt * 16 + (lane & 15);
    Pre p;
    p.b = ldg4(bias + ot * 16 + 4 * g);
    p.mw = 0;
    if (lrow < nvalid) p.mw = mask[(size_t)(row0 + lrow) * nt + ot];
    return p;
  }
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) { (*this)(ot, rt, acc, pre(ot, rt)); }
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc, const Pre& pr) {
    const int g = lane >> 4;
    const int lrow = rt * 16 + (lane & 15);
    const int f0 = ot * 16 + 4 * g;
    const f32x4 b = pr.b;
    const bool valid = lrow < nvalid;
    const unsigned nib = (pr.mw >> (4 * g)) & 0xFu;
    f32x4 h;
    unsigned act = 0;
    // (fp32 form: the reference's division, x / p; split form: a multiply -- the product is not bit-exact anyway
    //  and four IEEE divisions are a third of this epilogue's instructions)
    const float ik = NP > 0 ? 1.f / keep : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[r] + b[r];
      const bool a = ((nib >> r) & 1u) && (v > 0.f);
      if constexpr (NP > 0) h[r] = a ? v * ik : 0.f;
      else h[r] = a ? (keep == 1.f ? v : v / keep) : 0.f;
      act |= (a ? 1u : 0u) << r;
    }
    if constexpr (NP > 0) {
      pm_store_planes<NP, R, F16>(lds_out, (unsigned)ld, (unsigned)lrow, (unsigned)f0, h);
      // the K padding up to the next K32 block (the buffer held fp32 rows / partial tiles before: any bits)
      if (ot == nt - 1 && (nt & 1))
        pm_store_planes<NP, R, F16>(lds_out, (unsigned)ld, (unsigned)lrow, (unsigned)f0 + 16u, f32x4{0.f, 0.f, 0.f, 0.f});
      if constexpr (F16) {
        // an activation out of fp16's range (h >= 0 here); a WEIGHT out of range -- it would pack to inf, inf x 0 is
        // NaN, and a NaN pre-activation vanishes in the ReLU -- is caught when the weights are packed (A.wflag)
        if (fmaxf(fmaxf(h[0], h[1]), fmaxf(h[2], h[3])) > 65504.f) *ovf = 1;
      }
    } else {
      *reinterpret_cast<f32x4*>(lds_out + lrow * ld + f0) = h;
    }
#ifndef PM_EXP_GS_NOSTASH
    if (stash) {
#pragma unroll
      for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(h[r], stash + (size_t)(f0 + r) * Rw + lrow);   // write-once stream: keep the weights in L2
    }
#endif
#ifndef PM_EXP_GS_NOABITS
    unsigned w16 = act << (4 * g);
    w16 |= __shfl_xor(w16, 16);
    w16 |= __shfl_xor(w16, 32);
    if (g == 0 && valid) abits[(size_t)(row0 + lrow) * nt + ot] = (uint16_t)w16;
#endif
  }
};

// hidden layer, backward: g_pre = active ? acc / keep : 0
template <int NP = 0, int R = 0>
struct EpiHiddenBwdT {
  const uint16_t* abits;  // [B][nt] slice of step t
  float keep;
  float* lds_out;
  float* stash;           // gT block or nullptr
  int ld, Rw, row0, nvalid, nt, lane;
  struct Pre {
    unsigned mw;
  };
  __device__ __forceinline__ Pre pre(int ot, int rt) const {
    const int lrow = rt * 16 + (lane & 15);
    Pre p;
    p.mw = 0;
    if (lrow < nvalid) p.mw = abits[(size_t)(row0 + lrow) * nt + ot];
    return p;
  }
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) { (*this)(ot, rt, acc, pre(ot, rt)); }
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc, const Pre& pr) {
    const int g = lane >> 4;
    const int lrow = rt * 16 + (lane & 15);
    const int f0 = ot * 16 + 4 * g;
    const unsigned nib = (pr.mw >> (4 * g)) & 0xFu;
    f32x4 h;
    const float ik = NP > 0 ? 1.f / keep : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (NP > 0) h[r] = ((nib >> r) & 1u) ? acc[r] * ik : 0.f;
      else h[r] = ((nib >> r) & 1u) ? (keep == 1.f ? acc[r] : acc[r] / keep) : 0.f;
    }
    if constexpr (NP > 0) {
      pm_store_planes<NP, R, false>(lds_out, (unsigned)ld, (unsigned)lrow, (unsigned)f0, h);
      if (ot == nt - 1 && (nt & 1))
        pm_store_planes<NP, R, false>(lds_out, (unsigned)ld, (unsigned)lrow, (unsigned)f0 + 16u, f32x4{0.f, 0.f, 0.f, 0.f});
    } else {
      *reinterpret_cast<f32x4*>(lds_out + lrow * ld + f0) = h;
    }
    if (stash) {
#pragma unroll
      for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(h[r], stash + (size_t)(f0 + r) * Rw + lrow);   // write-once stream: keep the weights in L2
    }
  }
};

// ---------------------------------------------------------------------------
// reward (envs/<env>/env.py:*Reward.forward); one thread per row
// ---------------------------------------------------------------------------
__device__ inline float reward_row(const RewardDev* __restrict__ rw, const float* x, int D,
                                   const float* a, int U, float* delta_out) {
  float phi[PMBRL_MAX_DIM];
  int De;
  if (rw->expand) {
    const int no = rw->n_other, na = rw->n_angle;
    for (int i = 0; i < no; ++i) phi[i] = x[rw->other_dims[i]];
    for (int j = 0; j < na; ++j) {
      const float th = x[rw->angle_dims[j]];
      phi[no + j] = sinf(th);
      phi[no + na + j] = cosf(th);
    }
    De = no + 2 * na;
  } else {
    for (int i = 0; i < D; ++i) phi[i] = x[i];
    De = D;
  }
  const int k = rw->k;
  float delta[PMBRL_MAX_TIP];
  for (int i = 0; i < k; ++i) {
    float s = 0.f;
    for (int j = 0; j < De; ++j) s = fmaf(phi[j], rw->C[i * De + j], s);
    delta[i] = s - rw->tt[i];
    if (delta_out) delta_out[i] = delta[i];
  }
  float cost = 0.f;
  for (int i = 0; i < k; ++i) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) s = fmaf(delta[j], rw->Q[j * k + i], s);
    cost = fmaf(s, delta[i], cost);
  }
  for (int i = 0; i < U; ++i) {
    float s = 0.f;
    for (int j = 0; j < U; ++j) s = fmaf(a[j], rw->R[j * U + i], s);
    cost = fmaf(s, a[i], cost);
  }
  cost *= rw->w;
  return rw->kind == PMBRL_REWARD_EXP ? expf(-cost) : -cost;
}

// adjoint of reward_row: adds d r / d x into gx[D], writes d r / d a into ga[U],
// both scaled by the upstream gr.
__device__ inline void reward_row_bwd(const RewardDev* __restrict__ rw, const float* x, int D,
                                      const float* a, int U, float r, float gr, float* gx,
                                      float* ga) {
  float delta[PMBRL_MAX_TIP];
  (void)reward_row(rw, x, D, a, U, delta);   // recompute delta (cheap)
  const float gc = (rw->kind == PMBRL_REWARD_EXP ? -gr * r : -gr) * rw->w;
  const int k = rw->k;
  const int De = rw->expand ? rw->n_other + 2 * rw->n_angle : D;
  float gdelta[PMBRL_MAX_TIP];
  for (int i = 0; i < k; ++i) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) s = fmaf(delta[j], rw->QQ[j * k + i], s);
    gdelta[i] = gc * s;
  }
  for (int i = 0; i < U; ++i) {
    float s = 0.f;
    for (int j = 0; j < U; ++j) s = fmaf(a[j], rw->RR[j * U + i], s);
    ga[i] = gc * s;
  }
  float gphi[PMBRL_MAX_DIM];
  for (int j = 0; j < De; ++j) {
    float s = 0.f;
    for (int i = 0; i < k; ++i) s = fmaf(gdelta[i], rw->C[i * De + j], s);
    gphi[j] = s;
  }
  if (rw->expand) {
    const int no = rw->n_other, na = rw->n_angle;
    for (int i = 0; i < no; ++i) gx[rw->other_dims[i]] += gphi[i];
    for (int j = 0; j < na; ++j) {
      const float th = x[rw->angle_dims[j]];
      gx[rw->angle_dims[j]] += gphi[no + j] * cosf(th) - gphi[no + na + j] * sinf(th);
    }
  } else {
    for (int i = 0; i < D; ++i) gx[i] += gphi[i];
  }
}

// ---------------------------------------------------------------------------
// LDS carve-up (floats).  Must match pmbrl_lds_floats() on the host.
// ---------------------------------------------------------------------------
// moment-matching noise addressing: cyclic PEGASUS buffer or fresh rows per step
__device__ __forceinline__ const float* pm_zbase(const float* z, int ld, int t, int Bg, int flags) {
  return (flags & PMBRL_FLAG_ZMM_PER_STEP) ? z + (size_t)t * Bg * ld : z;
}
__device__ __forceinline__ int pm_zrow0(int t, int row, int flags) {
  return (flags & PMBRL_FLAG_ZMM_PER_STEP) ? row : t + row;
}

struct LdsMap {
  float *bufA, *bufB, *xa, *xb, *av, *gad, *rr, *gr, *part, *tbl;
  double* mm;
};
// ip: in-place layers (gemm_layer_inplace_s) -- ONE activation buffer, narrow GEMM outputs at column PM_IP_NOFF of it
#define PM_IP_NOFF 64
__host__ __device__ inline size_t pm_lds_floats(int R, int LD, int D, int U, int RT, int mm_d, bool ip = false) {
  size_t n = (ip ? 1 : 2) * (size_t)R * LD;          // bufA, bufB
  n += 2 * (size_t)R * D;                 // xa, xb
  n += (size_t)R * U;                     // av
  n += (size_t)R * 16;                    // gad (action gradient, U <= 16)
  n += 2 * (size_t)R;                     // rr, gr
  if (!ip && !pm_part_alias_ok(R, LD, RT)) n += (size_t)PM_NW * PM_KS_NT * RT * 256;  // K-split partials (else: in the output buffer)
  if (ip) n += 64;                        // nibble -> multipliers (pmbrl_wide.h)
  n = (n + 3) & ~(size_t)3;
  n += 2 * (size_t)PM_NW * pm_mm_scratch_doubles(mm_d);  // per-wave fp64 scratch
  return n;
}
__device__ inline LdsMap pm_lds_carve(float* base, int R, int LD, int D, int U, int RT, bool ip = false) {
  LdsMap m;
  m.bufA = base;
  m.bufB = ip ? m.bufA : m.bufA + (size_t)R * LD;
  m.xa = m.bufB + (size_t)R * LD;
  m.xb = m.xa + (size_t)R * D;
  m.av = m.xb + (size_t)R * D;
  m.gad = m.av + (size_t)R * U;
  m.rr = m.gad + (size_t)R * 16;
  m.gr = m.rr + R;
  m.part = m.gr + R;
  size_t n = (size_t)(m.part - base);
  m.tbl = m.part;
  if (ip) n += 64;
  if (ip || pm_part_alias_ok(R, LD, RT)) m.part = nullptr;
  else n += (size_t)PM_NW * PM_KS_NT * RT * 256;
  n = (n + 3) & ~(size_t)3;
  m.mm = reinterpret_cast<double*>(base + n);
  return m;
}

// ===========================================================================
// forward
// ===========================================================================
// PR = 0: exact fp32 MFMA; PR = 2: split operands (pmbrl_gsplit.h) -- two fp16 pieces
// IP: in-place layers (PR = 2 only; widths <= 512, narrow widths <= PM_IP_NOFF, no mixture head); IP = 2: every hidden
// layer is 512 wide and runs on pmbrl_wide.h
template <int RT, int PR = 0, int IP = 0>
__global__ __launch_bounds__(PM_NT, 1) void pm_rollout_fwd(const RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16 * RT;
  constexpr bool SP = PR != 0;
  const unsigned LDB = (unsigned)A.LD;   // split: elements per row of a piece plane (= floats per row)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int row0 = wg * A.rows_per_wg;
  const int nvalid = min(A.rows_per_wg, A.B - row0);
  const int D = A.D, U = A.U, LD = A.LD, B = A.B;
  LdsMap L = pm_lds_carve(smem, R, LD, D, U, RT, IP != 0);
  float* xa = L.xa;   // current state x_t
  float* xb = L.xb;   // pre-moment-matching next state
  // PR = 2: set when an activation leaves fp16's range (the action-gradient rows are idle in the forward sweep)
  int* const p_ovf = reinterpret_cast<int*>(L.gad);
  if (SP && tid == 0) *p_ovf = (A.wflag && *A.wflag == A.wgen) ? 1 : 0;   // (1: a weight did not fit fp16, pm_pack_all)
  if constexpr (IP >= 2) pw_table_init(L.tbl, tid);

  // initial state (states[t0] is x0 for t0 == 0, the previous launch's output otherwise)
  {
    const float* src = (A.t0 == 0) ? A.x0 : A.states + (size_t)A.t0 * B * D;
    for (int i = tid; i < R * D; i += PM_NT) {
      const int r = i / D, d = i - r * D;
      const float v = (r < nvalid) ? src[(size_t)(row0 + r) * D + d] : 0.f;
      xa[i] = v;
      if (A.t0 == 0 && r < nvalid) A.states[(size_t)(row0 + r) * D + d] = v;
    }
  }
  __syncthreads();

  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  // angle_dims of the policy / the dynamics model (models/core.py:233-234,173-174)
  const FeatMap* pmap = A.ang ? &A.ang->pol : nullptr;
  const FeatMap* dmap = A.ang ? &A.ang->dyn : nullptr;

  for (int t = A.t0; t < A.t1; ++t) {
    // (IP = 2: nothing derived from the thread index is carried across steps -- hoisted out of the step loop, the
    //  address arithmetic of the elementwise phases alone kept some sixty registers live through the layers, and what
    //  the layers then spilled came back through scratch loads behind vmcnt(0): a memory round trip each)
    int tid_s = tid;
    if constexpr (IP >= 2) asm volatile("" : "+v"(tid_s));
    const int tid = tid_s, lane = tid & 63;
    const size_t blk = (size_t)t * A.nwg + wg;
    float* X = L.bufA;
    float* Y = L.bufB;
    PM_MARK(0);
    // ---- policy input (no normalisation in Policy.forward, models/core.py:221-248)
    {
      const int K16 = P.nt[0] * 16;
      const int KW = SP ? pm_kb32(P.nt[0]) * 32 : K16;   // split: whole K32 blocks of the piece planes
      float* st = A.actT[0] + blk * (size_t)K16 * A.Rw;
      for (int i = tid; i < R * KW; i += PM_NT) {
        const int k = i / R, r = i - k * R;
        float v = 0.f;
        if (pmap) {
          if (k < pmap->n_feat) v = pm_feat(pmap, k, xa + r * D);
        } else if (k < D) {
          v = xa[r * D + k];
        }
        if constexpr (SP) {
          // an INPUT out of fp16's range would become inf, inf x 0-weight NaN, and vanish in the ReLU: reported like an
          // activation out of range (the host re-runs in fp32)
          if (!(fabsf(v) <= 65504.f)) *p_ovf = 1;
          pm_put_planes<R, true>(X, LDB, r, IP >= 2 ? pw_sw(r, k) : k, v);
        }
        else X[r * LD + k] = v;
        if (k < K16) st[(size_t)k * A.Rw + r] = v;
      }
    }
    __syncthreads();
    PM_MARK(1);
    // ---- policy hidden layers
    for (int l = 0; l < P.nl - 1; ++l) {
      const int nt = P.nt[l + 1];
      const uint16_t* mk = P.mask[l] + ((A.flags & PMBRL_FLAG_POL_MASKS_PER_STEP) ? (size_t)t * B * nt : 0);
      EpiHiddenFwdT<SP ? 2 : 0, SP, R> e{P.bias[l], mk, P.abits[l] + (size_t)t * B * nt, P.keep[l], IP ? X : Y,
                                        A.actT[l + 1] + blk * (size_t)nt * 16 * A.Rw, LD, A.Rw, row0, nvalid, nt, lane,
                                        p_ovf};
      if constexpr (IP >= 2) {
        const PwFwd w{P.bias[l], mk, P.abits[l] + (size_t)t * B * nt, P.keep[l], X,
                      A.actT[l + 1] + blk * (size_t)nt * 16 * A.Rw, L.tbl, row0, nvalid, p_ovf,
                      (A.prof && wg == 0 && l == 1) ? A.prof + (size_t)t * 32 + 5 : nullptr,
                      (int)((A.stash_pre >> (l + 1)) & 1u)};      // (pre-split where the next layer's dW GEMM is pm_dw_wide_pre_kernel's)
        // (IP = 3: every 512-wide stash pre-split -- ONE form of the stash stores per kernel instance: two forms in one
        //  kernel, as two instances of the layer or as a branch around the stores, cost the forward sweep 0.5-1 ms)
        pw_hidden_fwd<IP == 3 ? 2 : 1>(P.wf[l], pm_kb32(P.nt[l]), w, wid, lane);
      } else if constexpr (IP) gemm_layer_inplace_s<RT, true>(P.wf[l], nt, pm_kb32(P.nt[l]), X, LDB, wid, lane, e);
      else if constexpr (SP) gemm_tiles_s<RT, true>(P.wf[l], nt, pm_kb32(P.nt[l]), X, LDB, wid, lane, e);
      else gemm_tiles<RT>(P.wf[l], nt, P.nt[l], X, LD, wid, lane, e);
      if constexpr (IP >= 2) pw_lds_barrier();
      else __syncthreads();
      PM_MARK(2 + l);
      if constexpr (!IP) { float* tmp = X; X = Y; Y = tmp; }
    }
    // ---- policy head -> Y[r][0..2U)
    if constexpr (IP) {
      // fp32 rows at column PM_IP_NOFF of the one buffer: clear of the 64 plane columns the next phase writes
      Y = X + PM_IP_NOFF;
      EpiPlain e{P.bias[P.nl - 1], Y, LD, lane};
      if constexpr (IP >= 2) pw_narrow<true>(P.wf[P.nl - 1], P.nt[P.nl], X, P.bias[P.nl - 1], Y, LD, wid, lane);
      else gemm_layer_inplace_s<RT, true>(P.wf[P.nl - 1], P.nt[P.nl], pm_kb32(P.nt[P.nl - 1]), X, LDB, wid, lane, e);
      __syncthreads();
    } else if constexpr (SP)
      gemm_narrow_s<RT, true>(P.wf[P.nl - 1], P.nt[P.nl], pm_kb32(P.nt[P.nl - 1]), P.bias[P.nl - 1], X, LDB, Y, LD,
                              L.part, wid, lane, tid);
    else
      gemm_narrow<RT>(P.wf[P.nl - 1], P.nt[P.nl], P.nt[P.nl - 1], P.bias[P.nl - 1], X, Y, LD, L.part,
                      wid, lane, tid);
    PM_MARK(10);
    // ---- squash + dynamics input (models/densities.py:87-121, models/core.py:243,169-177)
    if constexpr (IP >= 2) {
      // (the loop below walks (row, column) pairs: eight iterations at 64 x 64, each with its own dependent loads of
      //  the normalisation constants and, in the action columns, of the noise -- 29 k cycles of a step.  Here every
      //  action element is ONE thread with all its loads up front, and a thread of the copy part keeps its column)
      const int K16 = pm_kb32(F.nt[0]) * 32;
      for (int i = tid; i < R * U; i += PM_NT) {
        const int r = i / U, j = i - r * U, k = D + j;
        const bool valid = r < nvalid;
        const float z0 = A.zpol[(size_t)t * A.zpol_ss + (size_t)(row0 + (valid ? r : 0)) * U + j];
        const float z = valid ? z0 : 0.f;
        const float sc = A.pscale[j], pb = A.pbias[j], mxk = A.mx[k], isk = A.iSx[k];
        const float mu = Y[r * LD + j];
        const float ls = Y[r * LD + U + j];
        const float lc = -softplusf(-ls + A.mls_pol) + A.mls_pol;
        const float e = expf(lc);
        const float u = mu + z * e;
        const float a = sc * tanhf(u) + pb;
        L.av[r * U + j] = a;
        if (valid) {
          const size_t o = ((size_t)t * B + row0 + r) * U + j;
          A.actions[o] = a;
          A.Tp[o] = z * e * sigmoidf(-ls + A.mls_pol);
        }
        const float v = (a - mxk) * isk;
        if (!(fabsf(v) <= 65504.f)) *p_ovf = 1;
        pm_put_planes<R, true>(X, LDB, r, pw_sw(r, k), v);
      }
      {
        const int k = tid % K16, rstep = PM_NT / K16;
        const bool state = k < D, pad = k >= D + U;
        const float mxk = state ? A.mx[k] : 0.f, isk = state ? A.iSx[k] : 0.f;
        if (state || pad) {
          for (int r = tid / K16; r < R; r += rstep) {
            const float v = state ? (xa[r * D + k] - mxk) * isk : 0.f;
            if (!(fabsf(v) <= 65504.f)) *p_ovf = 1;
            pm_put_planes<R, true>(X, LDB, r, pw_sw(r, k), v);
          }
        }
      }
    } else {
      const int K16 = SP ? pm_kb32(F.nt[0]) * 32 : F.nt[0] * 16;
      for (int i = tid; i < R * K16; i += PM_NT) {
        const int r = i / K16, k = i - r * K16;
        float v = 0.f;
        int src = k;
        bool live = k < D + U;
        if (dmap) {
          live = k < dmap->n_feat;
          src = live ? dmap->src[k] : 0;
        }
        if (live && src < D) {
          const float s = dmap ? pm_feat(dmap, k, xa + r * D) : xa[r * D + k];
          v = (s - A.mx[k]) * A.iSx[k];
        } else if (live) {
          const int j = src - D;
          const float mu = Y[r * LD + j];
          const float ls = Y[r * LD + U + j];
          const float z = (r < nvalid) ? A.zpol[(size_t)t * A.zpol_ss + (size_t)(row0 + r) * U + j] : 0.f;
          const float lc = -softplusf(-ls + A.mls_pol) + A.mls_pol;
          const float e = expf(lc);
          const float u = mu + z * e;
          const float a = A.pscale[j] * tanhf(u) + A.pbias[j];
          L.av[r * U + j] = a;
          if (r < nvalid) {
            const size_t o = ((size_t)t * B + row0 + r) * U + j;
            A.actions[o] = a;
            A.Tp[o] = z * e * sigmoidf(-ls + A.mls_pol);
          }
          v = (a - A.mx[k]) * A.iSx[k];
        }
        if constexpr (SP) {
          // an INPUT out of fp16's range would become inf, inf x 0-weight NaN, and vanish in the ReLU: reported like an
          // activation out of range (the host re-runs in fp32)
          if (!(fabsf(v) <= 65504.f)) *p_ovf = 1;
          pm_put_planes<R, true>(X, LDB, r, k, v);
        }
        else X[r * LD + k] = v;
      }
    }
    __syncthreads();
    PM_MARK(11);
    // ---- dynamics hidden layers
    for (int l = 0; l < F.nl - 1; ++l) {
      const int nt = F.nt[l + 1];
      const uint16_t* mk = F.mask[l] + ((A.flags & PMBRL_FLAG_DYN_MASKS_PER_STEP) ? (size_t)t * B * nt : 0);
      EpiHiddenFwdT<SP ? 2 : 0, SP, R> e{F.bias[l], mk, F.abits[l] + (size_t)t * B * nt, F.keep[l], IP ? X : Y,
                                        nullptr, LD, A.Rw, row0, nvalid, nt, lane, p_ovf};
      if constexpr (IP >= 2) {
        const PwFwd w{F.bias[l], mk, F.abits[l] + (size_t)t * B * nt, F.keep[l], X, nullptr, L.tbl, row0, nvalid, p_ovf,
                      (A.prof && wg == 0 && l == 1) ? A.prof + (size_t)t * 32 + 15 : nullptr, 0};
        pw_hidden_fwd<0>(F.wf[l], pm_kb32(F.nt[l]), w, wid, lane);
      } else if constexpr (IP) gemm_layer_inplace_s<RT, true>(F.wf[l], nt, pm_kb32(F.nt[l]), X, LDB, wid, lane, e);
      else if constexpr (SP) gemm_tiles_s<RT, true>(F.wf[l], nt, pm_kb32(F.nt[l]), X, LDB, wid, lane, e);
      else gemm_tiles<RT>(F.wf[l], nt, F.nt[l], X, LD, wid, lane, e);
      if constexpr (IP >= 2) pw_lds_barrier();
      else __syncthreads();
      PM_MARK(12 + l);
      if constexpr (!IP) { float* tmp = X; X = Y; Y = tmp; }
    }
    // ---- dynamics head -> Y[r][0..2D)
    if constexpr (IP) {
      Y = X + PM_IP_NOFF;
      EpiPlain e{F.bias[F.nl - 1], Y, LD, lane};
      if constexpr (IP >= 2) pw_narrow<true>(F.wf[F.nl - 1], F.nt[F.nl], X, F.bias[F.nl - 1], Y, LD, wid, lane);
      else gemm_layer_inplace_s<RT, true>(F.wf[F.nl - 1], F.nt[F.nl], pm_kb32(F.nt[F.nl - 1]), X, LDB, wid, lane, e);
      __syncthreads();
    } else if constexpr (SP)
      gemm_narrow_s<RT, true>(F.wf[F.nl - 1], F.nt[F.nl], pm_kb32(F.nt[F.nl - 1]), F.bias[F.nl - 1], X, LDB, Y, LD,
                              L.part, wid, lane, tid);
    else
      gemm_narrow<RT>(F.wf[F.nl - 1], F.nt[F.nl], F.nt[F.nl - 1], F.bias[F.nl - 1], X, Y, LD, L.part,
                      wid, lane, tid);
    PM_MARK(20);
    // ---- sample next state (models/densities.py:97-121 with scaling_params, core.py:298)
    if (A.gmm_n > 1) {
      // GaussianMixtureDensity head (models/densities.py:173-233): head row = [n D means | n D log-stds | n logits |
      // log-temperature]; straight-through one-hot component over the tempered Gumbel-softmax.  Every (row, d)
      // thread redoes the row's softmax (n <= 8) and leaves what the adjoint needs: Td (d/d log-std of the drawn
      // component) and the coefficients of dL/d(logits, log-temperature) -- both linear in dL/dx~_d.
      const int n = A.gmm_n, nD = n * D;
      for (int i = tid; i < R * D; i += PM_NT) {
        const int r = i / D, d = i - r * D;
        const float* o = Y + r * LD;
        const bool valid = r < nvalid;
        const size_t row = (size_t)t * B + row0 + r;
        const float lt = o[2 * nD + n];
        const float temp = 0.1f + softplusf(lt);
        float lg[PMBRL_MAX_COMP], ks[PMBRL_MAX_COMP], sm[PMBRL_MAX_COMP];
        float mxl = -3.0e38f;
        for (int c = 0; c < n; ++c) { lg[c] = o[2 * nD + c] / temp; mxl = fmaxf(mxl, lg[c]); }
        float se = 0.f;
        for (int c = 0; c < n; ++c) se += expf(lg[c] - mxl);
        const float lse = mxl + logf(se);
        float mxy = -3.0e38f;
        for (int c = 0; c < n; ++c) {
          sm[c] = expf(lg[c] - lse);
          ks[c] = ((lg[c] - lse) + (valid ? A.zpi[(size_t)(row0 + r) * n + c] : 0.f)) / 0.1f;
          mxy = fmaxf(mxy, ks[c]);
        }
        float sy = 0.f;
        for (int c = 0; c < n; ++c) { ks[c] = expf(ks[c] - mxy); sy += ks[c]; }
        for (int c = 0; c < n; ++c) ks[c] /= sy;
        // component: inverse CDF of the row's uniform
        const float u = valid ? A.ucat[row] : 0.f;
        int kc = 0;
        float cum = ks[0];
        while (kc < n - 1 && u >= cum) { ++kc; cum += ks[kc]; }
        const float Sy = A.Sy[d], lSy = logf(Sy), myd = A.my[d];
        const float z = valid ? A.zdyn[(size_t)t * A.zdyn_ss + (size_t)(row0 + r) * D + d] : 0.f;
        const float zg = (valid && A.zdyn_grad) ? A.zdyn_grad[(size_t)(row0 + r) * D + d] : z;
        const float lsr_k = o[nD + d * n + kc];
        const float E = expf(-softplusf(-lsr_k + A.mls_dyn) + A.mls_dyn + lSy);
        const float xn = xa[i] + ((o[d * n + kc] * Sy + myd) + z * E);
        xb[i] = xn;
        if (valid) {
          const size_t oo = row * D + d;
          A.Td[oo] = zg * E * sigmoidf(-lsr_k + A.mls_dyn);
          if (A.flags & PMBRL_FLAG_MM_STATES) A.xt[oo] = xn;
          else A.states[oo + (size_t)B * D] = xn;
          // gk_c = g_d A_c,  A_c = mean_c + zg E lsc_c;  k_soft = softmax(y / 0.1);  y = log_softmax(logit) + z_pi
          float Ac[PMBRL_MAX_COMP], abar = 0.f;
          for (int c = 0; c < n; ++c) {
            const float lsc = -softplusf(-o[nD + d * n + c] + A.mls_dyn) + A.mls_dyn + lSy;
            Ac[c] = (o[d * n + c] * Sy + myd) + zg * E * lsc;
            abar += ks[c] * Ac[c];
          }
          float scy = 0.f;
          for (int c = 0; c < n; ++c) { Ac[c] = 10.f * ks[c] * (Ac[c] - abar); scy += Ac[c]; }
          float ct = 0.f;
          float* cp = A.gmm_c + row * (size_t)(n + 1) * D + d;
          for (int c = 0; c < n; ++c) {
            const float cl = Ac[c] - sm[c] * scy;          // d/d logit_c
            cp[(size_t)c * D] = cl / temp;                 // d/d logit_pi_c (raw head output)
            ct -= cl * o[2 * nD + c];
          }
          cp[(size_t)n * D] = ct * sigmoidf(lt) / (temp * temp);   // d/d log-temperature
          if (d == 0) A.gmm_k[row] = kc;
        }
      }
    } else if (IP >= 2 && PM_NT % D == 0) {
      // (a thread keeps its state dimension: the constants once, the rows' noise requested together)
      const int d = tid % D, rstep = PM_NT / D;
      const float Sy = A.Sy[d], lSy = logf(Sy), myd = A.my[d];
#pragma unroll 4
      for (int r = tid / D; r < R; r += rstep) {
        const int i = r * D + d;
        const float mu = Y[r * LD + d];
        const float ls = Y[r * LD + D + d];
        const float z0 = A.zdyn[(size_t)t * A.zdyn_ss + (size_t)(row0 + (r < nvalid ? r : 0)) * D + d];
        const float z = (r < nvalid) ? z0 : 0.f;
        const float lc = -softplusf(-ls + A.mls_dyn) + A.mls_dyn + lSy;
        const float e = expf(lc);
        const float xn = xa[i] + (mu * Sy + myd + z * e);
        xb[i] = xn;
        if (r < nvalid) {
          const size_t o = ((size_t)t * B + row0 + r) * D + d;
          A.Td[o] = z * e * sigmoidf(-ls + A.mls_dyn);
          if (A.flags & PMBRL_FLAG_MM_STATES) A.xt[o] = xn;
          else A.states[o + (size_t)B * D] = xn;
        }
      }
    } else
    for (int i = tid; i < R * D; i += PM_NT) {
      const int r = i / D, d = i - r * D;
      const float mu = Y[r * LD + d];
      const float ls = Y[r * LD + D + d];
      const float z = (r < nvalid) ? A.zdyn[(size_t)t * A.zdyn_ss + (size_t)(row0 + r) * D + d] : 0.f;
      const float Sy = A.Sy[d];
      const float lc = -softplusf(-ls + A.mls_dyn) + A.mls_dyn + logf(Sy);
      const float e = expf(lc);
      const float xn = xa[i] + (mu * Sy + A.my[d] + z * e);
      xb[i] = xn;
      if (r < nvalid) {
        const size_t o = ((size_t)t * B + row0 + r) * D + d;
        A.Td[o] = z * e * sigmoidf(-ls + A.mls_dyn);
        if (A.flags & PMBRL_FLAG_MM_STATES) A.xt[o] = xn;
        else A.states[o + (size_t)B * D] = xn;
      }
    }
    __syncthreads();
    PM_MARK(21);
    // ---- reward on the sampled (pre-mm) next state; failure detection
    if (A.ext_reward) {
      // (the reward never feeds the state recursion: pm_reward_all_kernel evaluates it for all row-steps at once
      //  after the sweep -- one thread per row here was 34 k cycles of a 243 k-cycle step at D = 32)
      for (int i = tid; i < nvalid * D; i += PM_NT)
        if (!isfinite(xb[i]) || (SP && *p_ovf)) atomicMin(A.status, t);
    } else
    for (int r = tid; r < R; r += PM_NT) {
      float rv = 0.f;
      if (r < nvalid) {
        rv = reward_row(A.rew, xb + r * D, D, L.av + r * U, U, nullptr);
        bool ok = isfinite(rv);
        for (int d = 0; d < D; ++d) ok = ok && isfinite(xb[r * D + d]);
        if (SP && *p_ovf) ok = false;   // an activation left fp16's range in this step (or earlier): the host retries in fp32
        if (!ok) atomicMin(A.status, t);
        const size_t o = (size_t)t * B + row0 + r;
        if (A.flags & PMBRL_FLAG_MM_REWARDS) A.rt[o] = rv;
        else A.rewards[o] = rv;
      }
      L.rr[r] = rv;
    }
    if (A.mm_mode == 1) {
      __syncthreads();
      // moment matching inside the workgroup (utils/rollout.py:121-145)
      const int gpw = A.rows_per_wg / A.M;   // whole groups per workgroup
      for (int gi = wid; gi < gpw; gi += PM_NW) {
        const int lr0 = gi * A.M;
        if (lr0 >= nvalid) break;
        double* scr = L.mm + (size_t)wid * pm_mm_scratch_doubles(D);
        if (A.flags & PMBRL_FLAG_MM_STATES) {
          const bool ok = pm_mm_fwd(xb + lr0 * D, D, A.M, D, pm_zbase(A.zmm, D, t, A.Bg, A.flags), D,
                                    pm_zrow0(t, A.row_off + row0 + lr0, A.flags),
                                    A.Bg, (A.flags & PMBRL_FLAG_INFER_NS) != 0, xa + lr0 * D, D,
                                    scr, lane);
          if (!ok && lane == 0) atomicMin(A.status, t);
        }
        if (A.flags & PMBRL_FLAG_MM_REWARDS) {
          const bool ok = pm_mm_fwd(L.rr + lr0, 1, A.M, 1, pm_zbase(A.zrr, 1, t, A.Bg, A.flags), 1,
                                    pm_zrow0(t, A.row_off + row0 + lr0, A.flags),
                                    A.Bg, (A.flags & PMBRL_FLAG_INFER_NS) != 0, L.gr + lr0, 1,
                                    scr, lane);
          if (!ok && lane == 0) atomicMin(A.status, t);
        }
      }
      __syncthreads();
      // publish the moment-matched results
      if (A.flags & PMBRL_FLAG_MM_STATES) {
        for (int i = tid; i < nvalid * D; i += PM_NT)
          A.states[((size_t)(t + 1) * B + row0) * D + i] = xa[i];
      }
      if (A.flags & PMBRL_FLAG_MM_REWARDS) {
        for (int r = tid; r < nvalid; r += PM_NT) A.rewards[(size_t)t * B + row0 + r] = L.gr[r];
      }
      if (!(A.flags & PMBRL_FLAG_MM_STATES)) {
        float* tmp = xa; xa = xb; xb = tmp;
      }
    } else {
      float* tmp = xa; xa = xb; xb = tmp;
    }
    __syncthreads();
    PM_MARK(22);
  }
}

// ===========================================================================
// backward sweep.  Produces the G stash consumed by the
// dW GEMM; policy dW/db are NOT accumulated here.
// ===========================================================================
// (PR = 2: two bf16 pieces -- the adjoint is linear in the incoming gradient, profiles/r02_split_precision_study.txt)
template <int RT, int PR = 0, int IP = 0>
__global__ __launch_bounds__(PM_NT, PR ? 1 : 2) void pm_rollout_bwd(const RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16 * RT;
  constexpr bool SP = PR != 0;
  const unsigned LDB = (unsigned)A.LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int row0 = wg * A.rows_per_wg;
  const int nvalid = min(A.rows_per_wg, A.B - row0);
  const int D = A.D, U = A.U, LD = A.LD, B = A.B;
  LdsMap L = pm_lds_carve(smem, R, LD, D, U, RT, IP != 0);
  float* gx = L.xa;    // dL/dx_{t+1} on entry of a step, dL/dx_t on exit
  float* gxt = L.xb;   // dL/dx~ (pre-mm next state)
  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  const bool mms = (A.flags & PMBRL_FLAG_MM_STATES) != 0;
  const bool mmr = (A.flags & PMBRL_FLAG_MM_REWARDS) != 0;
  const FeatMap* pmap = A.ang ? &A.ang->pol : nullptr;
  const FeatMap* dmap = A.ang ? &A.ang->dyn : nullptr;
  // truncated horizon (utils/rollout.py:154-157): only the steps the forward sweep completed
  const int T1 = A.nvalid ? min(A.t1, *A.nvalid) : A.t1;

  if constexpr (IP >= 2) pw_table_init(L.tbl, tid);
  // dL/dx_T1: zero, the carried value from the previous launch, or the terminal grad_states
  for (int i = tid; i < R * D; i += PM_NT) {
    const int r = i / D, d = i - r * D;
    float v = 0.f;
    if (r < nvalid) {
      if (A.gx_from_carry) v = A.gx_carry[(size_t)(row0 + r) * D + d];
      else if (A.grad_states) v = A.grad_states[((size_t)T1 * B + row0 + r) * D + d];
    }
    gx[i] = v;
  }
  __syncthreads();

  for (int t = T1 - 1; t >= A.t0; --t) {
    int tid_s = tid;
    if constexpr (IP >= 2) asm volatile("" : "+v"(tid_s));   // (see the forward sweep)
    const int tid = tid_s, lane = tid & 63;
    const size_t blk = (size_t)t * A.nwg + wg;
    float* X = L.bufA;
    float* Y = L.bufB;
    PM_MARK(0);
    if constexpr (IP >= 2) {
      // The general family's in-place sweeps take their rewards' Jacobians from pm_reward_all_kernel and never do the
      // moment matching in the kernel (mm_mode 0 / 2), no mixture head: the four phases below (row loads, copy, reward
      // adjoint, head-adjoint input: 40 k cycles of dependent loads, a barrier each) are ONE pass --
      //   dL/dx~ = dL/dx_{t+1} + dL/dr J_x;  X = [dL/dx~ Sy | dL/dx~ Td | 0];  direct action gradient dL/dr J_a
      // every element's loads independent of every other's, so that an unrolled loop has them all in flight.
      const int K16 = pm_kb32(F.nt[F.nl]) * 32;
#pragma unroll 4
      for (int i = tid; i < R * D; i += PM_NT) {
        const int r = i / D, d = i - r * D;
        const bool v = r < nvalid;
        const size_t row = (size_t)t * B + row0 + (v ? r : 0);
        // (loads unconditional on a clamped row, the selection behind them: a load under `v ? .. : 0` is a branch,
        //  and a branch per load is what keeps an unrolled loop from having its loads in flight together)
        const float gr0 = A.grad_rewards[row], jx0 = A.Jx[row * D + d], td0 = A.Td[row * D + d], sy = A.Sy[d];
        const float gr = v ? gr0 : 0.f, jx = v ? jx0 : 0.f, td = v ? td0 : 0.f;
        const float g = gx[i] + gr * jx;
        gxt[i] = g;
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, d), v ? g * sy : 0.f);
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, D + d), g * td);
      }
      for (int i = tid; i < R * U; i += PM_NT) {
        const int r = i / U, j = i - r * U;
        const bool v = r < nvalid;
        const size_t row = (size_t)t * B + row0 + (v ? r : 0);
        const float gr0 = A.grad_rewards[row], ja0 = A.Ja[row * U + j], ac0 = A.actions[row * U + j];
        L.gad[r * 16 + j] = v ? gr0 * ja0 : 0.f;
        L.av[i] = v ? ac0 : 0.f;
      }
      for (int i = tid; i < R * (K16 - 2 * D); i += PM_NT) {
        const int r = i / (K16 - 2 * D), k = i - r * (K16 - 2 * D);
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, 2 * D + k), 0.f);
      }
      __syncthreads();
      PM_MARK(3);
    } else {
    // ---- load x~ rows (for reward / mm recompute) into Y scratch columns, actions, upstream gr
    //      Y[r][0..D) = x~ ; L.av = a ; L.gr = dL/dr ; L.rr = r~
    {
      const float* xsrc = mms ? A.xt + (size_t)t * B * D : A.states + (size_t)(t + 1) * B * D;
      for (int i = tid; i < R * D; i += PM_NT) {
        const int r = i / D, d = i - r * D;
        Y[r * LD + d] = (r < nvalid) ? xsrc[(size_t)(row0 + r) * D + d] : 0.f;
      }
      for (int i = tid; i < R * U; i += PM_NT) {
        const int r = i / U;
        L.av[i] = (r < nvalid) ? A.actions[((size_t)t * B + row0) * U + i] : 0.f;
      }
      const float* rsrc = mmr ? A.rt : A.rewards;
      for (int r = tid; r < R; r += PM_NT) {
        const bool v = r < nvalid;
        L.gr[r] = v ? A.grad_rewards[(size_t)t * B + row0 + r] : 0.f;
        L.rr[r] = v ? rsrc[(size_t)t * B + row0 + r] : 0.f;
      }
    }
    __syncthreads();
    // ---- adjoint of moment matching (in-kernel groups)
    if (A.mm_mode == 1) {
      const int gpw = A.rows_per_wg / A.M;
      for (int gi = wid; gi < gpw; gi += PM_NW) {
        const int lr0 = gi * A.M;
        if (lr0 >= nvalid) break;
        double* scr = L.mm + (size_t)wid * pm_mm_scratch_doubles(D);
        if (mms)
          pm_mm_bwd(Y + lr0 * LD, LD, A.M, D, pm_zbase(A.zmm, D, t, A.Bg, A.flags), D,
                    pm_zrow0(t, A.row_off + row0 + lr0, A.flags), A.Bg,
                    (A.flags & PMBRL_FLAG_INFER_NS) != 0, gx + lr0 * D, D, gxt + lr0 * D, D, scr,
                    lane);
        if (mmr)
          pm_mm_bwd(L.rr + lr0, 1, A.M, 1, pm_zbase(A.zrr, 1, t, A.Bg, A.flags), 1,
                    pm_zrow0(t, A.row_off + row0 + lr0, A.flags), A.Bg,
                    (A.flags & PMBRL_FLAG_INFER_NS) != 0, L.gr + lr0, 1, L.gr + lr0, 1, scr, lane);
      }
      __syncthreads();
      if (!mms) {
        for (int i = tid; i < R * D; i += PM_NT) gxt[i] = gx[i];
        __syncthreads();
      }
    } else if (A.mm_mode == 2) {
      // external mm-adjoint kernel already turned (gx_carry, grad_rewards) into
      // (dL/dx~, dL/dr~): gx holds dL/dx~ and L.gr holds dL/dr~ here.
      for (int i = tid; i < R * D; i += PM_NT) gxt[i] = gx[i];
      __syncthreads();
    } else {
      for (int i = tid; i < R * D; i += PM_NT) gxt[i] = gx[i];
      __syncthreads();
    }
    PM_MARK(1);
    // ---- reward adjoint: gxt += dr/dx~ * gr ; ga_direct -> X scratch column block
    //      (X[r][0..U) holds the direct action gradient until phase B)
    if (A.ext_reward) {
      // Jacobians of the reward from pm_reward_all_kernel: dL/dx~ += dL/dr~ J_x, direct action gradient dL/dr~ J_a
      for (int i = tid; i < R * max(D, U); i += PM_NT) {
        const int r = i / max(D, U), k = i - r * max(D, U);
        const bool v = r < nvalid;
        const size_t row = (size_t)t * B + row0 + r;
        if (k < D && v) gxt[r * D + k] += L.gr[r] * A.Jx[row * D + k];
        if (k < U) L.gad[r * 16 + k] = v ? L.gr[r] * A.Ja[row * U + k] : 0.f;
      }
    } else
    for (int r = tid; r < R; r += PM_NT) {
      float ga[16];
      float xrow[PMBRL_MAX_DIM], gxr[PMBRL_MAX_DIM];
      for (int j = 0; j < U; ++j) ga[j] = 0.f;
      if (r < nvalid) {
        for (int d = 0; d < D; ++d) { xrow[d] = Y[r * LD + d]; gxr[d] = 0.f; }
        reward_row_bwd(A.rew, xrow, D, L.av + r * U, U, L.rr[r], L.gr[r], gxr, ga);
        for (int d = 0; d < D; ++d) gxt[r * D + d] += gxr[d];
      }
      for (int j = 0; j < U; ++j) L.gad[r * 16 + j] = ga[j];
    }
    __syncthreads();
    PM_MARK(2);
    // ---- dynamics head adjoint input: [gxt*Sy | gxt*Td | 0] -> X
    {
      const int K16 = SP ? pm_kb32(F.nt[F.nl]) * 32 : F.nt[F.nl] * 16;
      for (int i = tid; i < R * K16; i += PM_NT) {
        const int r = i / K16, k = i - r * K16;
        float v = 0.f;
        if (r < nvalid && A.gmm_n > 1) {
          // mixture head: only the drawn component's mean / log-std see the state gradient; logits and
          // log-temperature through the coefficients the forward sweep left
          const int n = A.gmm_n, nD = n * D;
          const size_t row = (size_t)t * B + row0 + r;
          if (k < 2 * nD) {
            const int kk = k < nD ? k : k - nD, dd = kk / n, c = kk - dd * n;
            if (c == A.gmm_k[row]) v = gxt[r * D + dd] * (k < nD ? A.Sy[dd] : A.Td[row * D + dd]);
          } else if (k <= 2 * nD + n) {
            const float* cp = A.gmm_c + (row * (size_t)(n + 1) + (k - 2 * nD)) * D;
            for (int dd = 0; dd < D; ++dd) v = fmaf(gxt[r * D + dd], cp[dd], v);
          }
        } else if (r < nvalid) {
          if (k < D) v = gxt[r * D + k] * A.Sy[k];
          else if (k < 2 * D) v = gxt[r * D + k - D] * A.Td[((size_t)t * B + row0 + r) * D + k - D];
        }
        if constexpr (SP) pm_put_planes<R, false>(X, LDB, r, k, v);
        else X[r * LD + k] = v;
      }
    }
    __syncthreads();
    PM_MARK(3);
    }
    // ---- dynamics trunk, dX only (weights frozen: no dV)
    for (int l = F.nl - 1; l >= 1; --l) {
      const int nt = F.nt[l];
      EpiHiddenBwdT<SP ? 2 : 0, R> e{F.abits[l - 1] + (size_t)t * B * nt, F.keep[l - 1], IP ? X : Y, nullptr, LD, A.Rw,
                                    row0, nvalid, nt, lane};
      if constexpr (IP >= 2) {
        const PwBwd w{F.abits[l - 1] + (size_t)t * B * nt, F.keep[l - 1], X, nullptr, L.tbl, row0, nvalid,
                      (A.prof && wg == 0 && l == 2) ? A.prof + (size_t)t * 32 + 8 : nullptr, 0};
        pw_hidden_bwd<0>(F.wb[l], pm_kb32(F.nt[l + 1]), w, wid, lane);
      } else if constexpr (IP) gemm_layer_inplace_s<RT, false>(F.wb[l], nt, pm_kb32(F.nt[l + 1]), X, LDB, wid, lane, e);
      else if constexpr (SP) gemm_tiles_s<RT, false>(F.wb[l], nt, pm_kb32(F.nt[l + 1]), X, LDB, wid, lane, e);
      else gemm_tiles<RT>(F.wb[l], nt, F.nt[l + 1], X, LD, wid, lane, e);
      if constexpr (IP >= 2) pw_lds_barrier();
      else __syncthreads();
      PM_MARK(4 + l);
      if constexpr (!IP) { float* tmp = X; X = Y; Y = tmp; }
    }
    // grad wrt normalised dynamics input [x | a] -> Y[r][0..D+U)
    if constexpr (IP) {
      Y = X + PM_IP_NOFF;
      EpiPlain e{nullptr, Y, LD, lane};
      if constexpr (IP >= 2) pw_narrow<false>(F.wb[0], F.nt[0], X, nullptr, Y, LD, wid, lane);
      else gemm_layer_inplace_s<RT, false>(F.wb[0], F.nt[0], pm_kb32(F.nt[1]), X, LDB, wid, lane, e);
      __syncthreads();
    } else if constexpr (SP)
      gemm_narrow_s<RT, false>(F.wb[0], F.nt[0], pm_kb32(F.nt[1]), nullptr, X, LDB, Y, LD, L.part, wid, lane, tid);
    else
      gemm_narrow<RT>(F.wb[0], F.nt[0], F.nt[1], nullptr, X, Y, LD, L.part, wid, lane, tid);
    PM_MARK(12);
    // ---- phase B: split into state / action parts; policy head adjoint -> X
    if constexpr (IP >= 2) {
      // (as above: the state part, the action part and the padding as separate loops with independent elements)
      const int K16 = P.nt[P.nl] * 16;
      const int KP = pm_kb32(P.nt[P.nl]) * 32;
      float* gst = A.gT[P.nl - 1] + blk * (size_t)K16 * A.Rw;
#pragma unroll 4
      for (int i = tid; i < R * D; i += PM_NT) {
        const int r = i / D, k = i - r * D;
        gxt[i] += Y[r * LD + k] * A.iSx[k];
      }
      for (int i = tid; i < R * U; i += PM_NT) {
        const int r = i / U, j = i - r * U;
        float go_mu = 0.f, go_ls = 0.f;
        if (r < nvalid) {
          const size_t o = ((size_t)t * B + row0 + r) * U + j;
          const float tp = A.Tp[o];
          const float sc = A.pscale[j], pb = A.pbias[j];
          float ga = L.gad[r * 16 + j] + Y[r * LD + D + j] * A.iSx[D + j];
          if (A.grad_actions) ga += A.grad_actions[o];
          L.gad[r * 16 + j] = ga;   // total dL/da_t (for the priority hook below)
          const float th = (L.av[r * U + j] - pb) / sc;
          const float gu = ga * sc * (1.f - th * th);
          go_mu = gu;
          go_ls = gu * tp;
        }
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, j), go_mu);
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, U + j), go_ls);
        gst[(size_t)j * A.Rw + r] = go_mu;
        gst[(size_t)(U + j) * A.Rw + r] = go_ls;
      }
      for (int i = tid; i < R * (KP - 2 * U); i += PM_NT) {
        const int r = i / (KP - 2 * U), k = 2 * U + i - r * (KP - 2 * U);
        pm_put_planes<R, false>(X, LDB, r, pw_sw(r, k), 0.f);
        if (k < K16) gst[(size_t)k * A.Rw + r] = 0.f;
      }
    } else {
      const int K16 = P.nt[P.nl] * 16;
      const int KP = SP ? pm_kb32(P.nt[P.nl]) * 32 : K16;   // split: the head-gradient tile is whole K32 blocks wide
      float* gst = A.gT[P.nl - 1] + blk * (size_t)K16 * A.Rw;
      const int W = max(KP, D + U);
      const float* xcur = A.states + (size_t)t * B * D;   // x_t (the angles the input features were taken at)
      for (int i = tid; i < R * W; i += PM_NT) {
        const int r = i / W, k = i - r * W;
        if (k < D) {
          // dL/dx_t via identity + dynamics input
          if (!dmap) {
            gxt[r * D + k] += Y[r * LD + k] * A.iSx[k];
          } else if (r < nvalid) {
            float g = 0.f;
            const int fc = dmap->f_copy[k], fs = dmap->f_sin[k];
            if (fc >= 0) g = Y[r * LD + fc] * A.iSx[fc];
            if (fs >= 0) {
              const int fo = dmap->f_cos[k];
              const float th = xcur[(size_t)(row0 + r) * D + k];
              g += Y[r * LD + fs] * A.iSx[fs] * cosf(th) - Y[r * LD + fo] * A.iSx[fo] * sinf(th);
            }
            gxt[r * D + k] += g;
          }
        } else if (k < D + U) {
          const int j = k - D;
          const int fa = dmap ? dmap->f_copy[k] : k;     // where action j sits in the dynamics input
          float go_mu = 0.f, go_ls = 0.f;
          if (r < nvalid) {
            float ga = L.gad[r * 16 + j] + Y[r * LD + fa] * A.iSx[fa];
            if (A.grad_actions) ga += A.grad_actions[((size_t)t * B + row0 + r) * U + j];
            L.gad[r * 16 + j] = ga;   // total dL/da_t (for the priority hook below)
            const float sc = A.pscale[j];
            const float th = (L.av[r * U + j] - A.pbias[j]) / sc;
            const float gu = ga * sc * (1.f - th * th);
            go_mu = gu;
            go_ls = gu * A.Tp[((size_t)t * B + row0 + r) * U + j];
          }
          if constexpr (SP) {
            pm_put_planes<R, false>(X, LDB, r, j, go_mu);
            pm_put_planes<R, false>(X, LDB, r, U + j, go_ls);
          } else {
            X[r * LD + j] = go_mu;
            X[r * LD + U + j] = go_ls;
          }
          gst[(size_t)j * A.Rw + r] = go_mu;
          gst[(size_t)(U + j) * A.Rw + r] = go_ls;
        }
        // zero the K-padding of the head-gradient tile (columns 2U..K16)
        if (k >= 2 * U && k < KP) {
          if constexpr (SP) pm_put_planes<R, false>(X, LDB, r, k, 0.f);
          else X[r * LD + k] = 0.f;
          if (k < K16) gst[(size_t)k * A.Rw + r] = 0.f;
        }
      }
    }
    __syncthreads();
    if (A.agn) {   // ||dL/da_t|| per row: the prioritised-replay hook (mc_pilco.py:156-188)
      for (int r = tid; r < nvalid; r += PM_NT) {
        float s2 = 0.f;
        for (int j = 0; j < U; ++j) s2 = fmaf(L.gad[r * 16 + j], L.gad[r * 16 + j], s2);
        A.agn[(size_t)t * B + row0 + r] = sqrtf(s2);
      }
    }
    PM_MARK(13);
    // ---- policy trunk: dX chain + G stash
    for (int l = P.nl - 1; l >= 1; --l) {
      const int nt = P.nt[l];
      EpiHiddenBwdT<SP ? 2 : 0, R> e{P.abits[l - 1] + (size_t)t * B * nt, P.keep[l - 1], IP ? X : Y,
                                    A.gT[l - 1] + blk * (size_t)nt * 16 * A.Rw, LD, A.Rw, row0, nvalid, nt, lane};
      if constexpr (IP >= 2) {
        const PwBwd w{P.abits[l - 1] + (size_t)t * B * nt, P.keep[l - 1], X,
                      A.gT[l - 1] + blk * (size_t)nt * 16 * A.Rw, L.tbl, row0, nvalid,
                      (A.prof && wg == 0 && l == 2) ? A.prof + (size_t)t * 32 + 18 : nullptr,
                      (int)((A.stash_pre >> (l - 1)) & 1u)};
        pw_hidden_bwd<IP == 3 ? 2 : 1>(P.wb[l], pm_kb32(P.nt[l + 1]), w, wid, lane);
      } else if constexpr (IP) gemm_layer_inplace_s<RT, false>(P.wb[l], nt, pm_kb32(P.nt[l + 1]), X, LDB, wid, lane, e);
      else if constexpr (SP) gemm_tiles_s<RT, false>(P.wb[l], nt, pm_kb32(P.nt[l + 1]), X, LDB, wid, lane, e);
      else gemm_tiles<RT>(P.wb[l], nt, P.nt[l + 1], X, LD, wid, lane, e);
      if constexpr (IP >= 2) pw_lds_barrier();
      else __syncthreads();
      PM_MARK(14 + l);
      if constexpr (!IP) { float* tmp = X; X = Y; Y = tmp; }
    }
    if constexpr (IP) {
      Y = X + PM_IP_NOFF;
      EpiPlain e{nullptr, Y, LD, lane};
      if constexpr (IP >= 2) pw_narrow<false>(P.wb[0], P.nt[0], X, nullptr, Y, LD, wid, lane);
      else gemm_layer_inplace_s<RT, false>(P.wb[0], P.nt[0], pm_kb32(P.nt[1]), X, LDB, wid, lane, e);
      __syncthreads();
    } else if constexpr (SP)
      gemm_narrow_s<RT, false>(P.wb[0], P.nt[0], pm_kb32(P.nt[1]), nullptr, X, LDB, Y, LD, L.part, wid, lane, tid);
    else
      gemm_narrow<RT>(P.wb[0], P.nt[0], P.nt[1], nullptr, X, Y, LD, L.part, wid, lane, tid);
    PM_MARK(22);
    // ---- phase C: dL/dx_t
    for (int i = tid; i < R * D; i += PM_NT) {
      const int r = i / D, d = i - r * D;
      float v = gxt[i];
      if (!pmap) {
        v += Y[r * LD + d];
      } else if (r < nvalid) {
        const int fc = pmap->f_copy[d], fs = pmap->f_sin[d];
        if (fc >= 0) v += Y[r * LD + fc];
        if (fs >= 0) {
          const float th = A.states[((size_t)t * B + row0 + r) * D + d];
          v += Y[r * LD + fs] * cosf(th) - Y[r * LD + pmap->f_cos[d]] * sinf(th);
        }
      }
      if (A.grad_states && r < nvalid) v += A.grad_states[((size_t)t * B + row0 + r) * D + d];
      gx[i] = v;
    }
    __syncthreads();
    PM_MARK(23);
  }
  // hand dL/dx_{t0} to the next launch / the caller
  for (int i = tid; i < nvalid * D; i += PM_NT) {
    const size_t o = (size_t)row0 * D + i;
    if (A.gx_carry) A.gx_carry[o] = gx[i];
    if (A.t0 == 0 && A.grad_x0) A.grad_x0[o] = gx[i];
  }
}
