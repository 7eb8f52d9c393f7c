// Device-side building blocks shared by the rollout kernels (gfx950 only).
//
// The one matmul shape on this path is  out[R rows, O] = act[R, K] . W[O, K]^T
// with R = 16..64 particle rows owned by one workgroup and W streamed from L2.
// It is mapped on v_mfma_f32_16x16x4_f32 (exact fp32, 1e-4 parity bar) with the
// WEIGHTS as the A operand and the ACTIVATIONS (transposed) as the B operand:
//   A[i = out feature (lane&15)][k = lane>>4]   one f32 per lane
//   B[k = lane>>4][j = particle row (lane&15)]  one f32 per lane
//   D[i = 4*(lane>>4)+r][j = lane&15]           4 f32 per lane
// so every lane ends up with 4 CONSECUTIVE output features of ONE particle row:
// the epilogue (bias, ReLU, dropout bit, stash) is lane-local and the result
// goes back to LDS as one 16-byte store.
//
// k-permutation: four consecutive MFMAs cover 16 k-values; lane group g=lane>>4
// takes k = 16*kb + 4*g + j for the j-th MFMA.  Both operands use the same
// permutation (the sum over k is order-free up to rounding), which makes each
// lane's 4 operand values 16 contiguous bytes: one ds_read_b128 for the
// activations, one global_load_dwordx4 for the (pre-packed) weights.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pmbrl.h"

#define PM_NW 8                 // waves per workgroup (two per SIMD)
#define PM_NT (PM_NW * 64)      // threads per workgroup
#define PM_MAXL PMBRL_MAX_LAYERS
#define PM_KS_NT 3              // max out tiles handled by the K-split GEMM (< PM_NW)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NetDev {
  int nl;                         // Linear layers
  int dim[PM_MAXL + 1];           // true widths
  int nt[PM_MAXL + 1];            // ceil(width / 16)
  float keep[PM_MAXL];            // divide masked activations by this (1 = no-op)
  float inv_keep[PM_MAXL];        // 1 / keep (fast kernels multiply)
  const float* wf[PM_MAXL];       // forward fragments  [nt[l+1]][nt[l]][64][4]
  const float* wb[PM_MAXL];       // transposed fragments [nt[l]][nt[l+1]][64][4]
  const float* bias[PM_MAXL];     // zero-padded to nt[l+1]*16
  const uint16_t* mask[PM_MAXL];  // dropout bits of hidden layer l: [B][nt[l+1]]
  uint16_t* abits[PM_MAXL];       // stash: mask & (pre-activation > 0): [H][B][nt[l+1]]
};

// Reward constants (device copy, built at plan creation).
struct RewardDev {
  int kind, expand, n_angle, n_other, k, De;
  int angle_dims[PMBRL_MAX_ANGLE];
  int other_dims[PMBRL_MAX_DIM];
  float C[PMBRL_MAX_TIP * PMBRL_MAX_DIM];   // [k][De], already divided by norm
  float tt[PMBRL_MAX_TIP];                  // tip_target / norm
  float w;
  float Q[PMBRL_MAX_TIP * PMBRL_MAX_TIP];
  float QQ[PMBRL_MAX_TIP * PMBRL_MAX_TIP];  // Q + Q^T
  float R[16 * 16];
  float RR[16 * 16];                        // R + R^T
  // gather form of the feature map phi (pm_reward_all_kernel walks these instead of indexing
  // per-thread arrays): phi_j = x[phi_src[j]] | sin(x[..]) | cos(x[..]) for phi_mode 0 | 1 | 2;
  // state dim d feeds phi[d_copy[d]] (or -1) and, as an angle, phi[d_sin[d]] / phi[d_cos[d]]
  int phi_src[PMBRL_MAX_DIM], phi_mode[PMBRL_MAX_DIM];
  int d_copy[PMBRL_MAX_DIM], d_sin[PMBRL_MAX_DIM], d_cos[PMBRL_MAX_DIM];
};

// Network-input feature maps for angle_dims (utils/angles.py:39-42): feature k of a network's input is
// v[src[k]] | sin(v[..]) | cos(v[..]) for mode[k] = 0 | 1 | 2, v = state (policy) or [state | action]
// (dynamics); entry s of v feeds feature f_copy[s] (or -1) and, as an angle, f_sin[s] / f_cos[s].
struct FeatMap {
  int n_feat;
  int src[PMBRL_MAX_DIM], mode[PMBRL_MAX_DIM];
  int f_copy[PMBRL_MAX_DIM], f_sin[PMBRL_MAX_DIM], f_cos[PMBRL_MAX_DIM];
};
struct AngleDev {
  FeatMap pol, dyn;
};
__device__ __forceinline__ float pm_feat(const FeatMap* m, int k, const float* v) {
  const float s = v[m->src[k]];
  const int md = m->mode[k];
  return md == 0 ? s : md == 1 ? sinf(s) : cosf(s);
}

// weight stream of the fast kernels: the hidden->hidden layers in processing order
struct StreamDesc {
  int n;
  const float* wf[2 * PM_MAXL];
  int n_ot[2 * PM_MAXL], n_kb[2 * PM_MAXL];
  // K-split of the LAST output tile (see pm_fast_ksplit): n_ot above already excludes it;
  // tw_off = offset (floats) of its LDS-resident weight k-blocks, n_kb_real = unpadded K blocks
  int ks[2 * PM_MAXL], tw_off[2 * PM_MAXL], n_kb_real[2 * PM_MAXL];
};
// LDS offsets (floats) of the per-layer regions of the fast kernels
struct FastOff {
  int pbias[PM_MAXL], dbias[PM_MAXL], pmask[PM_MAXL], dmask[PM_MAXL];
};

struct RolloutArgs {
  int B, D, U, H, Bg, row_off, flags;
  int G, M;            // moment-matching groups on this device, rows per group
  int mm_mode;         // 0 none, 1 in-kernel (group fits a workgroup), 2 external kernel, 3 per-step launches with the
                       // moment matching in their prologue (fast family)
  int t0, t1;          // step range of this launch
  int rows_per_wg, nwg, Rw;   // Rw = 16*RT = stash block width
  int LD;              // LDS leading dimension of the activation buffers (floats)
  int LDB;             // split precision: leading dimension of the bf16 piece planes (elements; 0 = fp32 path)
  int res_tiles;       // split precision: the first res_tiles (0 or 8 = one per wave) output tiles of the sweep's FIRST
                       // streamed layer keep their weights in registers for the launch; the stream table then
                       // describes only the remaining tiles of that layer (res_w: the layer's full fragments)
  const float* res_w;
  // ... and the first 8 tiles of its SECOND streamed layer live in LDS (lds_tile_s; shape-specialised instances):
  // lds_w the layer's full fragments, wlds_off where the tiles go (floats from the LDS base), lds_last_lanes the
  // lanes of the last K32 block that are stored
  const float* lds_w;
  int wlds_off, lds_last_lanes;
  float mls_pol, mls_dyn;
  NetDev pol, dyn;
  const RewardDev* rew;
  const AngleDev* ang;   // angle_dims of the policy / the dynamics model (general family only); nullptr: none
  // GaussianMixtureDensity dynamics head (general family only): components (0: diagonal Gaussian head), frozen
  // Gumbel noise [B][n], uniforms [H][B], the Gaussian noise rows the derivative of the noise term uses (the last
  // step's: the reference's autograd; nullptr: each step's own), and the stashes the adjoint consumes:
  // component drawn [H][B], coefficients of dL/d(logits, log-temperature) [H][B][n + 1][D]
  int gmm_n;
  const float *zpi, *ucat, *zdyn_grad;
  // general family: rewards, their Jacobians (Jx, Ja) and their moment matching are computed for all row-steps
  // after the forward sweep (pm_reward_all_kernel), as in the latency-optimised family; the sweeps only check the
  // sampled states and consume the Jacobians
  int ext_reward;
  // fp16 weight pieces: *wflag == wgen if pm_pack_all saw a weight beyond fp16's range in this call's weights
  const int* wflag;
  int wgen;
  int* gmm_k;
  float* gmm_c;
  const float *x0, *mx, *iSx, *my, *Sy, *pscale, *pbias, *zpol, *zdyn, *zmm, *zrr;
  float *states, *actions, *rewards;
  float* actT[PM_MAXL];   // policy layer inputs, feature-major blocks [H][nwg][nt*16][Rw]
  float* gT[PM_MAXL];     // policy pre-activation grads, same layout
  unsigned stash_pre;     // bit l: layer l's dW GEMM reads PRE-SPLIT stashes (actT[l], gT[l]: pmbrl_dw.h, pm_dw_wide_pre_kernel)
  float *Tp, *Td;         // [H][B][U], [H][B][D]
  float *xt, *rt;         // pre-moment-matching next state / reward [H][B][D], [H][B]
  double* mmfac;          // in-kernel moment matching: statistics + factor per (step, group), forward -> adjoint
  int mmfac_groups;       // groups per step in mmfac (= B / M)
  float *Jx, *Ja;         // reward Jacobian d r~/d x~ [H][B][D], d r~/d a [H][B][U] (fast kernels)
  int* status;          // forward: number of valid steps (atomicMin); backward: failure flag of the sweep (or nullptr)
  const int* nvalid;    // backward: the forward's status word; the sweep covers steps t < min(t1, *nvalid) (nullptr: t1)
  // backward only
  const float *grad_rewards, *grad_states, *grad_actions;
  float* gx_carry_out;   // mm_mode 3: the carried gradient is written here (ping-pong with gx_carry)
  // in-kernel moment matching with a group SPLIT over mm_parts workgroups (>= 2; rows_per_wg = M / mm_parts):
  // the workgroups of a group exchange their rows through HBM and meet at a group-local flag barrier
  // (pmbrl_fast.h, pm_group_sync); every one of them factors the whole group, each keeps its own rows
  int mm_parts;
  // more groups than fit the chip at once (the statistics exchange needs every workgroup of a group resident): the launch
  // covers the groups in batches; wg0 = first workgroup of this launch's batch (workgroup = blockIdx.x + wg0)
  int wg0, launch_wg;      // (launch_wg: workgroups of this launch; 0 = nwg)
  // ... more than 8 parts: the sums travel over two levels (pm_xch_get) -- mm_fan consecutive parts per collector;
  // the z standardisation of the whole group then comes from mm_ztab ([H][groups][zm (D) | zi (D)] doubles,
  // pm_mm_ztable_kernel) instead of every part walking the group's noise rows
  int mm_fan;
  const double* mm_ztab;
  // mm_mode 3 with every workgroup resident at once: ONE launch over the horizon, the workgroups
  // meet at a device-wide barrier (arrival counter gsync) where the per-step launches ended
  int mm_grid;
  unsigned* gsync;
  unsigned long long* xch;      // split groups: granules of the statistics exchange (pmbrl_fast.h, pm_xch_sum); nullptr = rows + flags
  float *grad_x0, *agn, *gx_carry;   // gx_carry [B][D]: dL/dx_{t+1} between launches (mm_mode 2)
  int gx_from_carry;
  long long zpol_ss, zdyn_ss;   // per-step strides of z_pol / z_dyn (0 = frozen)
  // raw (row-major) weights the fast kernels stage in LDS: head layers [n_out][K], first layers [h][in]
  const float *pol_head_w, *dyn_head_w, *pol_first_w, *dyn_first_w;
  StreamDesc sd_fwd, sd_bwd;
  FastOff fo;
  long long* prof;              // debug: cycle stamps [H][32] of workgroup 0 (nullptr = off)
};

// kernel variants of the latency-optimised family (pmbrl_fast.h)
#define PF_VAR_LEAN 0
#define PF_VAR_EXT 1
#define PF_VAR_MM 2
#define PF_VAR_MMG 3   // mm_mode 3 as ONE launch: the workgroups meet at a device-wide barrier every step

#define PM_MARK(slot)                                                        \
  do {                                                                        \
    if (A.prof && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + (slot)] = (long long)__builtin_readcyclecounter(); \
  } while (0)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ldg4(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}

// ---------------------------------------------------------------------------
// Tile-split GEMM: wave `wid` owns output tiles wid, wid+NW, ...; NT of them are
// in flight at once (independent accumulator chains hide the 40-cycle dependent
// MFMA latency) and weight fragments are double-buffered in registers in chunks
// of CK=4 k-blocks (64 k-values), one chunk ahead of the MFMAs.
//   wf      : fragment-packed weights  [n_ot][n_kb][64 lanes][4]
//   lds_in  : activations [16*RT][ld]  (ld % 16 == 8 -> conflict-free b128 reads)
//   epi(ot, rt, acc) : lane holds rows rt*16+(lane&15), features ot*16+4*(lane>>4)+r
// ---------------------------------------------------------------------------
#define PM_CK 4

template <int RT, int NT>
struct FragBuf {
  f32x4 a[NT][PM_CK];
};

template <int RT, int NT, class Epi>
__device__ __forceinline__ void gemm_tiles_group(const float* __restrict__ wf, int n_kb,
                                                 int ot0, int n_ot, const float* lds_in,
                                                 int ld, int lane, Epi& epi) {
  const int arow = lane & 15, g = lane >> 4;
  const float* bbase = lds_in + arow * ld + 4 * g;
  f32x4 acc[NT][RT];
  bool val[NT];
  const float* wp[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW;
    val[k] = ot < n_ot;
    wp[k] = wf + ((size_t)(val[k] ? ot : ot0) * n_kb) * 256 + lane * 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  FragBuf<RT, NT> f0, f1;

  auto load = [&](FragBuf<RT, NT>& f, int kb0) {
#pragma unroll
    for (int c = 0; c < PM_CK; ++c) {
      if (kb0 + c < n_kb) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
          if (val[k]) f.a[k][c] = ldg4(wp[k] + (size_t)(kb0 + c) * 256);
      }
    }
  };
  auto compute = [&](const FragBuf<RT, NT>& f, int kb0) {
#pragma unroll
    for (int c = 0; c < PM_CK; ++c) {
      if (kb0 + c < n_kb) {
        f32x4 b[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          b[rt] = *reinterpret_cast<const f32x4*>(bbase + rt * 16 * ld + (kb0 + c) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            if (val[k]) {
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
                acc[k][rt] = mfma4(f.a[k][c][j], b[rt][j], acc[k][rt]);
            }
          }
        }
      }
    }
  };

  load(f0, 0);
  for (int kb0 = 0; kb0 < n_kb; kb0 += 2 * PM_CK) {
    load(f1, kb0 + PM_CK);
    compute(f0, kb0);
    load(f0, kb0 + 2 * PM_CK);
    compute(f1, kb0 + PM_CK);
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (val[k]) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epi(ot0 + k * PM_NW, rt, acc[k][rt]);
    }
  }
}

// The same with n_kb a multiple of 2 * PM_CK (every chunk whole) and absent tiles of the last group
// computed as duplicates of the group's first tile and dropped: no branch inside the chunk loop
// (the general form puts a scalar branch around every pair of MFMAs).
template <int RT, int NT, class Epi>
__device__ __forceinline__ void gemm_tiles_group_full(const float* __restrict__ wf, int n_kb,
                                                      int ot0, int n_ot, const float* lds_in,
                                                      int ld, int lane, Epi& epi) {
  const int arow = lane & 15, g = lane >> 4;
  const float* bbase = lds_in + arow * ld + 4 * g;
  f32x4 acc[NT][RT];
  const float* wp[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW;
    wp[k] = wf + ((size_t)(ot < n_ot ? ot : ot0) * n_kb) * 256 + lane * 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  FragBuf<RT, NT> f0, f1;
  auto load = [&](FragBuf<RT, NT>& f, int kb0) {
#pragma unroll
    for (int c = 0; c < PM_CK; ++c)
#pragma unroll
      for (int k = 0; k < NT; ++k) f.a[k][c] = ldg4(wp[k] + (size_t)(kb0 + c) * 256);
  };
  auto compute = [&](const FragBuf<RT, NT>& f, int kb0) {
#pragma unroll
    for (int c = 0; c < PM_CK; ++c) {
      f32x4 b[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        b[rt] = *reinterpret_cast<const f32x4*>(bbase + rt * 16 * ld + (kb0 + c) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[k][rt] = mfma4(f.a[k][c][j], b[rt][j], acc[k][rt]);
    }
  };
  load(f0, 0);
  for (int kb0 = 0; kb0 < n_kb; kb0 += 2 * PM_CK) {
    load(f1, kb0 + PM_CK);
    compute(f0, kb0);
    load(f0, kb0 + 2 * PM_CK < n_kb ? kb0 + 2 * PM_CK : 0);   // past the end: a harmless re-load of chunk 0
    compute(f1, kb0 + PM_CK);
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (ot0 + k * PM_NW < n_ot) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epi(ot0 + k * PM_NW, rt, acc[k][rt]);
    }
  }
}

template <int RT, class Epi>
__device__ __forceinline__ void gemm_tiles(const float* __restrict__ wf, int n_ot, int n_kb,
                                           const float* lds_in, int ld, int wid, int lane,
                                           Epi& epi) {
  constexpr int NT = (RT >= 4) ? 1 : 2;
  if (n_kb % (2 * PM_CK) == 0) {
    for (int ot0 = wid; ot0 < n_ot; ot0 += NT * PM_NW)
      gemm_tiles_group_full<RT, NT>(wf, n_kb, ot0, n_ot, lds_in, ld, lane, epi);
  } else {
    for (int ot0 = wid; ot0 < n_ot; ot0 += NT * PM_NW)
      gemm_tiles_group<RT, NT>(wf, n_kb, ot0, n_ot, lds_in, ld, lane, epi);
  }
}

// ---------------------------------------------------------------------------
// K-split GEMM for narrow outputs (n_ot <= PM_KS_NT tiles, e.g. the 2U / 2D
// heads): every wave reduces its slice of k-blocks for ALL output tiles, the
// partial tiles meet in LDS (`part`, [NW][PM_KS_NT][RT][64][4] floats) and are
// summed in fixed wave order (deterministic) by gemm_ksplit_combine.
// ---------------------------------------------------------------------------
// Where the partial tiles live: a dedicated LDS region (`part`), or -- wide networks, where LDS is
// what limits the workgroups per CU -- the columns >= 16 * PM_KS_NT of the OUTPUT buffer's rows,
// which a narrow GEMM leaves untouched (`part` == nullptr, `alias` = that buffer; element e of the
// region sits at row e / W, column 16 * PM_KS_NT + e % W with W = the free columns rounded down to 4).
__host__ __device__ inline int pm_part_alias_w(int ld) { return (ld - 16 * PM_KS_NT) & ~3; }
__host__ __device__ inline bool pm_part_alias_ok(int R, int ld, int RT) {
  const int w = pm_part_alias_w(ld);
  return w > 0 && (size_t)R * w >= (size_t)PM_NW * PM_KS_NT * RT * 256;
}
__device__ __forceinline__ float* pm_part_at(float* part, float* alias, int ld, int e) {
  if (part) return part + e;
  const int w = pm_part_alias_w(ld);
  const int r = e / w;
  return alias + (size_t)r * ld + 16 * PM_KS_NT + (e - r * w);
}
template <int RT>
__device__ __forceinline__ void gemm_ksplit(const float* __restrict__ wf, int n_ot, int n_kb,
                                            const float* lds_in, int ld, int wid, int lane,
                                            float* part, float* alias = nullptr) {
  const int arow = lane & 15, g = lane >> 4;
  const float* bbase = lds_in + arow * ld + 4 * g;
  const int per = (n_kb + PM_NW - 1) / PM_NW;
  const int k_lo = wid * per;
  const int k_hi = min(n_kb, k_lo + per);
  f32x4 acc[PM_KS_NT][RT];
#pragma unroll
  for (int k = 0; k < PM_KS_NT; ++k)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb0 = k_lo; kb0 < k_hi; kb0 += PM_CK) {
    f32x4 a[PM_KS_NT][PM_CK];
#pragma unroll
    for (int c = 0; c < PM_CK; ++c)
      if (kb0 + c < k_hi) {
#pragma unroll
        for (int k = 0; k < PM_KS_NT; ++k)
          if (k < n_ot) a[k][c] = ldg4(wf + ((size_t)k * n_kb + kb0 + c) * 256 + lane * 4);
      }
#pragma unroll
    for (int c = 0; c < PM_CK; ++c)
      if (kb0 + c < k_hi) {
        f32x4 b[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          b[rt] = *reinterpret_cast<const f32x4*>(bbase + rt * 16 * ld + (kb0 + c) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < PM_KS_NT; ++k)
            if (k < n_ot) {
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) acc[k][rt] = mfma4(a[k][c][j], b[rt][j], acc[k][rt]);
            }
      }
  }
#pragma unroll
  for (int k = 0; k < PM_KS_NT; ++k)
    if (k < n_ot) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        *reinterpret_cast<f32x4*>(pm_part_at(part, alias, ld, (((wid * PM_KS_NT + k) * RT + rt) * 64 + lane) * 4)) =
            acc[k][rt];
    }
}

// after __syncthreads(): out[row][feature] = bias + sum_w part[w]; caller syncs after.
template <int RT>
__device__ __forceinline__ void gemm_ksplit_combine(float* part, int n_ot,
                                                    const float* bias, float* lds_out, int ld,
                                                    int tid, float* alias = nullptr) {
  for (int item = tid; item < n_ot * RT * 64; item += PM_NT) {
    const int ln = item & 63;
    const int rt = (item >> 6) % RT;
    const int k = (item >> 6) / RT;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < PM_NW; ++w)
      s += *reinterpret_cast<const f32x4*>(pm_part_at(part, alias, ld, (((w * PM_KS_NT + k) * RT + rt) * 64 + ln) * 4));
    const int f0 = k * 16 + 4 * (ln >> 4);
    if (bias) s += ldg4(bias + f0);
    *reinterpret_cast<f32x4*>(lds_out + (rt * 16 + (ln & 15)) * ld + f0) = s;
  }
}

// plain "out = acc (+ bias)" epilogue of the tile-split GEMM (wide heads)
struct EpiPlain {
  const float* bias;
  float* lds_out;
  int ld, lane;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int f0 = ot * 16 + 4 * (lane >> 4);
    if (bias) acc += ldg4(bias + f0);
    *reinterpret_cast<f32x4*>(lds_out + (rt * 16 + (lane & 15)) * ld + f0) = acc;
  }
  // (operands fetched at the start of a tile group: see EpiHiddenFwdT::pre)
  struct Pre {
    f32x4 b;
  };
  __device__ __forceinline__ Pre pre(int ot, int) const {
    Pre p;
    p.b = bias ? ldg4(bias + ot * 16 + 4 * (lane >> 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    return p;
  }
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc, const Pre& pr) {
    const int f0 = ot * 16 + 4 * (lane >> 4);
    *reinterpret_cast<f32x4*>(lds_out + (rt * 16 + (lane & 15)) * ld + f0) = acc + pr.b;
  }
};

// head / tail layer: picks K-split or tile-split by width.  Contains barriers:
// must be called by all threads.  Result (with bias) in lds_out[row][0..n_ot*16).
template <int RT>
__device__ __forceinline__ void gemm_narrow(const float* __restrict__ wf, int n_ot, int n_kb,
                                            const float* bias, const float* lds_in,
                                            float* lds_out, int ld, float* part, int wid,
                                            int lane, int tid) {
  if (n_ot <= PM_KS_NT) {
    // (part == nullptr: the partial tiles go to the free columns of lds_out's rows)
    gemm_ksplit<RT>(wf, n_ot, n_kb, lds_in, ld, wid, lane, part, lds_out);
    __syncthreads();
    gemm_ksplit_combine<RT>(part, n_ot, bias, lds_out, ld, tid, lds_out);
  } else {
    EpiPlain e{bias, lds_out, ld, lane};
    gemm_tiles<RT>(wf, n_ot, n_kb, lds_in, ld, wid, lane, e);
  }
  __syncthreads();
}

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words (pmbrl_draw_masks, the BNN
// training step's in-kernel dropout noise)
__device__ __forceinline__ void pm_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                          unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// numerically careful scalar helpers (match torch CPU within fp32 rounding)
__device__ __forceinline__ float softplusf(float x) {
  // torch.nn.functional.softplus, threshold 20 (models/densities.py:97)
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// block-wide sum in a fixed order (sm: blockDim.x doubles of LDS); contains barriers
__device__ __forceinline__ double pm_block_sum(double s, double* sm) {
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  return sm[0];
}

