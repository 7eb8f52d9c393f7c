// Moment matching of a group whose rows are spread over several devices (the reference is
// single-process: this is utils/rollout.py:20-29 over the rows of ALL ranks, e.g. mm_groups=None of
// examples/deep_pilco_mm.py:31 on a sharded run).
//
// Per step and group every rank
//   1. leaves the statistics of its own rows in ITS slot of a [ranks][groups][slot] fp64 buffer and zeros in
//      the other slots (pm_mmx_stats): row count, mean, centred second moments, sum z, sum z^2;
//   2. the buffer is summed over the ranks in place (one small all-reduce on the compute stream: with disjoint
//      slots that is an all-gather, so every rank then combines the SAME numbers in the SAME order);
//   3. combines the slots in rank order with the pairwise update of the centred moments (no raw second moments:
//      a 1 x 1 reward "covariance" of nearly equal rewards would cancel to noise), factors the covariance
//      exactly as the single-device routine does (pm_mm_chol) and maps its own rows (pm_mmx_apply).
// The adjoint needs two sums over all rows of the group -- mbar = sum_r g_r and Lbar = tril(g^T zhat) -- which
// are plain sums (pm_mmx_bwd_sums, one all-reduce), then the single-device tail (pm_mm_bwd_solve / _rows) on its own
// rows with the group's 1 / M.  One workgroup per group -- the sums over the rows split over its waves, the d x d
// chain on wave 0; everything in fp64 like pmbrl_mm.h.
#pragma once
#include "pmbrl_mm.h"

__host__ __device__ inline size_t pm_mmx_slot_doubles(int d) { return (size_t)d * d + 3 * d + 1; }
__host__ __device__ inline size_t pm_mmx_bwd_doubles(int d) { return (size_t)d * d + d; }

// Waves per group: the sums over a rank's rows are split over the waves of one workgroup (rows wid * P + part,
// stride nw * P), the waves' partial sums meet in LDS and wave 0 adds them in wave order (deterministic).
// `part` holds nw x pm_mmx_part_doubles(d) doubles.
__host__ __device__ inline size_t pm_mmx_part_doubles(int d) { return (size_t)d * d + 3 * d; }
__host__ inline int pm_mmx_waves(int M, int d) {
  int nw = (M + 63) / 64;
  const int cap = (int)(4096 / pm_mmx_part_doubles(d));   // <= 32 KB of partial sums (with the scratch: < 64 KB of LDS)
  if (nw > cap) nw = cap;
  if (nw > 16) nw = 16;
  return nw < 1 ? 1 : nw;
}

// workgroups the sums over a rank's M rows of a group are spread over (each at least 128 rows, at most PM_MMX_NB)
#define PM_MMX_NB 16
__host__ inline int pm_mmx_blocks(int M) {
  int nb = M / 128;
  if (nb > PM_MMX_NB) nb = PM_MMX_NB;
  return nb < 1 ? 1 : nb;
}

// slot: [n | mean (d) | M2 (d*d, lower triangle used) | sum z (d) | sum z^2 (d)]
// (all waves of the workgroup call this; scr: d doubles, part: see above)
__device__ __forceinline__ void pm_mmx_stats(const float* s, int s_ld, int M, int d, const float* z, int z_ld,
                                             int zrow0, int Bg, double* slot, double* scr, double* part, int nw,
                                             int wid, int lane) {
  double* mean = scr;   // d doubles of scratch
  const double inv_m = 1.0 / (double)M;
  {
    const int e2 = pm_pow2ceil(d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int pt = lane % P;
    for (int base = 0; base < d; base += per) {
      const int j = base + lane / P;
      double m = 0.0, zm = 0.0, zz = 0.0;
      if (j < d)
        for (int i = wid * P + pt; i < M; i += nw * P) {
          m += (double)s[(size_t)i * s_ld + j];
          const double zv = (double)z[(size_t)pm_zidx(zrow0, i, Bg) * z_ld + j];
          zm += zv;
          zz += zv * zv;
        }
      m = pm_seg_sum(m, P);
      zm = pm_seg_sum(zm, P);
      zz = pm_seg_sum(zz, P);
      if (j < d && pt == 0) {
        double* pw = part + (size_t)wid * 3 * d;
        pw[j] = m;
        pw[d + j] = zm;
        pw[2 * d + j] = zz;
      }
    }
  }
  __syncthreads();
  if (wid == 0) {
    for (int j = lane; j < d; j += 64) {
      double m = 0.0, zm = 0.0, zz = 0.0;
      for (int w = 0; w < nw; ++w) {
        const double* pw = part + (size_t)w * 3 * d;
        m += pw[j];
        zm += pw[d + j];
        zz += pw[2 * d + j];
      }
      mean[j] = m * inv_m;
      slot[1 + j] = m * inv_m;
      slot[1 + d + d * d + j] = zm;
      slot[1 + 2 * d + d * d + j] = zz;
    }
    if (lane == 0) slot[0] = (double)M;
  }
  __syncthreads();
  {
    const int e2 = pm_pow2ceil(d * d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int pt = lane % P;
    for (int base = 0; base < d * d; base += per) {
      const int e = base + lane / P;
      const int i = e / d, j = e - i * d;
      double acc = 0.0;
      const bool live = e < d * d && j <= i;
      if (live) {
        const double mi = mean[i], mj = mean[j];
        for (int r = wid * P + pt; r < M; r += nw * P)
          acc += ((double)s[(size_t)r * s_ld + i] - mi) * ((double)s[(size_t)r * s_ld + j] - mj);
      }
      acc = pm_seg_sum(acc, P);
      if (e < d * d && pt == 0) part[(size_t)wid * d * d + e] = live ? acc : 0.0;
    }
  }
  __syncthreads();
  if (wid == 0)
    for (int e = lane; e < d * d; e += 64) {
      double acc = 0.0;
      for (int w = 0; w < nw; ++w) acc += part[(size_t)w * d * d + e];
      slot[1 + d + e] = acc;
    }
}

// ---------------------------------------------------------------------------
// Row-per-thread forms of the ELEMENTWISE halves for d <= PM_MMX_DF (every shape of the configurations; d = 1:
// the rewards): a thread owns rows tid, tid + nthreads, ..., loads a row once (d contiguous floats) and writes it once
// (apply 12.7 -> 8.4 us, adjoint rows 12.8 -> 9.5 us on a 2500-row slice at d = 4).  The same idea for the SUMS --
// every entry in registers, wave butterflies, LDS in wave order -- was built and measured slower (stats 17 -> 30 us:
// 39 fp64 butterflies per wave on the LDS crossbar); the sums are spread over several workgroups instead (below).
// ---------------------------------------------------------------------------
#define PM_MMX_DF 6
// out rows = mean + zhat L^T, one thread per row (same order of additions per element as pm_mmx_apply)
__device__ __forceinline__ void pm_mmx_apply_rows(int M, int d, const float* z, int z_ld, int zrow0, int Bg,
                                                  float* out, int out_ld, const MMScratch& q, int tid, int nth) {
  constexpr int DF = PM_MMX_DF;
  for (int r = tid; r < M; r += nth) {
    const float* zr = z + (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
    double zh[DF];
#pragma unroll
    for (int c = 0; c < DF; ++c) zh[c] = c < d ? ((double)zr[c] - q.zmean[c]) * q.zistd[c] : 0.0;
#pragma unroll
    for (int j = 0; j < DF; ++j)
      if (j < d) {
        double acc = q.mean[j];
#pragma unroll
        for (int c = 0; c <= j; ++c) acc += zh[c] * q.Lm[j * d + c];
        out[(size_t)r * out_ld + j] = (float)acc;
      }
  }
}

// dL/ds rows of the adjoint, one thread per row (pm_mm_bwd_rows' order of additions)
__device__ __forceinline__ void pm_mmx_bwd_rows(const float* s, int s_ld, int M, int d, double inv_m, float* gout,
                                                int gout_ld, const MMScratch& q, int tid, int nth) {
  constexpr int DF = PM_MMX_DF;
  for (int r = tid; r < M; r += nth) {
    const float* sr = s + (size_t)r * s_ld;
    double dl[DF];
#pragma unroll
    for (int c = 0; c < DF; ++c) dl[c] = c < d ? (double)sr[c] - q.mean[c] : 0.0;
#pragma unroll
    for (int j = 0; j < DF; ++j)
      if (j < d) {
        double acc = q.mbar[j] * inv_m;
#pragma unroll
        for (int c = 0; c < DF; ++c)
          if (c < d) acc += dl[c] * q.P[c * d + j];
        gout[(size_t)r * gout_ld + j] = (float)acc;
      }
  }
}

// slots of all ranks (summed buffer, `stride` doubles from one rank's slot to the next) -> means, z
// standardisation, covariance and its factor in q.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool pm_mmx_factor(const double* slots, size_t stride, int nranks, int d,
                                              const MMScratch& q, int lane) {
  double Mtot = 0.0;
  for (int w = 0; w < nranks; ++w) Mtot += slots[w * stride];
  const double inv_m = 1.0 / Mtot, inv_m1 = 1.0 / (Mtot - 1.0);
  for (int j = lane; j < d; j += 64) {
    double m = 0.0, zm = 0.0, zz = 0.0;
    for (int w = 0; w < nranks; ++w) {
      const double* sl = slots + w * stride;
      m += sl[0] * sl[1 + j];
      zm += sl[1 + d + d * d + j];
      zz += sl[1 + 2 * d + d * d + j];
    }
    m *= inv_m;
    zm *= inv_m;
    q.mean[j] = m;
    q.zmean[j] = zm;
    q.zistd[j] = pm_rsqrt((zz - Mtot * zm * zm) * inv_m1);
  }
  pm_wave_sync();
  for (int e = lane; e < d * d; e += 64) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
    if (j <= i) {
      const double mi = q.mean[i], mj = q.mean[j];
      for (int w = 0; w < nranks; ++w) {
        const double* sl = slots + w * stride;
        acc += sl[1 + d + e] + sl[0] * (sl[1 + i] - mi) * (sl[1 + j] - mj);
      }
      acc = acc * inv_m1 + (i == j ? 1e-12 : 0.0);
    }
    q.Lm[e] = acc;
  }
  pm_wave_sync();
  return pm_mm_chol(d, q, lane);
}

// out rows = mean + zhat L^T for this rank's M rows of the group (pm_mm_fwd's last loop), thread tid of nthreads;
// infer_noise_variables (utils/rollout.py:6-17): zhat = Delta L^-T, the rows come back as they are (s, s_ld)
__device__ __forceinline__ void pm_mmx_apply(int M, int d, const float* z, int z_ld, int zrow0, int Bg,
                                             float* out, int out_ld, const MMScratch& q, int tid, int nthreads,
                                             const float* s = nullptr, int s_ld = 0) {
  if (s) {
    for (int e = tid; e < M * d; e += nthreads) {
      const int r = e / d, j = e - r * d;
      out[(size_t)r * out_ld + j] = s[(size_t)r * s_ld + j];
    }
    return;
  }
  for (int e = tid; e < M * d; e += nthreads) {
    const int r = e / d, j = e - r * d;
    double acc = q.mean[j];
    const size_t zr = (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
    for (int c = 0; c <= j; ++c)
      acc += ((double)z[zr + c] - q.zmean[c]) * q.zistd[c] * q.Lm[j * d + c];
    out[(size_t)r * out_ld + j] = (float)acc;
  }
}

// this rank's part of mbar = sum_r g_r (d) and Lbar = tril(g^T zhat) (d*d) -> sums[d + d*d]
// (all waves of the workgroup; part: nw x (d + d*d) doubles)
__device__ __forceinline__ void pm_mmx_bwd_sums(int M, int d, const float* z, int z_ld, int zrow0, int Bg,
                                                const float* g, int g_ld, const MMScratch& q, double* sums,
                                                double* part, int nw, int wid, int lane,
                                                const float* s_inf = nullptr, int s_ld = 0) {
  // s_inf (infer_noise_variables): the second sum is A = g^T Delta (full d x d, Delta = s - the GROUP's mean)
  double* pw = part + (size_t)wid * (d + d * d);
  {
    const int e2 = pm_pow2ceil(d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int pt = lane % P;
    for (int base = 0; base < d; base += per) {
      const int j = base + lane / P;
      double a = 0.0;
      if (j < d)
        for (int r = wid * P + pt; r < M; r += nw * P) a += (double)g[(size_t)r * g_ld + j];
      a = pm_seg_sum(a, P);
      if (j < d && pt == 0) pw[j] = a;
    }
  }
  {
    const int e2 = pm_pow2ceil(d * d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int pt = lane % P;
    for (int base = 0; base < d * d; base += per) {
      const int e = base + lane / P;
      const int i = e / d, j = e - i * d;
      double acc = 0.0;
      const bool live = e < d * d && (j <= i || s_inf);
      if (live && s_inf) {
        const double mj = q.mean[j];
        for (int r = wid * P + pt; r < M; r += nw * P)
          acc += (double)g[(size_t)r * g_ld + i] * ((double)s_inf[(size_t)r * s_ld + j] - mj);
      } else if (live) {
        const double zm = q.zmean[j], zs = q.zistd[j];
        for (int r = wid * P + pt; r < M; r += nw * P)
          acc += (double)g[(size_t)r * g_ld + i] *
                 (((double)z[(size_t)pm_zidx(zrow0, r, Bg) * z_ld + j] - zm) * zs);
      }
      acc = pm_seg_sum(acc, P);
      if (e < d * d && pt == 0) pw[d + e] = live ? acc : 0.0;
    }
  }
  __syncthreads();
  if (wid == 0)
    for (int e = lane; e < d + d * d; e += 64) {
      double acc = 0.0;
      for (int w = 0; w < nw; ++w) acc += part[(size_t)w * (d + d * d) + e];
      sums[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// kernels.  One workgroup per (rank slot, group[, step]); the caller's rows of group gi are rows
// [gi * A.M, (gi + 1) * A.M) of this device; their global row (the cyclic noise index of utils/rollout.py:53-59)
// is gi * span_rows + span_off + i.
// ---------------------------------------------------------------------------
struct MmxArgs {
  int nranks, rank;
  int span_rows, span_off;   // rows of a group over all ranks; this rank's first row inside each group
  double* buf;               // forward [nranks][n_items][slot]; backward [n_items][d + d*d]
  double* fac;               // [n_items][pm_mm_fac_doubles(d)]: the factor, forward -> adjoint
  // the SUMS over a rank's rows of a group are spread over nb workgroups (rows [b M / nb, (b + 1) M / nb)): forward,
  // every workgroup fills a slot of its own -- the combination treats (rank, b) like a rank, buf is
  // [nranks][nb][n_items][slot]; adjoint, buf is [nb][n_items][d + d*d] and the second half adds the nb parts in order
  int nb;
};

// what == 0: the states sampled by step t (A.xt -> A.states[t+1], d = D, items = groups);
// what == 1: the rewards of all steps (A.rt -> A.rewards, d = 1, items = (step, group))
template <int WHAT>
struct MmxItem {
  int d, t, gi, item, n_items, zrow0;
  const float *src, *z;
  float* out;
  __device__ MmxItem(const RolloutArgs& A, const MmxArgs& X, int t_arg, int bid) {
    if (WHAT == 0) {
      d = A.D; t = t_arg; gi = bid; item = bid; n_items = A.G;
      src = A.xt + ((size_t)t * A.B + (size_t)gi * A.M) * A.D;
      out = A.states + ((size_t)(t + 1) * A.B + (size_t)gi * A.M) * A.D;
      z = pm_zbase(A.zmm, A.D, t, A.Bg, A.flags);
    } else {
      d = 1; t = bid / A.G; gi = bid - t * A.G; item = bid; n_items = A.H * A.G;
      src = A.rt + (size_t)t * A.B + (size_t)gi * A.M;
      out = A.rewards + (size_t)t * A.B + (size_t)gi * A.M;
      z = pm_zbase(A.zrr, 1, t, A.Bg, A.flags);
    }
    zrow0 = pm_zrow0(t, gi * X.span_rows + X.span_off, A.flags);
  }
};

// LDS of the forward's second half: the scratch, then the slots of every (rank, workgroup) of one item -- if that
// fits PM_MMX_LDS_MAX (else the combination reads them from memory)
#define PM_MMX_LDS_MAX (60 * 1024)
__host__ __device__ inline size_t pm_mmx_apply_lds_doubles(int d, int nslots) {
  return pm_mm_scratch_doubles(d) + (size_t)nslots * pm_mmx_slot_doubles(d);
}
// LDS of the kernels: the one-wave scratch of pmbrl_mm.h, then the waves' partial sums
__host__ __device__ inline size_t pm_mmx_lds_doubles(int d, int nw) {
  return pm_mm_scratch_doubles(d) + (size_t)nw * pm_mmx_part_doubles(d);
}

template <int WHAT>
__global__ __launch_bounds__(1024) void pm_mmx_stats_kernel(const RolloutArgs A, const MmxArgs X, int t_arg) {
  extern __shared__ __attribute__((aligned(16))) double mmx_scr[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int b = blockIdx.x % X.nb;
  const MmxItem<WHAT> I(A, X, t_arg, blockIdx.x / X.nb);
  const size_t ss = pm_mmx_slot_doubles(I.d);
  const int w = blockIdx.y;
  double* slot = X.buf + (((size_t)w * X.nb + b) * I.n_items + I.item) * ss;
  const int r_lo = (int)((long long)b * A.M / X.nb), r_hi = (int)((long long)(b + 1) * A.M / X.nb);
  if (w != X.rank) {
    for (int e = threadIdx.x; e < (int)ss; e += blockDim.x) slot[e] = 0.0;
    return;
  }
  pm_mmx_stats(I.src + (size_t)r_lo * I.d, I.d, r_hi - r_lo, I.d, I.z, I.d, I.zrow0 + r_lo, A.Bg, slot, mmx_scr,
               mmx_scr + pm_mm_scratch_doubles(I.d), nw, wid, lane);
}

template <int WHAT>
__global__ __launch_bounds__(1024) void pm_mmx_apply_kernel(const RolloutArgs A, const MmxArgs X, int t_arg) {
  extern __shared__ __attribute__((aligned(16))) double mmx_scr[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const MmxItem<WHAT> I(A, X, t_arg, blockIdx.x);
  const size_t ss = pm_mmx_slot_doubles(I.d);
  const MMScratch q = pm_mm_carve(mmx_scr, I.d);
  // the slots of every (rank, workgroup) in one round trip to memory, all threads; wave 0 combines them from LDS
  // (read one after the other from HBM the combination of 16 slots cost 8 us)
  const int nslots = X.nranks * X.nb;
  double* sl = mmx_scr + pm_mm_scratch_doubles(I.d);
  const bool staged = pm_mmx_apply_lds_doubles(I.d, nslots) * sizeof(double) <= PM_MMX_LDS_MAX;
  if (staged) {
    for (int e = threadIdx.x; e < nslots * (int)ss; e += blockDim.x) {
      const int k = e / (int)ss, o = e - k * (int)ss;
      sl[e] = X.buf[((size_t)k * I.n_items + I.item) * ss + o];
    }
    __syncthreads();
  }
  if (wid == 0) {
    const bool ok = staged ? pm_mmx_factor(sl, ss, nslots, I.d, q, lane)
                           : pm_mmx_factor(X.buf + (size_t)I.item * ss, (size_t)I.n_items * ss, nslots, I.d, q, lane);
    double* fac = X.fac + (size_t)I.item * pm_mm_fac_doubles(I.d);
    for (int e = lane; e < (int)pm_mm_fac_doubles(I.d); e += 64) fac[e] = mmx_scr[e];
    if (!ok && lane == 0) atomicMin(A.status, I.t);
  }
  __syncthreads();
  const bool ins = (A.flags & PMBRL_FLAG_INFER_NS) != 0;
  if (!ins && I.d <= PM_MMX_DF) pm_mmx_apply_rows(A.M, I.d, I.z, I.d, I.zrow0, A.Bg, I.out, I.d, q, threadIdx.x, blockDim.x);
  else pm_mmx_apply(A.M, I.d, I.z, I.d, I.zrow0, A.Bg, I.out, I.d, q, threadIdx.x, blockDim.x, ins ? I.src : nullptr, I.d);
}

// adjoint, first half: g = dL/d(moment-matched rows) of this rank -> its part of the two sums.
// WHAT == 0: g = A.gx_carry (dL/dx_{t+1}); WHAT == 1: g = A.grad_rewards.
template <int WHAT>
__global__ __launch_bounds__(1024) void pm_mmx_bwd_sums_kernel(const RolloutArgs A, const MmxArgs X, int t_arg) {
  extern __shared__ __attribute__((aligned(16))) double mmx_scr[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int b = blockIdx.x % X.nb;
  const MmxItem<WHAT> I(A, X, t_arg, blockIdx.x / X.nb);
  double* sums = X.buf + ((size_t)b * I.n_items + I.item) * pm_mmx_bwd_doubles(I.d);
  const int r_lo = (int)((long long)b * A.M / X.nb), r_hi = (int)((long long)(b + 1) * A.M / X.nb);
  if (A.nvalid && I.t >= *A.nvalid) {   // a step the forward sweep did not complete: nothing to add (the
    for (int e = threadIdx.x; e < (int)pm_mmx_bwd_doubles(I.d); e += blockDim.x) sums[e] = 0.0;   // collective still runs)
    return;
  }
  const double* fac = X.fac + (size_t)I.item * pm_mm_fac_doubles(I.d);
  for (int e = threadIdx.x; e < (int)pm_mm_fac_doubles(I.d); e += blockDim.x) mmx_scr[e] = fac[e];
  __syncthreads();
  const MMScratch q = pm_mm_carve(mmx_scr, I.d);
  const float* g = WHAT == 0 ? A.gx_carry + (size_t)I.gi * A.M * A.D
                             : A.grad_rewards + (size_t)I.t * A.B + (size_t)I.gi * A.M;
  const bool ins = (A.flags & PMBRL_FLAG_INFER_NS) != 0;
  double* part = mmx_scr + pm_mm_scratch_doubles(I.d);
  pm_mmx_bwd_sums(r_hi - r_lo, I.d, I.z, I.d, I.zrow0 + r_lo, A.Bg, g + (size_t)r_lo * I.d, I.d, q, sums, part, nw, wid, lane,
                  ins ? I.src + (size_t)r_lo * I.d : nullptr, I.d);
}

// adjoint, second half: the summed (mbar, Lbar) -> dL/d(rows before moment matching) for this rank's rows.
// WHAT == 0: in place in A.gx_carry; WHAT == 1: A.grad_rewards -> gr_tilde.
template <int WHAT>
__global__ __launch_bounds__(1024) void pm_mmx_bwd_apply_kernel(const RolloutArgs A, const MmxArgs X, int t_arg,
                                                                float* gr_tilde) {
  extern __shared__ __attribute__((aligned(16))) double mmx_scr[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const MmxItem<WHAT> I(A, X, t_arg, blockIdx.x);
  if (A.nvalid && I.t >= *A.nvalid) return;
  const MMScratch q = pm_mm_carve(mmx_scr, I.d);
  const double Mtot = (double)X.span_rows;
  // the nb parts of the two sums: every thread adds the parts of its entries in part order (nb independent loads
  // in flight, one round trip to memory) and leaves the totals in LDS -- d*d + d doubles, which fit the partial-sum
  // area of any launch (>= one wave's d*d + 3d), whatever nb is
  const int nbd = (int)pm_mmx_bwd_doubles(I.d);
  double* pl = mmx_scr + pm_mm_scratch_doubles(I.d);
  for (int e = threadIdx.x; e < nbd; e += blockDim.x) {
    double t = 0.0;
    for (int b = 0; b < X.nb; ++b) t += X.buf[((size_t)b * I.n_items + I.item) * nbd + e];
    pl[e] = t;
  }
  __syncthreads();
  if (wid == 0) {
    const double* fac = X.fac + (size_t)I.item * pm_mm_fac_doubles(I.d);
    for (int e = lane; e < (int)pm_mm_fac_doubles(I.d); e += 64) mmx_scr[e] = fac[e];
    pm_wave_sync();
    auto total = [&](int e) { return pl[e]; };
    for (int e = lane; e < I.d; e += 64) q.mbar[e] = total(e);
    if (A.flags & PMBRL_FLAG_INFER_NS) {
      for (int e = lane; e < I.d * I.d; e += 64) q.Sb[e] = total(I.d + e);   // A = g^T Delta over all ranks
      pm_wave_sync();
      pm_mm_infer_lbar(I.d, q, lane);
    } else {
      for (int e = lane; e < I.d * I.d; e += 64) q.P[e] = total(I.d + e);
    }
    pm_wave_sync();
    pm_mm_bwd_solve(I.d, 1.0 / (Mtot - 1.0), q, lane);
  }
  __syncthreads();
  float* gout = WHAT == 0 ? A.gx_carry + (size_t)I.gi * A.M * A.D
                          : gr_tilde + (size_t)I.t * A.B + (size_t)I.gi * A.M;
  if (I.d <= PM_MMX_DF) pm_mmx_bwd_rows(I.src, I.d, A.M, I.d, 1.0 / Mtot, gout, I.d, q, threadIdx.x, blockDim.x);
  else pm_mm_bwd_rows(I.src, I.d, A.M, I.d, 1.0 / Mtot, gout, I.d, q, threadIdx.x, blockDim.x);
}
