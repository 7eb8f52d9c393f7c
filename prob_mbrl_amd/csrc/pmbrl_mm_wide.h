// Moment matching of WIDE states (6 < D <= 32: beyond the compile-time-width routines of pmbrl_mm.h) between the
// per-step launches of the sweeps (mm_mode 2): one workgroup of PM_MMW_NT threads per group, the group's rows, noise
// rows and incoming gradient staged in LDS once, every d x d product spread over all threads, and the only serial
// parts -- the Cholesky factorisation and L^-1 -- done once, in the forward kernel, which hands (statistics, L, L^-1)
// to the adjoint kernel of the same step through HBM.  Mathematics: utils/rollout.py:20-29 (mm_resample_) and its
// adjoint, exactly as pm_mm_fwd / pm_mm_bwd_solve / pm_mm_bwd_rows of pmbrl_mm.h (all of it in fp64); the general
// one-wave routines those are take 40 (forward) / 60 (adjoint) ms of a 112 ms iteration at BASELINE.md's C5 shape
// (D = 32, 256 groups of 64 rows, H = 100).
#pragma once
#include "pmbrl_mm.h"

#define PM_MMW_NT 512
#define PM_MMW_PARTS 16         // row parts of the column sums (PM_MMW_NT / 32 columns)
#define PM_MMW_DMAX 32

// what the forward leaves for the adjoint: [mean d | zmean d | zistd d | (mbar d) | invd d | L d x (d+1) | L^-1 d x (d+1)]
// (rows of L and L^-1 padded to d + 1 doubles: a column of either is read by 32 lanes at once, and with a row
//  stride of 32 doubles those reads would all land in one LDS bank)
__host__ __device__ inline size_t pm_mmw_fac_doubles(int d) { return (size_t)5 * d + (size_t)2 * d * (d + 1); }
// LDS: the fac block, work matrices (adjoint: P, Sb, T), the column-sum parts, then the staged fp32 rows
__host__ __device__ inline size_t pm_mmw_lds_doubles(int d, bool bwd) {
  return pm_mmw_fac_doubles(d) + (size_t)(bwd ? 3 : 0) * d * d + (size_t)3 * PM_MMW_PARTS * 32;   // (forward: the covariance is built in L's place)
}
__host__ __device__ inline size_t pm_mmw_lds_bytes(int M, int d, bool bwd) {
  return pm_mmw_lds_doubles(d, bwd) * sizeof(double) + (size_t)(bwd ? 3 : 2) * M * d * sizeof(float);
}
__host__ inline bool pm_mmw_ok(int M, int d, unsigned flags) {
  return d > 6 && d <= 32 && !(flags & PMBRL_FLAG_INFER_NS) && pm_mmw_lds_bytes(M, d, true) <= (size_t)160 * 1024;
}

struct MmwLds {
  double *fac, *mean, *zmean, *zistd, *mbar, *invd, *L, *Li, *P, *Sb, *T, *part;
  float *X, *Z, *G;
};
__device__ __forceinline__ MmwLds pm_mmw_carve(double* base, int M, int d, bool bwd) {
  MmwLds q;
  q.fac = base;
  q.mean = base;
  q.zmean = q.mean + d;
  q.zistd = q.zmean + d;
  q.mbar = q.zistd + d;
  q.invd = q.mbar + d;
  q.L = q.invd + d;
  q.Li = q.L + d * (d + 1);
  double* w = q.Li + d * (d + 1);
  q.P = w;
  q.Sb = q.P + (bwd ? d * d : 0);
  q.T = q.Sb + (bwd ? d * d : 0);
  q.part = q.T + (bwd ? d * d : 0);
  q.X = reinterpret_cast<float*>(q.part + 3 * PM_MMW_PARTS * 32);
  q.Z = q.X + (size_t)M * d;
  q.G = q.Z + (size_t)M * d;
  return q;
}

__device__ __forceinline__ double pm_readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Cholesky factor of the d x d matrix at Sp (row stride ld, lower triangle used, upper initialised), in REGISTERS of
// one wave: lane i keeps row i (fully unrolled, so every index is a register name), the pivot and the column entries
// a row needs from other rows travel by v_readlane -- no LDS round trip on the d (d + 1) / 2 dependent steps.  Same
// pivot rule as pm_mm_chol.  Leaves L (lower, zeros above) at Lout (row stride ld) and 1 / diag(L) at invd.
// (Measured at d = 32, cycles of the one wave: this form 36 k -- the 30 KB of straight-line code run once per launch,
//  at the speed of cold instruction fetch; a rolled left-looking form on LDS with pipelined dot products 40 k; the
//  right-looking LDS form of pm_mm_chol 80 k.)
__device__ __forceinline__ bool pm_mmw_chol_regs(const double* Sp, int d, int ld, const double* diag0, double* Lout,
                                                 double* invd, int lane) {
  double a[PM_MMW_DMAX];
  const int row = lane < d ? lane : 0;
#pragma unroll
  for (int c = 0; c < PM_MMW_DMAX; ++c) a[c] = (c < d) ? Sp[row * ld + c] : 0.0;
  pm_wave_sync();
  bool ok = true;
#pragma unroll
  for (int k = 0; k < PM_MMW_DMAX; ++k) {
    if (k < d) {
      double piv = pm_readlane_f64(a[k], k);
      if (!(piv > 6e-8 * diag0[k])) {
        ok = false;
        piv = 1.0;
      }
      const double rs = pm_rsqrt(piv);
      a[k] = (lane == k) ? piv * rs : a[k] * rs;      // L[i][k], i >= k (rows above k: unused upper triangle)
      if (lane == 0) invd[k] = rs;
#pragma unroll
      for (int j = k + 1; j < PM_MMW_DMAX; ++j)
        if (j < d) a[j] -= a[k] * pm_readlane_f64(a[k], j);    // A[i][j] -= L[i][k] L[j][k]
    }
  }
  if (lane < d) {
#pragma unroll
    for (int c = 0; c < PM_MMW_DMAX; ++c)
      if (c < d) Lout[lane * ld + c] = (c <= lane) ? a[c] : 0.0;
  }
  return ok;
}

// L^-1 (lower) of the factor at L (row stride ld), column jc on lane jc with its entries in registers; the entries of
// L arrive as LDS broadcast reads whose addresses depend on nothing computed here (the compiler issues them ahead).
// (22 k cycles at d = 32; a rolled form that reads its own column back from LDS: 40 k.)
__device__ __forceinline__ void pm_mmw_linv(const double* L, const double* invd, int d, int ld, double* Li, int lane) {
  double y[PM_MMW_DMAX];
  const int jc = lane;
#pragma unroll
  for (int i = 0; i < PM_MMW_DMAX; ++i) {
    y[i] = 0.0;
    if (i < d) {
      double acc = (i == jc) ? 1.0 : 0.0;
#pragma unroll
      for (int c = 0; c < i; ++c) acc -= L[i * ld + c] * y[c];       // (y[c] = 0 above the diagonal)
      y[i] = (i >= jc) ? acc * invd[i] : 0.0;
    }
  }
  if (jc < d) {
#pragma unroll
    for (int i = 0; i < PM_MMW_DMAX; ++i)
      if (i < d) Li[i * ld + jc] = y[i];
  }
}

// forward: rows s[M][d] (pre-moment-matching samples of step t) -> out[M][d]; factor block -> fac_out
__device__ __forceinline__ bool pm_mmw_fwd(const float* __restrict__ s, int M, int d, const float* __restrict__ z,
                                           int zrow0, int Bg, float* __restrict__ out, double* __restrict__ fac_out,
                                           double* lds, long long* prof = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const MmwLds q = pm_mmw_carve(lds, M, d, false);
  __shared__ int ok_s;
  if (tid == 0) ok_s = 1;
#define PM_MMW_MARK(slot) do { if (prof && tid == 0) prof[slot] = (long long)__builtin_readcyclecounter(); } while (0)
  PM_MMW_MARK(24);
  for (int e = tid; e < M * d; e += PM_MMW_NT) {
    const int r = e / d, j = e - r * d;
    q.X[e] = s[e];
    q.Z[e] = z[(size_t)pm_zidx(zrow0, r, Bg) * d + j];
  }
  __syncthreads();
  PM_MMW_MARK(25);
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  // column sums in PM_MMW_PARTS row parts, added in part order
  {
    const int j = tid & 31, part = tid >> 5;
    double m = 0.0, zm = 0.0, zz = 0.0;
    if (j < d)
      for (int r = part; r < M; r += PM_MMW_PARTS) {
        m += (double)q.X[r * d + j];
        const double zv = (double)q.Z[r * d + j];
        zm += zv;
        zz += zv * zv;
      }
    q.part[(0 * PM_MMW_PARTS + part) * 32 + j] = m;
    q.part[(1 * PM_MMW_PARTS + part) * 32 + j] = zm;
    q.part[(2 * PM_MMW_PARTS + part) * 32 + j] = zz;
  }
  __syncthreads();
  if (tid < d) {
    double m = 0.0, zm = 0.0, zz = 0.0;
    for (int p = 0; p < PM_MMW_PARTS; ++p) {
      m += q.part[(0 * PM_MMW_PARTS + p) * 32 + tid];
      zm += q.part[(1 * PM_MMW_PARTS + p) * 32 + tid];
      zz += q.part[(2 * PM_MMW_PARTS + p) * 32 + tid];
    }
    m *= inv_m;
    zm *= inv_m;
    q.mean[tid] = m;
    q.zmean[tid] = zm;
    q.zistd[tid] = pm_rsqrt((zz - (double)M * zm * zm) * inv_m1);
  }
  __syncthreads();
  PM_MMW_MARK(26);
  // covariance (lower triangle) + jitter, in L's place (padded rows); its diagonal kept for the pivot test
  const int ld = d + 1;
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
    if (j <= i) {
      const double mi = q.mean[i], mj = q.mean[j];
#pragma unroll 4
      for (int r = 0; r < M; ++r) acc += ((double)q.X[r * d + i] - mi) * ((double)q.X[r * d + j] - mj);
      acc = acc * inv_m1 + (i == j ? 1e-12 : 0.0);
      if (i == j) q.mbar[i] = acc;
    }
    q.L[i * ld + j] = acc;
  }
  __syncthreads();
  PM_MMW_MARK(27);
  if (wid == 0) {
    const bool ok = pm_mmw_chol_regs(q.L, d, ld, q.mbar, q.L, q.invd, lane);
    if (!ok && lane == 0) ok_s = 0;
  }
  __syncthreads();
  PM_MMW_MARK(28);
  if (wid == 0) {
    pm_mmw_linv(q.L, q.invd, d, ld, q.Li, lane);
    PM_MMW_MARK(29);
  } else {
    // out = mean + zhat L^T on the other waves meanwhile
    for (int e = tid - 64; e < M * d; e += PM_MMW_NT - 64) {
      const int r = e / d, j = e - r * d;
      double acc = q.mean[j];
#pragma unroll 4
      for (int c = 0; c <= j; ++c) acc += ((double)q.Z[r * d + c] - q.zmean[c]) * q.zistd[c] * q.L[j * ld + c];
      out[e] = (float)acc;
    }
  }
  __syncthreads();
  PM_MMW_MARK(30);
  for (int e = tid; e < (int)pm_mmw_fac_doubles(d); e += PM_MMW_NT) fac_out[e] = q.fac[e];
  PM_MMW_MARK(31);
#undef PM_MMW_MARK
  return ok_s != 0;
}

// adjoint: g[M][d] = dL/d(out rows) -> gout[M][d] = dL/d(s rows); gout may be g
__device__ __forceinline__ void pm_mmw_bwd(const float* __restrict__ s, int M, int d, const float* __restrict__ z,
                                           int zrow0, int Bg, const float* g, float* gout,
                                           const double* __restrict__ fac, double* lds) {
  const int tid = threadIdx.x;
  const MmwLds q = pm_mmw_carve(lds, M, d, true);
  for (int e = tid; e < (int)pm_mmw_fac_doubles(d); e += PM_MMW_NT) q.fac[e] = fac[e];
  for (int e = tid; e < M * d; e += PM_MMW_NT) {
    const int r = e / d, j = e - r * d;
    q.X[e] = s[e];
    q.Z[e] = z[(size_t)pm_zidx(zrow0, r, Bg) * d + j];
    q.G[e] = g[e];
  }
  __syncthreads();
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  const int ld = d + 1;
  {
    const int j = tid & 31, part = tid >> 5;
    double m = 0.0;
    if (j < d)
      for (int r = part; r < M; r += PM_MMW_PARTS) m += (double)q.G[r * d + j];
    q.part[part * 32 + j] = m;
  }
  // Lbar = tril(g^T zhat) -> P
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
    if (j <= i) {
      const double zm = q.zmean[j], zs = q.zistd[j];
#pragma unroll 4
      for (int r = 0; r < M; ++r) acc += (double)q.G[r * d + i] * (((double)q.Z[r * d + j] - zm) * zs);
    }
    q.P[e] = acc;
  }
  __syncthreads();
  if (tid < d) {
    double m = 0.0;
    for (int p = 0; p < PM_MMW_PARTS; ++p) m += q.part[p * 32 + tid];
    q.mbar[tid] = m;
  }
  // Phi = tril(L^T Lbar), diagonal halved -> Sb
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
    if (j <= i) {
#pragma unroll 4
      for (int c = i; c < d; ++c) acc += q.L[c * ld + i] * q.P[c * d + j];
      if (i == j) acc *= 0.5;
    }
    q.Sb[e] = acc;
  }
  __syncthreads();
  // X = Phi L^-1 (lower x lower) -> T
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
#pragma unroll 4
    for (int c = j; c <= i; ++c) acc += q.Sb[i * d + c] * q.Li[c * ld + j];
    q.T[e] = acc;
  }
  __syncthreads();
  // Sbar = L^-T X -> Sb
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
#pragma unroll 4
    for (int c = (i > j ? i : j); c < d; ++c) acc += q.Li[c * ld + i] * q.T[c * d + j];
    q.Sb[e] = acc;
  }
  __syncthreads();
  // P = (Sbar + Sbar^T) / (M - 1)
  for (int e = tid; e < d * d; e += PM_MMW_NT) {
    const int i = e / d, j = e - i * d;
    q.P[e] = (q.Sb[i * d + j] + q.Sb[j * d + i]) * inv_m1;
  }
  __syncthreads();
  // rows: sbar[r][j] = mbar[j] / M + sum_c (s[r][c] - mean[c]) P[c][j]
  for (int e = tid; e < M * d; e += PM_MMW_NT) {
    const int r = e / d, j = e - r * d;
    double acc = q.mbar[j] * inv_m;
#pragma unroll 4
    for (int c = 0; c < d; ++c) acc += ((double)q.X[r * d + c] - q.mean[c]) * q.P[c * d + j];
    gout[e] = (float)acc;
  }
}
