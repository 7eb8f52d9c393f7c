// Moment matching of ONE group by ONE wavefront with the d x d algebra in registers (compile-time d <= 6,
// group of M <= 64 rows in LDS): the form the in-kernel moment matching of the latency-optimised sweeps
// uses when the state width is a compile-time constant.
//
// The general routines (pmbrl_mm.h: pm_mm_fwd / pm_mm_bwd) keep the statistics, the factor and the adjoint's
// triangular solves in an LDS scratch and walk them with wave-level barriers: ~7 k cycles per step and
// direction at d = 4, M = 25 -- all latency, on one wave while seven wait.  Here
//   * the group sums are ONE Gram tile on the fp64 matrix core (pm_mm_gram_rows / pm_mm_gram_h_rows),
//   * the handful of entries that matter are broadcast with v_readlane into wave-uniform doubles,
//   * means, covariance, Cholesky factor, the adjoint's Phi / triangular solves / symmetrisation are
//     unrolled scalar code on those doubles (every lane computes the same d x d numbers: no LDS, no barrier),
//   * each lane then finishes its own row.
// Same mathematics and the same pivot rule as pm_mm_factor (utils/rollout.py:20-29).
#pragma once
#include "pmbrl_mm.h"

template <int DD>
struct MMW {
  double mean[DD], zm[DD], zi[DD], invd[DD];
  double L[DD][DD];   // lower triangle used
};

// entry (row, col) of a Gram tile: lane ((row & 3) << 4) | col, register row >> 2
#define PM_G(G, row, col) pm_rl64((G)[(row) >> 2], ((((row) & 3) << 4) | (col)))

// means (relative to the reference row), z standardisation, covariance, Cholesky factor from the Gram tile of
// X = [s - ref | 1 | z].  Returns false on a lost pivot (same rule as pm_mm_factor).
// the z standardisation alone (the columns of z in the tile)
template <int DD>
__device__ __forceinline__ void pm_mmw_zstats(const pm_f64x4& G, int M, MMW<DD>& q) {
  const double dM = (double)M, inv_m = 1.0 / dM, inv_m1 = 1.0 / (double)(M - 1);
#pragma unroll
  for (int j = 0; j < DD; ++j) {
    q.zm[j] = PM_G(G, DD, DD + 1 + j) * inv_m;
    const double szz = PM_G(G, DD + 1 + j, DD + 1 + j);
    q.zi[j] = pm_rsqrt((szz - dM * q.zm[j] * q.zm[j]) * inv_m1);
  }
}
template <int DD, bool ZS = true>      // ZS = false: q.zm / q.zi are already there
__device__ __forceinline__ bool pm_mmw_factor(const pm_f64x4& G, int M, MMW<DD>& q) {
  const double dM = (double)M, inv_m = 1.0 / dM, inv_m1 = 1.0 / (double)(M - 1);
  double A[DD][DD];
  if constexpr (ZS) pm_mmw_zstats<DD>(G, M, q);
#pragma unroll
  for (int j = 0; j < DD; ++j) q.mean[j] = PM_G(G, DD, j) * inv_m;
#pragma unroll
  for (int i = 0; i < DD; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
      A[i][j] = (PM_G(G, i, j) - dM * q.mean[i] * q.mean[j]) * inv_m1 + (i == j ? 1e-12 : 0.0);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DD; ++k) {
    const double d0 = A[k][k];         // original diagonal entry
    double piv = d0;
#pragma unroll
    for (int c = 0; c < k; ++c) piv -= q.L[k][c] * q.L[k][c];
    if (!(piv > 6e-8 * d0)) {
      ok = false;
      piv = 1.0;
    }
    const double rs = pm_rsqrt(piv);
    q.L[k][k] = piv * rs;
    q.invd[k] = rs;
#pragma unroll
    for (int i = k + 1; i < DD; ++i) {
      double a = A[i][k];
#pragma unroll
      for (int c = 0; c < k; ++c) a -= q.L[i][c] * q.L[k][c];
      q.L[i][k] = a * rs;
    }
  }
  return ok;
}

// forward: out = m + zhat L^T for the group's M rows (lane r = row r); s, z, out in LDS
// fac_out (optional): the statistics and the factor in the layout of pm_mm_fac_doubles (pmbrl_mm.h), for the
// adjoint sweep's pm_mm_bwd
// xf / ref_in / zst: a group split over workgroups -- xf(G) delivers the s entries of the Gram tile (sums over the
// rows of ALL parts relative to the reference point: lane c < DD of *ref_in holds its column c), zst the z
// standardisation [zm | zi] (LDS); s is not read here then
struct PmNoGramXch {
  __device__ __forceinline__ bool operator()(pm_f64x4&) const { return true; }
};
template <int DD, class XF = PmNoGramXch>
__device__ __forceinline__ bool pm_mm_fwd_w(const float* s, int s_ld, int M, const float* z, int z_ld, float* out,
                                            int out_ld, int lane, double* fac_out = nullptr, XF xf = XF{},
                                            const double* ref_in = nullptr, float* mean_out = nullptr,
                                            const double* zst = nullptr) {
  double ref = 0.0;
  pm_f64x4 G;
  MMW<DD> q;
  bool ok;
  if constexpr (std::is_same<XF, PmNoGramXch>::value) {
    G = pm_mm_gram_rows<DD, false>(s, s_ld, z, z_ld, 0, 0, 0, M, lane, &ref, ref_in);
    ok = pm_mmw_factor<DD>(G, M, q);
  } else {
    ref = *ref_in;
    G = pm_f64x4{0.0, 0.0, 0.0, 0.0};
    const bool xok = xf(G);
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      q.zm[j] = zst[j];
      q.zi[j] = zst[DD + j];
    }
    ok = pm_mmw_factor<DD, false>(G, M, q) && xok;
  }
  // the reference row (what the Gram subtracted): lane j < DD holds column j
  double refj[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) refj[j] = pm_rl64(ref, j);
  if (fac_out && lane == 0) {
    // [mean | zmean | zistd | (mbar) | invd | L row-major]
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      fac_out[j] = q.mean[j] + refj[j];
      fac_out[DD + j] = q.zm[j];
      fac_out[2 * DD + j] = q.zi[j];
      fac_out[4 * DD + j] = q.invd[j];
#pragma unroll
      for (int c = 0; c < DD; ++c) fac_out[5 * DD + j * DD + c] = c <= j ? q.L[j][c] : 0.0;
    }
  }
  if (mean_out && lane == 0) {      // the next step's reference point
#pragma unroll
    for (int j = 0; j < DD; ++j) mean_out[j] = (float)(q.mean[j] + refj[j]);
  }
  if (lane < M) {
    double zh[DD];
#pragma unroll
    for (int c = 0; c < DD; ++c) zh[c] = ((double)z[lane * z_ld + c] - q.zm[c]) * q.zi[c];
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      double acc = q.mean[j] + refj[j];
#pragma unroll
      for (int c = 0; c <= j; ++c) acc += zh[c] * q.L[j][c];
      out[lane * out_ld + j] = (float)acc;
    }
  }
  pm_wave_sync();
  return ok;
}

// adjoint: g = dL/d out [M][DD] -> gout = dL/d s (may alias g); s, z, g, gout in LDS
template <int DD>
__device__ __forceinline__ void pm_mm_bwd_w(const float* s, int s_ld, int M, const float* z, int z_ld, const float* g,
                                            int g_ld, float* gout, int gout_ld, int lane) {
  pm_f64x4 G, H;
  double ref = 0.0;
  pm_mm_gram_h_rows<DD, false>(s, s_ld, z, z_ld, 0, 0, g, g_ld, 0, M, lane, G, H, &ref);
  MMW<DD> q;
  (void)pm_mmw_factor<DD>(G, M, q);
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  // H = g^T [z | 1]: mbar = sum_r g, Lbar = tril(g^T zhat)
  double mb[DD], Lb[DD][DD];
#pragma unroll
  for (int i = 0; i < DD; ++i) mb[i] = PM_G(H, i, DD);
#pragma unroll
  for (int i = 0; i < DD; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) Lb[i][j] = (PM_G(H, i, j) - q.zm[j] * mb[i]) * q.zi[j];
  // Phi = tril(L^T Lbar), diagonal halved
  double X[DD][DD];
#pragma unroll
  for (int i = 0; i < DD; ++i)
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      double acc = 0.0;
      if (j <= i) {
#pragma unroll
        for (int c = i; c < DD; ++c) acc += q.L[c][i] * Lb[c][j];
        if (i == j) acc *= 0.5;
      }
      X[i][j] = acc;
    }
  // X <- Phi L^-1 (row i solves x L = phi_i)
#pragma unroll
  for (int i = 0; i < DD; ++i)
#pragma unroll
    for (int j = DD - 1; j >= 0; --j) {
      double a = X[i][j];
#pragma unroll
      for (int c = j + 1; c < DD; ++c) a -= X[i][c] * q.L[c][j];
      X[i][j] = a * q.invd[j];
    }
  // Sbar <- L^-T X (column j solves L^T y = x_j)
#pragma unroll
  for (int j = 0; j < DD; ++j)
#pragma unroll
    for (int i = DD - 1; i >= 0; --i) {
      double a = X[i][j];
#pragma unroll
      for (int c = i + 1; c < DD; ++c) a -= q.L[c][i] * X[c][j];
      X[i][j] = a * q.invd[i];
    }
  // P = (Sbar + Sbar^T) / (M - 1);  sbar[r] = (s_r - mean) P + mbar / M
  double refj[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) refj[j] = pm_rl64(ref, j);
  if (lane < M) {
    double dl[DD];
#pragma unroll
    for (int c = 0; c < DD; ++c) dl[c] = (double)s[lane * s_ld + c] - (q.mean[c] + refj[c]);
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      double acc = mb[j] * inv_m;
#pragma unroll
      for (int c = 0; c < DD; ++c) acc += dl[c] * ((X[c][j] + X[j][c]) * inv_m1);
      gout[lane * gout_ld + j] = (float)acc;
    }
  }
  pm_wave_sync();
}
