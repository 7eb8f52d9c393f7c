// Moment matching inside the register-resident sweeps (pmbrl_reg.h), round 5: utils/rollout.py:20-29 and its adjoint
//  for a group that is split over the `parts` consecutive workgroups of <= 16 rows that hold it.
//
// The state of a row lives in the lanes of every wave as x[s] = dimension 2 g + s (g = lane >> 4, row = lane & 15).  ONE
// wave (wave 0) runs the chain while the other three wait at the step's fifth barrier; what it costs is the length of
// the dependent chain, so the chain is built from the matrix core wherever the algebra allows:
//
//   forward   Gram tile [x~ - c | 1]^T [x~ - c | 1] over the part's rows: operands gathered from the state registers
//             with ds_bpermute (no LDS round trip), 3-4 fp64 16x16x4 MFMAs;  sums exchanged as data-tagged granules
//             (pmbrl_xch.h);  mean, covariance, Cholesky factor as wave-uniform scalar code (pmbrl_mm_w.h: inherently
//             sequential);  the part's rows of m + zhat L^T lane-local;  L and L^-1 stashed for the adjoint.
//   adjoint   H = g^T [zhat | 1] the same way;  while the sums travel: Y1 = L^-1 Delta^T (what does not depend on
//             them);  then the whole d x d algebra as MFMAs whose OUTPUT layout (column in lane & 15, row
//             (lane >> 4) + 4 r) is at once the B operand of the next left-multiplication and, for a symmetric
//             matrix, its A operand:
//                 Phi   = L^T tril(H)          A = L (as stored: lane (a, k) holds L[k][a]),  B = tril(H)
//                 Phi^T = tril(H)^T L          A = tril(H) (its own output registers),        B = L (the same registers)
//                 Psi   = low(Phi) + up(Phi^T)  (diagonal halved in both: Psi = Phi~ + Phi~^T, symmetric)
//                 Y2    = Psi Y1 ,  Y3 = L^-T Y2 ,  dL/dx~^T = Y3 / (M - 1) + mbar / M
//             -- three dependent MFMA levels (one MFMA each for d <= 4, two for d <= 8) instead of two triangular
//             solves in scalar code: ~0.3 k cycles against 2-5 k in the latency-optimised family's one-wave chain.
//
// What wave 0 needs from memory (its row of the noise, the group's noise standardisation, the forward sweep's factor, the
// part's pre-mm rows) is requested by plain per-lane loads in exactly the operand layout it is used in, at the START
// of the chain -- the loads land while the sums travel (~1.6 k cycles), and nothing of the moment matching is live
// across the step's GEMM phases (whose register budget is spoken for); only the adjoint's noise operand, which the
// chain needs at once, is requested a step ahead (four registers).
#pragma once
#include "pmbrl_mm_w.h"
#include "pmbrl_xch.h"

struct RegMM {
  int on;                        // moment matching of states inside the sweep
  int M, parts, rpw, groups;     // rows per group, workgroups per group, rows per part (the last part: what is left), groups
  int Bg, row_off, flags;        // cyclic noise buffer: global rows, this device's first row; PMBRL_FLAG_ZMM_PER_STEP
  const float* zmm;              // noise rows [Bg][D] (or [H][Bg][D])
  const double* ztab;            // [H][groups][zm (D) | zi (D)]: pm_mm_ztable_kernel
  double* mmfac;                 // [H][groups][5 D + D D]: mean | zm | zi | - | 1 / diag L | L (pmbrl_mm.h, pm_mm_carve)
  double* linv;                  // [H][groups][D D]: L^-1
  float* xt;                     // [H][B][D]: the sampled (pre-mm) states
  unsigned long long* xch;       // granules of the statistics exchange (parts > 1)
  double inv_m, inv_m1;          // 1 / M, 1 / (M - 1) (from the host: an fp64 division is thirty instructions)
};

// mean (relative to the reference point), covariance and its Cholesky factor from the Gram tile of [s - ref | 1]:
// pm_mmw_factor (pmbrl_mm_w.h) with the reciprocals handed in.  Same pivot rule (a pivot that has shed more than fp32's
// precision counts as lost: the reference factors in fp32 and raises, utils/rollout.py:154-157).
template <int DD>
__device__ __forceinline__ bool pr_mm_factor(const pm_f64x4& G, double dM, double inv_m, double inv_m1, MMW<DD>& q) {
  double A[DD][DD];
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
    const double d0 = A[k][k];
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

// value `dim` of row `rr`'s state from the lanes' registers: it sits in lane rr + 16 (dim >> 1), slot dim & 1
__device__ __forceinline__ float pr_mm_gather(const float (&v)[2], int rr, int dim) {
  const int src = (rr + 16 * ((dim >> 1) & 3)) * 4;
  const int a = __builtin_amdgcn_ds_bpermute(src, __float_as_int(v[0]));
  const int b = __builtin_amdgcn_ds_bpermute(src, __float_as_int(v[1]));
  return __int_as_float((dim & 1) ? b : a);
}

// cyclic noise row of local row r of the group that starts at device row g0 (utils/rollout.py:53-59)
__device__ __forceinline__ const float* pr_mm_zrow(const RegMM& Q, int D, int t, int g0, int r) {
  const float* zb = (Q.flags & PMBRL_FLAG_ZMM_PER_STEP) ? Q.zmm + (size_t)t * Q.Bg * D : Q.zmm;
  const int z0 = (Q.flags & PMBRL_FLAG_ZMM_PER_STEP) ? Q.row_off + g0 : t + Q.row_off + g0;
  return zb + (size_t)pm_zidx(z0, r, Q.Bg) * D;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// xn: the sampled dimensions 2 g, 2 g + 1 of row lane & 15 (finite everywhere).  refl: column lane & 15 of the reference
// point every part of the group subtracts (in / out: the next step's is this step's mean).  lrow: this lane's row inside
// the group (clamped to a row of the part).  Returns false on a lost pivot or a partner that never arrived.
// xout: dimensions 2 g, 2 g + 1 of the moment-matched row.
// (What the tail needs from memory -- the lane's noise row, the group's noise standardisation -- is requested HERE and
//  lands while the sums travel: nothing of the moment matching is live across the step's GEMM phases, whose register
//  budget has no room for it.)
template <int DD>
__device__ __forceinline__ bool pr_mm_fwd_chain(const RegMM& Q, int t, unsigned kstep, int gi, int g0, int lrow, int me,
                                                int first_wg, int nvalid, int lane, const float (&xn)[2], double& refl,
                                                float (&xout)[2]) {
  static_assert(DD >= 2 && DD <= 6, "state widths 2..6");
  const int c = lane & 15, k = lane >> 4;
  float z[DD];
  double zt[2 * DD];      // zm | zi of the group at this step (the same in every lane)
  {
    const float* zr = pr_mm_zrow(Q, DD, t, g0, lrow);
#pragma unroll
    for (int cc = 0; cc < DD; ++cc) z[cc] = zr[cc];
    const double* ztp = Q.ztab + ((size_t)t * Q.groups + gi) * 2 * DD;
#pragma unroll
    for (int cc = 0; cc < 2 * DD; ++cc) zt[cc] = ztp[cc];
  }
  pm_f64x4 G0 = {0.0, 0.0, 0.0, 0.0}, G1 = {0.0, 0.0, 0.0, 0.0};
  {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = pr_mm_gather(xn, 4 * q + k, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double x = c < DD ? (double)v[q] - refl : (c == DD ? 1.0 : 0.0);
      x = 4 * q + k < nvalid ? x : 0.0;
      if (4 * q < nvalid) {      // (uniform)
        if (q & 1) G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G1, 0, 0, 0);
        else G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G0, 0, 0, 0);
      }
    }
  }
  pm_f64x4 G = G0 + G1;
  bool ok = true;
  if (Q.parts > 1) {
    double v[2] = {G[0], G[1]};
    pm_xch_put<2>(Q.xch, first_wg, me, kstep, v, lane);
    ok = pm_xch_get<2>(Q.xch, first_wg, Q.parts, me, kstep, v, lane);
    G[0] = v[0];
    G[1] = v[1];
  }
  MMW<DD> q;
  ok = pr_mm_factor<DD>(G, (double)Q.M, Q.inv_m, Q.inv_m1, q) && ok;
  double mean[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) mean[j] = q.mean[j] + pm_rl64(refl, j);
  // this lane's row: m + zhat L^T, all d entries (d (d + 1) / 2 products), then the two this lane group keeps
  double zh[DD], acc[DD + 2];
#pragma unroll
  for (int cc = 0; cc < DD; ++cc) zh[cc] = ((double)z[cc] - zt[cc]) * zt[DD + cc];
#pragma unroll
  for (int j = 0; j < DD; ++j) {
    double a = mean[j];
#pragma unroll
    for (int cc = 0; cc <= j; ++cc) a += zh[cc] * q.L[j][cc];
    acc[j] = a;
  }
  acc[DD] = acc[DD + 1] = 0.0;
  double o0 = acc[0], o1 = acc[1];
  if constexpr (DD > 2) {
    o0 = k == 1 ? acc[2] : o0;
    o1 = k == 1 ? acc[3] : o1;
  }
  if constexpr (DD > 4) {
    o0 = k == 2 ? acc[4] : o0;
    o1 = k == 2 ? acc[5] : o1;
  }
  xout[0] = (float)o0;
  xout[1] = (float)o1;
  // the next step's reference point (every part computes the same bits)
  {
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < DD; ++j) r = c == j ? (double)(float)mean[j] : r;
    refl = r;
  }
  // the factor for the adjoint sweep: mean | zm | zi | - | 1 / diag L | L row-major (what the latency-optimised family's
  // adjoint reads as well), and L^-1
  if (me == 0) {
    double Li[DD][DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) {
      Li[i][i] = q.invd[i];
#pragma unroll
      for (int j = 0; j < i; ++j) {
        double a = 0.0;
#pragma unroll
        for (int cc = j; cc < i; ++cc) a += q.L[i][cc] * Li[cc][j];
        Li[i][j] = -a * q.invd[i];
      }
    }
    if (lane == 0) {
      double* fac = Q.mmfac + ((size_t)t * Q.groups + gi) * pm_mm_fac_doubles(DD);
      double* li = Q.linv + ((size_t)t * Q.groups + gi) * (DD * DD);
#pragma unroll
      for (int j = 0; j < DD; ++j) {
        fac[j] = mean[j];
        fac[DD + j] = zt[j];
        fac[2 * DD + j] = zt[DD + j];
        fac[4 * DD + j] = q.invd[j];
#pragma unroll
        for (int cc = 0; cc < DD; ++cc) {
          fac[5 * DD + j * DD + cc] = cc <= j ? q.L[j][cc] : 0.0;
          li[j * DD + cc] = cc <= j ? Li[j][cc] : 0.0;
        }
      }
    }
  }
  return ok;
}

// ---------------------------------------------------------------------------
// adjoint
// ---------------------------------------------------------------------------
// the noise of rows 4 q + k, column c (lane: c = lane & 15, k = lane >> 4): the one input the chain needs at its very
// start -- requested a step ahead, four registers across the step
template <int DD>
__device__ __forceinline__ void pr_mm_bwd_fetch(const RegMM& Q, int t, int g0, int row0, int nvalid, int lane, float (&z)[4]) {
  const int c = lane & 15, k = lane >> 4, cc = c < DD ? c : DD - 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = 4 * q + k < nvalid ? 4 * q + k : 0;
    z[q] = pr_mm_zrow(Q, DD, t, g0, row0 - g0 + rr)[cc];
  }
}

// gx: dL/dx_{t+1}, dimensions 2 g, 2 g + 1 of row lane & 15 (zero in rows past the part).  out[rr]: dL/dx~ of dimension
// (lane >> 4) + 4 rr of row lane & 15.  Everything else the chain reads (the forward sweep's factor, the part's pre-mm
// rows) is requested at its start, in the operand layout it is used in, and lands while the sums travel.
template <int DD>
__device__ __forceinline__ bool pr_mm_bwd_chain(const RegMM& Q, int B, int t, unsigned kstep, int gi, int row0, int me,
                                                int first_wg, int nvalid, int lane, const float (&gx)[2], const float (&z)[4],
                                                float (&out)[(DD + 3) / 4]) {
  constexpr int NK = (DD + 3) / 4;
  const int c = lane & 15, k = lane >> 4, cc = c < DD ? c : DD - 1;
  // lane (c, k):  zm / zi of column c;  LA = L[4 kk + k][c];  LiA = L^-1[c][4 kk + k];  LiT = L^-1[4 kk + k][c];
  // xt = pre-mm sample of row c, dimension 4 kk + k;  mean of dimension 4 kk + k
  double zm, zi, LA[NK], LiA[NK], LiT[NK], mean[NK];
  float xt[NK];
  {
    const double* fac = Q.mmfac + ((size_t)t * Q.groups + gi) * pm_mm_fac_doubles(DD);
    const double* li = Q.linv + ((size_t)t * Q.groups + gi) * (DD * DD);
    zm = fac[DD + cc];
    zi = fac[2 * DD + cc];
    const float* xr = Q.xt + ((size_t)t * B + row0 + (c < nvalid ? c : 0)) * DD;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int i = 4 * kk + k, ii = i < DD ? i : DD - 1;
      LA[kk] = fac[5 * DD + ii * DD + cc];
      LiA[kk] = li[cc * DD + ii];
      LiT[kk] = li[ii * DD + cc];
      xt[kk] = xr[ii];
      mean[kk] = fac[ii];
    }
  }
  // H = g^T [z | 1] with the RAW noise (standardised behind the exchange: the sums are linear in z)
  pm_f64x4 H0 = {0.0, 0.0, 0.0, 0.0}, H1 = {0.0, 0.0, 0.0, 0.0};
  {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = pr_mm_gather(gx, 4 * q + k, c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool rok = 4 * q + k < nvalid;
      const double a = (c < DD && rok) ? (double)v[q] : 0.0;
      double b = c < DD ? (double)z[q] : (c == DD ? 1.0 : 0.0);
      b = rok ? b : 0.0;
      if (4 * q < nvalid) {
        if (q & 1) H1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H1, 0, 0, 0);
        else H0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H0, 0, 0, 0);
      }
    }
  }
  pm_f64x4 H = H0 + H1;
  double hs[2] = {H[0], H[1]};
  if (Q.parts > 1) pm_xch_put<2>(Q.xch, first_wg, me, kstep, hs, lane);
  // while the sums travel: Y1 = L^-1 Delta^T (rows of this part in the columns)
  pm_f64x4 Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const bool live = 4 * kk + k < DD;
    const double dl = (live && c < nvalid) ? (double)xt[kk] - mean[kk] : 0.0;
    const double la = (live && c < DD) ? LiA[kk] : 0.0;
    Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(la, dl, Y1, 0, 0, 0);
  }
  bool ok = true;
  if (Q.parts > 1) ok = pm_xch_get<2>(Q.xch, first_wg, Q.parts, me, kstep, hs, lane);
  // mbar[i] = H[i][d] sits in lane (d, i & 3), register i >> 2; this lane's rows are i = k + 4 r: lane d + 16 k, register r
  double mb[NK];
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(hs[r]);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((DD + 16 * k) * 4, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((DD + 16 * k) * 4, (int)(unsigned)(u >> 32));
    mb[r] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  // Lbar = tril((g^T z - mbar zm^T) diag(zi)) in the tile's own registers (row i = k + 4 kk, column c)
  double lb[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) lb[kk] = (c <= k + 4 * kk && k + 4 * kk < DD) ? (hs[kk] - zm * mb[kk]) * zi : 0.0;
  pm_f64x4 Ph = {0.0, 0.0, 0.0, 0.0}, Pt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const double la = (4 * kk + k < DD && c < DD) ? LA[kk] : 0.0;
    Ph = __builtin_amdgcn_mfma_f64_16x16x4f64(la, lb[kk], Ph, 0, 0, 0);
    Pt = __builtin_amdgcn_mfma_f64_16x16x4f64(lb[kk], la, Pt, 0, 0, 0);
  }
  // Psi = low(Phi) + up(Phi^T), diagonal halved in both (entry (a, b): a = k + 4 r, b = c)
  double ps[NK];
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const int a = k + 4 * r;
    const double m1 = a > c ? 1.0 : (a == c ? 0.5 : 0.0), m2 = c > a ? 1.0 : (a == c ? 0.5 : 0.0);
    ps[r] = (a < DD && c < DD) ? m1 * Ph[r] + m2 * Pt[r] : 0.0;
  }
  pm_f64x4 Y2 = {0.0, 0.0, 0.0, 0.0}, Y3 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ps[kk], Y1[kk], Y2, 0, 0, 0);
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const double lt = (4 * kk + k < DD && c < DD) ? LiT[kk] : 0.0;
    Y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(lt, Y2[kk], Y3, 0, 0, 0);
  }
#pragma unroll
  for (int rr = 0; rr < NK; ++rr) out[rr] = (float)__builtin_fma(Y3[rr], Q.inv_m1, mb[rr] * Q.inv_m);
  return ok;
}
