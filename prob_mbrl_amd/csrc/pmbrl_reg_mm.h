// Moment matching inside the register-resident sweeps (pmbrl_reg.h), round 5: utils/rollout.py:20-29 and its adjoint
//  for a group that is split over the `parts` consecutive workgroups of <= 16 rows that hold it.
// Round 6: also ONE group over the whole batch (mm_groups=None, the reference examples' default --
// examples/deep_pilco_mm.py:31, utils/rollout.py:127-128): 157 parts at 2 500 rows, the sums over two levels.
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
// One wave per SIMD issues an instruction every 5-7 cycles, so what the chain costs is wave 0's instruction count --
// and three waves idle while it runs.  Everything of a step's chain that does NOT depend on the recursion is therefore
// prepared by the idle waves DURING THE PREVIOUS STEP'S CHAIN and handed over through LDS (two buffers, step t in buffer
// t & 1; the barrier that ends a chain orders the hand-over): the standardised noise rows (forward: wave 1), the
// adjoint's noise operand [zhat | 1] in MFMA layout (wave 1) and, from the forward sweep's stashed factor, L / L^-T in
// operand layout and Y1 = L^-1 Delta^T (wave 2, with its own MFMAs).  Wave 0 touches global memory only for the
// exchange and the factor it stashes; nothing of the moment matching is live across the step's GEMM phases, whose
// register budget is spoken for.
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
  unsigned xt_off;               // ... as a byte offset in the workspace (the forward sweep's buffer stores)
  unsigned long long* xch;       // granules of the statistics exchange (parts > 1)
  int fan, nwg;                  // more than 8 parts (mm_groups=None: ONE group over the batch): the sums travel over two levels --
                                 // every `fan` consecutive parts have a collector (pmbrl_xch.h, pm_xch_get_tree); nwg: the
                                 // launch's logical workgroups (the collectors' slots follow the parts')
  int xcd;                       // 1: the launch's workgroups are dealt so that a group's parts share an XCD (pr_wg)
  unsigned tag0;                 // generation of this launch, shifted past the step count: the granules' tags are tag0 + step
                                 // (no two launches share a tag, so the buffer is never zeroed between them)
  double inv_m, inv_m1;          // 1 / M, 1 / (M - 1) (from the host: an fp64 division is thirty instructions)
};

// doubles per lane the parts of a group exchange: the Gram / H tile's register 0 (rows 0 .. 3) and, beyond d = 4, register 1
#define PR_MM_NVX(DD) ((DD) <= 4 ? 1 : 2)

// 1 / sqrt(x): the hardware estimate (2^-26) and TWO Newton steps (pm_rsqrt takes three: on the one-wave chain every
// fp64 instruction is six cycles of the step)
__device__ __forceinline__ double pr_mm_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
#pragma unroll
  for (int it = 0; it < 2; ++it) y = __builtin_fma(y, __builtin_fma(-hx * y, y, 0.5), y);
  return y;
}

// mean (relative to the reference point), covariance and its Cholesky factor from the Gram tile of [s - ref | 1]:
// pm_mmw_factor (pmbrl_mm_w.h) with the reciprocals handed in.  Same pivot rule (a pivot that has shed more than fp32's
// precision counts as lost: the reference factors in fp32 and raises, utils/rollout.py:154-157).
template <int DD>
__device__ __forceinline__ bool pr_mm_factor(const pm_f64x4& G, double dM, double inv_m, double inv_m1, MMW<DD>& q,
                                             double* ratio_out = nullptr) {
  double rmin = 1.0;
  // (the tile is symmetric bit for bit -- both operands of its products are the same registers -- so the column sums are read
  //  from COLUMN d of the rows 0 .. d - 1: at d = 4 nothing of the tile's second register is used, and the parts of a group
  //  exchange one double per lane instead of two: PR_MM_NVX)
#pragma unroll
  for (int j = 0; j < DD; ++j) q.mean[j] = (DD <= 4 ? PM_G(G, j, DD) : PM_G(G, DD, j)) * inv_m;
  bool ok = true;
  // (column by column: the covariance entries of column k leave the tile when the factor gets there -- forming the
  //  whole matrix first keeps d (d + 1) / 2 more doubles alive than a 6 x 6 factorisation has registers for)
#pragma unroll
  for (int k = 0; k < DD; ++k) {
    double col[DD];
#pragma unroll
    for (int i = k; i < DD; ++i) col[i] = (PM_G(G, i, k) - dM * q.mean[i] * q.mean[k]) * inv_m1 + (i == k ? 1e-12 : 0.0);
    const double d0 = col[k];
    double piv = d0;
#pragma unroll
    for (int c = 0; c < k; ++c) piv -= q.L[k][c] * q.L[k][c];
    rmin = fmin(rmin, piv / d0);
    if (!(piv > 6e-8 * d0)) {
      ok = false;
      piv = 1.0;
    }
    const double rs = pr_mm_rsqrt(piv);
    q.L[k][k] = piv * rs;
    q.invd[k] = rs;
#pragma unroll
    for (int i = k + 1; i < DD; ++i) {
      double a = col[i];
#pragma unroll
      for (int c = 0; c < k; ++c) a -= q.L[i][c] * q.L[k][c];
      q.L[i][k] = a * rs;
    }
  }
  if (ratio_out) *ratio_out = rmin;
  return ok;
}

// fp64 16x16x4 MFMA with the accumulator in VGPRs: through the builtin hipcc put these accumulators into the accumulator
// file -- where the sweeps' weight fragments live -- and saved / restored eight of them around every chain (32 moves a step)
// (hazards the compiler's recogniser does not see inside asm, padded by hand: an operand written by a VALU instruction
//  needs 2 wait states before the MFMA reads it; a 16x16 DGEMM result needs 9 before the next DGEMM reads it as SrcC, 18
//  before a VALU instruction reads it)
__device__ __forceinline__ void pr_mfma64(pm_f64x4& acc, double a, double b) {
  asm volatile("s_nop 9\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pr_mfma64_0(pm_f64x4& acc, double a, double b) {      // first product of a chain: C = 0
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pr_mfma64_fence(pm_f64x4& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }
__device__ __forceinline__ void pr_mfma64_fence(pm_f64x4& a, pm_f64x4& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void pr_mfma64_open() {}

// (pointers read out of the argument struct carry no address space: accesses through them would be FLAT, which completes
//  out of order with LDS and degrades every LDS wait of the wave to lgkmcnt(0))
typedef PM_GLOBAL double* pr_gd;
typedef const PM_GLOBAL double* pr_gcd;
typedef const PM_GLOBAL float* pr_gcf;

// value `dim` of row `rr`'s state from the lanes' registers: it sits in lane rr + 16 (dim >> 1), slot dim & 1
__device__ __forceinline__ float pr_mm_gather(const float (&v)[2], int rr, int dim) {
  const int src = (rr + 16 * ((dim >> 1) & 3)) * 4;
  const int a = __builtin_amdgcn_ds_bpermute(src, __float_as_int(v[0]));
  const int b = __builtin_amdgcn_ds_bpermute(src, __float_as_int(v[1]));
  return __int_as_float((dim & 1) ? b : a);
}

// cyclic noise row of local row r of the group that starts at device row g0 (utils/rollout.py:53-59)
__device__ __forceinline__ pr_gcf pr_mm_zrow(const RegMM& Q, int D, int t, int g0, int r) {
  pr_gcf zb = (Q.flags & PMBRL_FLAG_ZMM_PER_STEP) ? (pr_gcf)Q.zmm + (size_t)t * Q.Bg * D : (pr_gcf)Q.zmm;
  const int z0 = (Q.flags & PMBRL_FLAG_ZMM_PER_STEP) ? Q.row_off + g0 : t + Q.row_off + g0;
  return zb + (size_t)pm_zidx(z0, r, Q.Bg) * D;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// LDS hand-over of the forward sweep: [2 buffers][16 rows][DD] doubles -- the standardised noise of the part's rows
#define PR_MM_ZH_DOUBLES(DD) (2 * 16 * (DD))
// ... and of the factor record's inputs, wave 0 -> wave 2 (pr_mm_fwd_file): [Gram register 0 | 1 | reference point][64 lanes]
#define PR_MM_REC_DOUBLES (3 * 64)
// (an idle wave, during the previous step's chain) zhat of the part's rows at step t -> zh; part 0 also files the
// standardisation in the factor record the latency-optimised family's adjoint reads (pmbrl_mm.h: mean | zm | zi | ...)
template <int DD>
__device__ __forceinline__ void pr_mm_fwd_prep(const RegMM& Q, int t, int gi, int g0, int row0, int nvalid, int me, int lane,
                                               double* zh) {
  const int r = lane & 15, cg = lane >> 4;
  pr_gcf zr = pr_mm_zrow(Q, DD, t, g0, row0 - g0 + (r < nvalid ? r : 0));
  pr_gcd zt = (pr_gcd)Q.ztab + ((size_t)t * Q.groups + gi) * 2 * DD;
  pr_gd fac = (pr_gd)Q.mmfac + ((size_t)t * Q.groups + gi) * pm_mm_fac_doubles(DD);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = cg + 4 * u;
    if (c < DD) {
      const double zm = zt[c], zi = zt[DD + c];
      zh[r * DD + c] = ((double)zr[c] - zm) * zi;
      if (me == 0 && r == 0) {
        fac[DD + c] = zm;
        fac[2 * DD + c] = zi;
      }
    }
  }
}

// pf: cycle stamps of the chain's stations (slots 8 ..; nullptr: none)
#define PR_MM_STAMP(slot) do { if (pf && lane == 0) pf[slot] = (long long)__builtin_readcyclecounter(); } while (0)
// xn: the sampled dimensions 2 g, 2 g + 1 of row lane & 15 (finite everywhere).  refl: column lane & 15 of the reference
// point every part of the group subtracts (in / out: the next step's is this step's mean).  zh: this step's hand-over
// buffer.  Returns false on a lost pivot or a partner that never arrived.  xout: dimensions 2 g, 2 g + 1 of the
// moment-matched row.
// rec: what pr_mm_fwd_file needs to write this step's factor record -- the group's Gram sums (two registers of the tile) and
// the reference point they are relative to.
template <int DD, bool TREE = false>
__device__ __forceinline__ bool pr_mm_fwd_chain(const RegMM& Q, int t, unsigned kstep, int gi, int me, int first_wg, int nvalid,
                                                int lane, const float (&xn)[2], double& refl, const double* zh,
                                                float (&xout)[2], double (&rec)[3], const double* hp, volatile unsigned* tags,
                                                long long* pf = nullptr) {
  static_assert(DD >= 2 && DD <= 6, "state widths 2..6");
  const int c = lane & 15, k = lane >> 4;
  pm_f64x4 G0, G1;
  {
    // operand of quad q: x~ - c in the state columns, 1 in column d, 0 elsewhere and in rows past the part.  The masks act
    // on the FLOAT (the reference point is a float by construction: a masked lane holds it and subtracts to exactly 0)
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = pr_mm_gather(xn, 4 * q + k, c);
    const float fill = (float)refl, one = c == DD ? 1.f : 0.f;      // (refl = 0 in the columns >= d)
    double x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float u = c < DD ? v[q] : one;
      u = 4 * q + k < nvalid ? u : fill;
      x[q] = (double)u - refl;
    }
    pr_mfma64_open();
    pr_mfma64_0(G0, x[0], x[0]);
    pr_mfma64_0(G1, x[1], x[1]);
    pr_mfma64(G0, x[2], x[2]);
    pr_mfma64(G1, x[3], x[3]);
    pr_mfma64_fence(G0, G1);
  }
  pm_f64x4 G = G0 + G1;
  bool ok = true;
  PR_MM_STAMP(8);
  constexpr int NVX = PR_MM_NVX(DD);
  double v2[NVX];
#pragma unroll
  for (int i = 0; i < NVX; ++i) v2[i] = G[i];
  if (Q.parts > 1) pm_xch_put<NVX>(Q.xch, first_wg, me, kstep, v2, lane);
  // (this lane's row of the standardised noise: requested here, used behind the factorisation -- at d > 4 requested behind
  //  it: the factorisation of a 5 x 5 or 6 x 6 covariance has no registers to carry the row through)
  double zhr[DD];
  if constexpr (DD <= 4) {
#pragma unroll
    for (int cc = 0; cc < DD; ++cc) zhr[cc] = zh[c * DD + cc];
  }
  if (Q.parts > 1) {
    if constexpr (TREE) ok = pm_xch_get_tree_helped<NVX, 4>(Q.xch, Q.nwg, first_wg, Q.parts, Q.fan, me, kstep, v2, hp, tags, lane);
    else ok = Q.parts == 2 ? pm_xch_get_pair<NVX>(Q.xch, first_wg, me, kstep, v2, lane)
                           : pm_xch_get_all<NVX, 4>(Q.xch, first_wg, Q.parts, me, kstep, v2, lane);
#pragma unroll
    for (int i = 0; i < NVX; ++i) G[i] = v2[i];
  }
  PR_MM_STAMP(9);
  rec[0] = G[0];
  rec[1] = G[1];
  rec[2] = refl;
  MMW<DD> q;
#ifdef PR_MM_DEBUG_RATIO
  double dbg_ratio;
  ok = pr_mm_factor<DD>(G, (double)Q.M, Q.inv_m, Q.inv_m1, q, &dbg_ratio) && ok;
#else
  ok = pr_mm_factor<DD>(G, (double)Q.M, Q.inv_m, Q.inv_m1, q) && ok;
#endif
  PR_MM_STAMP(10);
  if constexpr (DD > 4) {
#pragma unroll
    for (int cc = 0; cc < DD; ++cc) zhr[cc] = zh[c * DD + cc];
  }
  double mean[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) mean[j] = q.mean[j] + pm_rl64(refl, j);
  // this lane's row: m + zhat L^T, all d entries (d (d + 1) / 2 products), then the two this lane group keeps
  double acc[DD + 2];
#pragma unroll
  for (int j = 0; j < DD; ++j) {
    double a = mean[j];
#pragma unroll
    for (int cc = 0; cc <= j; ++cc) a += zhr[cc] * q.L[j][cc];
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
  PR_MM_STAMP(11);
  // the next step's reference point (every part computes the same bits)
  {
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < DD; ++j) r = c == j ? (double)(float)mean[j] : r;
    refl = r;
  }
  PR_MM_STAMP(12);
  return ok;
}

// The factor record of step t for the adjoint sweep, from the chain's hand-over (rec: [3][64] doubles in LDS -- the Gram
// sums' two registers and the reference point, lane by lane): an IDLE wave runs this one step later, next to the next
// step's chain.  (Until round 5 the chain's own wave did it behind the moment-matched rows: 1.1 k cycles of a 13.9 k-cycle
// step at d = 4 that every wave of the workgroup -- and, one step later, every part of the group -- waited for.)  The same
// factorisation on the same sums: the bits the chain used.  The duty is dealt over the parts: part 0 files
// mean | . | . | . | 1 / diag L, part 1 (of three or more; else part 0) L row-major (pm_mm_carve's record), the LAST
// part L^-1.
template <int DD>
__device__ __forceinline__ void pr_mm_fwd_file(const RegMM& Q, int t, int gi, int me, int lane, const double* rec) {
  const int l_part = Q.parts >= 3 ? 1 : 0;
  if (me != 0 && me != l_part && me != Q.parts - 1) return;
  pm_f64x4 G;
  G[0] = rec[lane];
  G[1] = rec[64 + lane];
  G[2] = G[3] = 0.0;
  double refl = rec[128 + lane];
  asm volatile("" : "+v"(refl));
  MMW<DD> q;
  (void)pr_mm_factor<DD>(G, (double)Q.M, Q.inv_m, Q.inv_m1, q);
  // (the cross-lane reads HERE, with every lane active: inside the one-lane branch below the compiler sinks the LDS read of
  //  `refl` into the branch too, and the lanes 1 .. d - 1 it is read from never load it)
  double mean[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) mean[j] = q.mean[j] + pm_rl64(refl, j);
  pr_gd fac = (pr_gd)Q.mmfac + ((size_t)t * Q.groups + gi) * pm_mm_fac_doubles(DD);
  if (lane == 0 && me == 0) {
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      fac[j] = mean[j];
      fac[4 * DD + j] = q.invd[j];
    }
  }
  if (lane == 0 && me == l_part) {
#pragma unroll
    for (int j = 0; j < DD; ++j)
#pragma unroll
      for (int cc = 0; cc < DD; ++cc) fac[5 * DD + j * DD + cc] = cc <= j ? q.L[j][cc] : 0.0;
  }
  if (me == Q.parts - 1) {
    // (column by column, each stored as soon as it is solved)
    pr_gd li = (pr_gd)Q.linv + ((size_t)t * Q.groups + gi) * (DD * DD);
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      double x[DD];
      x[j] = q.invd[j];
#pragma unroll
      for (int i = j + 1; i < DD; ++i) {
        double a = 0.0;
#pragma unroll
        for (int cc = j; cc < i; ++cc) a += q.L[i][cc] * x[cc];
        x[i] = -a * q.invd[i];
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < DD; ++i) li[i * DD + j] = i >= j ? x[i] : 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// adjoint
// ---------------------------------------------------------------------------
// LDS hand-over of the adjoint sweep, per buffer: the noise operand [4 row quads][64 lanes] and, per lane,
// Y1 (NK registers) | L (NK) | L^-T (NK) -- doubles
#define PR_MM_BOP_DOUBLES (4 * 64)
#define PR_MM_YOP_DOUBLES(DD) (3 * (((DD) + 3) / 4) * 64)
// (wave 1, during the next-higher step's chain) the B operand of H = g^T [zhat | 1] at step t: lane (c, k), quad q
template <int DD>
__device__ __forceinline__ void pr_mm_bwd_prep_noise(const RegMM& Q, int t, int gi, int g0, int row0, int nvalid, int lane,
                                                     double* bop) {
  const int c = lane & 15, k = lane >> 4, cc = c < DD ? c : DD - 1;
  pr_gcd zt = (pr_gcd)Q.ztab + ((size_t)t * Q.groups + gi) * 2 * DD;
  const double zm = zt[cc], zi = zt[DD + cc];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool rok = 4 * q + k < nvalid;
    const float z = pr_mm_zrow(Q, DD, t, g0, row0 - g0 + (rok ? 4 * q + k : 0))[cc];
    const double b = c < DD ? ((double)z - zm) * zi : (c == DD ? 1.0 : 0.0);
    bop[q * 64 + lane] = rok ? b : 0.0;
  }
}
// (wave 2, likewise) from the forward sweep's factor at step t: Y1 = L^-1 Delta^T of the part's rows, L and L^-T in
// operand layout -- lane (c, k): L[4 kk + k][c], L^-1[4 kk + k][c]
template <int DD>
__device__ __forceinline__ void pr_mm_bwd_prep_factor(const RegMM& Q, int B, int t, int gi, int row0, int nvalid, int lane,
                                                      double* yop) {
  constexpr int NK = (DD + 3) / 4;
  const int c = lane & 15, k = lane >> 4, cc = c < DD ? c : DD - 1;
  pr_gcd fac = (pr_gcd)Q.mmfac + ((size_t)t * Q.groups + gi) * pm_mm_fac_doubles(DD);
  pr_gcd li = (pr_gcd)Q.linv + ((size_t)t * Q.groups + gi) * (DD * DD);
  pr_gcf xr = (pr_gcf)Q.xt + ((size_t)t * B + row0 + (c < nvalid ? c : 0)) * DD;
  double la[NK], lia[NK], lit[NK], dl[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int i = 4 * kk + k, ii = i < DD ? i : DD - 1;
    const bool live = i < DD && c < DD;
    const double a = fac[5 * DD + ii * DD + cc], b = li[cc * DD + ii], e = li[ii * DD + cc];
    la[kk] = live ? a : 0.0;
    lia[kk] = live ? b : 0.0;
    lit[kk] = live ? e : 0.0;
    dl[kk] = (i < DD && c < nvalid) ? (double)xr[ii] - fac[ii] : 0.0;
  }
  pm_f64x4 Y1;
  pr_mfma64_open();
  pr_mfma64_0(Y1, lia[0], dl[0]);
  if constexpr (NK > 1) pr_mfma64(Y1, lia[1], dl[1]);
  pr_mfma64_fence(Y1);
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    yop[(0 * NK + kk) * 64 + lane] = Y1[kk];
    yop[(1 * NK + kk) * 64 + lane] = la[kk];
    yop[(2 * NK + kk) * 64 + lane] = lit[kk];
  }
}

// gx: dL/dx_{t+1}, dimensions 2 g, 2 g + 1 of row lane & 15 (zero in rows past the part).  out[rr]: dL/dx~ of dimension
// (lane >> 4) + 4 rr of row lane & 15.  bop / yop: this step's hand-over buffers.
template <int DD, bool TREE = false>
__device__ __forceinline__ bool pr_mm_bwd_chain(const RegMM& Q, unsigned kstep, int me, int first_wg, int nvalid, int lane,
                                                const float (&gx)[2], const double* bop, const double* yop,
                                                float (&out)[(DD + 3) / 4], const double* hp, volatile unsigned* tags,
                                                long long* pf = nullptr) {
  constexpr int NK = (DD + 3) / 4;
  const int c = lane & 15, k = lane >> 4;
  // H = g^T [zhat | 1]
  pm_f64x4 H0, H1;
  {
    // (rows past the part: the noise operand is zero there, whatever the gradient lanes hold -- dL/dx is kept at zero in
    //  them anyway; the columns >= d of the gradient operand are masked on the float)
    float v[4];
    double b[4], a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = pr_mm_gather(gx, 4 * q + k, c);
      b[q] = bop[q * 64 + lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = (double)(c < DD ? v[q] : 0.f);
    pr_mfma64_open();
    pr_mfma64_0(H0, a[0], b[0]);
    pr_mfma64_0(H1, a[1], b[1]);
    pr_mfma64(H0, a[2], b[2]);
    pr_mfma64(H1, a[3], b[3]);
    pr_mfma64_fence(H0, H1);
  }
  pm_f64x4 H = H0 + H1;
  // (rows i < d of H: register i >> 2 -- at d = 4 the first alone)
  constexpr int NVX = PR_MM_NVX(DD);
  static_assert(NVX == NK, "the H tile's registers the algebra reads are the ones exchanged");
  double hs[NVX];
#pragma unroll
  for (int i = 0; i < NVX; ++i) hs[i] = H[i];
  PR_MM_STAMP(8);
  if (Q.parts > 1) pm_xch_put<NVX>(Q.xch, first_wg, me, kstep, hs, lane);
  double Y1[NK], LA[NK], LiT[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    Y1[kk] = yop[(0 * NK + kk) * 64 + lane];
    LA[kk] = yop[(1 * NK + kk) * 64 + lane];
    LiT[kk] = yop[(2 * NK + kk) * 64 + lane];
  }
  bool ok = true;
  PR_MM_STAMP(9);
  if (Q.parts > 1)
  {
    if constexpr (TREE) ok = pm_xch_get_tree_helped<NVX, 4>(Q.xch, Q.nwg, first_wg, Q.parts, Q.fan, me, kstep, hs, hp, tags, lane);
    else ok = Q.parts == 2 ? pm_xch_get_pair<NVX>(Q.xch, first_wg, me, kstep, hs, lane)
                           : pm_xch_get_all<NVX, 4>(Q.xch, first_wg, Q.parts, me, kstep, hs, lane);
  }
  PR_MM_STAMP(10);
  // Lbar = tril(H[:, :d]) in the tile's own registers (row i = k + 4 kk, column c)
  double lb[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) lb[kk] = (c <= k + 4 * kk && k + 4 * kk < DD) ? hs[kk] : 0.0;
  pm_f64x4 Ph, Pt;
  pr_mfma64_open();
  pr_mfma64_0(Ph, LA[0], lb[0]);
  pr_mfma64_0(Pt, lb[0], LA[0]);
  if constexpr (NK > 1) {
    pr_mfma64(Ph, LA[1], lb[1]);
    pr_mfma64(Pt, lb[1], LA[1]);
  }
  pr_mfma64_fence(Ph, Pt);
  // mbar[i] = H[i][d] sits in lane (d, i & 3), register i >> 2; this lane's rows are i = k + 4 r: lane d + 16 k, register r
  // (behind the first products: only the last line needs it)
  double mb[NK];
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(hs[r]);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((DD + 16 * k) * 4, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((DD + 16 * k) * 4, (int)(unsigned)(u >> 32));
    mb[r] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  // Psi = low(Phi) + up(Phi^T), diagonal halved in both (entry (a, b): a = k + 4 r, b = c)
  double ps[NK];
#pragma unroll
  for (int r = 0; r < NK; ++r) {
    const int a = k + 4 * r;
    const double m1 = a > c ? 1.0 : (a == c ? 0.5 : 0.0), m2 = c > a ? 1.0 : (a == c ? 0.5 : 0.0);
    ps[r] = (a < DD && c < DD) ? m1 * Ph[r] + m2 * Pt[r] : 0.0;
  }
  pm_f64x4 Y2, Y3;
  pr_mfma64_open();
  pr_mfma64_0(Y2, ps[0], Y1[0]);
  if constexpr (NK > 1) pr_mfma64(Y2, ps[1], Y1[1]);
  pr_mfma64_fence(Y2);
  double y2[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) y2[kk] = Y2[kk];
  pr_mfma64_0(Y3, LiT[0], y2[0]);
  if constexpr (NK > 1) pr_mfma64(Y3, LiT[1], y2[1]);
  pr_mfma64_fence(Y3);
#pragma unroll
  for (int rr = 0; rr < NK; ++rr) out[rr] = (float)__builtin_fma(Y3[rr], Q.inv_m1, mb[rr] * Q.inv_m);
  PR_MM_STAMP(11);
  return ok;
}
