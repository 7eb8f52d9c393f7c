// Moment matching (utils/rollout.py:20-29, mm_resample_) and its adjoint
// (SURVEY.md Appendix A), one WAVEFRONT per group.
//
//   m = mean(s); Delta = s - m; S = Delta^T Delta/(M-1) + 1e-12 I; L = chol(S)
//   zhat = (z - mean z)/std_unbiased(z)   (constant w.r.t. the gradient)
//   out = m + zhat L^T
//
// The statistics, the Cholesky factor and the adjoint's two triangular solves
// are carried in fp64: the reference's own fp32 result is ill-conditioned here
// (the 1x1 reward "covariance" underflows when a group's rewards are nearly
// equal and the adjoint divides by it), fp64 on a dxd matrix per group is free
// on this part, and it moves the result towards the fp64 reference rather than
// away from it.  Rows may live in LDS or in HBM (generic pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
typedef double pm_f64x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t pm_mm_scratch_doubles(int d) {
  return d > 0 ? (size_t)3 * d * d + 5 * d : 0;
}

// leading part of the scratch that the forward pass hands to the adjoint: mean, zmean, zistd,
// (mbar), invd, L  -- see pm_mm_carve
__host__ __device__ inline size_t pm_mm_fac_doubles(int d) { return (size_t)5 * d + (size_t)d * d; }

// 1/sqrt(x) in fp64 from the hardware estimate + Newton steps (the IEEE sqrt and divide
// expansions cost several hundred cycles each and sat on the serial path of every pivot and
// every triangular-solve element)
__device__ __forceinline__ double pm_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 3; ++it) y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// LDS traffic between lanes of ONE wave: make prior writes visible / ordered.
__device__ __forceinline__ void pm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct MMScratch {
  double *mean, *zmean, *zistd, *mbar, *invd, *Lm, *P, *Sb;   // invd: 1 / diag(L)
};
__device__ inline MMScratch pm_mm_carve(double* scr, int d) {
  MMScratch s;
  s.mean = scr;
  s.zmean = s.mean + d;
  s.zistd = s.zmean + d;
  s.mbar = s.zistd + d;
  s.invd = s.mbar + d;
  s.Lm = s.invd + d;
  s.P = s.Lm + d * d;
  s.Sb = s.P + d * d;
  return s;
}

// cyclic noise row of utils/rollout.py:53-59
// (Bg == 0: the rows were already gathered into a dense block, no wrap)
__device__ __forceinline__ int pm_zidx(int zrow0, int i, int Bg) { return Bg ? (zrow0 + i) % Bg : zrow0 + i; }

// Sums over the M rows of a group are split P ways: lane = entry * P + part, a part walks rows
// part, part+P, ..., the P partial sums of an entry meet in a butterfly (fixed order ->
// deterministic).  P = 64 / (entries rounded up to a power of two), 1 when there are >= 64.
__device__ __forceinline__ int pm_pow2ceil(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}
__device__ __forceinline__ double pm_seg_sum(double v, int P) {
  for (int o = P >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Standardisation of a split group's noise rows, mean and 1 / std per column and step (what every part of a split group
// needs of the WHOLE group), tabulated per launch: [H][groups][2 D] doubles.  Same formula and summation scheme as the
// sweeps' own prologue (pmbrl_fast.h): lane l adds the rows l, l + 64, ... in that order, a butterfly joins the lanes.
// Workgroups of 256 threads; ONE definition for the launch of its own (pm_mm_ztable_kernel) and for the extra workgroups
// of the register-resident family's pack launch (pm_reg_pack_kernel: one launch less per iteration).
//   pack = 0: workgroup = (step, group), wave w takes the columns w, w + 4, ... (large groups: eight rows' loads in flight
//             per lane -- one at a time, a 2 500-row group was 39 memory round trips in a row, 14 us);
//   pack = 1: groups of <= 64 rows and D <= 8: a WAVE per (step, group), four to a workgroup, the columns' loads issued
//             together.
struct ZtabArgs {
  const float* zmm;
  double* tab;
  int n_blocks;        // workgroups the table takes (0: no table)
  int pack;
  int H, G, M, D, Bg, per_step, row_off;
};
__host__ __device__ inline int pm_ztab_blocks(int H, int G, int pack) { return pack ? (H * G + 3) / 4 : H * G; }
__device__ __forceinline__ void pm_ztab_block(const ZtabArgs& Z, int blk, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  const int D = Z.D, M = Z.M;
  const double dM = (double)M, inv_m = 1.0 / dM, inv_m1 = 1.0 / (double)(M - 1);
  const int item = Z.pack ? blk * 4 + wid : blk;
  if (item >= Z.H * Z.G) return;
  const int t = item / Z.G, gi = item - t * Z.G;
  const float* zb = Z.per_step ? Z.zmm + (size_t)t * Z.Bg * D : Z.zmm;
  const int row = Z.row_off + gi * M;
  const int z0 = Z.per_step ? row : t + row;
  double* out = Z.tab + (size_t)item * 2 * D;
  if (Z.pack) {
    const float* zr = zb + (size_t)pm_zidx(z0, min(lane, M - 1), Z.Bg) * D;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = zr[min(j, D - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < D) {
        double s1 = 0.0, s2 = 0.0;
        if (lane < M) {
          const double zv = (double)v[j];
          s1 += zv;
          s2 += zv * zv;
        }
        const double sm = pm_seg_sum(s1, 64), sq = pm_seg_sum(s2, 64);
        const double zm = sm * inv_m;
        if (lane == 0) {
          out[j] = zm;
          out[D + j] = pm_rsqrt((sq - dM * zm * zm) * inv_m1);
        }
      }
    }
    return;
  }
  for (int j = wid; j < D; j += 4) {
    double s1 = 0.0, s2 = 0.0;
    for (int r0 = lane; r0 < M; r0 += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = zb[(size_t)pm_zidx(z0, min(r0 + 64 * u, M - 1), Z.Bg) * D + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 64 * u < M) {
          const double zv = (double)v[u];
          s1 += zv;
          s2 += zv * zv;
        }
    }
    const double sm = pm_seg_sum(s1, 64), sq = pm_seg_sum(s2, 64);
    const double zm = sm * inv_m;
    if (lane == 0) {
      out[j] = zm;
      out[D + j] = pm_rsqrt((sq - dM * zm * zm) * inv_m1);
    }
  }
}

// In-place Cholesky factor of the covariance in q.Lm (lower triangle), 1 / diag(L) into q.invd.
// Returns false (wave-uniform) on a non-positive pivot.
__device__ __forceinline__ bool pm_mm_chol(int d, const MMScratch& q, int lane) {
  // The reference factors in fp32 and raises (-> RuntimeError, utils/rollout.py:154-157)
  // when a pivot is lost to rounding; reproduce that contract: a pivot that has shed
  // more than fp32 precision relative to its diagonal entry counts as non-positive.
  for (int j = lane; j < d; j += 64) q.mbar[j] = q.Lm[j * d + j];
  pm_wave_sync();
  bool ok = true;
  for (int k = 0; k < d; ++k) {
    double piv = q.Lm[k * d + k];
    if (!(piv > 6e-8 * q.mbar[k])) {
      ok = false;
      piv = 1.0;
    }
    const double rs = pm_rsqrt(piv);
    const double lkk = piv * rs;
    pm_wave_sync();   // everyone has read the pivot before lane 0 overwrites it
    for (int i = k + 1 + lane; i < d; i += 64) q.Lm[i * d + k] *= rs;
    if (lane == 0) {
      q.Lm[k * d + k] = lkk;
      q.invd[k] = rs;
    }
    pm_wave_sync();
    for (int e = lane; e < d * d; e += 64) {
      const int i = e / d, j = e - i * d;
      if (j > k && j <= i) q.Lm[e] -= q.Lm[i * d + k] * q.Lm[j * d + k];
    }
    pm_wave_sync();
  }
  return ok;
}

// means, z standardisation, covariance and its Cholesky factor.  Returns false
// (wave-uniform) on a non-positive pivot.
__device__ __forceinline__ bool pm_mm_factor(const float* s, int s_ld, int M, int d, const float* z,
                                    int z_ld, int zrow0, int Bg, const MMScratch& q, int lane) {
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  {
    const int e2 = pm_pow2ceil(d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int part = lane % P;
    for (int base = 0; base < d; base += per) {
      const int j = base + lane / P;
      double m = 0.0, zm = 0.0, zz = 0.0;
      if (j < d)
        for (int i = part; i < M; i += P) {
          m += (double)s[i * s_ld + j];
          const double zv = (double)z[(size_t)pm_zidx(zrow0, i, Bg) * z_ld + j];
          zm += zv;
          zz += zv * zv;
        }
      m = pm_seg_sum(m, P);
      zm = pm_seg_sum(zm, P);
      zz = pm_seg_sum(zz, P);
      if (j < d && part == 0) {
        m *= inv_m;
        zm *= inv_m;
        q.mean[j] = m;
        q.zmean[j] = zm;
        q.zistd[j] = pm_rsqrt((zz - (double)M * zm * zm) * inv_m1);   // fp64: no cancellation issue for N(0,1) rows
      }
    }
  }
  pm_wave_sync();
  {
    const int e2 = pm_pow2ceil(d * d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int part = lane % P;
    for (int base = 0; base < d * d; base += per) {
      const int e = base + lane / P;
      const int i = e / d, j = e - i * d;
      double acc = 0.0;
      const bool live = e < d * d && j <= i;
      if (live) {
        const double mi = q.mean[i], mj = q.mean[j];
        for (int r = part; r < M; r += P)
          acc += ((double)s[r * s_ld + i] - mi) * ((double)s[r * s_ld + j] - mj);
      }
      acc = pm_seg_sum(acc, P);
      if (e < d * d && part == 0) q.Lm[e] = live ? acc * inv_m1 + (i == j ? 1e-12 : 0.0) : 0.0;
    }
  }
  pm_wave_sync();
  return pm_mm_chol(d, q, lane);
}

__device__ __forceinline__ bool pm_mm_fwd(const float* s, int s_ld, int M, int d, const float* z, int z_ld,
                                 int zrow0, int Bg, bool infer_ns, float* out, int out_ld,
                                 double* scr, int lane, double* fac_out = nullptr) {
  const MMScratch q = pm_mm_carve(scr, d);
  const bool ok = pm_mm_factor(s, s_ld, M, d, z, z_ld, zrow0, Bg, q, lane);
  // the adjoint needs the same means / standardisation / factor: hand them over instead of
  // having it redo the statistics and the factorisation (pm_mm_fac_doubles values)
  if (fac_out)
    for (int e = lane; e < (int)pm_mm_fac_doubles(d); e += 64) fac_out[e] = scr[e];
  if (infer_ns) {
    // utils/rollout.py:6-17 (mm_resample_infer_ns_): zhat = Delta L^-T, so m + zhat L^T is the input
    // again (the reference gets it back up to rounding); only the gradient differs -- pm_mm_bwd
    for (int e = lane; e < M * d; e += 64) {
      const int r = e / d, j = e - r * d;
      out[r * out_ld + j] = s[r * s_ld + j];
    }
    pm_wave_sync();
    return ok;
  }
  for (int e = lane; e < M * d; e += 64) {
    const int r = e / d, j = e - r * d;
    double acc = q.mean[j];
    const size_t zr = (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
    for (int c = 0; c <= j; ++c)
      acc += ((double)z[zr + c] - q.zmean[c]) * q.zistd[c] * q.Lm[j * d + c];
    out[r * out_ld + j] = (float)acc;
  }
  pm_wave_sync();
  return ok;
}

// The adjoint behind the sums over the rows: (q.mbar = sum_r g, q.P = Lbar = tril(g^T zhat), the factor in q)
// -> gout = dL/d s for the M rows at s; inv_m / inv_m1 = 1 / M and 1 / (M - 1) of the WHOLE group (the rows
// at s may be a part of it: groups spread over devices, pmbrl_mmx.h);
// first the serial part on ONE wave (pm_mm_bwd_solve: q.P <- the symmetric matrix of the row formula), then the
// rows (pm_mm_bwd_rows: any number of threads, thread `tid` of `nthreads`).
__device__ __forceinline__ void pm_mm_bwd_solve(int d, double inv_m1, const MMScratch& q, int lane) {
  // Phi = tril(L^T Lbar), diagonal halved -> q.Sb
  for (int e = lane; e < d * d; e += 64) {
    const int i = e / d, j = e - i * d;
    double acc = 0.0;
    if (j <= i) {
      for (int c = i; c < d; ++c) acc += q.Lm[c * d + i] * q.P[c * d + j];
      if (i == j) acc *= 0.5;
    }
    q.Sb[e] = acc;
  }
  pm_wave_sync();
  // X = Phi L^-1  (row i of X solves x L = phi_i), in place in q.Sb
  for (int i = lane; i < d; i += 64) {
    for (int j = d - 1; j >= 0; --j) {
      double a = q.Sb[i * d + j];
      for (int c = j + 1; c < d; ++c) a -= q.Sb[i * d + c] * q.Lm[c * d + j];
      q.Sb[i * d + j] = a * q.invd[j];
    }
  }
  pm_wave_sync();
  // Sbar = L^-T X  (column j solves L^T y = x_j), in place
  for (int j = lane; j < d; j += 64) {
    for (int i = d - 1; i >= 0; --i) {
      double a = q.Sb[i * d + j];
      for (int c = i + 1; c < d; ++c) a -= q.Lm[c * d + i] * q.Sb[c * d + j];
      q.Sb[i * d + j] = a * q.invd[i];
    }
  }
  pm_wave_sync();
  // symmetrise into q.P:  P = (Sbar + Sbar^T) / (M-1)   (= 2 * sym(Sbar) / (M-1))
  for (int e = lane; e < d * d; e += 64) {
    const int i = e / d, j = e - i * d;
    q.P[e] = (q.Sb[i * d + j] + q.Sb[j * d + i]) * inv_m1;
  }
  pm_wave_sync();
}
__device__ __forceinline__ void pm_mm_bwd_rows(const float* s, int s_ld, int M, int d, double inv_m, float* gout,
                                               int gout_ld, const MMScratch& q, int tid, int nthreads) {
  // sbar[r][j] = sum_c Delta[r][c] P[c][j] + mbar[j]/M      (mean_r of the first term is 0)
  for (int e = tid; e < M * d; e += nthreads) {
    const int r = e / d, j = e - r * d;
    double acc = q.mbar[j] * inv_m;
    for (int c = 0; c < d; ++c) acc += ((double)s[r * s_ld + c] - q.mean[c]) * q.P[c * d + j];
    // all reads of g happened before the first pm_wave_sync above: in-place is safe
    gout[r * gout_ld + j] = (float)acc;
  }
}
__device__ __forceinline__ void pm_mm_bwd_finish(const float* s, int s_ld, int M, int d, double inv_m, double inv_m1,
                                                 float* gout, int gout_ld, const MMScratch& q, int lane) {
  pm_mm_bwd_solve(d, inv_m1, q, lane);
  pm_mm_bwd_rows(s, s_ld, M, d, inv_m, gout, gout_ld, q, lane, 64);
  pm_wave_sync();
}

// infer_noise_variables: q.Sb = A = g^T Delta (full d x d) -> q.P = Lbar = tril(A L^-T): row i of A L^-T by forward
// substitution (x L^T = A_i)
__device__ __forceinline__ void pm_mm_infer_lbar(int d, const MMScratch& q, int lane) {
  for (int i = lane; i < d; i += 64) {
    for (int j = 0; j < d; ++j) {
      double a = q.Sb[i * d + j];
      for (int c = 0; c < j; ++c) a -= q.P[i * d + c] * q.Lm[j * d + c];
      q.P[i * d + j] = a * q.invd[j];       // full row first (x_c for c < j feeds x_j) ...
    }
    for (int j = i + 1; j < d; ++j) q.P[i * d + j] = 0.0;   // ... then keep the lower triangle
  }
  pm_wave_sync();
}

// g: upstream dL/d out [M][d]; gout: dL/d s [M][d] (may alias g).
__device__ __forceinline__ void pm_mm_bwd(const float* s, int s_ld, int M, int d, const float* z, int z_ld,
                                 int zrow0, int Bg, bool infer_ns, const float* g, int g_ld,
                                 float* gout, int gout_ld, double* scr, int lane,
                                 const double* fac = nullptr) {
  const MMScratch q = pm_mm_carve(scr, d);
  if (fac) {
    for (int e = lane; e < (int)pm_mm_fac_doubles(d); e += 64) scr[e] = fac[e];
    pm_wave_sync();
  } else {
    (void)pm_mm_factor(s, s_ld, M, d, z, z_ld, zrow0, Bg, q, lane);
  }
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  // mbar = sum_r g ;  Lbar = tril(g^T zhat)  -> q.P
  {
    const int e2 = pm_pow2ceil(d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int part = lane % P;
    for (int base = 0; base < d; base += per) {
      const int j = base + lane / P;
      double a = 0.0;
      if (j < d)
        for (int r = part; r < M; r += P) a += (double)g[r * g_ld + j];
      a = pm_seg_sum(a, P);
      if (j < d && part == 0) q.mbar[j] = a;
    }
  }
  {
    const int e2 = pm_pow2ceil(d * d);
    const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
    const int part = lane % P;
    for (int base = 0; base < d * d; base += per) {
      const int e = base + lane / P;
      const int i = e / d, j = e - i * d;
      double acc = 0.0;
      const bool live = e < d * d && j <= i;
      if (live) {
        const double zm = q.zmean[j], zs = q.zistd[j];
        for (int r = part; r < M; r += P)
          acc += (double)g[r * g_ld + i] *
                 (((double)z[(size_t)pm_zidx(zrow0, r, Bg) * z_ld + j] - zm) * zs);
      }
      acc = pm_seg_sum(acc, P);
      if (e < d * d && part == 0) q.P[e] = live ? acc : 0.0;
    }
  }
  pm_wave_sync();
  if (infer_ns) {
    // zhat = Delta L^-T (constant): Lbar = tril(g^T zhat) = tril((g^T Delta) L^-T).
    // A = g^T Delta -> q.Sb, then row i of A L^-T by forward substitution (x L^T = A_i) -> q.P
    {
      const int e2 = pm_pow2ceil(d * d);
      const int P = e2 >= 64 ? 1 : 64 / e2, per = 64 / P;
      const int part = lane % P;
      for (int base = 0; base < d * d; base += per) {
        const int e = base + lane / P;
        const int i = e / d, j = e - i * d;
        double acc = 0.0;
        if (e < d * d) {
          const double mj = q.mean[j];
          for (int r = part; r < M; r += P) acc += (double)g[r * g_ld + i] * ((double)s[r * s_ld + j] - mj);
        }
        acc = pm_seg_sum(acc, P);
        if (e < d * d && part == 0) q.Sb[e] = acc;
      }
    }
    pm_wave_sync();
    pm_mm_infer_lbar(d, q, lane);
  }
  pm_mm_bwd_finish(s, s_ld, M, d, inv_m, inv_m1, gout, gout_ld, q, lane);
}


// The adjoint with a compile-time width (the shape-specialised sweep kernels) and the factor already in the
// scratch (the caller copies pm_mm_fac_doubles values there while the group's rows are still on their way from
// HBM: that load is a full memory round trip and used to open this routine).  Same mathematics and summation
// order as pm_mm_bwd; what changes is where the serial part runs: the d lanes that do the two triangular
// solves keep L, 1 / diag(L) and their own row / column in REGISTERS -- the LDS form pays an LDS round trip
// for every one of the d (d + 1) / 2 dependent steps of a solve.
// Gram tiles of a group whose rows sit in LDS, row-major with row length DD (what the split-group paths of the
// latency-optimised sweeps stage: pmbrl_fast.h).  Branch-free operands, all of a pass's LDS reads issued before
// its MFMAs -- the general routine above (cyclic noise rows, device-scope loads, row ranges) compiles to a load
// and a wait inside a divergent branch per MFMA: 2.2 k cycles for 25 rows, 4.4 k for 50.
//   pm_mm_gram_lds:   X^T X for X = [s - ref | 1 | z]          (ref: in lane l the reference point's column l & 15)
//   pm_mm_gram_h_lds: g^T [zhat | 1], zhat = (z - zm) zi       (zm / zi: in lane l those of column l & 15)
// (one wave alone on its SIMD issues an instruction every 4-5 cycles: what these routines cost is their instruction
//  count -- operands are formed with per-lane constant multipliers instead of selects the compiler turns into branches)
// MODE 0: X = [s - ref | 1 | z];  1: [s - ref | 1 | 0] (the sums several workgroups add up);  2: [0 | 1 | z]
// Row quads q0, q0 + qs, ... (rows 4 q + lane / 16): eight waves take two quads each of a 64-row group.
template <int DD, int MODE = 0, int UB = 8>
__device__ __forceinline__ pm_f64x4 pm_mm_gram_lds(const float* s, const float* z, int M, int lane, double ref,
                                                   int q0 = 0, int qs = 1) {
  static_assert(2 * DD + 1 <= 16, "the Gram tile holds 2d+1 columns");
  const int g = lane >> 4, c = lane & 15;
  const bool is_s = MODE != 2 && c < DD, is_z = MODE != 1 && c > DD && c <= 2 * DD;
  const float* base = (is_z ? z + (c - DD - 1) : s + (c < DD ? c : 0)) + g * DD;
  // x = (v - sub) * mul + one: data columns mul = 1, the column of ones mul = 0 / one = 1, unused columns 0 / 0
  double mul = (is_s || is_z) ? 1.0 : 0.0, one = c == DD ? 1.0 : 0.0, sub = is_s ? ref : 0.0;
  asm volatile("" : "+v"(mul), "+v"(one), "+v"(sub));
  pm_f64x4 G0 = {0.0, 0.0, 0.0, 0.0}, G1 = {0.0, 0.0, 0.0, 0.0};
  for (int qb = q0; 4 * qb < M; qb += UB * qs) {
    float v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = qb + u * qs;
      v[u] = base[(4 * q + g < M ? 4 * q : 0) * DD];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) asm volatile("" : "+v"(v[u]));     // the reads stay here, together
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = qb + u * qs;
      double x = __builtin_fma((double)v[u] - sub, mul, one);
      x = 4 * q + g < M ? x : 0.0;
      asm volatile("" : "+v"(x));
      if (4 * q < M) {      // (uniform) two accumulator chains
        if (u & 1) G1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G1, 0, 0, 0);
        else G0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G0, 0, 0, 0);
      }
    }
  }
  return G0 + G1;
}
template <int DD, int UB = 8>
__device__ __forceinline__ pm_f64x4 pm_mm_gram_h_lds(const float* g_rows, const float* z, int M, int lane, double zm,
                                                     double zi, int q0 = 0, int qs = 1) {
  const int gq = lane >> 4, c = lane & 15;
  const int cc = c < DD ? c : 0;
  const float* gb = g_rows + gq * DD + cc;
  const float* zb = z + gq * DD + cc;
  // a = g * am; b = (z - zm) * bm + one
  double am = c < DD ? 1.0 : 0.0, bm = c < DD ? zi : 0.0, one = c == DD ? 1.0 : 0.0;
  asm volatile("" : "+v"(am), "+v"(bm), "+v"(one));
  pm_f64x4 H0 = {0.0, 0.0, 0.0, 0.0}, H1 = {0.0, 0.0, 0.0, 0.0};
  for (int qb = q0; 4 * qb < M; qb += UB * qs) {
    float gv[UB], zv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = qb + u * qs;
      const int o = (4 * q + gq < M ? 4 * q : 0) * DD;
      gv[u] = gb[o];
      zv[u] = zb[o];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) asm volatile("" : "+v"(gv[u]), "+v"(zv[u]));
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = qb + u * qs;
      double a = (double)gv[u] * am;
      double b = __builtin_fma((double)zv[u] - zm, bm, one);
      a = 4 * q + gq < M ? a : 0.0;
      b = 4 * q + gq < M ? b : 0.0;
      asm volatile("" : "+v"(a), "+v"(b));
      if (4 * q < M) {
        if (u & 1) H1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H1, 0, 0, 0);
        else H0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H0, 0, 0, 0);
      }
    }
  }
  return H0 + H1;
}

// xf (a group split over workgroups): xf(H) delivers the sums over g as a tile g^T [zhat | 1] (pm_mm_gram_h_lds over
// the parts' rows, summed by the caller: pmbrl_fast.h); g is not read here then
struct PmNoXch {
  __device__ __forceinline__ bool operator()(pm_f64x4&) const { return true; }
};
// the d x d part of the adjoint: q.P = Lbar (lower triangle), q.Lm / q.invd the factor  ->  q.P = (Sbar + Sbar^T) / (M - 1)
template <int DD>
__device__ __forceinline__ void pm_mm_bwd_l_tail(const MMScratch& q, int lane, int M) {
  constexpr int d = DD;
  const double inv_m1 = 1.0 / (double)(M - 1);
  // L and 1 / diag(L) in registers (every lane: the loads are broadcasts, issued back to back)
  double Lr[DD][DD], idg[DD];
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    idg[i] = q.invd[i];
#pragma unroll
    for (int j = 0; j <= i; ++j) Lr[i][j] = q.Lm[i * d + j];
  }
  // Phi = tril(L^T Lbar), diagonal halved: lane i builds ROW i, solves x L = phi_i in registers -> row i of X
  if (lane < d) {
    const int i = lane;
    double lb[DD][DD];   // Lbar rows c >= i are needed: read the whole lower triangle (broadcast reads)
#pragma unroll
    for (int c = 0; c < DD; ++c)
#pragma unroll
      for (int j = 0; j <= c; ++j) lb[c][j] = q.P[c * d + j];
    // column i of L below the diagonal (i = lane: selected by comparison, zero above): L[c][i], c >= i
    double lci[DD];
#pragma unroll
    for (int c = 0; c < DD; ++c) {
      double v = 0.0;
#pragma unroll
      for (int ii = 0; ii <= c; ++ii) v = (ii == i) ? Lr[c][ii] : v;
      lci[c] = v;
    }
    double x[DD];
#pragma unroll
    for (int j = 0; j < DD; ++j) {
      double acc = 0.0;
#pragma unroll
      for (int c = j; c < DD; ++c) acc += lci[c] * lb[c][j];   // (terms with c < i are exact zeros)
      if (j == i) acc *= 0.5;
      x[j] = (j <= i) ? acc : 0.0;
    }
#pragma unroll
    for (int j = DD - 1; j >= 0; --j) {
      double a = x[j];
#pragma unroll
      for (int c = j + 1; c < DD; ++c) a -= x[c] * Lr[c][j];
      x[j] = a * idg[j];
    }
#pragma unroll
    for (int j = 0; j < DD; ++j) q.Sb[i * d + j] = x[j];
  }
  pm_wave_sync();
  // Sbar = L^-T X: lane j solves L^T y = x_j (COLUMN j of X) in registers
  if (lane < d) {
    const int j = lane;
    double y[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) y[i] = q.Sb[i * d + j];
#pragma unroll
    for (int i = DD - 1; i >= 0; --i) {
      double a = y[i];
#pragma unroll
      for (int c = i + 1; c < DD; ++c) a -= Lr[c][i] * y[c];
      y[i] = a * idg[i];
    }
#pragma unroll
    for (int i = 0; i < DD; ++i) q.Sb[i * d + j] = y[i];
  }
  pm_wave_sync();
  for (int e = lane; e < d * d; e += 64) {
    const int i = e / d, j = e - i * d;
    q.P[e] = (q.Sb[i * d + j] + q.Sb[j * d + i]) * inv_m1;
  }
  pm_wave_sync();
}
template <int DD, class XF = PmNoXch>
__device__ __forceinline__ void pm_mm_bwd_l(const float* s, int s_ld, int M, const float* z, int z_ld,
                                            const float* g, int g_ld, float* gout, int gout_ld, double* scr,
                                            int lane, const double* fac, bool fac_in_scr, XF xf = XF{}) {
  constexpr int d = DD;
  const MMScratch q = pm_mm_carve(scr, d);
  if (!fac_in_scr) {
    for (int e = lane; e < (int)pm_mm_fac_doubles(d); e += 64) scr[e] = fac[e];
    pm_wave_sync();
  }
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  if constexpr (!std::is_same<XF, PmNoXch>::value) {
    const int gq = lane >> 4, c = lane & 15;
    pm_f64x4 H = {0.0, 0.0, 0.0, 0.0};
    (void)xf(H);
#pragma unroll
    for (int r = 0; r < (DD + 3) / 4; ++r) {
      const int i = gq + 4 * r;
      if (i < DD && c < DD) q.P[i * d + c] = c <= i ? H[r] : 0.0;
      if (i < DD && c == DD) q.mbar[i] = H[r];
    }
  } else {
  {
    constexpr int e2 = DD <= 1 ? 1 : DD <= 2 ? 2 : DD <= 4 ? 4 : 8;
    constexpr int P = 64 / e2;
    const int part = lane % P, j = lane / P;
    double a = 0.0;
    if (j < d)
      for (int r = part; r < M; r += P) a += (double)g[r * g_ld + j];
    a = pm_seg_sum(a, P);
    if (j < d && part == 0) q.mbar[j] = a;
  }
  {
    constexpr int e2 = DD * DD <= 1 ? 1 : DD * DD <= 4 ? 4 : DD * DD <= 16 ? 16 : DD * DD <= 32 ? 32 : 64;
    constexpr int P = 64 / e2;
    const int part = lane % P, e = lane / P;
    const int i = e / d, j = e - i * d;
    const bool live = e < d * d && j <= i;
    double acc = 0.0;
    if (live) {
      const double zm = q.zmean[j], zs = q.zistd[j];
      for (int r = part; r < M; r += P) acc += (double)g[r * g_ld + i] * (((double)z[r * z_ld + j] - zm) * zs);
    }
    acc = pm_seg_sum(acc, P);
    if (e < d * d && part == 0) q.P[e] = live ? acc : 0.0;
  }
  }
  pm_wave_sync();
  pm_mm_bwd_l_tail<DD>(q, lane, M);
  for (int e = lane; e < M * d; e += 64) {
    const int r = e / d, j = e - r * d;
    double acc = q.mbar[j] * inv_m;
#pragma unroll
    for (int c = 0; c < DD; ++c) acc += ((double)s[r * s_ld + c] - q.mean[c]) * q.P[c * d + j];
    gout[r * gout_ld + j] = (float)acc;
  }
  pm_wave_sync();
}


// ===========================================================================
// Compile-time-d variants (2d+1 <= 16) for the stand-alone moment-matching kernels (groups
// larger than a workgroup's rows: the reference examples' default, one group over all
// particles).  The sums over the group's rows run on the fp64 matrix core: with
// X = [s - s_0 | 1 | z] (one row per particle), ONE accumulator tile G = X^T X holds
// sum s s^T, sum s, M, sum z and sum z^2 (v_mfma_f64_16x16x4_f64, a = b = X[row 4kb + (lane>>4)]
// [col lane&15]; G[i][j] sits in lane ((i&3)<<4)|j, register i>>2).  The covariance entries stay
// in the lanes that hold them and the Cholesky factorisation runs across lanes: per pivot the
// pivot and its column are broadcast with v_readlane, every lane updates its own entries -- no
// LDS round trip and no wave barrier inside the factorisation.  Products of fp32 inputs are
// exact in fp64 and the rows are shifted by the group's first row, so the single-pass
// covariance (sum s s^T - M m m^T) loses nothing that matters.
// (Not used inside the sweep kernels: there the extra registers cost the other phases more than
// the shorter phase gains -- DESIGN.md section 6.)

__device__ __forceinline__ double pm_rl64(double v, int src_lane) {   // src_lane: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ double pm_sel(const double (&v)[N], int idx) {
  double o = v[0];
#pragma unroll
  for (int k = 1; k < N; ++k) {
    o = (idx == k) ? v[k] : o;
    // keep this a chain of v_cndmask: left alone the compiler turns it into a private array indexed
    // by lane -- a store / load round trip through scratch memory on the serial path of every pivot
    asm volatile("" : "+v"(o));
  }
  return o;
}

// COH: the rows were written by OTHER workgroups of this launch (device-wide barrier form of the
// sweeps): read them with device-scope loads, which do not hit this XCD's possibly stale L2 / L1 lines
template <bool COH>
__device__ __forceinline__ float pm_ldc(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
// Gram tile of X = [s - ref | 1 | z] over rows [r_lo, r_hi) of the group (ref: the group's first row)
template <int DD, bool COH = false>
__device__ __forceinline__ pm_f64x4 pm_mm_gram_rows(const float* s, int s_ld, const float* z, int z_ld,
                                                    int zrow0, int Bg, int r_lo, int r_hi, int lane,
                                                    double* ref_out = nullptr, const double* ref_in = nullptr) {
  // (ref_in: lane c < DD holds the reference point's column c instead -- sums that several workgroups add up)
  static_assert(2 * DD + 1 <= 16, "the Gram tile holds 2d+1 columns");
  const int g = lane >> 4, c = lane & 15;
  const int cs = c < DD ? c : 0;
  const int cz = (c > DD && c <= 2 * DD) ? c - DD - 1 : 0;
  const double ref = ref_in ? *ref_in : (double)pm_ldc<COH>(s + cs);
  if (ref_out) *ref_out = ref;     // lane j < DD: the group's first row, column j
  pm_f64x4 G = {0.0, 0.0, 0.0, 0.0};
  if (r_hi <= r_lo) return G;
  // rows r_lo + g, + 4, ...: the cyclic noise row advances with them (one modulo up front, then
  // a conditional wrap).  The rows of UB row quads are loaded together, then their MFMAs run: written
  // as two fixed-count inner loops because the compiler does not unroll the single loop (it left one
  // memory round trip per row quad: 80 k cycles for a 2500-row group)
  constexpr int UB = 8;
  int zr = pm_zidx(zrow0, min(r_lo + g, r_hi - 1), Bg);
  const int zwrap = Bg ? Bg : 0x7fffffff;
  for (int r0 = r_lo; r0 < r_hi; r0 += 4 * UB) {
    float sv[UB], zv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + 4 * u + g, rr = r < r_hi ? r : r_hi - 1;
      sv[u] = pm_ldc<COH>(s + (size_t)rr * s_ld + cs);
      zv[u] = z[(size_t)zr * z_ld + cz];
      if (r + 4 < r_hi) {
        zr += 4;
        if (zr >= zwrap) zr -= zwrap;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + 4 * u + g;
      double x = c < DD ? (double)sv[u] - ref : (c == DD ? 1.0 : (c <= 2 * DD ? (double)zv[u] : 0.0));
      if (r >= r_hi) x = 0.0;
      if (r0 + 4 * u < r_hi) G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G, 0, 0, 0);
    }
  }
  return G;
}

// means / standardisation / Cholesky factor from the group's Gram tile -> q (LDS scratch of this wave)
// ref_lane: in lane j < DD the reference row's column j (what pm_mm_gram_rows subtracted)
template <int DD>
__device__ __forceinline__ bool pm_mm_factor_from_gram_ref(pm_f64x4 G, int M, double ref_lane, const MMScratch& q,
                                                           int lane) {
  constexpr int NR = (DD + 3) / 4;             // accumulator registers that hold covariance rows
  const int g = lane >> 4, c = lane & 15;
  const double dM = (double)M, inv_m = 1.0 / dM, inv_m1 = 1.0 / (double)(M - 1);
  // wave-uniform first moments and the z standardisation
  double sm[DD], zm[DD], zi[DD];
#pragma unroll
  for (int j = 0; j < DD; ++j) {
    constexpr int rowS = DD;
    const int rowZ = DD + 1 + j;
    sm[j] = pm_rl64(G[rowS >> 2], ((rowS & 3) << 4) | j) * inv_m;
    zm[j] = pm_rl64(G[rowS >> 2], ((rowS & 3) << 4) | (DD + 1 + j)) * inv_m;
    const double szz = pm_rl64(G[rowZ >> 2], ((rowZ & 3) << 4) | rowZ);
    zi[j] = pm_rsqrt((szz - dM * zm[j] * zm[j]) * inv_m1);
  }
  // covariance entry (i = g + 4r, j = c) in the lane that holds the Gram entry
  double A[NR];
  const double mj = pm_sel(sm, c);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = g + 4 * r;
    double mi = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (4 * r + u < DD) mi = (g == u) ? sm[4 * r + u] : mi;
    const double cov = (G[r] - dM * mi * mj) * inv_m1 + (i == c ? 1e-12 : 0.0);
    A[r] = (i < DD && c <= i) ? cov : 0.0;
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DD; ++k) {
    double piv = pm_rl64(A[k >> 2], ((k & 3) << 4) | k);
    // same rule as pm_mm_factor: a pivot that has shed more than fp32 precision relative to its
    // diagonal entry (= pivot + the squares eliminated so far) counts as non-positive
    double d0 = piv;
#pragma unroll
    for (int c2 = 0; c2 < k; ++c2) {
      const double l = pm_rl64(A[k >> 2], ((k & 3) << 4) | c2);
      d0 += l * l;
    }
    if (!(piv > 6e-8 * d0)) {
      ok = false;
      piv = 1.0;
    }
    const double rs = pm_rsqrt(piv);
    const double lkk = piv * rs;
    double col[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) col[i] = i > k ? pm_rl64(A[i >> 2], ((i & 3) << 4) | k) * rs : 0.0;
    const double lj = pm_sel(col, c);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = g + 4 * r;
      double li = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * r + u < DD) li = (g == u) ? col[4 * r + u] : li;
      if (i < DD) {
        if (c == k) A[r] = (i == k) ? lkk : (i > k ? li : A[r]);
        else if (c > k && c <= i) A[r] -= li * lj;
      }
    }
    if (lane == k) q.invd[k] = rs;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = g + 4 * r;
    if (i < DD && c < DD) q.Lm[i * DD + c] = A[r];
  }
  if (lane < DD) {
    q.mean[lane] = pm_sel(sm, lane) + ref_lane;
    q.zmean[lane] = pm_sel(zm, lane);
    q.zistd[lane] = pm_sel(zi, lane);
  }
  pm_wave_sync();
  return ok;
}
template <int DD, bool COH = false>
__device__ __forceinline__ bool pm_mm_factor_from_gram(pm_f64x4 G, int M, const float* s, const MMScratch& q,
                                                       int lane) {
  return pm_mm_factor_from_gram_ref<DD>(G, M, (double)pm_ldc<COH>(s + (lane < DD ? lane : 0)), q, lane);
}
template <int DD>
__device__ __forceinline__ bool pm_mm_factor_t(const float* s, int s_ld, int M, const float* z, int z_ld,
                                               int zrow0, int Bg, const MMScratch& q, int lane) {
  return pm_mm_factor_from_gram<DD>(pm_mm_gram_rows<DD>(s, s_ld, z, z_ld, zrow0, Bg, 0, M, lane), M, s, q, lane);
}

template <int DD>
__device__ __forceinline__ bool pm_mm_fwd_t(const float* s, int s_ld, int M, const float* z, int z_ld,
                                            int zrow0, int Bg, float* out, int out_ld, double* scr, int lane) {
  const MMScratch q = pm_mm_carve(scr, DD);
  const bool ok = pm_mm_factor_t<DD>(s, s_ld, M, z, z_ld, zrow0, Bg, q, lane);
  for (int e = lane; e < M * DD; e += 64) {
    const int r = e / DD, j = e - r * DD;
    double acc = q.mean[j];
    const size_t zr = (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
    for (int c = 0; c <= j; ++c)
      acc += ((double)z[zr + c] - q.zmean[c]) * q.zistd[c] * q.Lm[j * DD + c];
    out[(size_t)r * out_ld + j] = (float)acc;
  }
  pm_wave_sync();
  return ok;
}

// adjoint's d x d algebra on the scratch: Lbar (q.P) -> P = (Sbar + Sbar^T) / (M - 1)  (q.P)
template <int DD>
__device__ __forceinline__ void pm_mm_bwd_tail(const MMScratch& q, int lane, int M) {
  const double inv_m1 = 1.0 / (double)(M - 1);
  // Phi = tril(L^T Lbar), diagonal halved -> q.Sb
  for (int e = lane; e < DD * DD; e += 64) {
    const int i = e / DD, j = e - i * DD;
    double acc = 0.0;
    if (j <= i) {
      for (int c = i; c < DD; ++c) acc += q.Lm[c * DD + i] * q.P[c * DD + j];
      if (i == j) acc *= 0.5;
    }
    q.Sb[e] = acc;
  }
  pm_wave_sync();
  // X = Phi L^-1  (row i of X solves x L = phi_i), in place in q.Sb
  for (int i = lane; i < DD; i += 64) {
    for (int j = DD - 1; j >= 0; --j) {
      double a = q.Sb[i * DD + j];
      for (int c = j + 1; c < DD; ++c) a -= q.Sb[i * DD + c] * q.Lm[c * DD + j];
      q.Sb[i * DD + j] = a * q.invd[j];
    }
  }
  pm_wave_sync();
  // Sbar = L^-T X  (column j solves L^T y = x_j), in place
  for (int j = lane; j < DD; j += 64) {
    for (int i = DD - 1; i >= 0; --i) {
      double a = q.Sb[i * DD + j];
      for (int c = i + 1; c < DD; ++c) a -= q.Lm[c * DD + i] * q.Sb[c * DD + j];
      q.Sb[i * DD + j] = a * q.invd[i];
    }
  }
  pm_wave_sync();
  for (int e = lane; e < DD * DD; e += 64) {
    const int i = e / DD, j = e - i * DD;
    q.P[e] = (q.Sb[i * DD + j] + q.Sb[j * DD + i]) * inv_m1;
  }
  pm_wave_sync();
}

template <int DD>
__device__ __forceinline__ void pm_mm_bwd_t(const float* s, int s_ld, int M, const float* z, int z_ld,
                                            int zrow0, int Bg, const float* g, int g_ld, float* gout,
                                            int gout_ld, double* scr, int lane) {
  constexpr int NR = (DD + 3) / 4;
  const MMScratch q = pm_mm_carve(scr, DD);
  (void)pm_mm_factor_t<DD>(s, s_ld, M, z, z_ld, zrow0, Bg, q, lane);
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  // H = g^T [z | 1]  ->  mbar = sum_r g,  Lbar = tril(g^T zhat)
  {
    const int gq = lane >> 4, c = lane & 15;
    const int cc = c < DD ? c : 0;
    pm_f64x4 H = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int r0 = 0; r0 < M; r0 += 4) {
      const int r = r0 + gq, rr = r < M ? r : M - 1;
      const double gv = (double)g[(size_t)rr * g_ld + cc];
      const double zv = (double)z[(size_t)pm_zidx(zrow0, rr, Bg) * z_ld + cc];
      const double a = (c < DD && r < M) ? gv : 0.0;
      const double b = r < M ? (c < DD ? zv : (c == DD ? 1.0 : 0.0)) : 0.0;
      H = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H, 0, 0, 0);
    }
    double mb[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) mb[i] = pm_rl64(H[i >> 2], ((i & 3) << 4) | DD);
    const double zmc = q.zmean[cc], zsc = q.zistd[cc];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = gq + 4 * r;
      double mi = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * r + u < DD) mi = (gq == u) ? mb[4 * r + u] : mi;
      if (i < DD && c < DD) q.P[i * DD + c] = (c <= i) ? (H[r] - zmc * mi) * zsc : 0.0;
    }
    if (lane < DD) q.mbar[lane] = pm_sel(mb, lane);
  }
  pm_wave_sync();
  pm_mm_bwd_tail<DD>(q, lane, M);
  for (int e = lane; e < M * DD; e += 64) {
    const int r = e / DD, j = e - r * DD;
    double acc = q.mbar[j] * inv_m;
    for (int c = 0; c < DD; ++c) acc += ((double)s[(size_t)r * s_ld + c] - q.mean[c]) * q.P[c * DD + j];
    gout[(size_t)r * gout_ld + j] = (float)acc;
  }
  pm_wave_sync();
}

// Rows [o_lo, o_hi) ... see pm_mm_fwd_rows / pm_mm_bwd_rows.
// --- multi-wave forms (one workgroup of NW waves per group): every wave takes a slice of the
// group's rows for the row sums (partial Gram tiles meet in LDS, added in wave order by every
// wave -> identical statistics everywhere, no further exchange) and for the per-row outputs; the
// d x d algebra in between is repeated by every wave in its own scratch.  `part` holds NW * 256
// doubles.  A group of any size costs a few microseconds instead of a walk by one wave.
__device__ __forceinline__ pm_f64x4 pm_mm_sum_parts(pm_f64x4 mine, double* part, int nw, int wid, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) part[((size_t)wid * 64 + lane) * 4 + r] = mine[r];
  __syncthreads();
  pm_f64x4 t = {0.0, 0.0, 0.0, 0.0};
  for (int w = 0; w < nw; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] += part[((size_t)w * 64 + lane) * 4 + r];
  return t;
}
__device__ __forceinline__ void pm_mm_slice(int M, int nw, int wid, int& r_lo, int& r_hi) {
  const int per = ((M + nw - 1) / nw + 3) & ~3;
  r_lo = min(M, wid * per);
  r_hi = min(M, r_lo + per);
}
// Rows [o_lo, o_hi) of the group are written by THIS wave, to out[(r - o_shift) * out_ld + j].
template <int DD, bool COH = false>
__device__ __forceinline__ bool pm_mm_fwd_rows(const float* s, int s_ld, int M, const float* z, int z_ld,
                                               int zrow0, int Bg, float* out, int out_ld, int o_lo, int o_hi,
                                               int o_shift, double* scr, double* part, int nw, int wid,
                                               int lane) {
  int r_lo, r_hi;
  pm_mm_slice(M, nw, wid, r_lo, r_hi);
  // the noise row of this lane's first output element: loaded with the Gram's rows, not after the algebra
  const int e0 = o_lo * DD + lane;
  float zfirst[DD];
  {
    const int r = (e0 < o_hi * DD) ? e0 / DD : 0;
    const size_t zr = (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
#pragma unroll
    for (int c = 0; c < DD; ++c) zfirst[c] = z[zr + c];
  }
  double ref = 0.0;
  const pm_f64x4 G = pm_mm_sum_parts(pm_mm_gram_rows<DD, COH>(s, s_ld, z, z_ld, zrow0, Bg, r_lo, r_hi, lane, &ref),
                                     part, nw, wid, lane);
  const MMScratch q = pm_mm_carve(scr + (size_t)wid * pm_mm_scratch_doubles(DD), DD);
  const bool ok = pm_mm_factor_from_gram_ref<DD>(G, M, ref, q, lane);
  for (int e = e0; e < o_hi * DD; e += 64) {
    const int r = e / DD, j = e - r * DD;
    double acc = q.mean[j];
    const size_t zr = (size_t)pm_zidx(zrow0, r, Bg) * z_ld;
#pragma unroll
    for (int c = 0; c < DD; ++c) {
      const float zc = (e == e0) ? zfirst[c] : z[zr + c];
      if (c <= j) acc += ((double)zc - q.zmean[c]) * q.zistd[c] * q.Lm[j * DD + c];
    }
    out[(size_t)(r - o_shift) * out_ld + j] = (float)acc;
  }
  return ok;
}
template <int DD>
__device__ __forceinline__ bool pm_mm_fwd_mw(const float* s, int s_ld, int M, const float* z, int z_ld,
                                             int zrow0, int Bg, float* out, int out_ld, double* scr,
                                             double* part, int nw, int wid, int lane) {
  int r_lo, r_hi;
  pm_mm_slice(M, nw, wid, r_lo, r_hi);
  return pm_mm_fwd_rows<DD>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, r_lo, r_hi, 0, scr, part, nw, wid, lane);
}
// Gram tile (as pm_mm_gram_rows) and H = g^T [z | 1] over the same rows in ONE pass: the rows of s, z
// and g are in flight together (one memory round trip instead of two), same accumulation order
template <int DD, bool COH>
__device__ __forceinline__ void pm_mm_gram_h_rows(const float* s, int s_ld, const float* z, int z_ld, int zrow0,
                                                  int Bg, const float* g, int g_ld, int r_lo, int r_hi, int lane,
                                                  pm_f64x4& G, pm_f64x4& H, double* ref_out) {
  const int gq = lane >> 4, c = lane & 15;
  const int cs = c < DD ? c : 0;
  const int cz = (c > DD && c <= 2 * DD) ? c - DD - 1 : 0;
  const double ref = (double)s[cs];
  *ref_out = ref;
  G = pm_f64x4{0.0, 0.0, 0.0, 0.0};
  H = pm_f64x4{0.0, 0.0, 0.0, 0.0};
  if (r_hi <= r_lo) return;
  int zr = pm_zidx(zrow0, min(r_lo + gq, r_hi - 1), Bg);
  const int zwrap = Bg ? Bg : 0x7fffffff;
  constexpr int UB = 4;        // row quads whose loads are in flight together (see pm_mm_gram_rows)
  for (int r0 = r_lo; r0 < r_hi; r0 += 4 * UB) {
    float sv[UB], zv[UB], zh[UB], gv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + 4 * u + gq, rr = r < r_hi ? r : r_hi - 1;
      sv[u] = s[(size_t)rr * s_ld + cs];
      zv[u] = z[(size_t)zr * z_ld + cz];
      zh[u] = z[(size_t)zr * z_ld + cs];
      gv[u] = pm_ldc<COH>(g + (size_t)rr * g_ld + cs);
      if (r + 4 < r_hi) {
        zr += 4;
        if (zr >= zwrap) zr -= zwrap;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int r = r0 + 4 * u + gq;
      double x = c < DD ? (double)sv[u] - ref : (c == DD ? 1.0 : (c <= 2 * DD ? (double)zv[u] : 0.0));
      if (r >= r_hi) x = 0.0;
      const double a = (c < DD && r < r_hi) ? (double)gv[u] : 0.0;
      const double b = r < r_hi ? (c < DD ? (double)zh[u] : (c == DD ? 1.0 : 0.0)) : 0.0;
      if (r0 + 4 * u < r_hi) {
        G = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, G, 0, 0, 0);
        H = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, H, 0, 0, 0);
      }
    }
  }
}
template <int DD, bool COH = false>   // COH: the carried gradient g comes from other workgroups of this launch
__device__ __forceinline__ void pm_mm_bwd_rows(const float* s, int s_ld, int M, const float* z, int z_ld,
                                               int zrow0, int Bg, const float* g, int g_ld, float* gout,
                                               int gout_ld, int o_lo, int o_hi, int o_shift, double* scr,
                                               double* part, int nw, int wid, int lane) {
  constexpr int NR = (DD + 3) / 4;
  int r_lo, r_hi;
  pm_mm_slice(M, nw, wid, r_lo, r_hi);
  // this lane's first output element needs its row of s: loaded with the statistics' rows
  const int e0 = o_lo * DD + lane;
  float sfirst[DD];
  {
    const int r = (e0 < o_hi * DD) ? e0 / DD : 0;
#pragma unroll
    for (int c = 0; c < DD; ++c) sfirst[c] = s[(size_t)r * s_ld + c];
  }
  pm_f64x4 G0, H;
  double ref = 0.0;
  pm_mm_gram_h_rows<DD, COH>(s, s_ld, z, z_ld, zrow0, Bg, g, g_ld, r_lo, r_hi, lane, G0, H, &ref);
  const pm_f64x4 G = pm_mm_sum_parts(G0, part, nw, wid, lane);
  const MMScratch q = pm_mm_carve(scr + (size_t)wid * pm_mm_scratch_doubles(DD), DD);
  (void)pm_mm_factor_from_gram_ref<DD>(G, M, ref, q, lane);
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  {
    const int gq = lane >> 4, c = lane & 15;
    const int cc = c < DD ? c : 0;
    __syncthreads();            // the Gram partials have been consumed; every read of g is done
    H = pm_mm_sum_parts(H, part, nw, wid, lane);
    double mb[DD];
#pragma unroll
    for (int i = 0; i < DD; ++i) mb[i] = pm_rl64(H[i >> 2], ((i & 3) << 4) | DD);
    const double zmc = q.zmean[cc], zsc = q.zistd[cc];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = gq + 4 * r;
      double mi = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * r + u < DD) mi = (gq == u) ? mb[4 * r + u] : mi;
      if (i < DD && c < DD) q.P[i * DD + c] = (c <= i) ? (H[r] - zmc * mi) * zsc : 0.0;
    }
    if (lane < DD) q.mbar[lane] = pm_sel(mb, lane);
  }
  pm_wave_sync();
  pm_mm_bwd_tail<DD>(q, lane, M);
  for (int e = e0; e < o_hi * DD; e += 64) {
    const int r = e / DD, j = e - r * DD;
    double acc = q.mbar[j] * inv_m;
#pragma unroll
    for (int c = 0; c < DD; ++c) {
      const float sc = (e == e0) ? sfirst[c] : s[(size_t)r * s_ld + c];
      acc += ((double)sc - q.mean[c]) * q.P[c * DD + j];
    }
    gout[(size_t)(r - o_shift) * gout_ld + j] = (float)acc;
  }
  (void)inv_m1;
}

template <int DD>
__device__ __forceinline__ void pm_mm_bwd_mw(const float* s, int s_ld, int M, const float* z, int z_ld,
                                             int zrow0, int Bg, const float* g, int g_ld, float* gout,
                                             int gout_ld, double* scr, double* part, int nw, int wid,
                                             int lane) {
  int r_lo, r_hi;
  pm_mm_slice(M, nw, wid, r_lo, r_hi);
  pm_mm_bwd_rows<DD>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, r_lo, r_hi, 0, scr, part, nw, wid,
                     lane);
}
// d -> template dispatch; the general code for widths without an instantiation
__device__ __forceinline__ bool pm_mm_fwd_auto(const float* s, int s_ld, int M, int d, const float* z,
                                               int z_ld, int zrow0, int Bg, float* out, int out_ld,
                                               double* scr, int lane) {
  switch (d) {
    case 1: return pm_mm_fwd_t<1>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    case 2: return pm_mm_fwd_t<2>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    case 3: return pm_mm_fwd_t<3>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    case 4: return pm_mm_fwd_t<4>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    case 5: return pm_mm_fwd_t<5>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    case 6: return pm_mm_fwd_t<6>(s, s_ld, M, z, z_ld, zrow0, Bg, out, out_ld, scr, lane);
    default: break;
  }
  return pm_mm_fwd(s, s_ld, M, d, z, z_ld, zrow0, Bg, false, out, out_ld, scr, lane);
}
__device__ __forceinline__ void pm_mm_bwd_auto(const float* s, int s_ld, int M, int d, const float* z,
                                               int z_ld, int zrow0, int Bg, const float* g, int g_ld,
                                               float* gout, int gout_ld, double* scr, int lane) {
  switch (d) {
    case 1: return pm_mm_bwd_t<1>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    case 2: return pm_mm_bwd_t<2>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    case 3: return pm_mm_bwd_t<3>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    case 4: return pm_mm_bwd_t<4>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    case 5: return pm_mm_bwd_t<5>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    case 6: return pm_mm_bwd_t<6>(s, s_ld, M, z, z_ld, zrow0, Bg, g, g_ld, gout, gout_ld, scr, lane);
    default: break;
  }
  pm_mm_bwd(s, s_ld, M, d, z, z_ld, zrow0, Bg, false, g, g_ld, gout, gout_ld, scr, lane);
}
