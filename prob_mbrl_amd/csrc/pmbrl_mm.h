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

__device__ __forceinline__ bool pm_mm_fwd(const float* s, int s_ld, int M, int d, const float* z, int z_ld,
                                 int zrow0, int Bg, bool infer_ns, float* out, int out_ld,
                                 double* scr, int lane, double* fac_out = nullptr) {
  (void)infer_ns;   // value of the infer_ns variant equals s; not offered on the device path
  const MMScratch q = pm_mm_carve(scr, d);
  const bool ok = pm_mm_factor(s, s_ld, M, d, z, z_ld, zrow0, Bg, q, lane);
  // the adjoint needs the same means / standardisation / factor: hand them over instead of
  // having it redo the statistics and the factorisation (pm_mm_fac_doubles values)
  if (fac_out)
    for (int e = lane; e < (int)pm_mm_fac_doubles(d); e += 64) fac_out[e] = scr[e];
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

// g: upstream dL/d out [M][d]; gout: dL/d s [M][d] (may alias g).
__device__ __forceinline__ void pm_mm_bwd(const float* s, int s_ld, int M, int d, const float* z, int z_ld,
                                 int zrow0, int Bg, bool infer_ns, const float* g, int g_ld,
                                 float* gout, int gout_ld, double* scr, int lane,
                                 const double* fac = nullptr) {
  (void)infer_ns;
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
  // sbar[r][j] = sum_c Delta[r][c] P[c][j] + mbar[j]/M      (mean_r of the first term is 0)
  for (int e = lane; e < M * d; e += 64) {
    const int r = e / d, j = e - r * d;
    double acc = q.mbar[j] * inv_m;
    for (int c = 0; c < d; ++c) acc += ((double)s[r * s_ld + c] - q.mean[c]) * q.P[c * d + j];
    // all reads of g happened before the first pm_wave_sync above: in-place is safe
    gout[r * gout_ld + j] = (float)acc;
  }
  pm_wave_sync();
}

