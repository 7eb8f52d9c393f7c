// Stand-alone evaluation of one Bayesian MLP + diagonal-Gaussian head on B independent rows
// (include/pmbrl.h: pmbrl_mlp_forward; reference: models/core.py:169-187, 221-248, 265-303,
// models/densities.py:87-121).  Same MFMA tile routines and fragment-packed weights as the
// rollout kernels; a workgroup (4 waves) owns 16 rows and walks the layers through two LDS
// activation buffers.  This is the acting / model-evaluation path (a handful of rows per call),
// not a throughput kernel.
#pragma once
#include "pmbrl_dev.h"

struct MlpArgs {
  int B, nl, LD, n_out;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  const float* wf[PM_MAXL];
  const float* bias[PM_MAXL];
  const uint16_t* mask[PM_MAXL];    // bit rows [B][nt[l+1]] or nullptr
  float keep[PM_MAXL];
  const float *x, *z, *in_shift, *in_iscale, *out_scale, *out_shift, *sq_scale, *sq_bias;
  float mls;
  float *sample, *mean, *log_std;
};

struct EpiMlpHidden {
  const float* bias;
  const uint16_t* mask;
  float keep;
  float* lds_out;
  int ld, row0, nvalid, nt, lane;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int g = lane >> 4;
    const int lrow = rt * 16 + (lane & 15);
    const int f0 = ot * 16 + 4 * g;
    const f32x4 b = ldg4(bias + f0);
    unsigned nib = 0xFu;
    if (mask) {
      unsigned mw = 0;
      if (lrow < nvalid) mw = mask[(size_t)(row0 + lrow) * nt + ot];
      nib = (mw >> (4 * g)) & 0xFu;
    }
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[r] + b[r];
      const bool a = ((nib >> r) & 1u) && (v > 0.f);
      h[r] = a ? (keep == 1.f ? v : v / keep) : 0.f;
    }
    *reinterpret_cast<f32x4*>(lds_out + lrow * ld + f0) = h;
  }
};

__global__ __launch_bounds__(PM_NT, 1) void pm_mlp_fwd_kernel(const MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * R;
  const int nvalid = min(R, A.B - row0);
  const int LD = A.LD;
  float* X = smem;
  float* Y = smem + (size_t)R * LD;
  float* part = smem + 2 * (size_t)R * LD;
  const int K0 = A.dim[0];
  for (int i = tid; i < R * A.nt[0] * 16; i += PM_NT) {
    const int r = i / (A.nt[0] * 16), k = i - r * (A.nt[0] * 16);
    float v = 0.f;
    if (r < nvalid && k < K0) {
      v = A.x[(size_t)(row0 + r) * K0 + k];
      if (A.in_shift) v = (v - A.in_shift[k]) * A.in_iscale[k];
    }
    X[r * LD + k] = v;
  }
  __syncthreads();
  for (int l = 0; l < A.nl - 1; ++l) {
    EpiMlpHidden e{A.bias[l], A.mask[l], A.keep[l], Y, LD, row0, nvalid, A.nt[l + 1], lane};
    gemm_tiles<1>(A.wf[l], A.nt[l + 1], A.nt[l], X, LD, wid, lane, e);
    __syncthreads();
    float* t = X; X = Y; Y = t;
  }
  gemm_narrow<1>(A.wf[A.nl - 1], A.nt[A.nl], A.nt[A.nl - 1], A.bias[A.nl - 1], X, Y, LD, part, wid, lane, tid);
  const int n = A.n_out;
  for (int i = tid; i < nvalid * n; i += PM_NT) {
    const int r = i / n, j = i - r * n;
    float mu = Y[r * LD + j];
    float ls = Y[r * LD + n + j];
    ls = -softplusf(-ls + A.mls) + A.mls;            // models/densities.py:97-98
    if (A.out_scale) {
      ls += logf(A.out_scale[j]);
      mu = mu * A.out_scale[j] + A.out_shift[j];
    }
    const size_t o = (size_t)(row0 + r) * n + j;
    float s = mu;
    if (A.z) s = mu + A.z[o] * expf(ls);
    if (A.sq_scale) s = A.sq_scale[j] * tanhf(s) + A.sq_bias[j];
    if (A.sample) A.sample[o] = s;
    if (A.mean) A.mean[o] = mu;
    if (A.log_std) A.log_std[o] = ls;
  }
}
