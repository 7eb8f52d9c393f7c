// Stand-alone evaluation of one Bayesian MLP + diagonal-Gaussian head on B independent rows
// (include/pmbrl.h: pmbrl_mlp_forward; reference: models/core.py:169-187, 221-248, 265-303,
// models/densities.py:87-121).  Same MFMA tile routines and fragment-packed weights as the
// rollout kernels; a workgroup (PM_NW waves) owns 16 rows and walks the layers through two LDS
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

// ---------------------------------------------------------------------------
// gradient of the outputs of pm_mlp_fwd_kernel with respect to its INPUT rows (the network is
// constant): what autograd needs when a stand-alone network sits inside a differentiable
// computation, e.g. the terminal value V(x_H) of algorithms/mc_pilco.py:136-140.  The forward
// is recomputed in the same launch (layer inputs stay in LDS), then the dX chain runs back.
// ---------------------------------------------------------------------------
struct MlpBwdArgs {
  MlpArgs f;                       // the forward's arguments (outputs unused)
  const float* wb[PM_MAXL];        // transposed fragments
  const float *g_sample, *g_mean, *g_log_std;   // [B][n_out], any may be null
  float* grad_x;                   // [B][n_in]
};

struct EpiMlpBwd {
  const float* h_in;               // that layer's output (zero where masked / inactive)
  float inv_keep;
  float* g_out;
  int ld, lane;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int g = lane >> 4, lrow = lane & 15;
    const int f0 = ot * 16 + 4 * g;
    const f32x4 h = *reinterpret_cast<const f32x4*>(h_in + lrow * ld + f0);
    f32x4 gp;
#pragma unroll
    for (int r = 0; r < 4; ++r) gp[r] = h[r] > 0.f ? acc[r] * inv_keep : 0.f;
    *reinterpret_cast<f32x4*>(g_out + lrow * ld + f0) = gp;
  }
};

__host__ __device__ inline size_t pm_mlp_bwd_lds_floats(int nl, int LD) {
  return ((size_t)nl + 2) * 16 * LD + (size_t)PM_NW * PM_KS_NT * 256;   // H[0..nl-1], two gradient buffers, K-split scratch
}

__global__ __launch_bounds__(PM_NT, 1) void pm_mlp_bwdx_kernel(const MlpBwdArgs B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16;
  const MlpArgs& A = B.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * R;
  const int nvalid = min(R, A.B - row0);
  const int LD = A.LD, nl = A.nl;
  float* H = smem;                                   // [nl][16][LD]
  float* G0 = H + (size_t)nl * R * LD;
  float* G1 = G0 + (size_t)R * LD;
  float* part = G1 + (size_t)R * LD;
  const int K0 = A.dim[0];
  for (int i = tid; i < R * A.nt[0] * 16; i += PM_NT) {
    const int r = i / (A.nt[0] * 16), k = i - r * (A.nt[0] * 16);
    float v = 0.f;
    if (r < nvalid && k < K0) {
      v = A.x[(size_t)(row0 + r) * K0 + k];
      if (A.in_shift) v = (v - A.in_shift[k]) * A.in_iscale[k];
    }
    H[r * LD + k] = v;
  }
  __syncthreads();
  for (int l = 0; l < nl - 1; ++l) {
    EpiMlpHidden e{A.bias[l], A.mask[l], A.keep[l], H + (size_t)(l + 1) * R * LD, LD, row0, nvalid, A.nt[l + 1], lane};
    gemm_tiles<1>(A.wf[l], A.nt[l + 1], A.nt[l], H + (size_t)l * R * LD, LD, wid, lane, e);
    __syncthreads();
  }
  gemm_narrow<1>(A.wf[nl - 1], A.nt[nl], A.nt[nl - 1], A.bias[nl - 1], H + (size_t)(nl - 1) * R * LD, G1, LD, part,
                 wid, lane, tid);
  // gradient wrt the raw head outputs [mu | log_std]
  {
    const int n = A.n_out, W16 = A.nt[nl] * 16;
    for (int i = tid; i < R * W16; i += PM_NT) {
      const int r = i / W16, j = i - r * W16;
      float gv = 0.f;
      if (r < nvalid && j < 2 * n) {
        const int d = j < n ? j : j - n;
        const size_t o = (size_t)(row0 + r) * n + d;
        const float mu0 = G1[r * LD + d], ls0 = G1[r * LD + n + d];
        float ls = -softplusf(-ls0 + A.mls) + A.mls;
        float mu = mu0, osc = 1.f;
        if (A.out_scale) {
          osc = A.out_scale[d];
          ls += logf(osc);
          mu = mu0 * osc + A.out_shift[d];
        }
        const float zz = A.z ? A.z[o] : 0.f;
        const float e = expf(ls);
        float gs = B.g_sample ? B.g_sample[o] : 0.f;
        if (A.sq_scale && B.g_sample) {
          const float th = tanhf(mu + zz * e);
          gs *= A.sq_scale[d] * (1.f - th * th);
        }
        if (j < n) gv = (gs + (B.g_mean ? B.g_mean[o] : 0.f)) * osc;
        else gv = (gs * zz * e + (B.g_log_std ? B.g_log_std[o] : 0.f)) * sigmoidf(A.mls - ls0);
      }
      G0[r * LD + j] = gv;
    }
  }
  __syncthreads();
  float* Gin = G0;
  float* Gout = G1;
  for (int l = nl - 1; l >= 1; --l) {
    EpiMlpBwd e{H + (size_t)l * R * LD, 1.f / A.keep[l - 1], Gout, LD, lane};
    gemm_tiles<1>(B.wb[l], A.nt[l], A.nt[l + 1], Gin, LD, wid, lane, e);
    __syncthreads();
    float* t = Gin; Gin = Gout; Gout = t;
  }
  // first layer: d / d x_in, then the input normalisation
  {
    EpiPlain e{nullptr, Gout, LD, lane};
    gemm_tiles<1>(B.wb[0], A.nt[0], A.nt[1], Gin, LD, wid, lane, e);
    __syncthreads();
    for (int i = tid; i < nvalid * K0; i += PM_NT) {
      const int r = i / K0, k = i - r * K0;
      float v = Gout[r * LD + k];
      if (A.in_iscale) v *= A.in_iscale[k];
      B.grad_x[(size_t)(row0 + r) * K0 + k] = v;
    }
  }
}
