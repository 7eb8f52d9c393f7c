// BNN maximum-likelihood training step of the dynamics model: loss and gradient
// of one minibatch, the body of utils/train_regressor.py:113-131 with the model in train() mode
//   Regressor.forward(x, normalize=False, resample=True)        models/core.py:169-187
//   CDropout.forward, training branch (concrete relaxation,     models/modules.py:102-118,120-160
//     straight-through Bernoulli sample)
//   DiagGaussianDensity.forward (mean, clamped log-std)          models/densities.py:87-121
//   losses.gaussian_log_likelihood                               losses.py:16-37
//   CDropout / BSequential regularisers                          models/modules.py:88-93,30-35,234-274
//
// pm_bnn_fwd_bwd: a workgroup (PM_NW waves) owns 16 minibatch rows; all layer inputs and the
// dropout derivative terms stay in LDS between the forward and the backward pass; the
// pre-activation gradients and layer inputs are stashed feature-major, exactly the layout the
// policy-gradient dW GEMM (pmbrl_dw.h) consumes, so dW / db come from the same kernel.
// pm_bnn_finish adds the regulariser, reduces the dropout-logit gradients and the loss.
#pragma once
#include "pmbrl_dev.h"

struct BnnArgs {
  int M, nl, LD, n_out, nwg, n_in, sum_h;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  const float* wf[PM_MAXL];
  const float* wb[PM_MAXL];
  const float* bias[PM_MAXL];
  const float* logit_p[PM_MAXL];   // hidden layer l: [dim[l+1]] or nullptr (no dropout)
  const float* u[PM_MAXL];         // [M][dim[l+1]] uniform noise of the concrete relaxation
  const float* bvar[PM_MAXL];      // [M][dim[l+1]] uniform variate of the Bernoulli draw (hard = bvar < probs)
  float inv_temp[PM_MAXL];
  int lp_off[PM_MAXL];             // offset of layer l's logits in the concatenated logit vector
  const float *X, *Y;              // normalised dataset [N][n_in], [N][n_out]
  const int* idx;                  // [M] minibatch rows
  float mls, inv_M;
  int mse;                         // loss: mean squared error of the mean head instead of the Gaussian NLL
  int gmm_n;                       // > 1: mixture-of-Gaussians NLL, head = [n D means | n D log-stds | n logits | log-T]
  const float* row_w;              // [M] per-row weight of the log-likelihood (importance sampling) or nullptr
  float* row_lp;                   // [M] out: per-row log-likelihood (unweighted) or nullptr
  float* actT[PM_MAXL];            // stash: input of layer l   [wg][nt[l]*16][16]
  float* gT[PM_MAXL];              // stash: grad wrt pre-activation of layer l [wg][nt[l+1]*16][16]
  float* part_lp;                  // [nwg][sum_h]
  float* part_loss;                // [nwg]
};

// hidden layer forward: out = relu(acc + b) * hard ;  q = relu(.) * d mask / d logit_p
struct EpiBnnFwd {
  const float* bias;
  const float* logit_p;
  const float *u, *bvar;
  float inv_temp;
  float *h_out, *q_out, *stash;
  int ld, row0, nvalid, width, lane;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int g = lane >> 4, lrow = lane & 15;
    const int f0 = ot * 16 + 4 * g;
    const f32x4 b = ldg4(bias + f0);
    f32x4 h, q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + r;
      const float v = fmaxf(acc[r] + b[r], 0.f);
      float hv = v, qv = 0.f;
      if (logit_p && f < width && lrow < nvalid) {
        const size_t o = (size_t)(row0 + lrow) * width + f;
        const float uu = u[o];
        const float cp = logit_p[f] + logf((uu + 1e-7f) / (1.f - (uu - 1e-7f)));
        const float pr = sigmoidf(cp * inv_temp);
        const bool hard = bvar[o] < pr;
        hv = hard ? v : 0.f;
        qv = v * pr * (1.f - pr) * inv_temp;
      }
      if (f >= width || lrow >= nvalid) { hv = 0.f; qv = 0.f; }
      h[r] = hv;
      q[r] = qv;
    }
    *reinterpret_cast<f32x4*>(h_out + lrow * ld + f0) = h;
    *reinterpret_cast<f32x4*>(q_out + lrow * ld + f0) = q;
#pragma unroll
    for (int r = 0; r < 4; ++r) stash[(size_t)(f0 + r) * 16 + lrow] = h[r];
  }
};

// backward through hidden layer l-1's output: acc = dL/d out ; gq = acc * q (dropout-logit term),
// g_pre = out > 0 ? acc : 0
struct EpiBnnBwd {
  const float *h_in, *q_in;   // out and q of that layer
  float *g_out, *gq_out, *stash;
  int ld, lane;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int g = lane >> 4, lrow = lane & 15;
    const int f0 = ot * 16 + 4 * g;
    const f32x4 h = *reinterpret_cast<const f32x4*>(h_in + lrow * ld + f0);
    const f32x4 q = *reinterpret_cast<const f32x4*>(q_in + lrow * ld + f0);
    f32x4 gp, gq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gp[r] = h[r] > 0.f ? acc[r] : 0.f;
      gq[r] = acc[r] * q[r];
    }
    *reinterpret_cast<f32x4*>(g_out + lrow * ld + f0) = gp;
    *reinterpret_cast<f32x4*>(gq_out + lrow * ld + f0) = gq;
#pragma unroll
    for (int r = 0; r < 4; ++r) stash[(size_t)(f0 + r) * 16 + lrow] = gp[r];
  }
};

__host__ __device__ inline size_t pm_bnn_lds_floats(int nl, int LD) {
  // H[0..nl-1], Q[0..nl-2], two gradient buffers, one gq buffer, K-split scratch, loss scratch
  return ((size_t)nl + (nl - 1) + 3) * 16 * LD + (size_t)PM_NW * PM_KS_NT * 256 + PM_NT;
}

__global__ __launch_bounds__(PM_NT, 1) void pm_bnn_fwd_bwd(const BnnArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int row0 = wg * R;
  const int nvalid = min(R, A.M - row0);
  const int LD = A.LD, nl = A.nl;
  float* H = smem;                                      // [nl][16][LD]
  float* Q = H + (size_t)nl * R * LD;                   // [nl-1][16][LD]
  float* G0 = Q + (size_t)(nl - 1) * R * LD;            // gradient ping
  float* G1 = G0 + (size_t)R * LD;                      // gradient pong
  float* GQ = G1 + (size_t)R * LD;
  float* part = GQ + (size_t)R * LD;
  float* red = part + (size_t)PM_NW * PM_KS_NT * 256;

  // ---- gather the minibatch rows (+ dW stash of the first layer's input)
  {
    const int K16 = A.nt[0] * 16;
    float* st = A.actT[0] + (size_t)wg * K16 * 16;
    for (int i = tid; i < R * K16; i += PM_NT) {
      const int k = i / R, r = i - k * R;
      float v = 0.f;
      if (r < nvalid && k < A.n_in) v = A.X[(size_t)A.idx[row0 + r] * A.n_in + k];
      H[r * LD + k] = v;
      st[(size_t)k * 16 + r] = v;
    }
  }
  __syncthreads();
  // ---- forward
  for (int l = 0; l < nl - 1; ++l) {
    EpiBnnFwd e{A.bias[l], A.logit_p[l], A.u[l], A.bvar[l], A.inv_temp[l],
                H + (size_t)(l + 1) * R * LD, Q + (size_t)l * R * LD,
                A.actT[l + 1] + (size_t)wg * A.nt[l + 1] * 16 * 16, LD, row0, nvalid, A.dim[l + 1], lane};
    gemm_tiles<1>(A.wf[l], A.nt[l + 1], A.nt[l], H + (size_t)l * R * LD, LD, wid, lane, e);
    __syncthreads();
  }
  gemm_narrow<1>(A.wf[nl - 1], A.nt[nl], A.nt[nl - 1], A.bias[nl - 1], H + (size_t)(nl - 1) * R * LD, G1, LD, part,
                 wid, lane, tid);
  // ---- mixture negative log-likelihood (losses.py:40-64; head parametrisation models/densities.py:173-207):
  //      ll = logsumexp_c [log_softmax(logit / T)_c - sum_d lsc_dc - D/2 log 2 pi - 1/2 sum_d t_dc^2]
  if (A.gmm_n > 1) {
    const int n = A.gmm_n, D = A.n_out, nD = n * D, W16 = A.nt[nl] * 16;
    float* gst = A.gT[nl - 1] + (size_t)wg * W16 * 16;
    __shared__ float s_resp[16][PMBRL_MAX_COMP], s_sm[16][PMBRL_MAX_COMP], s_temp[16], s_ll[16];
    // responsibilities: one thread per row (n <= 8, D <= 64: a few hundred flops)
    if (tid < R) {
      const int r = tid;
      float ll = 0.f;
      if (r < nvalid) {
        const float* o = G1 + r * LD;
        const float* y = A.Y + (size_t)A.idx[row0 + r] * D;
        const float temp = 0.1f + softplusf(o[2 * nD + n]);
        float lg[PMBRL_MAX_COMP], lp[PMBRL_MAX_COMP], mx = -3.0e38f;
        for (int c = 0; c < n; ++c) { lg[c] = o[2 * nD + c] / temp; mx = fmaxf(mx, lg[c]); }
        float se = 0.f;
        for (int c = 0; c < n; ++c) se += expf(lg[c] - mx);
        const float lse = mx + logf(se);
        float mxp = -3.0e38f;
        for (int c = 0; c < n; ++c) {
          float sls = 0.f, sq = 0.f;
          for (int d = 0; d < D; ++d) {
            const float lsc = -softplusf(-o[nD + d * n + c] + A.mls) + A.mls;
            const float t = (o[d * n + c] - y[d]) * expf(-lsc);
            sls += lsc;
            sq = fmaf(t, t, sq);
          }
          lp[c] = (lg[c] - lse) + (-(float)D * 0.9189385332046727f - sls) - 0.5f * sq;
          s_sm[r][c] = expf(lg[c] - lse);
          mxp = fmaxf(mxp, lp[c]);
        }
        float sp = 0.f;
        for (int c = 0; c < n; ++c) sp += expf(lp[c] - mxp);
        ll = mxp + logf(sp);
        for (int c = 0; c < n; ++c) s_resp[r][c] = expf(lp[c] - ll);
        s_temp[r] = temp;
      }
      s_ll[r] = ll;
    }
    __syncthreads();
    float lsum = 0.f;
    for (int i = tid; i < R * W16; i += PM_NT) {
      const int r = i / W16, j = i - r * W16;
      float gval = 0.f;
      if (r < nvalid && j <= 2 * nD + n) {
        const float* o = G1 + r * LD;
        const float rw = (A.row_w ? A.row_w[row0 + r] : 1.f) * A.inv_M;
        if (j < 2 * nD) {
          const int jj = j < nD ? j : j - nD, d = jj / n, c = jj - d * n;
          const float ls = o[nD + jj];
          const float lsc = -softplusf(-ls + A.mls) + A.mls;
          const float sd = expf(-lsc);
          const float t = (o[jj] - A.Y[(size_t)A.idx[row0 + r] * D + d]) * sd;
          gval = j < nD ? rw * s_resp[r][c] * t * sd : rw * s_resp[r][c] * (1.f - t * t) * sigmoidf(A.mls - ls);
        } else if (j < 2 * nD + n) {
          const int c = j - 2 * nD;
          gval = -rw * (s_resp[r][c] - s_sm[r][c]) / s_temp[r];
        } else {
          float a = 0.f;
          for (int c = 0; c < n; ++c) a += (s_resp[r][c] - s_sm[r][c]) * o[2 * nD + c];
          gval = rw * a * sigmoidf(o[2 * nD + n]) / (s_temp[r] * s_temp[r]);
        }
        if (j == 0) lsum -= (A.row_w ? A.row_w[row0 + r] : 1.f) * s_ll[r];
      }
      G0[r * LD + j] = gval;
      gst[(size_t)j * 16 + r] = gval;
    }
    red[tid] = lsum;
    __syncthreads();
    for (int o = PM_NT / 2; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) A.part_loss[wg] = red[0];
    if (A.row_lp && tid < nvalid) A.row_lp[row0 + tid] = s_ll[tid];
  } else
  // ---- Gaussian negative log-likelihood and its gradient wrt the head outputs
  {
    const int n = A.n_out;
    float* gst = A.gT[nl - 1] + (size_t)wg * A.nt[nl] * 16 * 16;
    float lsum = 0.f;
    __shared__ float rowlp[16];
    if (tid < 16) rowlp[tid] = 0.f;
    __syncthreads();
    for (int i = tid; i < R * A.nt[nl] * 16; i += PM_NT) {
      const int r = i / (A.nt[nl] * 16), j = i - r * (A.nt[nl] * 16);
      float gval = 0.f;
      if (r < nvalid && j < 2 * n) {
        const int d = j < n ? j : j - n;
        const float mu = G1[r * LD + d];
        const float ls = G1[r * LD + n + d];
        if (A.mse) {
          // torch.nn.functional.mse_loss: mean over the M x n entries
          const float dl = mu - A.Y[(size_t)A.idx[row0 + r] * n + d];
          if (j < n) {
            gval = 2.f * dl * A.inv_M / (float)n;
            lsum += dl * dl / (float)n;
          }
        } else {
        const float lsc = -softplusf(-ls + A.mls) + A.mls;
        const float s = expf(-lsc);
        const float t = (mu - A.Y[(size_t)A.idx[row0 + r] * n + d]) * s;
        const float rw = A.row_w ? A.row_w[row0 + r] : 1.f;
        if (j < n) {
          gval = rw * t * s * A.inv_M;
          const float nl_d = 0.5f * t * t + lsc + 0.9189385332046727f;   // + 1/2 log(2 pi)
          lsum += rw * nl_d;
          if (A.row_lp) atomicAdd(&rowlp[r], -nl_d);
        } else {
          gval = rw * (1.f - t * t) * sigmoidf(A.mls - ls) * A.inv_M;
        }
        }
      }
      G0[r * LD + j] = gval;
      gst[(size_t)j * 16 + r] = gval;
    }
    red[tid] = lsum;
    __syncthreads();
    for (int o = PM_NT / 2; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) A.part_loss[wg] = red[0];
    if (A.row_lp && tid < nvalid) A.row_lp[row0 + tid] = rowlp[tid];
  }
  __syncthreads();
  // ---- backward: dX chain, dropout-logit terms, G stash
  float* Gin = G0;
  float* Gout = G1;
  for (int l = nl - 1; l >= 1; --l) {
    EpiBnnBwd e{H + (size_t)l * R * LD, Q + (size_t)(l - 1) * R * LD, Gout, GQ,
                A.gT[l - 1] + (size_t)wg * A.nt[l] * 16 * 16, LD, lane};
    gemm_tiles<1>(A.wb[l], A.nt[l], A.nt[l + 1], Gin, LD, wid, lane, e);
    __syncthreads();
    if (A.logit_p[l - 1]) {
      float* pl = A.part_lp + (size_t)wg * A.sum_h + A.lp_off[l - 1];
      for (int f = tid; f < A.dim[l]; f += PM_NT) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) s += GQ[r * LD + f];
        pl[f] = s;
      }
    }
    __syncthreads();
    float* t = Gin; Gin = Gout; Gout = t;
  }
}

// regulariser + reductions.  Flat layouts: params / grad = [W0, b0, W1, b1, ...] at w_off / b_off,
// logits of hidden layer l at lp_poff[l] of the SAME flat vectors (the module's parameter order).
struct BnnFinishArgs {
  int nl, nwg, sum_h, N;
  int dim[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL], lp_poff[PM_MAXL], lp_off[PM_MAXL];
  int has_drop[PM_MAXL];
  float reg_scale[PM_MAXL], drop_reg[PM_MAXL];
  float reg_weight, inv_M;
  int reg_only;      // gradient / loss of the regulariser alone (the likelihood partials are not read)
  const float* params;
  float* grad;
  const float* part_lp;
  const float* part_loss;
  float* loss_out;   // [3]: loss, Enlml, reg
  float* reg_part;   // [gridDim.x] scratch
};

// One workgroup per (dropout layer l, group of 32 hidden units): thread = (unit k, one of 8
// slices of the next layer's output rows).  A single coalesced pass over the 32 weight columns
// adds the weight-decay gradient in place and accumulates the column sums of squares; the
// slices meet in LDS, slice 0 finishes the unit's logit gradient and regulariser terms.  The
// group 0 of a layer also handles that layer's biases.
#define PM_BNN_CG 32
__global__ __launch_bounds__(256) void pm_bnn_finish(const BnnFinishArgs A) {
  __shared__ float s2s[8][PM_BNN_CG + 1];
  __shared__ double sm[256];
  const int tid = threadIdx.x, col = tid & (PM_BNN_CG - 1), sl = tid / PM_BNN_CG;
  // locate (layer, column group) of this workgroup
  int l = -1, grp = 0, b = blockIdx.x;
  for (int q = 0; q < A.nl - 1; ++q) {
    if (!A.has_drop[q]) continue;
    const int ng = (A.dim[q + 1] + PM_BNN_CG - 1) / PM_BNN_CG;
    if (b < ng) { l = q; grp = b; break; }
    b -= ng;
  }
  double reg = 0.0;
  if (l >= 0) {
    const float c = A.reg_weight / (float)A.N;
    const int h = A.dim[l + 1], O = A.dim[l + 2];
    const float* W = A.params + A.w_off[l + 1];      // [O][h]
    float* gW = A.grad + A.w_off[l + 1];
    const float rs = A.reg_scale[l], dr = A.drop_reg[l];
    const int k = grp * PM_BNN_CG + col;
    float p = 0.f, s2 = 0.f;
    if (k < h) {
      p = sigmoidf(A.params[A.lp_poff[l] + k]);
      const float cw = 2.f * c * rs * p;
      for (int o = sl; o < O; o += 8) {
        const float w = W[(size_t)o * h + k];
        s2 = fmaf(w, w, s2);
        gW[(size_t)o * h + k] += cw * w;
      }
    }
    s2s[sl][col] = s2;
    __syncthreads();
    if (sl == 0 && k < h) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) tot += s2s[q][col];
      float gdata = 0.f;
      if (!A.reg_only)
        for (int w = 0; w < A.nwg; ++w) gdata += A.part_lp[(size_t)w * A.sum_h + A.lp_off[l] + k];
      const float lgp = logf(p), lg1 = logf(1.f - p);
      A.grad[A.lp_poff[l] + k] = gdata + c * p * (1.f - p) * (rs * tot + dr * (lgp - lg1));
      reg += (double)(rs * p * tot + dr * (p * lgp + (1.f - p) * lg1));
    }
    if (grp == 0) {
      const float* bb = A.params + A.b_off[l + 1];
      for (int o = tid; o < O; o += 256) {
        const float bv = bb[o];
        A.grad[A.b_off[l + 1] + o] += 2.f * c * rs * bv;
        reg += (double)(rs * bv * bv);
      }
    }
  }
  sm[tid] = reg;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sm[tid] += sm[tid + o];
    __syncthreads();
  }
  if (tid == 0) A.reg_part[blockIdx.x] = (float)sm[0];
}

__global__ void pm_bnn_loss(const BnnFinishArgs A, int n_reg_part) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double reg = 0.0, nll = 0.0;
  for (int i = 0; i < n_reg_part; ++i) reg += (double)A.reg_part[i];
  if (!A.reg_only)
    for (int w = 0; w < A.nwg; ++w) nll += (double)A.part_loss[w];
  reg *= (double)A.reg_weight;
  const double en = nll * (double)A.inv_M;
  A.loss_out[0] = (float)(en + reg / (double)A.N);
  A.loss_out[1] = (float)en;
  A.loss_out[2] = (float)reg;
}
