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
  // u / bvar drawn inside the kernel (pmbrl_bnn_train_steps without recorded draws): Philox keyed by the caller's seed,
  // counter = (minibatch row, feature quad, hidden layer | stream << 8, step)
  int rng;
  unsigned rng_k0, rng_k1, rng_step;
  long long* prof;                 // debugging: cycle stamps of workgroup 0 (nullptr: none)
};

// 1 / x to an ulp or two: v_rcp_f32 and one Newton step
__device__ __forceinline__ float pm_fast_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
}

// hidden layer forward: out = relu(acc + b) * hard ;  q = relu(.) * d mask / d logit_p
struct EpiBnnFwd {
  const float* bias;
  const float* logit_p;
  const float *u, *bvar;
  float inv_temp;
  float *h_out, *q_out, *stash;
  int ld, row0, nvalid, width, lane;
  int rng;                       // 1: u / bvar from the generator (rk0, rk1, rstep; rlayer = this hidden layer)
  unsigned rk0, rk1, rstep, rlayer;
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc) {
    const int g = lane >> 4, lrow = lane & 15;
    const int f0 = ot * 16 + 4 * g;
    const f32x4 b = ldg4(bias + f0);
    f32x4 h, q;
    unsigned ru[4] = {0u, 0u, 0u, 0u}, rv[4] = {0u, 0u, 0u, 0u};
    if (rng && logit_p) {      // 24-bit uniforms in [0, 1), what torch.rand gives a float tensor
      pm_philox((unsigned)(row0 + lrow), (unsigned)(f0 >> 2), rlayer, rstep, rk0, rk1, ru);
      pm_philox((unsigned)(row0 + lrow), (unsigned)(f0 >> 2), rlayer | 0x100u, rstep, rk0, rk1, rv);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + r;
      const float v = fmaxf(acc[r] + b[r], 0.f);
      float hv = v, qv = 0.f;
      if (logit_p && f < width && lrow < nvalid) {
        const size_t o = (size_t)(row0 + lrow) * width + f;
        const float uu = rng ? (float)(ru[r] >> 8) * (1.0f / 16777216.0f) : u[o];
        // the logistic noise and the keep probability on the hardware's log2 / exp2 / reciprocal (1 ulp each, what the
        // library routines wrap in range handling this argument range does not need: u in [0, 1), the exponent clamped
        // so that the reciprocal never sees an infinity): 12.2 k -> 9.3 k cycles for the 5 -> 200 layer's 13 tile
        // epilogues, 19.6 k -> 17.9 k for the 200 -> 200 layer's -- the epilogues are instruction issue, four tiles a SIMD.
        // (Staging bias and logits in LDS at the head of the kernel, so that the epilogues load nothing from memory, was
        //  measured with it: the staging cost what the epilogues gained; not kept.)
        const float cp = logit_p[f] + 0.6931471805599453f * (__builtin_amdgcn_logf(uu + 1e-7f) - __builtin_amdgcn_logf(1.f - (uu - 1e-7f)));
        const float pr = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fminf(-1.4426950408889634f * (cp * inv_temp), 80.f)));
        const bool hard = (rng ? (float)(rv[r] >> 8) * (1.0f / 16777216.0f) : bvar[o]) < pr;
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

// (A GEMM variant that requests ALL of a wave's weight fragments of a layer at once was measured: pm_bnn_fwd_bwd 27.7 ->
//  34.3 us -- the first MFMA then waits for 32 loads where gemm_tiles starts on four; the layer chain is not what bounds it.)
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
#define BNN_STAMP(i) do { if (A.prof && wg == 0 && tid == 0) A.prof[i] = (long long)__builtin_readcyclecounter(); } while (0)
  BNN_STAMP(0);

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
  BNN_STAMP(1);
  // ---- forward
  for (int l = 0; l < nl - 1; ++l) {
    EpiBnnFwd e{A.bias[l], A.logit_p[l], A.u[l], A.bvar[l], A.inv_temp[l],
                H + (size_t)(l + 1) * R * LD, Q + (size_t)l * R * LD,
                A.actT[l + 1] + (size_t)wg * A.nt[l + 1] * 16 * 16, LD, row0, nvalid, A.dim[l + 1], lane,
                A.rng, A.rng_k0, A.rng_k1, A.rng_step, (unsigned)l};
    gemm_tiles<1>(A.wf[l], A.nt[l + 1], A.nt[l], H + (size_t)l * R * LD, LD, wid, lane, e);
    __syncthreads();
    BNN_STAMP(2 + l);
  }
  gemm_narrow<1>(A.wf[nl - 1], A.nt[nl], A.nt[nl - 1], A.bias[nl - 1], H + (size_t)(nl - 1) * R * LD, G1, LD, part,
                 wid, lane, tid);
  BNN_STAMP(10);
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
  BNN_STAMP(11);
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
    BNN_STAMP(12 + (nl - 1 - l));
    float* t = Gin; Gin = Gout; Gout = t;
  }
  BNN_STAMP(20);
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


// ---------------------------------------------------------------------------
// The rest of a training iteration in ONE launch (round 5; pmbrl_bnn_train_steps): dW / db from the stashes of
// pm_bnn_fwd_bwd, the regulariser's gradient (pm_bnn_finish's arithmetic), Adam (torch.optim.Adam's, no clipping --
// utils/train_regressor.py:113-131 clips nothing), the updated weights written back BOTH as the flat parameter vector
// and as the MFMA fragments the next iteration's pm_bnn_fwd_bwd reads (pm_pack_all's layouts), and the loss.
// The iteration used to be eight launches around 3 us of arithmetic (pack, forward + backward, dW, dW reduce,
// regulariser, loss, Adam + the two torch.rand): 77 us.  Adam without a global norm is element-wise, so the
// workgroup that forms a block of dW can finish those parameters on the spot.
//
// A workgroup = (layer L, 16 input columns kt of W_L): all output tiles of those columns, dealt to its four waves.
// It owns what the regulariser couples: the columns' sums of squares feed the gradient of the dropout logits of the
// hidden units that ARE those columns (hidden layer L - 1).  Column tile 0 of a layer also does the layer's bias.
// ---------------------------------------------------------------------------
#define PM_BNT_NW 8
#define PM_BNT_NT (PM_BNT_NW * 64)
struct BnnTailArgs {
  int nl, n_chunks, sum_h, N;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL], lp_poff[PM_MAXL], lp_off[PM_MAXL];
  int has_drop[PM_MAXL];            // dropout behind hidden layer l (the output of Linear l)
  int unit0[PM_MAXL + 1];           // first workgroup of layer l (prefix sums of nt[l])
  float reg_scale[PM_MAXL], drop_reg[PM_MAXL];
  float reg_weight, inv_M;
  float *params, *m, *v;            // flat: parameters, Adam moments
  float* wf[PM_MAXL];
  float* wb[PM_MAXL];
  float* bias[PM_MAXL];
  const float* actT[PM_MAXL];
  const float* gT[PM_MAXL];
  const float* part_lp;
  const float* part_loss;
  int n_part_loss;
  float* reg_part;                  // [gridDim.x]
  unsigned* counter;                // arrival counter of the launch (left at zero)
  float* loss_out;                  // [3]: loss, -E[lml], reg
  long long* step;                  // device-side Adam step counter: this launch applies step[0] + 1, its last workgroup stores it
  float lr, b1, b2, eps;
  double ln_b1, ln_b2;              // ln beta1, ln beta2 (from the host)
};

struct BnnAdam {
  float b1, b2, omb1, omb2, eps, step_size, bc2_sqrt, inv_bc2;
  // torch.optim.Adam (pm_clip_adam_kernel's arithmetic): returns the new parameter
  __device__ __forceinline__ float operator()(float p, float g, float& m, float& v) const {
    m = m * b1 + omb1 * g;
    v = v * b2 + omb2 * g * g;
    return p - step_size * (m * pm_fast_rcp(sqrtf(v) * inv_bc2 + eps));
  }
};

__global__ __launch_bounds__(PM_BNT_NT) void pm_bnn_tail(const BnnTailArgs A) {
  __shared__ float s_s2[PM_BNT_NW][16];
  __shared__ double s_reg[PM_BNT_NT];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  int L = 0;
  while (L + 1 < A.nl && (int)blockIdx.x >= A.unit0[L + 1]) ++L;
  const int kt = (int)blockIdx.x - A.unit0[L];
  const int O = A.dim[L + 1], K = A.dim[L], n_ot = A.nt[L + 1], n_kb = A.nt[L];
  const int O16 = n_ot * 16, K16 = n_kb * 16;
  BnnAdam ad;
  {
    // bias corrections 1 - beta^t from the device-side step count as -expm1(t ln beta), the exponent formed in double
    // (two double-precision pow calls were several hundred instructions at the head of every workgroup)
    const double st = (double)(A.step[0] + 1);
    ad.b1 = A.b1; ad.b2 = A.b2; ad.omb1 = 1.f - A.b1; ad.omb2 = 1.f - A.b2; ad.eps = A.eps;
    ad.step_size = A.lr / -expm1f((float)(st * A.ln_b1));
    ad.bc2_sqrt = sqrtf(-expm1f((float)(st * A.ln_b2)));
    ad.inv_bc2 = 1.f / ad.bc2_sqrt;
  }
  // regulariser of the dropout layer in FRONT of Linear L (hidden layer q = L - 1): weight decay scaled by the units' keep
  // logits; none in front of the first layer
  const int q = L - 1;
  const bool drop = L >= 1 && A.has_drop[q] != 0;
  const float c = A.reg_weight / (float)A.N;
  const float rs = drop ? A.reg_scale[q] : 0.f, dr = drop ? A.drop_reg[q] : 0.f;
  const int k = kt * 16 + c16;      // this lane's input column
  float p_k = 0.f;
  if (drop && k < K) p_k = sigmoidf(A.params[A.lp_poff[q] + k]);
  const float cw = 2.f * c * rs * p_k;
  double reg = 0.0;
  float s2 = 0.f;
  const float* gbase = A.gT[L];
  const float* abase = A.actT[L] + (size_t)(kt * 16 + c16) * 16 + 4 * g;
  // A wave's tiles two at a time, EVERYTHING they read requested before the first value is used -- the gradient chunks
  // (eight at a time), the parameters and both moments: taken in program order the kernel was a chain of dependent memory
  // round trips (seven per tile for the chunks, one more for the optimiser state: 26 us for 3 us of arithmetic).
  constexpr int CB = 8, TB = 2;
  f32x4 bv[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) bv[u] = ldg4(abase + (size_t)min(u, A.n_chunks - 1) * K16 * 16);
  for (int ot0 = wid; ot0 < n_ot; ot0 += TB * PM_BNT_NW) {
    f32x4 a[TB][CB];
    float pw[TB][4], pm[TB][4], pv[TB][4], pb[TB], pbm[TB], pbv[TB];
    int pidx[TB][4];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int ot = min(ot0 + tb * PM_BNT_NW, n_ot - 1);
      const float* gp = gbase + (size_t)(ot * 16 + c16) * 16 + 4 * g;
#pragma unroll
      for (int u = 0; u < CB; ++u) a[tb][u] = ldg4(gp + (size_t)min(u, A.n_chunks - 1) * O16 * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = min(ot * 16 + 4 * g + r, O - 1);
        pidx[tb][r] = A.w_off[L] + o * K + min(k, K - 1);
        pw[tb][r] = A.params[pidx[tb][r]];
        pm[tb][r] = A.m[pidx[tb][r]];
        pv[tb][r] = A.v[pidx[tb][r]];
      }
      const int ob = A.b_off[L] + min(ot * 16 + c16, O - 1);
      pb[tb] = A.params[ob];
      pbm[tb] = A.m[ob];
      pbv[tb] = A.v[ob];
    }
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int ot = ot0 + tb * PM_BNT_NW;
      if (ot >= n_ot) break;
      // dW tile: sum over the minibatch rows of g[o][row] act[k][row]; a lane's four k-values of a chunk are its rows
      // 4 g .. 4 g + 3 (a sum over rows does not care about their order: one 16-byte load per operand and chunk)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      float bsum = 0.f;
      const float* gp = gbase + (size_t)(ot * 16 + c16) * 16 + 4 * g;
      for (int c0 = 0; c0 < A.n_chunks; c0 += CB) {
        if (c0 > 0) {      // (minibatches beyond 128 rows: the further chunks as they come)
#pragma unroll
          for (int u = 0; u < CB; ++u) {
            a[tb][u] = ldg4(gp + (size_t)min(c0 + u, A.n_chunks - 1) * O16 * 16);
            bv[u] = ldg4(abase + (size_t)min(c0 + u, A.n_chunks - 1) * K16 * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          if (c0 + u < A.n_chunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma4(a[tb][u][j], bv[u][j], acc);
            bsum += (a[tb][u][0] + a[tb][u][1]) + (a[tb][u][2] + a[tb][u][3]);
          }
        }
      }
      if (A.n_chunks > CB) {
#pragma unroll
        for (int u = 0; u < CB; ++u) bv[u] = ldg4(abase + (size_t)u * K16 * 16);
      }
      // lane: dW[o = 16 ot + 4 g + r][k]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = ot * 16 + 4 * g + r;
        if (o < O && k < K) {
          const int idx = pidx[tb][r];
          const float w = pw[tb][r];
          float mi = pm[tb][r], vi = pv[tb][r];
          const float wn = ad(w, acc[r] + cw * w, mi, vi);
          s2 = fmaf(w, w, s2);
          A.params[idx] = wn;
          A.m[idx] = mi;
          A.v[idx] = vi;
          // forward fragments [ot][kb][lane (o % 16, (k % 16) / 4)][k % 4]; transposed ones [kt][ob][lane (k % 16, (o % 16) / 4)][o % 4]
          A.wf[L][(((size_t)ot * n_kb + kt) * 64 + ((4 * g + r) + 16 * (c16 >> 2))) * 4 + (c16 & 3)] = wn;
          A.wb[L][(((size_t)kt * n_ot + ot) * 64 + (c16 + 16 * g)) * 4 + r] = wn;
        }
      }
      if (kt == 0) {
        // bias of Linear L: db[o] = sum over rows of g[o][row] (lanes (o, g): partial over their rows)
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        const int o = ot * 16 + c16;
        if (g == 0 && o < O) {
          const int idx = A.b_off[L] + o;
          const float bb = pb[tb];
          float mi = pbm[tb], vi = pbv[tb];
          const float bn = ad(bb, bsum + (drop ? 2.f * c * rs * bb : 0.f), mi, vi);
          if (drop) reg += (double)(rs * bb * bb);
          A.params[idx] = bn;
          A.m[idx] = mi;
          A.v[idx] = vi;
          A.bias[L][o] = bn;
        }
      }
    }
  }
  // the columns' sums of squares: over the lane groups, then over the waves
  s2 += __shfl_xor(s2, 16);
  s2 += __shfl_xor(s2, 32);
  if (g == 0) s_s2[wid][c16] = s2;
  __syncthreads();
  if (drop && wid == 0 && g == 0 && k < K) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < PM_BNT_NW; ++w) tot += s_s2[w][c16];
    float gdata = 0.f;
    for (int w = 0; w < A.n_chunks; ++w) gdata += A.part_lp[(size_t)w * A.sum_h + A.lp_off[q] + k];
    const float lgp = logf(p_k), lg1 = logf(1.f - p_k);
    const int idx = A.lp_poff[q] + k;
    float mi = A.m[idx], vi = A.v[idx];
    const float ln = ad(A.params[idx], gdata + c * p_k * (1.f - p_k) * (rs * tot + dr * (lgp - lg1)), mi, vi);
    reg += (double)(rs * p_k * tot + dr * (p_k * lgp + (1.f - p_k) * lg1));
    A.params[idx] = ln;
    A.m[idx] = mi;
    A.v[idx] = vi;
  }
  s_reg[tid] = reg;
  __syncthreads();
  for (int o = PM_BNT_NT / 2; o > 0; o >>= 1) {
    if (tid < o) s_reg[tid] += s_reg[tid + o];
    __syncthreads();
  }
  // the loss: the last workgroup to arrive adds the partials in a fixed order and advances the step counter
  if (tid == 0) {
    A.reg_part[blockIdx.x] = (float)s_reg[0];
    __threadfence();
    s_last = atomicAdd(A.counter, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (s_last && wid == 0) {
    // (one partial per lane and a fixed butterfly: as one thread's loop these were gridDim.x dependent round trips to L2 at
    //  the end of every iteration -- 27 at the shipped shape, 2.5 us of this launch's 18.5)
    __threadfence();
    double rsum = 0.0, nll = 0.0;
    for (unsigned i = lane; i < gridDim.x; i += 64)
      rsum += (double)__hip_atomic_load(A.reg_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int w = lane; w < A.n_part_loss; w += 64) nll += (double)A.part_loss[w];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      rsum += __shfl_xor(rsum, o);
      nll += __shfl_xor(nll, o);
    }
    if (lane != 0) return;
    rsum *= (double)A.reg_weight;
    const double en = nll * (double)A.inv_M;
    A.loss_out[0] = (float)(en + rsum / (double)A.N);
    A.loss_out[1] = (float)en;
    A.loss_out[2] = (float)rsum;
    A.step[0] += 1;
    *A.counter = 0u;
  }
}
