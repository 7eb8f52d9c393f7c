// Policy weight gradient as ONE batched GEMM over every (time step, particle
// row) pair:   dW_l[o][k] = sum_n G_l[n][o] * act_l[n][k],   db_l[o] = sum_n G_l[n][o]
// with n running over H*B row-steps.  The forward / backward sweeps stash both
// operands FEATURE-MAJOR in blocks of 16*RT rows ([t][wg][feature][row]), which
// makes the reduction index n the contiguous one: each lane's MFMA operand for
// four consecutive k-steps is one 16-byte global load, no LDS staging needed.
//
// Decomposition: a workgroup = 8 waves (2 per SIMD) owns a contiguous range of
// row-step chunks (split-K) and walks ALL output wave blocks of all layers in
// `npass` passes of 8 wave blocks.  A wave block is up to 4 x 8 tiles (128
// accumulator registers); the 8 blocks of a pass are neighbours in the output
// tile grid of one layer, so a stash chunk is fetched from HBM once per
// workgroup and re-read by the other waves through L1/L2 (the first version,
// 4x4 blocks spread over independent workgroups, read every chunk ~2.5 times
// and was bound by that traffic).  Partial sums go to part[split][n_params] and
// are added in a fixed order by pm_dw_reduce -> bit-reproducible gradients.
#pragma once
#include "pmbrl_dev.h"

#define PM_DW_NW 8
#define PM_DW_NT (PM_DW_NW * 64)
#define PM_DW_TM 4
#define PM_DW_TN 8

struct DwBlock {      // one wave block; layer < 0: idle slot
  int16_t layer, ot0, it0, n_ot, n_it, pad;
};

struct DwArgs {
  int nl, npass, nsplit, n_chunks, chunks_per_split;
  int RT, Rw, n_params;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL];    // offsets in the flat parameter vector
  const float* actT[PM_MAXL];
  const float* gT[PM_MAXL];
  const DwBlock* blocks;  // [npass][PM_DW_NW]
  float* part;            // [nsplit][n_params]
};

// one wave block with at most NJ input-tile columns (NJ is static so that narrow layers --
// the first layer's K <= 16, the head's 2U outputs -- do not pay for 8 columns)
template <int NJ>
__device__ __forceinline__ void pm_dw_block(const DwArgs& A, const DwBlock& blk, int c_lo, int c_hi,
                                            float* part, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int l = blk.layer;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  // per-lane offsets inside a (t,wg) block for row tile rt: (feature)*Rw + rt*16 + 4g
  const int goff0 = (blk.ot0 * 16 + c16) * A.Rw + 4 * g;
  const int aoff0 = (blk.it0 * 16 + c16) * A.Rw + 4 * g;
  const int tstride = 16 * A.Rw;

  f32x4 acc[PM_DW_TM][NJ];
  float bsum[PM_DW_TM];
#pragma unroll
  for (int i = 0; i < PM_DW_TM; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = blk.it0 == 0;
  // register double buffer: the loads of chunk c+1 are in flight while chunk c feeds the MFMAs
  f32x4 ga[2][PM_DW_TM], aa[2][NJ];
  auto load = [&](int buf, int c) {
    if (c < c_hi) {
      const int b = c / A.RT, rt = c - b * A.RT;
      const float* gp = gbase + (size_t)b * gblk + rt * 16 + goff0;
      const float* ap = abase + (size_t)b * ablk + rt * 16 + aoff0;
#pragma unroll
      for (int i = 0; i < PM_DW_TM; ++i)
        ga[buf][i] = (i < blk.n_ot) ? ldg4(gp + i * tstride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        aa[buf][j] = (j < blk.n_it) ? ldg4(ap + j * tstride) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto compute = [&](int buf, int c) {
    if (c < c_hi) {
      // absent columns carry zero operands; absent rows are skipped with one uniform branch
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < PM_DW_TM; ++i)
          if (i < blk.n_ot) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(ga[buf][i][kk], aa[buf][j][kk], acc[i][j]);
          }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < PM_DW_TM; ++i)
          bsum[i] += (ga[buf][i][0] + ga[buf][i][1]) + (ga[buf][i][2] + ga[buf][i][3]);
      }
    }
  };
  load(0, c_lo);
  for (int c = c_lo; c < c_hi; c += 2) {
    load(1, c + 1);
    compute(0, c);
    load(0, c + 2);
    compute(1, c + 1);
  }
  // write the partial tile: lane holds dW[o = ot*16 + 4g + r][k = it*16 + c16]
  const int O = A.dim[l + 1], K = A.dim[l];
#pragma unroll
  for (int i = 0; i < PM_DW_TM; ++i)
    if (i < blk.n_ot) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (j < blk.n_it) {
          const int k = (blk.it0 + j) * 16 + c16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (blk.ot0 + i) * 16 + 4 * g + r;
            if (o < O && k < K) part[A.w_off[l] + (size_t)o * K + k] = acc[i][j][r];
          }
        }
      if (do_bias) {
        float s = bsum[i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const int o = (blk.ot0 + i) * 16 + c16;
        if (g == 0 && o < O) part[A.b_off[l] + o] = s;
      }
    }
}

__global__ __launch_bounds__(PM_DW_NT, 2) void pm_dw_kernel(const DwArgs A) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int c_lo = split * A.chunks_per_split;
  const int c_hi = min(A.n_chunks, c_lo + A.chunks_per_split);
  float* part = A.part + (size_t)split * A.n_params;
  for (int pass = 0; pass < A.npass; ++pass) {
    const DwBlock blk = A.blocks[pass * PM_DW_NW + wid];
    if (blk.layer < 0) continue;
    if (blk.n_it <= 1) pm_dw_block<1>(A, blk, c_lo, c_hi, part, lane);
    else if (blk.n_it <= 4) pm_dw_block<4>(A, blk, c_lo, c_hi, part, lane);
    else pm_dw_block<PM_DW_TN>(A, blk, c_lo, c_hi, part, lane);
  }
}

// grad[i] = sum_s part[s][i] in fixed order; 32 parameters x 8 split-slices per workgroup
__global__ __launch_bounds__(256) void pm_dw_reduce(const float* __restrict__ part, int nsplit, int n,
                                                    float* __restrict__ grad) {
  __shared__ float sm[8][33];
  const int pi = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + pi;
  const int per = (nsplit + 7) / 8;
  float s = 0.f;
  if (i < n) {
    const int k_hi = min(nsplit, (sl + 1) * per);
    for (int k = sl * per; k < k_hi; ++k) s += part[(size_t)k * n + i];
  }
  sm[sl][pi] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][pi];
    grad[i] = t;
  }
}
