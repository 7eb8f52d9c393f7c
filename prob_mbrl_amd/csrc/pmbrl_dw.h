// Policy weight gradient as ONE batched GEMM over every (time step, particle
// row) pair:   dW_l[o][k] = sum_n G_l[n][o] * act_l[n][k],   db_l[o] = sum_n G_l[n][o]
// with n running over H*B row-steps.  The forward / backward sweeps stash both
// operands FEATURE-MAJOR in blocks of 16*RT rows ([t][wg][feature][row]), which
// makes the reduction index n the contiguous one: each lane's MFMA operand for
// four consecutive k-steps is one 16-byte global load, no LDS staging needed.
//
// Decomposition: the output tile grid of every layer is cut into wave blocks of
// up to 4x4 tiles (64 accumulator registers); a workgroup takes PM_NW
// consecutive wave blocks; the n range is split `nsplit` ways across
// workgroups.  Partial sums go to `part[split][n_params]` and are summed in a
// fixed order by pm_dw_reduce -> deterministic gradients.
#pragma once
#include "pmbrl_dev.h"

#define PM_DW_TM 4
#define PM_DW_TN 4

struct DwBlock {      // one wave block
  int16_t layer, ot0, it0, n_ot, n_it, pad;
};

struct DwArgs {
  int nl, n_blocks, n_wg_per_split, nsplit, n_chunks, chunks_per_split;
  int nwg_rollout, RT, Rw, n_params;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL];    // offsets in the flat parameter vector
  const float* actT[PM_MAXL];
  const float* gT[PM_MAXL];
  const DwBlock* blocks;
  float* part;            // [nsplit][n_params]
};

__global__ __launch_bounds__(PM_NT, 1) void pm_dw_kernel(const DwArgs A) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x / A.n_wg_per_split;
  const int bidx = (blockIdx.x - split * A.n_wg_per_split) * PM_NW + wid;
  if (bidx >= A.n_blocks) return;
  const DwBlock blk = A.blocks[bidx];
  const int l = blk.layer;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const int g = lane >> 4, c16 = lane & 15;
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  // per-lane offsets inside a (t,wg) block for row tile rt: (feature)*Rw + rt*16 + 4g
  int goff[PM_DW_TM], aoff[PM_DW_TN];
#pragma unroll
  for (int i = 0; i < PM_DW_TM; ++i) goff[i] = ((blk.ot0 + i) * 16 + c16) * A.Rw + 4 * g;
#pragma unroll
  for (int j = 0; j < PM_DW_TN; ++j) aoff[j] = ((blk.it0 + j) * 16 + c16) * A.Rw + 4 * g;

  f32x4 acc[PM_DW_TM][PM_DW_TN];
  float bsum[PM_DW_TM];
#pragma unroll
  for (int i = 0; i < PM_DW_TM; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < PM_DW_TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int c_lo = split * A.chunks_per_split;
  const int c_hi = min(A.n_chunks, c_lo + A.chunks_per_split);
  const bool do_bias = blk.it0 == 0;

  f32x4 ga[2][PM_DW_TM], aa[2][PM_DW_TN];
  auto load = [&](int buf, int c) {
    if (c < c_hi) {
      const int b = c / A.RT, rt = c - b * A.RT;
      const float* gp = gbase + (size_t)b * gblk + rt * 16;
      const float* ap = abase + (size_t)b * ablk + rt * 16;
#pragma unroll
      for (int i = 0; i < PM_DW_TM; ++i)
        if (i < blk.n_ot) ga[buf][i] = ldg4(gp + goff[i]);
#pragma unroll
      for (int j = 0; j < PM_DW_TN; ++j)
        if (j < blk.n_it) aa[buf][j] = ldg4(ap + aoff[j]);
    }
  };
  auto compute = [&](int buf, int c) {
    if (c < c_hi) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < PM_DW_TM; ++i)
          if (i < blk.n_ot) {
#pragma unroll
            for (int j = 0; j < PM_DW_TN; ++j)
              if (j < blk.n_it) acc[i][j] = mfma4(ga[buf][i][kk], aa[buf][j][kk], acc[i][j]);
          }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < PM_DW_TM; ++i)
          if (i < blk.n_ot) bsum[i] += (ga[buf][i][0] + ga[buf][i][1]) + (ga[buf][i][2] + ga[buf][i][3]);
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
  // write the partial tile: lane holds dW[o = (ot)*16 + 4g + r][k = it*16 + c16]
  float* part = A.part + (size_t)split * A.n_params;
  const int O = A.dim[l + 1], K = A.dim[l];
#pragma unroll
  for (int i = 0; i < PM_DW_TM; ++i)
    if (i < blk.n_ot) {
#pragma unroll
      for (int j = 0; j < PM_DW_TN; ++j)
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

// grad[i] = sum_s part[s][i]  in fixed order
__global__ void pm_dw_reduce(const float* __restrict__ part, int nsplit, int n, float* __restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
  grad[i] = s;
}
