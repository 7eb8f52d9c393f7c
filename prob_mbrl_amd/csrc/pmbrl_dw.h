// Policy weight gradient as ONE batched GEMM over every (time step, particle
// row) pair:   dW_l[o][k] = sum_n G_l[n][o] * act_l[n][k],   db_l[o] = sum_n G_l[n][o]
// with n running over H*B row-steps.  The forward / backward sweeps stash both
// operands FEATURE-MAJOR in blocks of 16*RT rows ([t][wg][feature][row]), which
// makes the reduction index n the contiguous one: each lane's MFMA operand for
// four consecutive k-steps is one 16-byte global load, no LDS staging needed.
//
// Decomposition: a workgroup = 8 waves (2 per SIMD) owns a contiguous range of
// row-step chunks (split-K) and covers ALL output tiles of all layers.  The
// output tile grid of every layer is cut into wave blocks of at most 4 x 7
// tiles (112 accumulator registers); the host deals the blocks to the four SIMDs
// by decreasing size (longest-processing-time first) so that every SIMD carries
// the same number of MFMAs, and a wave walks its list of blocks one after the
// other, each over the whole chunk range.  A stash chunk is fetched from HBM once
// per workgroup and re-read by the other waves through L1/L2.  Partial sums go to
// part[split][n_params] and are added in a fixed order by pm_dw_reduce ->
// bit-reproducible gradients.
//
// Inner loop: a block is a compile-time NI x NJ tile shape (edge blocks repeat
// their last tile: branch-free loads and MFMAs, the duplicates are not stored).
// Operands are double-buffered in registers with inline-asm loads and explicit
// s_waitcnt (same rules as the weight stream of pmbrl_fast.h, checked by
// tools/check_inflight.py): the compiler's own waitcnt placement drained the
// queue (vmcnt(0)) before every chunk, i.e. one exposed HBM round trip per chunk.
#pragma once
#include "pmbrl_dev.h"
#include "pmbrl_split.h"

#include <utility>
#include <type_traits>
template <class F, int... I>
__device__ __forceinline__ void pr_for_dw_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void pr_for_dw(F&& f) { pr_for_dw_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }
typedef unsigned pr_u32x4_dw __attribute__((ext_vector_type(4)));
typedef unsigned long long pr_u64x2_dw __attribute__((ext_vector_type(2)));
#define PM_DW_NW 8
#define PM_DW_NT (PM_DW_NW * 64)
#define PM_DW_TM 4
#define PM_DW_TN 7
#define PM_DW_MAXBLK 256
#define PM_DW_TN_S 4   // split-operand form: blocks of at most 4 x 4 tiles (see pm_dw_block_s)

struct DwBlock {      // one wave block
  int16_t layer, ot0, it0, n_ot, n_it, pad;
};

struct PmTrue { static constexpr bool value = true; };
struct PmFalse { static constexpr bool value = false; };
// one 256 x 128 output tile of a wide layer (pm_dw_wide_kernel)
struct DwUnit {
  int16_t layer, m0, n0, pad;      // first output feature / first input feature of the tile
};
#define PM_DWW_TM 128
#define PM_DWW_TN 128
#define PM_DWW_LDK 40
#define PM_DWW_STAGE ((PM_DWW_TM + PM_DWW_TN) * 2 * PM_DWW_LDK)      // 16-bit elements per stage
#define PM_DWW_LDS_BYTES (2 * PM_DWW_STAGE * 2)

struct DwArgs {
  int nl, nsplit, n_chunks, chunks_per_split;
  int RT, Rw, n_params;
  int part_stride;                       // floats between the partial vectors of two splits (n_params rounded up to 4)
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL];    // offsets in the flat parameter vector
  const float* actT[PM_MAXL];
  const float* gT[PM_MAXL];
  const int* nvalid;                     // forward status word: only chunks of steps t < *nvalid exist (nullptr: all)
  int chunks_per_step;                   // nwg * RT
  const DwBlock* blocks;                 // sorted by wave
  int wave_first[PM_DW_NW + 1];          // wave w owns blocks [wave_first[w], wave_first[w+1])
  float* part;                           // [nsplit][n_params]
  int split_prec;                        // 1: two bf16 pieces per operand on the bf16 matrix core (pm_dw_kernel_s)
  // Pipelined form (pmbrl.hip: "dW behind the adjoint sweep"): the adjoint sweep runs as K launches over
  // descending step ranges and launch k of this GEMM covers the chunks of range k only, [c_begin, c_end),
  // dealt to the same nsplit partial rows every time; the first launch that owns a valid step writes its
  // row (zeros if the row got no chunk), the later ones add to it -- still one fixed order of additions.
  int pipe;                              // 0: one launch over all chunks (fields below unused)
  int pipe_k;                            // index of this launch, 0 = the highest steps
  int c_begin, c_end;                    // chunk range of this launch (c_end = the previous launch's c_begin)
  int rows_before;                       // partial rows the earlier launches own (the last launch has more)
};

struct DwRange {
  int c_lo, c_hi;
  bool add;        // the partial row already holds the sum of earlier launches
  bool zero;       // no chunk for this row in the launch that defines the rows: write zeros
};

#ifdef PM_MAIN_TU   // the kernels below are compiled into pmbrl.hip only (pmbrl_host.h needs just the structs above)
__device__ __forceinline__ void pm_dw_ld(f32x4& d, unsigned voff, const float* sbase) {
#ifdef PM_EXP_DW_NOLOAD
  asm volatile("" : "=v"(d) : "v"(voff), "s"(sbase));
#else
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(sbase));
#endif
}

// chunk range and write mode of partial row `row` in this launch; false: nothing to do
__device__ __forceinline__ bool pm_dw_range(const DwArgs& A, int row, DwRange& r) {
  long long nv = 0x7fffffffll;      // valid steps of the forward sweep (the status word is INT_MAX when complete)
  if (A.nvalid) nv = max(0, __builtin_amdgcn_readfirstlane(*A.nvalid));
  const int n_chunks = A.nvalid ? (int)min((long long)A.n_chunks, nv * A.chunks_per_step) : A.n_chunks;
  r.add = r.zero = false;
  if (!A.pipe) {
    r.c_lo = row * A.chunks_per_split;
    r.c_hi = min(n_chunks, r.c_lo + A.chunks_per_split);
    return r.c_lo < r.c_hi;                 // (pm_dw_reduce skips the rows without chunks)
  }
  const long long first_valid = max(nv, 1ll) * A.chunks_per_step;   // a launch below this bound defines the rows
  r.add = A.pipe_k > 0 && (long long)A.c_end < first_valid && row < A.rows_before;
  r.c_lo = A.c_begin + row * A.chunks_per_split;
  r.c_hi = min(min(A.c_end, n_chunks), r.c_lo + A.chunks_per_split);
  if (r.c_lo < r.c_hi) return true;
  r.zero = !r.add && (long long)A.c_begin < first_valid;
  return r.zero;
}
__device__ __forceinline__ void pm_dw_put(float* q, float v, bool add) { *q = add ? *q + v : v; }

template <int NI, int NJ>
__device__ __forceinline__ void pm_dw_block(const DwArgs& A, const DwBlock& blk, int c_lo, int c_hi,
                                            float* part, int lane, bool add) {
  const int g = lane >> 4, c16 = lane & 15;
  const int l = blk.layer;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  // per-lane byte offsets inside a (t, wg) block: (feature) * Rw + 4g ; tiles past the edge of
  // the block repeat its last tile
  unsigned goff[NI], aoff[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ot = blk.ot0 + (i < blk.n_ot ? i : blk.n_ot - 1);
    goff[i] = (unsigned)(((ot * 16 + c16) * A.Rw + 4 * g) * 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int it = blk.it0 + (j < blk.n_it ? j : blk.n_it - 1);
    aoff[j] = (unsigned)(((it * 16 + c16) * A.Rw + 4 * g) * 4);
  }

  f32x4 acc[NI][NJ];
  float bsum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = blk.it0 == 0;
  // register double buffer: the loads of chunk c+1 are in flight while chunk c feeds the MFMAs
  f32x4 ga[2][NI], aa[2][NJ];
  auto load = [&](int buf, int c) {
    const int cc = c < c_hi ? c : c_hi - 1;     // past the end: harmless re-load of the last chunk
    const int b = cc / A.RT, rt = cc - b * A.RT;
    const float* gp = gbase + (size_t)b * gblk + rt * 16;
    const float* ap = abase + (size_t)b * ablk + rt * 16;
#pragma unroll
    for (int i = 0; i < NI; ++i) pm_dw_ld(ga[buf][i], goff[i], gp);
#pragma unroll
    for (int j = 0; j < NJ; ++j) pm_dw_ld(aa[buf][j], aoff[j], ap);
  };
  // everything but the NI+NJ loads just issued (the other buffer) has landed
  auto wait = [&](int buf) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI + NJ));
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(ga[buf][i]));
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(aa[buf][j]));
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#ifdef PM_EXP_DW_NOMFMA
          if (kk == 0) acc[i][j][0] += ga[buf][i][0] * aa[buf][j][0];
#else
          acc[i][j] = mfma4(ga[buf][i][kk], aa[buf][j][kk], acc[i][j]);
#endif
        }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        bsum[i] += (ga[buf][i][0] + ga[buf][i][1]) + (ga[buf][i][2] + ga[buf][i][3]);
    }
  };
  load(0, c_lo);
  for (int c = c_lo; c < c_hi; c += 2) {
    load(1, c + 1);
    wait(0);
    compute(0);
    load(0, c + 2);
    wait(1);
    if (c + 1 < c_hi) compute(1);
  }
  asm volatile("s_waitcnt vmcnt(0)");   // drain the look-ahead loads before the registers are reused
#pragma unroll
  for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(ga[0][i]));
#pragma unroll
  for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(aa[0][j]));
  // write the partial tile: lane holds dW[o = ot*16 + 4g + r][k = it*16 + c16]
  const int O = A.dim[l + 1], K = A.dim[l];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (i < blk.n_ot) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (j < blk.n_it) {
          const int k = (blk.it0 + j) * 16 + c16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (blk.ot0 + i) * 16 + 4 * g + r;
            if (o < O && k < K) pm_dw_put(part + A.w_off[l] + (size_t)o * K + k, acc[i][j][r], add);
          }
        }
      if (do_bias) {
        float s = bsum[i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const int o = (blk.ot0 + i) * 16 + c16;
        if (g == 0 && o < O) pm_dw_put(part + A.b_off[l] + o, s, add);
      }
    }
}

// block shape classes: NI in {1, 3, 4}, NJ in {1, 4, 6, 7} (a block runs in the smallest
// class that holds it)
template <int NI>
__device__ __forceinline__ void pm_dw_dispatch_j(const DwArgs& A, const DwBlock& blk, int c_lo, int c_hi,
                                                 float* part, int lane, bool add) {
  if (blk.n_it <= 1) pm_dw_block<NI, 1>(A, blk, c_lo, c_hi, part, lane, add);
  else if (blk.n_it <= 4) pm_dw_block<NI, 4>(A, blk, c_lo, c_hi, part, lane, add);
  else if (blk.n_it <= 6) pm_dw_block<NI, 6>(A, blk, c_lo, c_hi, part, lane, add);
  else pm_dw_block<NI, 7>(A, blk, c_lo, c_hi, part, lane, add);
}

// (two waves per SIMD at up to 256 registers per lane: a workgroup takes a CU's whole register file, so a grid
//  of n workgroups occupies n CUs and never shares one with the adjoint sweep's workgroups -- the pipelined
//  form relies on that to stay out of the sweep's way)
__global__ __launch_bounds__(PM_DW_NT, 2) void pm_dw_kernel(const DwArgs A) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  float* part = A.part + (size_t)row * A.part_stride;
  if (R.zero) {
    for (int i = tid; i < A.part_stride; i += PM_DW_NT) part[i] = 0.f;
    return;
  }
  for (int bi = A.wave_first[wid]; bi < A.wave_first[wid + 1]; ++bi) {
    const DwBlock blk = A.blocks[bi];
    if (blk.n_ot <= 1) pm_dw_dispatch_j<1>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
    else if (blk.n_ot <= 3) pm_dw_dispatch_j<3>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
    else pm_dw_dispatch_j<4>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
  }
}


// ---------------------------------------------------------------------------
// The same GEMM on split operands: every stashed fp32 value becomes two bf16 pieces in registers
// (round, residual, round: 16 bits of significand -- the adjoint's precision class,
// profiles/r02_split_precision_study.txt) and a product is three v_mfma_f32_16x16x32_bf16 per 32 row-steps
// instead of eight fp32 MFMAs of twice the issue time.  K = 32 row-steps = two consecutive 16-row chunks
// (c even, c + 1): lane (c16, g) supplies rows 8(g&1)..+7 of chunk c + (g>>1).  One raw operand set is in
// flight while the previous one, already split into pieces, feeds the MFMAs; blocks are at most 4 x 4 tiles
// (64 accumulator + 64 piece + 64 raw registers).  Used where the fp32 form is bound by the matrix core
// (32- and 64-row workgroups, wide networks); the 16-row form is bound by HBM and stays fp32.
template <int NI, int NJ>
__device__ __forceinline__ void pm_dw_block_s(const DwArgs& A, const DwBlock& blk, int c_lo, int c_hi,
                                              float* part, int lane, bool add) {
  const int g = lane >> 4, c16 = lane & 15;
  const int l = blk.layer;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  // bytes from chunk c (even) to chunk c + 1: the next 16 rows of the same block, or (16-row blocks) the next block
  const unsigned gd = A.RT >= 2 ? 64u : (unsigned)(gblk * 4), ad = A.RT >= 2 ? 64u : (unsigned)(ablk * 4);
  unsigned goff[NI], aoff[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ot = blk.ot0 + (i < blk.n_ot ? i : blk.n_ot - 1);
    goff[i] = (unsigned)(((ot * 16 + c16) * A.Rw + 8 * (g & 1)) * 4) + (unsigned)(g >> 1) * gd;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int it = blk.it0 + (j < blk.n_it ? j : blk.n_it - 1);
    aoff[j] = (unsigned)(((it * 16 + c16) * A.Rw + 8 * (g & 1)) * 4) + (unsigned)(g >> 1) * ad;
  }
  f32x4 acc[NI][NJ];
  float bsum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = blk.it0 == 0;
  f32x4 gr[NI][2], ar[NJ][2];     // raw fp32 rows, in flight
  f32x4 gp[NI][2], ap[NJ][2];     // pieces: [.][0] = 8 high bf16, [.][1] = 8 low bf16
  auto load = [&](int c) {
    const int cc = c < c_hi ? c : c_lo;         // past the end: harmless re-load of the first pair
    const int b = cc / A.RT, rt = cc - b * A.RT;
    const float* gq = gbase + (size_t)b * gblk + rt * 16;
    const float* aq = abase + (size_t)b * ablk + rt * 16;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(gr[i][0]) : "v"(goff[i]), "s"(gq));
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(gr[i][1]) : "v"(goff[i]), "s"(gq));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(ar[j][0]) : "v"(aoff[j]), "s"(aq));
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(ar[j][1]) : "v"(aoff[j]), "s"(aq));
    }
  };
  // 8 fp32 -> 8 high + 8 low bf16; rows of an absent second chunk (odd tail) contribute zeros
  auto split8 = [&](const f32x4 (&r)[2], f32x4 (&p)[2], bool live) {
    unsigned hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = live ? r[e >> 1][2 * (e & 1)] : 0.f, b = live ? r[e >> 1][2 * (e & 1) + 1] : 0.f;
      hi[e] = pm_pk_bf16(a, b);
      lo[e] = pm_pk_bf16(a - pm_bf_lo(hi[e]), b - pm_bf_hi(hi[e]));
    }
    p[0] = __builtin_bit_cast(f32x4, (uint4){hi[0], hi[1], hi[2], hi[3]});
    p[1] = __builtin_bit_cast(f32x4, (uint4){lo[0], lo[1], lo[2], lo[3]});
  };
  load(c_lo);
  for (int c = c_lo; c < c_hi; c += 2) {
    asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      asm volatile("" : "+v"(gr[i][0]));
      asm volatile("" : "+v"(gr[i][1]));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      asm volatile("" : "+v"(ar[j][0]));
      asm volatile("" : "+v"(ar[j][1]));
    }
    const bool live = g < 2 || c + 1 < c_hi;
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const f32x4 t = gr[i][0] + gr[i][1];
        bsum[i] += live ? (t[0] + t[1]) + (t[2] + t[3]) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) split8(gr[i], gp[i], live);
#pragma unroll
    for (int j = 0; j < NJ; ++j) split8(ar[j], ap[j], live);
    load(c + 2);
    // smallest contributions first: lo x hi, hi x lo, hi x hi
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = pm_mfma_bf<false>(gp[i][1], ap[j][0], acc[i][j]);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = pm_mfma_bf<false>(gp[i][0], ap[j][1], acc[i][j]);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = pm_mfma_bf<false>(gp[i][0], ap[j][0], acc[i][j]);
  }
  asm volatile("s_waitcnt vmcnt(0)");   // drain the look-ahead loads before the registers are reused
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    asm volatile("" : "+v"(gr[i][0]));
    asm volatile("" : "+v"(gr[i][1]));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    asm volatile("" : "+v"(ar[j][0]));
    asm volatile("" : "+v"(ar[j][1]));
  }
  const int O = A.dim[l + 1], K = A.dim[l];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (i < blk.n_ot) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (j < blk.n_it) {
          const int k = (blk.it0 + j) * 16 + c16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = (blk.ot0 + i) * 16 + 4 * g + r;
            if (o < O && k < K) pm_dw_put(part + A.w_off[l] + (size_t)o * K + k, acc[i][j][r], add);
          }
        }
      if (do_bias) {
        float s = bsum[i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const int o = (blk.ot0 + i) * 16 + c16;
        if (g == 0 && o < O) pm_dw_put(part + A.b_off[l] + o, s, add);
      }
    }
}

template <int NI>
__device__ __forceinline__ void pm_dw_dispatch_js(const DwArgs& A, const DwBlock& blk, int c_lo, int c_hi,
                                                  float* part, int lane, bool add) {
  if (blk.n_it <= 1) pm_dw_block_s<NI, 1>(A, blk, c_lo, c_hi, part, lane, add);
  else pm_dw_block_s<NI, PM_DW_TN_S>(A, blk, c_lo, c_hi, part, lane, add);
}

// (chunks_per_split and c_begin are even: a chunk pair never straddles two rows or two launches; up to 256
//  registers per lane, i.e. one workgroup per CU -- a CU holds this kernel or the adjoint sweep, never both)
__global__ __launch_bounds__(PM_DW_NT, 1) void pm_dw_kernel_s(const DwArgs A) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  float* part = A.part + (size_t)row * A.part_stride;
  if (R.zero) {
    for (int i = tid; i < A.part_stride; i += PM_DW_NT) part[i] = 0.f;
    return;
  }
  for (int bi = A.wave_first[wid]; bi < A.wave_first[wid + 1]; ++bi) {
    const DwBlock blk = A.blocks[bi];
    if (blk.n_ot <= 1) pm_dw_dispatch_js<1>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
    else if (blk.n_ot <= 3) pm_dw_dispatch_js<3>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
    else pm_dw_dispatch_js<4>(A, blk, R.c_lo, R.c_hi, part, lane, R.add);
  }
}

// ---------------------------------------------------------------------------
// Wide layers (both widths >= 128: the 512 x 512 layers of the stress shape).  The block kernels above give every
// wave its own 4 x 4-tile block and let it fetch and split its operands itself: at 32 x 32 tiles per layer every
// stash tile is fetched and converted by eight waves, with one memory round trip exposed per chunk pair (15 ms
// for 1.8 TFLOP).  Here a workgroup owns a 128 x 128 output tile of ONE layer over its row-step range; per K = 32
// row-steps the two operand tiles are fetched once (registers, a step ahead), split once into two bf16 pieces
// and staged in LDS ([piece][feature][32 + 8 bf16]: 80-byte rows, conflict-free b128 reads), double-buffered,
// one barrier per step; every wave accumulates 2 x 4 tiles.  116 registers and 80 KB of LDS: two workgroups per
// CU, one in its MFMAs while the other converts and stages.  Same partial-row scheme as the block kernels
// (row = split of the row-step range).  Measured at the stress shape: 15.3 -> 7.7 ms for the whole dW step
// (6.5 ms this kernel: 52 GB from L2 at the ~31 B/clk a CU pulls -- that, not the matrix pipe, is its bound;
// 256 x 128 tiles with one workgroup per CU: 6.9 ms, phases serialised).
__global__ __launch_bounds__(PM_DW_NT, 4) void pm_dw_wide_kernel(const DwArgs A, const DwUnit* __restrict__ units,
                                                                  int n_units) {
  extern __shared__ __attribute__((aligned(16))) unsigned short dww_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroups go round-robin to the 8 XCDs (each with its own L2): the tiles of one row-step range share their
  // operands -- a 256-feature delta tile is read by every input tile of its layer -- so all tiles of a range are
  // dealt to ONE XCD, next to each other in its dispatch order (XCD x works on the ranges x, x + 8, ...).  Dealt
  // to eight XCDs they fetched eight copies: 40 GB through HBM instead of 13 at the 3 x 512 stress shape.
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int row = (jx / n_units) * 8 + xcd, unit = jx % n_units;
  if (row >= A.nsplit) return;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  const DwUnit U = units[unit];
  const int l = U.layer;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const int O = A.dim[l + 1], K = A.dim[l];
  float* part = A.part + (size_t)row * A.part_stride;
  const int g = lane >> 4, c16 = lane & 15;
  const int wm = wid & 3, wn = wid >> 2;      // 32 output features (2 tiles) x 64 input features (4 tiles) per wave
  if (R.zero) {
    for (int e = tid; e < PM_DWW_TM * PM_DWW_TN; e += PM_DW_NT) {
      const int o = U.m0 + e / PM_DWW_TN, k = U.n0 + e % PM_DWW_TN;
      if (o < O && k < K) part[A.w_off[l] + (size_t)o * K + k] = 0.f;
    }
    if (U.n0 == 0)
      for (int e = tid; e < PM_DWW_TM; e += PM_DW_NT)
        if (U.m0 + e < O) part[A.b_off[l] + U.m0 + e] = 0.f;
    return;
  }
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  // staging: thread -> (feature f = idx >> 3, k quad kq = idx & 7), idx = tid + 512 j
  f32x4 ra0[2], rb0[2];
  float bs[2] = {0.f, 0.f};
  // Whole tiles of 32- / 64-row stash blocks (every tile of the 3 x 512 stress shape): the step's operands are
  //   base + [block b = c / RT, row tile c % RT: scalar] + [feature row, k quad: a per-thread constant]
  // -- loads with a scalar base and one 32-bit lane offset.  (The general form below recomputed c / RT by a software
  // division and a 64-bit address per load, behind a branch per load for the edge tiles: ~150 vector and ~170 scalar
  // instructions a wave-step beside 24 MFMAs, four waves a SIMD -- the kernel was bound by that instruction stream,
  // MfmaUtil 33 %.)
  const bool whole = A.RT >= 2 && (A.RT & (A.RT - 1)) == 0 && U.m0 + PM_DWW_TM <= Fo16 && U.n0 + PM_DWW_TN <= Fi16;
  const int rt_sh = A.RT >= 4 ? 2 : 1;
  unsigned offA[2], offB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + PM_DW_NT * j;
    offA[j] = (unsigned)((U.m0 + (idx >> 3)) * A.Rw + (idx & 7) * 4);
    offB[j] = (unsigned)((U.n0 + (idx >> 3)) * A.Rw + (idx & 7) * 4);
  }
  auto fetch = [&](auto fast, f32x4 (&ra)[2], f32x4 (&rb)[2], int c) {
    if constexpr (decltype(fast)::value) {
      const int b = c >> rt_sh, rt = c & (A.RT - 1);
      const float* pg = gbase + (size_t)b * gblk + rt * 16;
      const float* pa = abase + (size_t)b * ablk + rt * 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pg + offA[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) rb[j] = *reinterpret_cast<const f32x4*>(pa + offB[j]);
      return;
    }
    const bool second = c + 1 < R.c_hi;      // (16-row blocks: rows 16..31 of the step are the next chunk)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + PM_DW_NT * j, f = U.m0 + (idx >> 3), kq = idx & 7;
      const float* p;
      bool live = f < Fo16;
      if (A.RT >= 2) {
        const int b = c / A.RT, rt = c - b * A.RT;
        p = gbase + (size_t)b * gblk + (size_t)f * A.Rw + rt * 16 + kq * 4;
      } else {
        p = gbase + (size_t)(c + (kq >> 2)) * gblk + (size_t)f * 16 + (kq & 3) * 4;
        live = live && (kq < 4 || second);
      }
      ra[j] = live ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + PM_DW_NT * j, f = U.n0 + (idx >> 3), kq = idx & 7;
      const float* p;
      bool live = f < Fi16;
      if (A.RT >= 2) {
        const int b = c / A.RT, rt = c - b * A.RT;
        p = abase + (size_t)b * ablk + (size_t)f * A.Rw + rt * 16 + kq * 4;
      } else {
        p = abase + (size_t)(c + (kq >> 2)) * ablk + (size_t)f * 16 + (kq & 3) * 4;
        live = live && (kq < 4 || second);
      }
      rb[j] = live ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // fp32 x 4 -> 4 high + 4 low bf16, written to the two piece planes of `plane` (feature row f, k quad kq)
  auto put = [&](unsigned short* plane, int nfeat, int f, int kq, const f32x4& v) {
    const unsigned h0 = pm_pk_bf16(v[0], v[1]), h1 = pm_pk_bf16(v[2], v[3]);
    const unsigned l0 = pm_pk_bf16(v[0] - pm_bf_lo(h0), v[1] - pm_bf_hi(h0));
    const unsigned l1 = pm_pk_bf16(v[2] - pm_bf_lo(h1), v[3] - pm_bf_hi(h1));
    unsigned short* q = plane + f * PM_DWW_LDK + kq * 4;
    *reinterpret_cast<uint2*>(q) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(q + nfeat * PM_DWW_LDK) = uint2{l0, l1};
  };
  auto stage = [&](const f32x4 (&ra)[2], const f32x4 (&rb)[2], int st) {
    unsigned short* sa = dww_lds + st * PM_DWW_STAGE;
    unsigned short* sb = sa + 2 * PM_DWW_TM * PM_DWW_LDK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + PM_DW_NT * j;
      put(sa, PM_DWW_TM, idx >> 3, idx & 7, ra[j]);
      bs[j] += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + PM_DW_NT * j;
      put(sb, PM_DWW_TN, idx >> 3, idx & 7, rb[j]);
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfmas = [&](int st) {
    const unsigned short* sa = dww_lds + st * PM_DWW_STAGE;
    const unsigned short* sb = sa + 2 * PM_DWW_TM * PM_DWW_LDK;
    f32x4 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* q = sa + (wm * 32 + i * 16 + c16) * PM_DWW_LDK + g * 8;
      ah[i] = *reinterpret_cast<const f32x4*>(q);
      al[i] = *reinterpret_cast<const f32x4*>(q + PM_DWW_TM * PM_DWW_LDK);
    }
    // one B fragment pair at a time (registers); per accumulator the smallest contributions first:
    // lo x hi, hi x lo, hi x hi
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned short* q = sb + (wn * 64 + j * 16 + c16) * PM_DWW_LDK + g * 8;
      const f32x4 bh = *reinterpret_cast<const f32x4*>(q);
      const f32x4 bl = *reinterpret_cast<const f32x4*>(q + PM_DWW_TN * PM_DWW_LDK);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = pm_mfma_bf<false>(al[i], bh, acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = pm_mfma_bf<false>(ah[i], bl, acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = pm_mfma_bf<false>(ah[i], bh, acc[i][j]);
    }
  };
  // step c from LDS stage st; the registers receive step c + 2 meanwhile (the CU's other workgroup covers what of
  // that round trip the MFMAs of one step do not)
  // (two steps ahead -- a second register set, the loop unrolled by two -- does not fit the 128 registers of two
  //  workgroups per CU: the allocator spilled freshly loaded operands behind vmcnt(0); with one A fragment pair at a
  //  time to make room, 24 spills.  Not run.)
  // (nor does issuing the next fetch right behind the staging, in front of the barrier: 6.5 -> 7.2 ms.)
  auto sweep = [&](auto fast) {
    fetch(fast, ra0, rb0, R.c_lo);
    stage(ra0, rb0, 0);
    __syncthreads();
    int st = 0;
    for (int c = R.c_lo; c < R.c_hi; c += 2, st ^= 1) {
      const bool more = c + 2 < R.c_hi;
      if (more) fetch(fast, ra0, rb0, c + 2);
      mfmas(st);
      if (more) stage(ra0, rb0, st ^ 1);
      __syncthreads();
    }
  };
  if (whole) sweep(PmTrue{});
  else sweep(PmFalse{});
  // partial tile: lane holds dW[o = m0 + 32 wm + 16 i + 4 g + r][k = n0 + 64 wn + 16 j + c16]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = U.n0 + wn * 64 + j * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = U.m0 + wm * 32 + i * 16 + 4 * g + r;
        if (o < O && k < K) pm_dw_put(part + A.w_off[l] + (size_t)o * K + k, acc[i][j][r], R.add);
      }
    }
  if (U.n0 == 0) {
    // bias gradient: the eight threads that staged one delta feature hold its partial row sums
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v = bs[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const int idx = tid + PM_DW_NT * j, o = U.m0 + (idx >> 3);
      if ((idx & 7) == 0 && o < O) pm_dw_put(part + A.b_off[l] + o, v, R.add);
    }
  }
}

// ---------------------------------------------------------------------------
// Wide layers from a PRE-SPLIT stash (round 6; the wide sweeps of pmbrl_wide.h write it: pw_stash_pre).  The kernel above
// fetches fp32 tiles into registers, splits them into two bf16 pieces and stages them through LDS: per K = 32 step and
// thread ~80 vector instructions beside 24 MFMAs, one register set (no fetch in flight while it converts), MfmaUtil 46 %.
// The split is what the sweeps' epilogues have in registers anyway (the adjoint's bit for bit: its LDS planes ARE these
// pieces).  Stash block of one (step, workgroup), 128 KB as before:
//     [piece (hi, lo)][tile of 16 features (32)][row (64)][16 features] bf16
// -- row-major inside a tile: an epilogue lane holds four consecutive features of a row (ONE 8-byte store per piece
// instead of four 4-byte ones), and a [32 rows][16 features] sub-tile -- what one K = 32 step needs of a tile -- is 1 KB
// CONTIGUOUS in HBM: one global_load_lds_dwordx4 per sub-tile and wave, no register, no conversion, no ds_write.  The
// MFMA operands (8 consecutive rows of one feature per lane) come out of the row-major image by ds_read_b64_tr_b16
// (lane t of a 16-lane group addresses row t >> 2, 8-byte chunk t & 3 of a [4 rows][16] block and receives column t:
// tools/ubench/tr_probe.hip); a sub-tile's rows are 32 bytes apart: the conflict-free image of the transposing read.
// A workgroup = 8 waves (2 x 4), ONE per CU, a 256 x 256 output tile of one layer over its row-step range: 64 KB of
// operand pieces per K = 32 step for 3 k cycles of MFMAs per SIMD -- 21 B/clk, inside the ~30 B/clk a CU pulls from L2
// (the first build, 256 x 128 tiles on 16 waves, needed 31 B/clk: 3.1 ms with its MFMAs compiled out, 4.3 with them).
// Two LDS stages of 64 KB: the transfers of step i + 1 run while the matrix core works on step i; one raw barrier per step.
// A wave keeps 8 x 4 accumulator tiles, the four B sub-tiles' operands for the whole step and the A operands of one
// sub-tile a read ahead.  Bias gradients by two extra MFMAs per A sub-tile against a column of ones (waves of the first
// column of wave tiles, units of the first input tile).
#define PM_DWP_TM 256
#define PM_DWP_TN 256
#define PM_DWP_NW 8
#define PM_DWP_NT (PM_DWP_NW * 64)
#define PM_DWP_STAGE (64 * 1024)                 // bytes: A 2 pieces x 16 sub-tiles + B 2 pieces x 16 sub-tiles, 1 KB each
#define PM_DWP_LDS_BYTES (2 * PM_DWP_STAGE)
#define PM_DWP_BLOCK 131072u                     // bytes of a stash block: 2 pieces x 32 tiles x 64 rows x 16 x 2
__global__ __launch_bounds__(PM_DWP_NT, 1) void pm_dw_wide_pre_kernel(const DwArgs A, const DwUnit* __restrict__ units,
                                                                      int n_units) {
  extern __shared__ __attribute__((aligned(16))) unsigned short dww_lds[];
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;       // (the tiles of a row-step range on ONE XCD: see pm_dw_wide_kernel)
  const int row = (jx / n_units) * 8 + xcd, unit = jx % n_units;
  if (row >= A.nsplit) return;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  const DwUnit U = units[unit];
  const int l = U.layer;
  const int O = A.dim[l + 1], K = A.dim[l];
  float* part = A.part + (size_t)row * A.part_stride;
  const int g = lane >> 4, c16 = lane & 15;
  const int wm = wid & 1, wn = wid >> 1;      // 128 output features (8 tiles) x 64 input features (4 tiles) per wave
  if (R.zero) {
    for (int e = tid; e < PM_DWP_TM * PM_DWP_TN; e += PM_DWP_NT) {
      const int o = U.m0 + e / PM_DWP_TN, k = U.n0 + e % PM_DWP_TN;
      if (o < O && k < K) part[A.w_off[l] + (size_t)o * K + k] = 0.f;
    }
    if (U.n0 == 0)
      for (int e = tid; e < PM_DWP_TM; e += PM_DWP_NT)
        if (U.m0 + e < O) part[A.b_off[l] + U.m0 + e] = 0.f;
    return;
  }
  // (uniform bases of the two stashes at this unit's first tile; a step = chunks c, c + 1 = rows 32 hh .. 32 hh + 31 of block c >> 2)
  const char* gb = reinterpret_cast<const char*>(A.gT[l]) + (size_t)(U.m0 >> 4) * 2048;
  const char* ab = reinterpret_cast<const char*>(A.actT[l]) + (size_t)(U.n0 >> 4) * 2048;
  const unsigned lds0 = (unsigned)(unsigned long long)(const void*)dww_lds;
  // A step's 64 KB reach LDS over BOTH paths a CU has: the A operand's 32 sub-tile slots (piece x 16) by LDS-DMA, four
  // instructions a wave; the B operand's 32 through registers -- four 16-byte chunks a thread, loaded a step ahead, written
  // (ds_write_b128, the same lane-linear image) behind the step's barrier.  (All 64 slots by DMA: the kernel ran at the
  // DMA's rate, ~16 B/clk/CU -- 4.0 ms, 2.9 ms with its MFMAs compiled out; touching the lines a few steps ahead changed
  // nothing: it is a rate, not a latency.)
  auto dma = [&](int c, int st) {
    const size_t boff = (size_t)(c >> 2) * PM_DWP_BLOCK + (size_t)((c >> 1) & 1) * 1024 + (size_t)lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = wid + 8 * j;                    // slot: piece idx >> 4, sub-tile idx & 15
      const char* src = gb + (size_t)(j >> 1) * 65536 + (size_t)(idx & 15) * 2048;
      const unsigned dst = (unsigned)st * PM_DWP_STAGE + (unsigned)idx * 1024u;
#ifndef PM_EXP_DWP_NODMA
      __builtin_amdgcn_global_load_lds((glb_vp)(src + boff), (lds_vp)(reinterpret_cast<char*>(dww_lds) + dst), 16, 0, 0);
#endif
    }
  };
  f32x4 breg[4];
  auto bload = [&](int c) {
    const size_t boff = (size_t)(c >> 2) * PM_DWP_BLOCK + (size_t)((c >> 1) & 1) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + PM_DWP_NT * j, slot = q >> 6;      // chunk q of the B region: slot q >> 6 (piece slot >> 4), 16 bytes (q & 63)
      breg[j] = *reinterpret_cast<const f32x4*>(ab + (size_t)(slot >> 4) * 65536 + (size_t)(slot & 15) * 2048 + boff + (size_t)(q & 63) * 16);
    }
  };
  auto bwrite = [&](int st) {
    char* dst = reinterpret_cast<char*>(dww_lds) + (size_t)st * PM_DWP_STAGE + 32768;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (size_t)(tid + PM_DWP_NT * j) * 16) = breg[j];
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool bias = U.n0 == 0 && wn == 0;
  const f32x4 ones = __builtin_bit_cast(f32x4, (pr_u32x4_dw){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  // transposing reads: lane (t, kq) addresses row 8 kq + (t >> 2) (+ 4: the second read), 8-byte chunk t & 3 of a sub-tile
  const unsigned la = lds0 + (unsigned)((8 * g + (c16 >> 2)) * 32 + (c16 & 3) * 8);
  auto tr2 = [&](f32x4& d, unsigned addr, auto offc) {
    constexpr int off = decltype(offc)::value;
    unsigned long long r0, r1;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r0) : "v"(addr), "n"(off));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(off + 128));
    d = __builtin_bit_cast(f32x4, (pr_u64x2_dw){r0, r1});
  };
  const int n_steps = (R.c_hi - R.c_lo + 1) >> 1;
  dma(R.c_lo, 0);
  bload(R.c_lo);
  bwrite(0);
  if (n_steps > 1) bload(R.c_lo + 2);
  int st = 0;
  for (int i = 0; i < n_steps; ++i) {
    // this wave's transfers and loads for step i (+ 1) have landed, then everybody's -- and everybody is done with step
    // i - 1, whose stage takes step i + 1 while the matrix core works on step i
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(breg[0]), "+v"(breg[1]), "+v"(breg[2]), "+v"(breg[3]) : : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + 1 < n_steps) {
      bwrite(st ^ 1);
      dma(R.c_lo + 2 * (i + 1), st ^ 1);
    }
    if (i + 2 < n_steps) bload(R.c_lo + 2 * (i + 2));
    const unsigned sa = la + (unsigned)st * PM_DWP_STAGE + (unsigned)wm * 8192u;             // A: sub-tiles 8 wm .. 8 wm + 7
    const unsigned sb = la + (unsigned)st * PM_DWP_STAGE + 32768u + (unsigned)wn * 4096u;    // B: sub-tiles 4 wn .. 4 wn + 3
    f32x4 bh[4], bl[4], ah[2], al[2];
    tr2(bh[0], sb, std::integral_constant<int, 0>{});      tr2(bl[0], sb, std::integral_constant<int, 16384>{});
    tr2(bh[1], sb, std::integral_constant<int, 1024>{});   tr2(bl[1], sb, std::integral_constant<int, 16384 + 1024>{});
    tr2(bh[2], sb, std::integral_constant<int, 2048>{});   tr2(bl[2], sb, std::integral_constant<int, 16384 + 2048>{});
    tr2(bh[3], sb, std::integral_constant<int, 3072>{});   tr2(bl[3], sb, std::integral_constant<int, 16384 + 3072>{});
    tr2(ah[0], sa, std::integral_constant<int, 0>{});      tr2(al[0], sa, std::integral_constant<int, 16384>{});
    pr_for_dw<8>([&](auto ic) {
      constexpr int i2 = decltype(ic)::value;
      constexpr int cur = i2 & 1, nxt = cur ^ 1;
      if constexpr (i2 < 7) {
        tr2(ah[nxt], sa, std::integral_constant<int, (i2 + 1) * 1024>{});
        tr2(al[nxt], sa, std::integral_constant<int, 16384 + (i2 + 1) * 1024>{});
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[cur]), "+v"(al[cur]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]),
                     "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al[cur]));
      }
#ifdef PM_EXP_DWP_NOMFMA
      acc[i2][0] += ah[cur] + al[cur] + bh[0] + bl[0] + bh[1] + bl[1] + bh[2] + bl[2] + bh[3] + bl[3];
#else
      // per accumulator the smallest contributions first: lo x hi, hi x lo, hi x hi (pm_dw_wide_kernel's order)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i2][j] = pm_mfma_bf<false>(al[cur], bh[j], acc[i2][j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i2][j] = pm_mfma_bf<false>(ah[cur], bl[j], acc[i2][j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i2][j] = pm_mfma_bf<false>(ah[cur], bh[j], acc[i2][j]);
      if (bias) {
        accb[i2] = pm_mfma_bf<false>(al[cur], ones, accb[i2]);
        accb[i2] = pm_mfma_bf<false>(ah[cur], ones, accb[i2]);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
    st ^= 1;
  }
  // partial tile: lane holds dW[o = m0 + 128 wm + 16 i + 4 g + r][k = n0 + 64 wn + 16 j + c16]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = U.n0 + wn * 64 + j * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = U.m0 + wm * 128 + i * 16 + 4 * g + r;
        if (o < O && k < K) pm_dw_put(part + A.w_off[l] + (size_t)o * K + k, acc[i][j][r], R.add);
      }
    }
  if (bias && c16 == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = U.m0 + wm * 128 + i * 16 + 4 * g + r;
        if (o < O) pm_dw_put(part + A.b_off[l] + o, accb[i][r], R.add);
      }
  }
}

// The two NARROW layers beside the 512 x 512 ones -- the first layer (512 x <= 32: its delta stash is 512 wide) and the head
// (<= 16 x 512: its input stash is) -- from the same pre-split stashes: with every 512-wide stash of the wide sweeps in ONE
// form the sweeps carry one form of the stash stores (two forms in one kernel cost the forward sweep 0.5 ms).  C = Wide^T
// Narrow over the row-steps: the wide operand (all 32 tiles x 2 pieces = 64 KB per K = 32 step: the high pieces by LDS-DMA,
// the low ones through registers) is the MFMA's A operand, wave w its tiles 4 w .. 4 w + 3; the narrow one (fp32 feature-major
// rows, as every family writes it) is read straight into B operand registers and split there.  wide_is_g: C = dW (first
// layer), bias sums from the wide operand; else C = dW^T (head), bias sums from the narrow one.  Bound by the stash read.
template <int NT>
__global__ __launch_bounds__(PM_DWP_NT, 1) void pm_dw_narrow_pre_kernel(const DwArgs A, int l, int wide_is_g) {
  extern __shared__ __attribute__((aligned(16))) unsigned short dww_lds[];
  typedef __attribute__((address_space(3))) void* lds_vp;
  typedef const __attribute__((address_space(1))) void* glb_vp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  const int O = A.dim[l + 1], K = A.dim[l];
  float* part = A.part + (size_t)row * A.part_stride;
  const int g = lane >> 4, c16 = lane & 15;
  if (R.zero) {
    for (int e = tid; e < O * K; e += PM_DWP_NT) part[A.w_off[l] + e] = 0.f;
    for (int e = tid; e < O; e += PM_DWP_NT) part[A.b_off[l] + e] = 0.f;
    return;
  }
  const char* wb = reinterpret_cast<const char*>(wide_is_g ? A.gT[l] : A.actT[l]);
  const float* nb = wide_is_g ? A.actT[l] : A.gT[l];
  const int NF16 = (wide_is_g ? A.nt[l] : A.nt[l + 1]) * 16;      // the narrow stash's feature rows per block
  const unsigned lds0 = (unsigned)(unsigned long long)(const void*)dww_lds;
  auto dma = [&](int c, int st) {      // high pieces: slots 0 .. 31, four a wave
    const size_t boff = (size_t)(c >> 2) * PM_DWP_BLOCK + (size_t)((c >> 1) & 1) * 1024 + (size_t)lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = wid + 8 * j;
      __builtin_amdgcn_global_load_lds((glb_vp)(wb + (size_t)idx * 2048 + boff),
                                       (lds_vp)(reinterpret_cast<char*>(dww_lds) + (unsigned)st * PM_DWP_STAGE + (unsigned)idx * 1024u), 16, 0, 0);
    }
  };
  f32x4 breg[4];
  auto bload = [&](int c) {            // low pieces: slots 32 .. 63, four 16-byte chunks a thread
    const size_t boff = (size_t)(c >> 2) * PM_DWP_BLOCK + (size_t)((c >> 1) & 1) * 1024 + 65536;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + PM_DWP_NT * j;
      breg[j] = *reinterpret_cast<const f32x4*>(wb + (size_t)(q >> 6) * 2048 + boff + (size_t)(q & 63) * 16);
    }
  };
  auto bwrite = [&](int st) {
    char* dst = reinterpret_cast<char*>(dww_lds) + (size_t)st * PM_DWP_STAGE + 32768;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (size_t)(tid + PM_DWP_NT * j) * 16) = breg[j];
  };
  // the narrow operand of step c: lane (n, kq) holds rows 8 kq .. 8 kq + 7 of feature 16 nt + n -- 32 contiguous bytes of the
  // feature-major block
  f32x4 nraw[NT][2];
  auto nload = [&](int c) {
    const float* blk = nb + (size_t)(c >> 2) * NF16 * A.Rw + ((c >> 1) & 1) * 32 + 8 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float* q = blk + (size_t)(16 * nt + c16) * A.Rw;
      nraw[nt][0] = *reinterpret_cast<const f32x4*>(q);
      nraw[nt][1] = *reinterpret_cast<const f32x4*>(q + 4);
    }
  };
  f32x4 acc[4][NT], accb[4], accn[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) accn[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 ones = __builtin_bit_cast(f32x4, (pr_u32x4_dw){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  const unsigned la = lds0 + (unsigned)((8 * g + (c16 >> 2)) * 32 + (c16 & 3) * 8);
  auto tr2 = [&](f32x4& d, unsigned addr, auto offc) {
    constexpr int off = decltype(offc)::value;
    unsigned long long r0, r1;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r0) : "v"(addr), "n"(off));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(off + 128));
    d = __builtin_bit_cast(f32x4, (pr_u64x2_dw){r0, r1});
  };
  const int n_steps = (R.c_hi - R.c_lo + 1) >> 1;
  dma(R.c_lo, 0);
  bload(R.c_lo);
  bwrite(0);
  if (n_steps > 1) bload(R.c_lo + 2);
  nload(R.c_lo);
  int st = 0;
  for (int i = 0; i < n_steps; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(breg[0]), "+v"(breg[1]), "+v"(breg[2]), "+v"(breg[3]) : : "memory");
    // (this step's narrow operand: split into two bf16 pieces here -- a few hundred values a wave)
    f32x4 nh[NT], nl[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      pm_u32x2 p0[2], p1[2];
      pm_split4<2, false>(nraw[nt][0], p0);
      pm_split4<2, false>(nraw[nt][1], p1);
      nh[nt] = __builtin_bit_cast(f32x4, (pr_u32x4_dw){p0[0][0], p0[0][1], p1[0][0], p1[0][1]});
      nl[nt] = __builtin_bit_cast(f32x4, (pr_u32x4_dw){p0[1][0], p0[1][1], p1[1][0], p1[1][1]});
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + 1 < n_steps) {
      bwrite(st ^ 1);
      dma(R.c_lo + 2 * (i + 1), st ^ 1);
      nload(R.c_lo + 2 * (i + 1));
    }
    if (i + 2 < n_steps) bload(R.c_lo + 2 * (i + 2));
    const unsigned sa = la + (unsigned)st * PM_DWP_STAGE + (unsigned)wid * 4096u;      // sub-tiles 4 wid .. 4 wid + 3, high pieces
    f32x4 ah[4], al[4];
    tr2(ah[0], sa, std::integral_constant<int, 0>{});      tr2(al[0], sa, std::integral_constant<int, 32768>{});
    tr2(ah[1], sa, std::integral_constant<int, 1024>{});   tr2(al[1], sa, std::integral_constant<int, 32768 + 1024>{});
    tr2(ah[2], sa, std::integral_constant<int, 2048>{});   tr2(al[2], sa, std::integral_constant<int, 32768 + 2048>{});
    tr2(ah[3], sa, std::integral_constant<int, 3072>{});   tr2(al[3], sa, std::integral_constant<int, 32768 + 3072>{});
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) acc[i2][j] = pm_mfma_bf<false>(al[i2], nh[j], acc[i2][j]);
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) acc[i2][j] = pm_mfma_bf<false>(ah[i2], nl[j], acc[i2][j]);
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) acc[i2][j] = pm_mfma_bf<false>(ah[i2], nh[j], acc[i2][j]);
    }
    if (wide_is_g) {
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) accb[i2] = pm_mfma_bf<false>(al[i2], ones, accb[i2]);
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2) accb[i2] = pm_mfma_bf<false>(ah[i2], ones, accb[i2]);
    } else if (wid == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        accn[j] = pm_mfma_bf<false>(ones, nl[j], accn[j]);
        accn[j] = pm_mfma_bf<false>(ones, nh[j], accn[j]);
      }
    }
    st ^= 1;
  }
  // lane holds C[m = 64 wid + 16 i + 4 g + r][n = 16 j + c16]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = 16 * j + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 64 * wid + 16 * i + 4 * g + r;
        if (wide_is_g) {
          if (m < O && n < K) pm_dw_put(part + A.w_off[l] + (size_t)m * K + n, acc[i][j][r], R.add);
        } else {
          if (n < O && m < K) pm_dw_put(part + A.w_off[l] + (size_t)n * K + m, acc[i][j][r], R.add);
        }
      }
    }
  if (wide_is_g) {
    if (c16 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 64 * wid + 16 * i + 4 * g + r;
          if (m < O) pm_dw_put(part + A.b_off[l] + m, accb[i][r], R.add);
        }
    }
  } else if (wid == 0 && g == 0) {      // (every row of the ones product is the column sum: row 0 = lane group 0, register 0)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      if (16 * j + c16 < O) pm_dw_put(part + A.b_off[l] + 16 * j + c16, accn[j][0], R.add);
  }
}

// A WHOLE layer of up to 13 x 13 tiles (the 200 x 200 layer of the shipped shapes) per workgroup: at that width
// the 128 x 128 tiles above leave a workgroup a dozen steps of one exposed memory round trip each, in two rounds
// over the chip (65 us for 173 MB at the cart-pole shape).  One workgroup per row-step range and CU instead:
// 53 KB per step in flight per CU, every stash byte fetched once, wave w accumulates tile rows w and w + 8.
#define PM_DWL_NT 13
#define PM_DWL_F (PM_DWL_NT * 16)
#define PM_DWL_STAGE (2 * PM_DWL_F * 2 * PM_DWW_LDK)
#define PM_DWL_LDS_BYTES (2 * PM_DWL_STAGE * 2)
#define PM_DWL_Q ((PM_DWL_F * 8 + PM_DW_NT - 1) / PM_DW_NT)      // float4 quads per thread and operand: 4
__global__ __launch_bounds__(PM_DW_NT, 2) void pm_dw_layer_kernel(const DwArgs A, int l) {
  extern __shared__ __attribute__((aligned(16))) unsigned short dww_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x;
  DwRange R;
  if (!pm_dw_range(A, row, R)) return;
  const int Fo16 = A.nt[l + 1] * 16, Fi16 = A.nt[l] * 16;
  const int O = A.dim[l + 1], K = A.dim[l];
  float* part = A.part + (size_t)row * A.part_stride;
  const int g = lane >> 4, c16 = lane & 15;
  if (R.zero) {
    for (int e = tid; e < O * K; e += PM_DW_NT) part[A.w_off[l] + e] = 0.f;
    for (int e = tid; e < O; e += PM_DW_NT) part[A.b_off[l] + e] = 0.f;
    return;
  }
  const float* gbase = A.gT[l];
  const float* abase = A.actT[l];
  const size_t gblk = (size_t)Fo16 * A.Rw, ablk = (size_t)Fi16 * A.Rw;
  f32x4 ra[PM_DWL_Q], rb[PM_DWL_Q];
  float bs[PM_DWL_Q];
#pragma unroll
  for (int j = 0; j < PM_DWL_Q; ++j) bs[j] = 0.f;
  auto fetch1 = [&](f32x4& dst, const float* base, size_t blk, int F16, int idx, int c, bool second) {
    const int f = idx >> 3, kq = idx & 7;
    const float* p;
    bool live = f < F16;
    if (A.RT >= 2) {
      const int b = c / A.RT, rt = c - b * A.RT;
      p = base + (size_t)b * blk + (size_t)f * A.Rw + rt * 16 + kq * 4;
    } else {
      p = base + (size_t)(c + (kq >> 2)) * blk + (size_t)f * 16 + (kq & 3) * 4;
      live = live && (kq < 4 || second);
    }
    dst = live ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // A step with both its 16-row chunks (every step but an odd range's last): scalar base of the step + a per-thread
  // constant -- loads with a scalar base and one 32-bit lane offset, no branch per load (a feature row past the layer's
  // is clamped to the last one: it is never staged).  fetch1 above is what remains for the odd step; it was ~25
  // instructions and a branch per load, eight loads a thread and step (as in pm_dw_wide_kernel).
  const bool pow2 = (A.RT & (A.RT - 1)) == 0;
  const int rt_sh = A.RT >= 4 ? 2 : (A.RT >= 2 ? 1 : 0);
  unsigned offA[PM_DWL_Q], offB[PM_DWL_Q];
#pragma unroll
  for (int j = 0; j < PM_DWL_Q; ++j) {
    const int idx = tid + PM_DW_NT * j, f = idx >> 3, kq = idx & 7;
    const int fa = min(f, Fo16 - 1), fb = min(f, Fi16 - 1);
    if (A.RT >= 2) {
      offA[j] = (unsigned)(fa * A.Rw + kq * 4);
      offB[j] = (unsigned)(fb * A.Rw + kq * 4);
    } else {
      offA[j] = (unsigned)((kq >> 2) * (int)gblk + fa * 16 + (kq & 3) * 4);
      offB[j] = (unsigned)((kq >> 2) * (int)ablk + fb * 16 + (kq & 3) * 4);
    }
  }
  auto fetch = [&](int c) {
    const bool second = c + 1 < R.c_hi;
    if (second && pow2) {
      const int b = A.RT >= 2 ? c >> rt_sh : c, rt = A.RT >= 2 ? (c & (A.RT - 1)) : 0;
      const float* pg = gbase + (size_t)b * gblk + rt * 16;
      const float* pa = abase + (size_t)b * ablk + rt * 16;
#pragma unroll
      for (int j = 0; j < PM_DWL_Q; ++j) ra[j] = *reinterpret_cast<const f32x4*>(pg + offA[j]);
#pragma unroll
      for (int j = 0; j < PM_DWL_Q; ++j) rb[j] = *reinterpret_cast<const f32x4*>(pa + offB[j]);
      return;
    }
#pragma unroll
    for (int j = 0; j < PM_DWL_Q; ++j) {
      const int idx = tid + PM_DW_NT * j;      // (idx >= 13 * 16 * 8: feature >= 208 >= F16 -> zeros, never staged)
      fetch1(ra[j], gbase, gblk, Fo16, idx, c, second);
      fetch1(rb[j], abase, ablk, Fi16, idx, c, second);
    }
  };
  auto put = [&](unsigned short* plane, int f, int kq, const f32x4& v) {
    const unsigned h0 = pm_pk_bf16(v[0], v[1]), h1 = pm_pk_bf16(v[2], v[3]);
    const unsigned l0 = pm_pk_bf16(v[0] - pm_bf_lo(h0), v[1] - pm_bf_hi(h0));
    const unsigned l1 = pm_pk_bf16(v[2] - pm_bf_lo(h1), v[3] - pm_bf_hi(h1));
    unsigned short* q = plane + f * PM_DWW_LDK + kq * 4;
    *reinterpret_cast<uint2*>(q) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(q + PM_DWL_F * PM_DWW_LDK) = uint2{l0, l1};
  };
  auto stage = [&](int st) {
    unsigned short* sa = dww_lds + st * PM_DWL_STAGE;
    unsigned short* sb = sa + 2 * PM_DWL_F * PM_DWW_LDK;
#pragma unroll
    for (int j = 0; j < PM_DWL_Q; ++j) {
      const int idx = tid + PM_DW_NT * j;
      if (idx < PM_DWL_F * 8) {
        put(sa, idx >> 3, idx & 7, ra[j]);
        put(sb, idx >> 3, idx & 7, rb[j]);
        bs[j] += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
      }
    }
  };
  f32x4 acc[2][PM_DWL_NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PM_DWL_NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool two = wid + 8 < PM_DWL_NT;      // this wave owns tile rows wid and (waves 0..4) wid + 8
  auto mfmas = [&](int st) {
    const unsigned short* sa = dww_lds + st * PM_DWL_STAGE;
    const unsigned short* sb = sa + 2 * PM_DWL_F * PM_DWW_LDK;
    f32x4 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tr = (i == 0 || two) ? wid + 8 * i : wid;
      const unsigned short* q = sa + (tr * 16 + c16) * PM_DWW_LDK + g * 8;
      ah[i] = *reinterpret_cast<const f32x4*>(q);
      al[i] = *reinterpret_cast<const f32x4*>(q + PM_DWL_F * PM_DWW_LDK);
    }
#pragma unroll
    for (int j = 0; j < PM_DWL_NT; ++j) {
      const unsigned short* q = sb + (j * 16 + c16) * PM_DWW_LDK + g * 8;
      const f32x4 bh = *reinterpret_cast<const f32x4*>(q);
      const f32x4 bl = *reinterpret_cast<const f32x4*>(q + PM_DWL_F * PM_DWW_LDK);
      // per accumulator the smallest contributions first: lo x hi, hi x lo, hi x hi
      acc[0][j] = pm_mfma_bf<false>(al[0], bh, acc[0][j]);
      if (two) acc[1][j] = pm_mfma_bf<false>(al[1], bh, acc[1][j]);
      acc[0][j] = pm_mfma_bf<false>(ah[0], bl, acc[0][j]);
      if (two) acc[1][j] = pm_mfma_bf<false>(ah[1], bl, acc[1][j]);
      acc[0][j] = pm_mfma_bf<false>(ah[0], bh, acc[0][j]);
      if (two) acc[1][j] = pm_mfma_bf<false>(ah[1], bh, acc[1][j]);
    }
  };
  fetch(R.c_lo);
  stage(0);
  __syncthreads();
  int st = 0;
  for (int c = R.c_lo; c < R.c_hi; c += 2, st ^= 1) {
    const bool more = c + 2 < R.c_hi;
    if (more) fetch(c + 2);
    mfmas(st);
    if (more) stage(st ^ 1);
    __syncthreads();
  }
  // lane holds dW[o = 16 (wid + 8 i) + 4 g + r][k = 16 j + c16].  Through LDS and out in rows (round 6): written from the
  // accumulators a store instruction covered four 64-byte pieces of four different rows -- 160 KB per workgroup in 2 560
  // such pieces, all 252 workgroups at the end of the launch at once.  Two halves (tile rows 0..6, 7..12: 7 x 16 rows of
  // 212 floats fit the stages' LDS), each copied out as whole rows, 16 bytes per lane where the row starts allow it.
  {
    constexpr int LDE = PM_DWL_F + 4;            // (row stride 212: the four lane groups' rows on different banks)
    float* ep = reinterpret_cast<float*>(dww_lds);
    float* wpart = part + A.w_off[l];
    const bool vec4 = (K & 3) == 0 && (A.w_off[l] & 3) == 0;
    __syncthreads();                             // (the last step's operands have been read)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int tr0 = 7 * half;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int tr = wid + 8 * i;
        if ((i == 0 || two) && (half == 0 ? tr < 7 : tr >= 7)) {
#pragma unroll
          for (int j = 0; j < PM_DWL_NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[((tr - tr0) * 16 + 4 * g + r) * LDE + j * 16 + c16] = acc[i][j][r];
        }
      }
      __syncthreads();
      const int o0 = tr0 * 16, nro = min(half == 0 ? 7 * 16 : 6 * 16, max(0, O - o0));   // rows of this half that exist
      if (vec4) {
        const int K4 = K >> 2;
        for (int e = tid; e < nro * K4; e += PM_DW_NT) {
          const int ro = e / K4, kq = e - ro * K4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(ep + ro * LDE + 4 * kq);
          f32x4* q = reinterpret_cast<f32x4*>(wpart + (size_t)(o0 + ro) * K + 4 * kq);
          *q = R.add ? *q + v : v;
        }
      } else {
        for (int e = tid; e < nro * K; e += PM_DW_NT) {
          const int ro = e / K, k = e - ro * K;
          pm_dw_put(wpart + (size_t)(o0 + ro) * K + k, ep[ro * LDE + k], R.add);
        }
      }
      __syncthreads();
    }
  }
  // bias gradient: the eight threads that staged one delta feature hold its partial row sums
#pragma unroll
  for (int j = 0; j < PM_DWL_Q; ++j) {
    float v = bs[j];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    const int idx = tid + PM_DW_NT * j, o = idx >> 3;
    if ((idx & 7) == 0 && o < O) pm_dw_put(part + A.b_off[l] + o, v, R.add);
  }
}

// Squared norm of the gradient for clip_grad_norm_ (algorithms/mc_pilco.py:199-200): partial sums over blocks of 256
// consecutive elements, ONE definition for the two kernels that form them -- pm_gradnorm_kernel (its own launch) and
// pm_dw_reduce (round 6: the fused iteration's reduction forms them on the way, one launch less) -- so that the two
// forms of an iteration walk the same trajectory bit for bit.  Lane c of a wave holds elements 4 c .. 4 c + 3 of the
// block: squares added in element order, then a fixed butterfly over the 64 lanes.
#define PM_NORM_MAXB 8192      // blocks of 256 elements: gradients of up to 2 M parameters (beyond: pm_gradnorm_kernel strides)
__device__ double g_norm_part[PM_NORM_MAXB];
__device__ double g_loss_part[PM_NORM_MAXB];   // the loss' partial sums, one per workgroup of pm_dw_reduce (loss_r != nullptr)
__device__ int g_adam_go;
__device__ __forceinline__ double pm_sq4_wave(const f32x4& t, long long i0, long long n) {
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double x = i0 + r < n ? (double)t[r] : 0.0;
    s += x * x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  return s;
}
// the optimiser step of a guarded iteration is taken if the rollout completed: status[0] = steps the forward sweep
// completed, status[1] = set by the adjoint sweep when one of its exchanges timed out (the gradient is garbage then)
__device__ __forceinline__ void pm_adam_decide(const int* status, int expect, long long* step) {
  const int go = (!status || (status[0] >= expect && status[1] == 0)) ? 1 : 0;
  g_adam_go = go;
  if (go && step) step[0] += 1;
}

// grad[i] = sum_s part[s][i] in fixed order.  A workgroup = 64 float4 columns x 8 slices of the
// split range: every wave reads whole 1 KiB rows, a thread keeps 16 independent loads in flight.
// norm_on: also the partial sums of squares of ITS 256 elements and (block 0) the decision whether the guarded optimiser
// step is taken -- what pm_gradnorm_kernel does in a launch of its own.
__global__ __launch_bounds__(512) void pm_dw_reduce(const float* __restrict__ part, int nsplit, int n,
                                                    int stride, float* __restrict__ grad,
                                                    const int* __restrict__ nvalid = nullptr, int chunks_per_step = 0,
                                                    int chunks_per_split = 1, int norm_on = 0,
                                                    const int* __restrict__ status = nullptr, int expect = 0,
                                                    long long* __restrict__ step = nullptr,
                                                    const float* __restrict__ loss_r = nullptr,
                                                    const float* __restrict__ loss_w = nullptr, long long loss_n = 0,
                                                    long long loss_per_step = 0) {
  __shared__ f32x4 sm[8][64];
  __shared__ double sl_[8];
  // loss_r: the fused iteration's loss on the way -- this workgroup's slice of sum_i r_i w_i over the valid steps (requested
  // here, added below: pm_clip_adam_kernel adds the workgroups' sums in workgroup order).  One launch less per iteration.
  float lr4[4] = {0.f, 0.f, 0.f, 0.f}, lw4[4] = {0.f, 0.f, 0.f, 0.f};
  long long l_n = 0;
  const long long l_i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, l_st = (long long)gridDim.x * blockDim.x;
  if (loss_r) {
    l_n = nvalid ? min(loss_n, (long long)max(0, *nvalid) * loss_per_step) : loss_n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = min(l_i0 + u * l_st, loss_n - 1);
      lr4[u] = loss_r[j];
      lw4[u] = loss_w[j];
    }
  }
  if (nvalid) {   // truncated horizon: only the splits that own a chunk of a valid step wrote a partial
    const long long nc = (long long)max(0, *nvalid) * chunks_per_step;
    nsplit = (int)min((long long)nsplit, (nc + chunks_per_split - 1) / chunks_per_split);
  }
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + col;            // float4 column
  const int per = (nsplit + 7) / 8;
  const int k_lo = sl * per, k_hi = min(nsplit, k_lo + per);
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c4 * 4 < n) {
    const float* p = part + (size_t)c4 * 4;
    // (sixteen rows in flight per thread, the tail as one predicated batch of eight: at C2 a slice is 33 rows -- as
    //  8 + 8 + 8 + 8 + 1 it was five memory round trips in a row, now three; the order of the additions is the rows')
    int k = k_lo;
    for (; k + 16 <= k_hi; k += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ldg4(p + (size_t)(k + u) * stride);
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; k < k_hi; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldg4(p + (size_t)min(k + u, k_hi - 1) * stride);   // (beyond: dropped below)
#pragma unroll
      for (int u = 0; u < 8; ++u) s += k + u < k_hi ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  sm[sl][col] = s;
  if (loss_r) {
    double ls = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) ls += l_i0 + u * l_st < l_n ? (double)lr4[u] * (double)lw4[u] : 0.0;
    for (long long j = l_i0 + 4 * l_st; j < l_n; j += l_st) ls += (double)loss_r[j] * (double)loss_w[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o);
    if (col == 0) sl_[sl] = ls;
  }
  __syncthreads();
  if (loss_r && threadIdx.x == 0) {
    double t = sl_[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sl_[k];
    g_loss_part[blockIdx.x] = t;
  }
  if (sl == 0) {
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < n) {
      t = sm[0][col];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += sm[k][col];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c4 * 4 + r < n) grad[c4 * 4 + r] = t[r];
    }
    if (norm_on) {
      const double s2 = pm_sq4_wave(t, (long long)c4 * 4, n);
      if (col == 0) {
        g_norm_part[blockIdx.x] = s2;
        if (blockIdx.x == 0) pm_adam_decide(status, expect, step);
      }
    }
  }
}
#endif   // PM_MAIN_TU
