// Split-operand matrix-core GEMMs for the GENERAL kernel family (pmbrl_rollout.h): wide networks (the
// 3 x 512 stress shape), three-layer nets, angle_dims, per-step masks -- everything the latency-optimised
// sweeps (pmbrl_fast.h) do not cover.  Same arithmetic as there (pmbrl_split.h): every fp32 operand is two
// 16-bit pieces, a product is the three piece products of combined order < 2 on v_mfma_f32_16x16x32_{f16,bf16}
// with fp32 accumulation -- two fp16 pieces (22 bits) in the forward sweep, two bf16 pieces in the adjoint.
//
// What changes against the fp32 form of this family:
//   * an activation buffer [R][LD] floats is read as two piece planes [2][R][LDB] of 16-bit elements with
//     LDB = LD (same bytes), LDB = 16 (mod 32): conflict-free ds_read_b128 B operands;
//   * weights [tile][K32 block][piece][lane][8 x 16 bit] (pm_pack_all, planes = 2): the same 4 bytes per
//     weight as fp32, one global_load_dwordx4 per (tile, block, piece) and lane;
//   * per K = 32 three MFMAs of 16 cycles instead of eight fp32 MFMAs of 32: the hidden-width GEMMs stop being
//     bound by the matrix core (0.55 of the fp32 peak at 3 x 512) and become bound by the weight stream L2 -> CU;
//   * narrow GEMMs (heads, first-layer adjoints) still leave fp32 rows in the output buffer, which the
//     elementwise phases read as before; those phases write network inputs as piece planes.
// The weight loads of the tile GEMM are inline asm with explicit waits (see gemm_tiles_group_s).
#pragma once
#include "pmbrl_split.h"

#define PM_GS_CK 1   // K32 blocks per chunk (a ring of four chunks: three in flight while one feeds the MFMAs)

// scalar store of one value into the piece planes (the elementwise phases' network inputs)
template <int R, bool F16>
__device__ __forceinline__ void pm_put_planes(float* buf, unsigned ldb, unsigned row, unsigned col, float v) {
  unsigned short* pb = reinterpret_cast<unsigned short*>(buf);
  if constexpr (F16) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    pb[row * ldb + col] = __builtin_bit_cast(unsigned short, h);
    pb[(R + row) * ldb + col] = __builtin_bit_cast(unsigned short, l);
  } else {
    const unsigned a = pm_pk_bf16(v, 0.f) & 0xffffu;
    const unsigned b = pm_pk_bf16(v - pm_bf_lo(a), 0.f) & 0xffffu;
    pb[row * ldb + col] = (unsigned short)a;
    pb[(R + row) * ldb + col] = (unsigned short)b;
  }
}

// K32 blocks of a layer input of `nt` 16-wide tiles
__host__ __device__ inline int pm_kb32(int nt) { return (nt + 1) / 2; }

template <int NT>
struct GsFrag {
  f32x4 a[NT][PM_GS_CK][2];
};

// NT output tiles (ot0, ot0 + PM_NW, ...) x RT row tiles per wave; absent tiles of the last group are computed
// as duplicates of the group's first tile and dropped.
template <int RT, int NT, bool F16, class Epi>
__device__ __forceinline__ void gemm_tiles_group_s(const float* __restrict__ wf, int n_kb, int ot0, int n_ot,
                                                   const float* buf_in, unsigned ldb, int lane, Epi& epi) {
  typedef PmPairs<2> PP;
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  f32x4 acc[NT][RT];
  const float* wp[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW;
    wp[k] = wf + ((size_t)(ot < n_ot ? ot : ot0) * n_kb) * 512 + lane * 4;   // 2 pieces x 256 floats per block
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the epilogue's HBM / L2 operands of every (tile, row tile) of the group, in flight behind the K loop
  typename Epi::Pre pre[NT][RT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) pre[k][rt] = epi.pre(ot < n_ot ? ot : ot0, rt);
  }
  // A ring of four one-block chunks: three (NT * 2 KB each = 12 KB per wave at NT = 2) in flight while one feeds
  // the MFMAs.  At 3 x 512 the weights (6 MB a sweep) do not fit an XCD's L2 and a load takes ~2 k cycles to
  // return: with one two-block chunk ahead (8 KB in flight) the K loop ran at one chunk per round trip,
  // 32 B/clk/CU.  (Four TWO-block buffers were measured slower: 128 registers of landing space spill.)
  // The loads are inline asm with explicit waits (the rules of pmbrl_fast.h's weight stream; checked by
  // tools/check_inflight.py): left to the compiler, the wait in front of the first chunk of every loop iteration
  // came out as vmcnt(1) -- everything issued so far -- and the ring never had more than its last chunk in flight.
  GsFrag<NT> f0, f1, f2, f3;
  constexpr int NLD = NT * PM_GS_CK * 2;   // loads per chunk
  auto load = [&](GsFrag<NT>& f, int kb0) {
#pragma unroll
    for (int c = 0; c < PM_GS_CK; ++c) {
      const int kb = kb0 + c < n_kb ? kb0 + c : 0;   // past the end: a harmless re-load of block 0
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const float* q = wp[k] + (size_t)kb * 512;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(f.a[k][c][0]) : "v"(q));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=&v"(f.a[k][c][1]) : "v"(q));
      }
    }
  };
  // chunk f has landed: at most the three younger chunks are still outstanding
  auto wait = [&](GsFrag<NT>& f) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLD));
#pragma unroll
    for (int c = 0; c < PM_GS_CK; ++c)
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(f.a[k][c][p]));
  };
  auto compute = [&](GsFrag<NT>& f, int kb0) {
    wait(f);
#pragma unroll
    for (int c = 0; c < PM_GS_CK; ++c) {
      if (kb0 + c < n_kb) {
        BQ<RT, 2> b;
        bq_load<RT, 2>(b, lb, ldb, kb0 + c);
        BScaled<RT, F16> bs(b);
#pragma unroll
        for (int q = 0; q < PP::N; ++q)
#pragma unroll
          for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[k][rt] = pm_mfma_bf<F16>(f.a[k][c][PP::W[q]], pm_bsel<F16>(q, b, bs, rt), acc[k][rt]);
      }
    }
  };
  constexpr int S = PM_GS_CK;
  load(f0, 0);
  load(f1, S);
  load(f2, 2 * S);
  for (int kb0 = 0; kb0 < n_kb; kb0 += 4 * S) {
    load(f3, kb0 + 3 * S);
    compute(f0, kb0);
    load(f0, kb0 + 4 * S);
    compute(f1, kb0 + S);
    load(f1, kb0 + 5 * S);
    compute(f2, kb0 + 2 * S);
    load(f2, kb0 + 6 * S);
    compute(f3, kb0 + 3 * S);
  }
  // the look-ahead chunks (harmless re-loads past the end) land before their registers are reused
  asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
  for (int c = 0; c < PM_GS_CK; ++c)
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        asm volatile("" : "+v"(f0.a[k][c][p]));
        asm volatile("" : "+v"(f1.a[k][c][p]));
        asm volatile("" : "+v"(f2.a[k][c][p]));
      }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (ot0 + k * PM_NW < n_ot) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epi(ot0 + k * PM_NW, rt, acc[k][rt], pre[k][rt]);
    }
  }
}

// Layers with K <= 64 (n_kb <= 2 K32 blocks): the first layers (state / [state | action] -> hidden) and the adjoints
// of the heads (2D / 2U -> hidden).  The ring above is built for long K loops: at n_kb = 1 it issues seven chunk loads
// per tile group of which one is used and drains them all before the epilogue -- two exposed memory round trips per
// group, two groups per wave, 14-21 k cycles for a layer whose MFMAs take 0.4 k (profiles/r03a_phase_prof_stress32_mm.txt:
// 21 + 18 k forward, 14 + 17 k adjoint of a 435 k-cycle step at the C5 shape).  Here a wave takes FOUR tiles at once:
// every weight fragment and every epilogue operand of the group is requested up front, one round trip, then the MFMAs
// and the epilogues.
template <int RT, int NT, bool F16, class Epi>
__device__ __forceinline__ void gemm_tiles_group_small_s(const float* __restrict__ wf, int n_kb, int ot0, int n_ot,
                                                         const float* buf_in, unsigned ldb, int lane, Epi& epi) {
  typedef PmPairs<2> PP;
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  f32x4 acc[NT][RT];
  f32x4 a[NT][2][2];
  typename Epi::Pre pre[NT][RT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW < n_ot ? ot0 + k * PM_NW : ot0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int kb = c < n_kb ? c : 0;
#pragma unroll
      for (int p = 0; p < 2; ++p) a[k][c][p] = ldg4(wf + (((size_t)ot * n_kb + kb) * 2 + p) * 256 + lane * 4);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      pre[k][rt] = epi.pre(ot, rt);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c < n_kb) {
      BQ<RT, 2> b;
      bq_load<RT, 2>(b, lb, ldb, c);
      BScaled<RT, F16> bs(b);
#pragma unroll
      for (int q = 0; q < PP::N; ++q)
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[k][rt] = pm_mfma_bf<F16>(a[k][c][PP::W[q]], pm_bsel<F16>(q, b, bs, rt), acc[k][rt]);
    }
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (ot0 + k * PM_NW < n_ot) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) epi(ot0 + k * PM_NW, rt, acc[k][rt], pre[k][rt]);
    }
  }
}

template <int RT, bool F16, class Epi>
__device__ __forceinline__ void gemm_tiles_s(const float* __restrict__ wf, int n_ot, int n_kb, const float* buf_in,
                                             unsigned ldb, int wid, int lane, Epi& epi) {
  if (n_kb <= 2) {
    constexpr int NTS = RT <= 2 ? 4 : 2;
    for (int ot0 = wid; ot0 < n_ot; ot0 += NTS * PM_NW)
      gemm_tiles_group_small_s<RT, NTS, F16>(wf, n_kb, ot0, n_ot, buf_in, ldb, lane, epi);
    return;
  }
  constexpr int NT = 2;
  for (int ot0 = wid; ot0 < n_ot; ot0 += NT * PM_NW)
    gemm_tiles_group_s<RT, NT, F16>(wf, n_kb, ot0, n_ot, buf_in, ldb, lane, epi);
}

// ---------------------------------------------------------------------------
// IN-PLACE layers: the output overwrites the input buffer.  A wave keeps ALL its output tiles (NT tiles x RT row
// tiles) in accumulators over the whole K loop, the workgroup meets once every wave has read the input, and only
// then do the epilogues write.  One activation buffer instead of two is what lets a workgroup hold 64 rows of a
// 512-wide layer in LDS (135 KB) -- and 64 rows per weight fetch is what the wide sweeps are bound by: at 32 rows a
// 512 x 512 layer streams its 1 MB of weight pieces at the ~30 B/clk a CU pulls from L2 and the matrix pipe idles
// two thirds of the time.
// Contains a barrier: called by all waves, also those without a tile.
// ---------------------------------------------------------------------------
template <int RT, int NT, bool F16, class Epi>
__device__ __forceinline__ void gemm_tiles_inplace_s(const float* __restrict__ wf, int n_ot, int n_kb, const float* buf,
                                                     unsigned ldb, int wid, int lane, Epi& epi) {
  typedef PmPairs<2> PP;
  const unsigned short* lb = pm_plane_lane(buf, ldb, lane);
  const int ot0 = wid;
  f32x4 acc[NT][RT];
  const float* wp[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int ot = ot0 + k * PM_NW;
    wp[k] = wf + ((size_t)(ot < n_ot ? ot : (ot0 < n_ot ? ot0 : 0)) * n_kb) * 512 + lane * 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  typename Epi::Pre pre[NT][RT];
  if (ot0 < n_ot) {
    GsFrag<NT> f0, f1;
    constexpr int NLD = NT * 2;   // loads per one-block chunk
    auto load = [&](GsFrag<NT>& f, int kb0) {
      const int kb = kb0 < n_kb ? kb0 : 0;   // past the end: a harmless re-load of block 0
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const float* q = wp[k] + (size_t)kb * 512;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(f.a[k][0][0]) : "v"(q));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=&v"(f.a[k][0][1]) : "v"(q));
      }
    };
    auto compute = [&](GsFrag<NT>& f, int kb0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD));      // this chunk has landed: the younger one may be out
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(f.a[k][0][p]));
      if (kb0 < n_kb) {
        BQ<RT, 2> b;
        bq_load<RT, 2>(b, lb, ldb, kb0);
        BScaled<RT, F16> bs(b);
#pragma unroll
        for (int q = 0; q < PP::N; ++q)
#pragma unroll
          for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[k][rt] = pm_mfma_bf<F16>(f.a[k][0][PP::W[q]], pm_bsel<F16>(q, b, bs, rt), acc[k][rt]);
      }
    };
    // two one-block chunks (NT x 2 KB each): one feeds the MFMAs while the other is in flight
    load(f0, 0);
    for (int kb0 = 0; kb0 < n_kb; kb0 += 2) {
      load(f1, kb0 + 1);
      compute(f0, kb0);
      load(f0, kb0 + 2);
      compute(f1, kb0 + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
      for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(f0.a[k][0][p]));
    // the epilogues' HBM / L2 operands (bias, dropout / activity bits): requested now, in flight across the barrier
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int ot = ot0 + k * PM_NW < n_ot ? ot0 + k * PM_NW : ot0;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) pre[k][rt] = epi.pre(ot, rt);
    }
  }
  __syncthreads();          // every wave has read its last operand: the buffer may be overwritten
  if (ot0 < n_ot) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (ot0 + k * PM_NW < n_ot) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) epi(ot0 + k * PM_NW, rt, acc[k][rt], pre[k][rt]);
      }
    }
  }
}

// a layer of up to 4 * PM_NW output tiles in place: four tiles per wave, or one where the layer is narrow
template <int RT, bool F16, class Epi>
__device__ __forceinline__ void gemm_layer_inplace_s(const float* __restrict__ wf, int n_ot, int n_kb, const float* buf,
                                                     unsigned ldb, int wid, int lane, Epi& epi) {
  if (n_ot <= PM_NW) gemm_tiles_inplace_s<RT, 1, F16>(wf, n_ot, n_kb, buf, ldb, wid, lane, epi);
  else if (n_ot <= 2 * PM_NW) gemm_tiles_inplace_s<RT, 2, F16>(wf, n_ot, n_kb, buf, ldb, wid, lane, epi);
  else gemm_tiles_inplace_s<RT, 4, F16>(wf, n_ot, n_kb, buf, ldb, wid, lane, epi);
}

// K-split GEMM for narrow outputs (see gemm_ksplit): every wave reduces its slice of K32 blocks for all
// output tiles; partial tiles in the layout gemm_ksplit_combine() reads.
template <int RT, bool F16>
__device__ __forceinline__ void gemm_ksplit_s(const float* __restrict__ wf, int n_ot, int n_kb, const float* buf_in,
                                              unsigned ldb, int wid, int lane, float* part, float* alias, int ld) {
  typedef PmPairs<2> PP;
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  const int per = (n_kb + PM_NW - 1) / PM_NW;
  const int k_lo = wid * per;
  const int k_hi = min(n_kb, k_lo + per);
  f32x4 acc[PM_KS_NT][RT];
#pragma unroll
  for (int k = 0; k < PM_KS_NT; ++k)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = k_lo; kb < k_hi; ++kb) {
    f32x4 a[PM_KS_NT][2];
#pragma unroll
    for (int k = 0; k < PM_KS_NT; ++k)
      if (k < n_ot) {
#pragma unroll
        for (int p = 0; p < 2; ++p) a[k][p] = ldg4(wf + (((size_t)k * n_kb + kb) * 2 + p) * 256 + lane * 4);
      }
    BQ<RT, 2> b;
    bq_load<RT, 2>(b, lb, ldb, kb);
    BScaled<RT, F16> bs(b);
#pragma unroll
    for (int q = 0; q < PP::N; ++q)
#pragma unroll
      for (int k = 0; k < PM_KS_NT; ++k)
        if (k < n_ot) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[k][rt] = pm_mfma_bf<F16>(a[k][PP::W[q]], pm_bsel<F16>(q, b, bs, rt), acc[k][rt]);
        }
  }
#pragma unroll
  for (int k = 0; k < PM_KS_NT; ++k)
    if (k < n_ot) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        *reinterpret_cast<f32x4*>(pm_part_at(part, alias, ld, (((wid * PM_KS_NT + k) * RT + rt) * 64 + lane) * 4)) =
            acc[k][rt];
    }
}

// head / tail layer on the piece planes; fp32 rows (with bias) in lds_out[row][0 .. n_ot*16), leading
// dimension ld floats.  Contains barriers: called by all threads.
template <int RT, bool F16>
__device__ __forceinline__ void gemm_narrow_s(const float* __restrict__ wf, int n_ot, int n_kb, const float* bias,
                                              const float* buf_in, unsigned ldb, float* lds_out, int ld, float* part,
                                              int wid, int lane, int tid) {
  if (n_ot <= PM_KS_NT) {
    gemm_ksplit_s<RT, F16>(wf, n_ot, n_kb, buf_in, ldb, wid, lane, part, lds_out, ld);
    __syncthreads();
    gemm_ksplit_combine<RT>(part, n_ot, bias, lds_out, ld, tid, lds_out);
  } else {
    EpiPlain e{bias, lds_out, ld, lane};
    gemm_tiles_s<RT, F16>(wf, n_ot, n_kb, buf_in, ldb, wid, lane, e);
  }
  __syncthreads();
}
