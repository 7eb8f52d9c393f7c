// WIDE hidden layers of the general family: 512-wide layers on 64-row workgroups, in place (pm_rollout_fwd / _bwd
// <4, 2, 2>; round 4).
//
// Where the 3 x 512 stress shape (C5) stood (profiles/r03b_inplace_*, r04h_phase_prof_stress32.txt): at 32 rows per
// workgroup a 512 x 512 layer streams 1 MB of weight pieces L2 -> CU for 12.3 k cycles of MFMAs -- 35 k cycles per
// layer-step, 512 workgroups in two rounds.  64 rows per workgroup (one workgroup per CU, ONE round) halve the bytes
// per row; round 3's in-place form (gemm_layer_inplace_s: ONE activation buffer of 64 x 512 two-piece values = 135
// KB, a barrier between the last operand read and the first output write) lost that again in its epilogue: generic
// code, 150 instructions per 16 x 16 output tile, 150 spilled registers whose reloads sit behind vmcnt(0) (a memory
// round trip each) -- 33 k cycles for the 16 tiles of a wave with nothing to overlap them.
//
// This file is that layer with a compile-time shape (32 output tiles; wave w owns tiles 4 w .. 4 w + 3 for all four
// row tiles) and an epilogue written for its instruction count (45 per tile):
//   * dropout / activity bits reach a lane as ONE 8-byte word per row (its four tiles are adjacent) through a buffer
//     load with range checking (absent rows read 0: no branches), requested before the K loop, and become multipliers
//     through a 16-entry LDS table (nibble -> four 0/1 floats: one v_bfe, one ds_read_b128);
//   * the wave's 64 bias values live in ONE register per lane and are fetched by DPP row_share;
//   * h = max(acc + b, 0) * (m / keep): the other forms' bits;
//   * fp16 pieces: v_cvt_pk_f16_f32 for the high pair, v_fma_mix_f32 (reads the fp16 half directly) for the residuals;
//   * activity bits from the bit pattern of h (h >= +0: min(bits, 1)), one 16-bit field per tile, OR-ed over the four
//     16-lane groups once per layer with v_permlane32_swap / v_permlane16_swap (VALU, no LDS), one 8-byte store per lane;
//   * every stash store is buffer_store_dword v, v_lane, s[desc], s_tile offset:r*Rw*4 -- no address arithmetic;
//   * the fp16 range guard is a running integer max, tested once per layer;
//   * ALL of that runs in front of the barrier (results parked as packed pieces in the accumulators' registers);
//     behind it only the 32 ds_write_b64.
// K loop: a ring of three one-block chunks per wave (inline-asm loads through scalar tile bases, explicit waits, a fixed
// shape tools/check_inflight.py can follow; chunks past the end read 16 bytes per lane group), one row tile's operands
// at a time so that nothing the loop uses is spilled.  Nothing derived from the thread index is carried across steps
// (pmbrl_rollout.h launders it per step): the address arithmetic the compiler hoisted out of the step loop was what
// the layers spilled around.  The piece planes are chunk-swizzled (pw_sw): the epilogue's writes two-way instead of
// four-way conflicted.  Narrow layers (heads, first-layer adjoints): pw_narrow.
//
// Measured at C5 (profiles/r04i_notes.txt has every step): a hidden layer-step of 64 rows 86 k -> 42 k cycles (forward,
// with the stash) and 66 k -> 38-42 k (adjoint); forward sweep 19.3 -> 11.9 ms, adjoint 18.2 -> 10.8 ms, MfmaUtil
// 24 % -> 41 % / 44 %.  What binds a layer now, from per-wave stamps and elimination builds: the older wave of each SIMD
// leaves the K loop after 24 k cycles, the younger after 36 k, for 24.6 k cycles of MFMAs per SIMD; with the weight
// loads compiled out 20 k / 32 k, with the MFMAs compiled out 21 k / 27 k -- the instruction stream beside the MFMAs
// (per block and wave 8 ds_read_b128, 16 v_pk_mul_f16 for the scaled operand, 8 loads, waits: ~50 issue slots for 48
// MFMAs) and the fetch each cost about a third on top and overlap imperfectly; and the part clocks down as the duty
// rises (2.3 -> 2.0 GHz between the first and the last build of this file).
// Not kept (measured, no change): B operands read a half ahead across block boundaries; K order rotated per wave or
// per workgroup (no L2-channel hot spot); stash stores as sc0 sc1; the bias staged through LDS.
// Same LDS buffer, same stashes, same activity-bit layout as the other forms of the family: the sweeps of the three
// forms are interchangeable (tests/test_gpu_wide.py).
#pragma once
#include "pmbrl_gsplit.h"

#ifndef PW_STASH_AUX
#define PW_STASH_AUX 2   // nt
#endif
#define PW_RW 64      // rows per workgroup (= the stashes' row pitch)
#define PW_STAMP(i) do { if (e.prof && lane == 0 && wid == 0) e.prof[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define PW_LDB 528u   // elements per row of a piece plane (512 + 16: conflict-free ds_read_b128)

typedef __amdgpu_buffer_rsrc_t pw_rsrc;
__device__ __forceinline__ pw_rsrc pw_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int pw_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Where feature column `col` of row `row` sits in a piece plane of the wide layers: the index of its 16-byte chunk is
// XOR-ed with bits 2..3 of the row.  The plain layout (row pitch 528 elements = 8 banks mod 32) is conflict-free for the
// ds_read_b128 of the B operands, but the epilogue's ds_write_b64 covers 16 ROWS per 16-lane group and rows r, r + 4,
// r + 8, r + 12 meet in one bank pair: a four-way conflict, 4.4 k cycles for the 32 writes of a wave behind every
// layer.  With the XOR the reads stay conflict-free (a chunk moves inside its K32 block only) and the writes are
// two-way (the floor for 8-byte writes at this pitch: tools/ubench/lds_swizzle.py enumerates the lane groups).
__device__ __forceinline__ unsigned pw_sw(unsigned row, unsigned col) {
  return (((col >> 3) ^ ((row >> 2) & 3u)) << 3) | (col & 7u);
}
// this lane's base for the B operands of row tile 0 (row lane & 15, chunk lane >> 4 of a K32 block)
__device__ __forceinline__ const unsigned short* pw_plane_lane(const float* buf, int lane) {
  const unsigned c = (unsigned)lane & 15u, g = (unsigned)lane >> 4;
  return reinterpret_cast<const unsigned short*>(buf) + c * PW_LDB + 8u * (g ^ ((c >> 2) & 3u));
}

// nibble -> four multipliers; 64 floats of LDS, written once per launch
__device__ __forceinline__ void pw_table_init(float* tbl, int tid) {
  if (tid < 64) tbl[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 1.f : 0.f;
}

// acc[k][rt] = sum over K of W[tile 4 wid + k] x X[row tile rt]; weights [tile][K32 block][piece][lane][4 floats].
// A ring of THREE one-block chunks (4 tiles x 2 pieces = 8 KB per wave each): two in flight while one feeds the
// MFMAs -- with one in flight (round 3's in-place loop) a block took a memory round trip (2.1 k cycles against the
// 1.5 k its 96 MFMAs keep a SIMD's two waves busy: 34 k cycles per 512 x 512 layer, profiles/r04i_*).  NR = 4 (round
// 6, the adjoint): a fourth chunk where the registers allow it -- 255, no spill; the forward sweep's layers spill 79
// around the loop with it and lose what the loop gains (profiles/r06_c5_notes.txt): 10.69 -> 10.27 ms per adjoint sweep.
template <bool F16, int NR = 3>
__device__ __forceinline__ void pw_kloop(const float* __restrict__ wf, int n_kb, const float* buf, int wid, int lane,
                                         f32x4 (&acc)[4][4]) {
  typedef PmPairs<2> PP;
  constexpr unsigned ldb = PW_LDB;
  const unsigned short* lb = pw_plane_lane(buf, lane);
  // the wave's four tile bases as scalars: every load is  global_load_dwordx4 v, v_offset, s[base]  -- one 32-bit
  // lane offset per chunk, no 64-bit address arithmetic per load
  const char* tb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long a = (unsigned long long)(wf + ((size_t)(4 * wid + k) * n_kb) * 512);
    tb[k] = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  GsFrag<4> f0, f1, f2, f3;
  constexpr int NLD = 8;   // loads per one-block chunk
  // A chunk past the last block is loaded all the same -- the ring keeps its fixed shape: every wait is "all but the two
  // youngest chunks", which tools/check_inflight.py can follow on every path -- but with lane offset 0: every lane
  // reads the first 16 bytes of its tile, two cache lines per load instead of sixteen, from L1 after the first.
  auto load = [&](GsFrag<4>& f, int kb) {
    // (the lane offset is rebuilt from the lane id here: kept in a register across the loop it was the one value the
    //  allocator spilled INSIDE the loop, and a scratch reload between the ring's loads is waited for with vmcnt(0))
    unsigned ones = ~0u;
    asm volatile("" : "+s"(ones));
    const unsigned lid = __builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    const unsigned vo = kb < n_kb ? (lid << 4) + (unsigned)kb * 2048u : 0u;
    // (s_nop 4: an SGPR a VALU instruction -- v_readfirstlane -- has just written needs 5 wait states before a
    //  vector-memory instruction reads it, and the hazard recogniser does not look inside an asm statement)
    asm volatile("s_nop 4");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(f.a[k][0][0]) : "v"(vo), "s"(tb[k]));
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(f.a[k][0][1]) : "v"(vo), "s"(tb[k]));
    }
  };
  // B operands (PF, the adjoint): TWO sets of {high, low} piece (16 registers), the reads of the NEXT row tile (behind the block's last:
  // of the next block's first) issued in front of a row tile's 12 MFMAs.  Round 6: until then a row tile's operands were
  // read right in front of its MFMAs -- ds_read, s_waitcnt lgkmcnt(0), MFMAs -- and only the SIMD's other wave covered
  // the LDS latency: a wave alone kept the matrix core busy half of the time, and of a SIMD's two waves the younger ran
  // its last third alone (24 k / 36 k cycles for 24.6 k of MFMAs).
  // The reads are inline asm with hand-written waits, like the weight ring: LDS returns in order, every row tile's wait
  // is "all but the two youngest" (lgkmcnt(2)) -- left to the compiler the first wait of every block came out as
  // lgkmcnt(0) (the pending reads of the block before are behind a join of the control flow), one exposed LDS round trip
  // per block.  Nothing else of the loop touches LDS or scalar memory; a compiler-issued wait can only be more
  // conservative for the asm reads it does not know of.
  constexpr bool PF = NR == 4;      // (the adjoint: look-ahead operands and a fourth chunk; the forward sweep: neither)
  f32x4 bq[2][2];
  const unsigned la = (unsigned)(unsigned long long)(const void*)lb;
  auto bread = [&](f32x4 (&b)[2], auto rtc, int kb) {
    constexpr int rt = decltype(rtc)::value;
    const unsigned a0 = la + (unsigned)kb * 64u, a1 = a0 + 64u * ldb * 2u;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(b[0]) : "v"(a0), "n"(rt * 16 * (int)PW_LDB * 2));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(b[1]) : "v"(a1), "n"(rt * 16 * (int)PW_LDB * 2));
  };
  auto row_tile = [&](GsFrag<4>& f, auto rtc, int kb, int kbn) {
    constexpr int rt = decltype(rtc)::value;
    if constexpr (rt < 3) bread(bq[(rt + 1) & 1], std::integral_constant<int, (rt + 1) & 3>{}, kb);
    else bread(bq[0], std::integral_constant<int, 0>{}, kbn);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[rt & 1][0]), "+v"(bq[rt & 1][1]));
    // (fp16 pieces: the product with the weights' scaled low piece takes act.hi * 2^-11 -- pmbrl_split.h, BScaled)
    BQ<1, 2> b;
    b.v[0][0] = bq[rt & 1][0];
    b.v[1][0] = bq[rt & 1][1];
    BScaled<1, F16> bs(b);
#pragma unroll
    for (int q = 0; q < PP::N; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k][rt] = pm_mfma_bf<F16>(f.a[k][0][PP::W[q]], pm_bsel<F16>(q, b, bs, 0), acc[k][rt]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto compute = [&](GsFrag<4>& f, int kb) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NR - 1) * NLD));      // this chunk has landed: the NR - 1 younger ones may be out
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(f.a[k][0][p]));
    if (kb < n_kb) {
      if constexpr (PF) {
        const int kbn = kb + 1 < n_kb ? kb + 1 : kb;      // (past the end: a harmless re-read)
        row_tile(f, std::integral_constant<int, 0>{}, kb, kbn);
        row_tile(f, std::integral_constant<int, 1>{}, kb, kbn);
        row_tile(f, std::integral_constant<int, 2>{}, kb, kbn);
        row_tile(f, std::integral_constant<int, 3>{}, kb, kbn);
      } else {
        // the forward sweep: a row tile's two B operands (8 registers) and the scaled one (4) read right in front of its
        // 12 MFMAs -- the SIMD's other wave covers the LDS latency (with the look-ahead form the forward layers spill 50
        // registers around the loop and gain nothing: profiles/r06_c5_notes.txt)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          BQ<1, 2> b;
#pragma unroll
          for (int p = 0; p < 2; ++p)
            b.v[p][0] = *reinterpret_cast<const f32x4*>(lb + ((unsigned)(p * 64) + rt * 16u) * ldb + kb * 32);
          BScaled<1, F16> bs(b);
#pragma unroll
          for (int q = 0; q < PP::N; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              acc[k][rt] = pm_mfma_bf<F16>(f.a[k][0][PP::W[q]], pm_bsel<F16>(q, b, bs, 0), acc[k][rt]);
        }
      }
    }
  };
  if constexpr (PF) bread(bq[0], std::integral_constant<int, 0>{}, 0);
  load(f0, 0);
  load(f1, 1);
  if constexpr (NR == 4) {
    load(f2, 2);
    for (int kb0 = 0; kb0 < n_kb; kb0 += 4) {
      load(f3, kb0 + 3);
      compute(f0, kb0);
      load(f0, kb0 + 4);
      compute(f1, kb0 + 1);
      load(f1, kb0 + 5);
      compute(f2, kb0 + 2);
      load(f2, kb0 + 6);
      compute(f3, kb0 + 3);
    }
  } else {
    for (int kb0 = 0; kb0 < n_kb; kb0 += 3) {
      load(f2, kb0 + 2);
      compute(f0, kb0);
      load(f0, kb0 + 3);
      compute(f1, kb0 + 1);
      load(f1, kb0 + 4);
      compute(f2, kb0 + 2);
    }
  }
  // the look-ahead chunks (past the end: zeros, no traffic) and the last block's look-ahead operands land before their
  // registers are reused
  if constexpr (PF) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]));
  else asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      asm volatile("" : "+v"(f0.a[k][0][p]));
      asm volatile("" : "+v"(f1.a[k][0][p]));
      asm volatile("" : "+v"(f2.a[k][0][p]));
      if constexpr (NR == 4) asm volatile("" : "+v"(f3.a[k][0][p]));
    }
}

// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads also drains its stash stores: 9 k cycles
// behind a policy layer)
__device__ __forceinline__ void pw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// OR over the four 16-lane groups of a wave (every lane ends with the OR of lanes c, c + 16, c + 32, c + 48)
__device__ __forceinline__ unsigned pw_or_groups(unsigned w) {
  auto a = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  w = a[0] | a[1];
  auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  return b[0] | b[1];
}

// the 16-bit words (dropout bits / activity bits, [row][32 tiles]) of the wave's 16 tiles: lane (g, c) gets row
// 16 rt + c, tiles 4 wid .. 4 wid + 3 -- eight contiguous bytes, one load per row tile; rows past nvalid read as 0
// (the descriptor's range check), no branches, no 64-bit address arithmetic
__device__ __forceinline__ void pw_load_words(pm_u32x2 (&mw)[4], const uint16_t* words, int row0, int nvalid, int wid,
                                              int c) {
  const pw_rsrc rd = pw_make_rsrc(words + (size_t)row0 * 32, (unsigned)nvalid * 64u);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
    mw[rt] = __builtin_bit_cast(pm_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, (rt * 16 + c) * 64, pw_uni(wid * 8), 0));
}
// nibble g of tile k's word
__device__ __forceinline__ unsigned pw_nibble(const pm_u32x2& w, int k, int g) {
  return (w[k >> 1] >> (16 * (k & 1) + 4 * g)) & 15u;
}
// 1 for a non-zero bit pattern (v_min_u32; written as min(u, 1) the compiler makes a compare and a select of it)
__device__ __forceinline__ unsigned pw_nz(unsigned u) {
  unsigned r;
  asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(u));
  return r;
}

struct PwFwd {
  const float* bias;
  const uint16_t* mask;   // [B][32] of this step (or the frozen masks)
  uint16_t* abits;        // [B][32] slice of step t
  float keep;
  float* buf;             // the one activation buffer (piece planes, 64 rows each)
  float* stash;           // feature-major block [512][Rw] of this (step, workgroup) or nullptr
  float* tbl;             // LDS: nibble -> multipliers (64 floats), then the layer's bias (512 floats)
  int row0, nvalid;
  int* ovf;
  long long* prof;        // debug: four cycle stamps of this layer (workgroup 0, thread 0) or nullptr
  int pre;                // the stash in the pre-split form (pw_stash_pre) instead of fp32 rows
};

// lane (g, c) -> the value lane (g, n) holds, n a constant: DPP row_share (a 16-lane row is one lane group g)
template <int N>
__device__ __forceinline__ float pw_row_share(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x150 + N, 0xf, 0xf, false));
}

// hidden layer, forward, in place:  h = relu(W x + b) * mask / keep   (models/modules.py:46-61,120-160)
// Contains the barrier between the last read of the input and the first write of the output; the caller
// synchronises behind it.  Everything but the LDS writes happens IN FRONT of that barrier: a wave that leaves the K
// loop early (of a SIMD's two waves one does) runs its epilogue while its partner still feeds the matrix core, and the
// results wait for the barrier as packed pieces in the accumulators' registers.
// STASH: 0 none; 1 fp32, feature-major [512][Rw] (what the fp32-stash kernels of pmbrl_dw.h read); 2 PRE-SPLIT for
// pm_dw_wide_pre_kernel / pm_dw_narrow_pre_kernel: [piece][tile][row (64)][16 features] bf16 -- a lane's four features of a
// row are one 8-byte store per piece (pw_stash_pre).  One form per kernel instance (pm_rollout_fwd / _bwd <4, 2, 2 | 3>).
__device__ __forceinline__ void pw_stash_pre(const pw_rsrc& srd, int vo, int so, const pm_u32x2& hi, const pm_u32x2& lo) {
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(srd, 0, 0, 0)) b64_t;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64_t, hi), srd, vo, so, PW_STASH_AUX);
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64_t, lo), srd, vo, so + 65536, PW_STASH_AUX);
}
template <int STASH>
__device__ __forceinline__ void pw_hidden_fwd(const float* __restrict__ wf, int n_kb, const PwFwd& e, int wid, int lane) {
  constexpr int R = 64;
  const int g = lane >> 4, c = lane & 15;
  // the dropout words of the wave's 16 tiles: in flight behind the K loop
  pm_u32x2 mw[4];
  pw_load_words(mw, e.mask, e.row0, e.nvalid, wid, c);
  // the wave's 64 bias values, ONE register a lane (lane (g, c) holds value 4 g + (c & 3) of tile c >> 2; the lanes
  // of a group fetch each other's by DPP row_share): 16 registers a lane the K loop does not have
  const float breg = e.bias[(4 * wid + (c >> 2)) * 16 + 4 * g + (c & 3)];
  f32x4 acc[4][4];
  pw_kloop<true>(wf, n_kb, e.buf, wid, lane, acc);
  PW_STAMP(0);
  const float ik = 1.f / e.keep;
  const f32x4 ik4 = {ik, ik, ik, ik};
  f32x4 bk[4];
  bk[0] = f32x4{pw_row_share<0>(breg), pw_row_share<1>(breg), pw_row_share<2>(breg), pw_row_share<3>(breg)};
  bk[1] = f32x4{pw_row_share<4>(breg), pw_row_share<5>(breg), pw_row_share<6>(breg), pw_row_share<7>(breg)};
  bk[2] = f32x4{pw_row_share<8>(breg), pw_row_share<9>(breg), pw_row_share<10>(breg), pw_row_share<11>(breg)};
  bk[3] = f32x4{pw_row_share<12>(breg), pw_row_share<13>(breg), pw_row_share<14>(breg), pw_row_share<15>(breg)};
  // multipliers of a tile's four row tiles: looked up one tile ahead of their use
  auto lookup = [&](f32x4 (&m)[4], int k) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) m[rt] = *reinterpret_cast<const f32x4*>(e.tbl + 4 * pw_nibble(mw[rt], k, g));
  };
  f32x4 m[2][4];
  lookup(m[0], 0);
  const pw_rsrc srd = pw_make_rsrc(e.stash, 0xffffffffu);
  const int vo_st = (4 * g * PW_RW + c) * 4;
  unsigned aw[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};   // [row tile][tile pair]: 16-bit field per tile
  unsigned hmax = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < 3) lookup(m[(k + 1) & 1], k + 1);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      // (the two-buffer form's arithmetic, bit for bit: (acc + b) * (1 / keep) for an active unit)
      const f32x4 v = acc[k][rt] + bk[k];
      f32x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = fmaxf(v[r], 0.f);
      h *= m[k & 1][rt] * ik4;
      const unsigned u0 = __float_as_uint(h[0]), u1 = __float_as_uint(h[1]), u2 = __float_as_uint(h[2]),
                     u3 = __float_as_uint(h[3]);
      hmax = max(max(hmax, max(u0, u1)), max(u2, u3));
      const unsigned nb = pw_nz(u0) | (pw_nz(u1) << 1) | (pw_nz(u2) << 2) | (pw_nz(u3) << 3);
      aw[rt][k >> 1] |= nb << (16 * (k & 1));
      if constexpr (STASH != 0) {
        if constexpr (STASH == 2) {
          // the dW GEMM's operands: two bf16 pieces of h -- what pm_dw_wide_kernel made of the fp32 value, bit for bit
          pm_u32x2 pc[2];
          pm_split4<2, false>(h, pc);
          pw_stash_pre(srd, (c * 32 + g * 8), pw_uni((4 * wid + k) * 2048 + rt * 512), pc[0], pc[1]);
        } else {
          const int so = pw_uni(((4 * wid + k) * 16 * PW_RW + rt * 16) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[r]), srd, vo_st + r * PW_RW * 4, so, PW_STASH_AUX);
        }
      }
      // two fp16 pieces: residuals h - hi in fp32 (v_fma_mix_f32 reads the fp16 half directly), then the same RNE pair
      // conversion as the other forms (the low piece of |h| < 0.125 is an fp16 subnormal: kept)
      const unsigned ha = pm_pk_f16(h[0], h[1]), hb = pm_pk_f16(h[2], h[3]);
      float l0, l1, l2, l3;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ha), "v"(h[0]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ha), "v"(h[1]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(hb), "v"(h[2]));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(hb), "v"(h[3]));
      // ... parked where the accumulator was
      acc[k][rt] = f32x4{__uint_as_float(ha), __uint_as_float(hb), __uint_as_float(pm_pk_f16(l0, l1)),
                         __uint_as_float(pm_pk_f16(l2, l3))};
    }
  }
  if (hmax > 0x477fe000u) *e.ovf = 1;   // an activation beyond fp16's range (65504; h >= +0, inf and NaN compare above)
  // activity bits: every lane's nibble goes to bit 4 g of its tile's 16-bit field; OR over the four lane groups;
  // lane group g stores row tile g: the row's four words are eight contiguous bytes
  const pw_rsrc ard = pw_make_rsrc(e.abits + (size_t)e.row0 * 32, (unsigned)e.nvalid * 64u);
  pm_u32x2 out = {0u, 0u};
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const unsigned w0 = pw_or_groups(aw[rt][0] << (4 * g)), w1 = pw_or_groups(aw[rt][1] << (4 * g));
    if (g == rt) out = pm_u32x2{w0, w1};
  }
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(ard, 0, 0, 0)), out), ard,
                                        (g * 16 + c) * 64, pw_uni(wid * 8), 0);
  PW_STAMP(1);
  pw_lds_barrier();          // every wave has read its last operand: the buffer may be overwritten
  PW_STAMP(2);
  unsigned short* pb = reinterpret_cast<unsigned short*>(e.buf) + (unsigned)c * PW_LDB;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned short* q0 = pb + pw_sw((unsigned)c, (unsigned)(wid * 64 + k * 16 + 4 * g));   // (row tile: same bits 2..3)
    unsigned short* q1 = q0 + (unsigned)R * PW_LDB;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const unsigned eo = (unsigned)(rt * 16) * PW_LDB;
      *reinterpret_cast<pm_u32x2*>(q0 + eo) = pm_u32x2{__float_as_uint(acc[k][rt][0]), __float_as_uint(acc[k][rt][1])};
      *reinterpret_cast<pm_u32x2*>(q1 + eo) = pm_u32x2{__float_as_uint(acc[k][rt][2]), __float_as_uint(acc[k][rt][3])};
    }
  }
}

struct PwBwd {
  const uint16_t* abits;  // [B][32] slice of step t
  float keep;
  float* buf;
  float* stash;           // gT block [512][Rw] or nullptr
  const float* tbl;
  int row0, nvalid;
  long long* prof;
  int pre;                // the stash in the pre-split form (pw_stash_pre)
};

// hidden layer, adjoint, in place:  g_pre = active ? (W^T g) / keep : 0   (bf16 pieces; as the forward layer)
template <int STASH>
__device__ __forceinline__ void pw_hidden_bwd(const float* __restrict__ wb, int n_kb, const PwBwd& e, int wid, int lane) {
  constexpr int R = 64;
  const int g = lane >> 4, c = lane & 15;
  pm_u32x2 mw[4];
  pw_load_words(mw, e.abits, e.row0, e.nvalid, wid, c);
  f32x4 acc[4][4];
  pw_kloop<false, 4>(wb, n_kb, e.buf, wid, lane, acc);      // (the adjoint has the registers for a fourth chunk: 255, no spill)
  PW_STAMP(0);
  const float ik = 1.f / e.keep;
  auto lookup = [&](f32x4 (&m)[4], int k) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) m[rt] = *reinterpret_cast<const f32x4*>(e.tbl + 4 * pw_nibble(mw[rt], k, g));
  };
  f32x4 m[2][4];
  lookup(m[0], 0);
  const pw_rsrc srd = pw_make_rsrc(e.stash, 0xffffffffu);
  const int vo_st = (4 * g * PW_RW + c) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < 3) lookup(m[(k + 1) & 1], k + 1);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const f32x4 h = acc[k][rt] * ik * m[k & 1][rt];
      pm_u32x2 pc[2];
      pm_split4<2, false>(h, pc);
      if constexpr (STASH != 0) {
        // (pre-split stash: the pieces the next layer reads from LDS are the dW GEMM's operands)
        if constexpr (STASH == 2) {
          pw_stash_pre(srd, (c * 32 + g * 8), pw_uni((4 * wid + k) * 2048 + rt * 512), pc[0], pc[1]);
        } else {
          const int so = pw_uni(((4 * wid + k) * 16 * PW_RW + rt * 16) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[r]), srd, vo_st + r * PW_RW * 4, so, PW_STASH_AUX);
        }
      }
      acc[k][rt] = f32x4{__uint_as_float(pc[0][0]), __uint_as_float(pc[0][1]), __uint_as_float(pc[1][0]),
                         __uint_as_float(pc[1][1])};
    }
  }
  PW_STAMP(1);
  pw_lds_barrier();
  PW_STAMP(2);
  unsigned short* pb = reinterpret_cast<unsigned short*>(e.buf) + (unsigned)c * PW_LDB;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned short* q0 = pb + pw_sw((unsigned)c, (unsigned)(wid * 64 + k * 16 + 4 * g));   // (row tile: same bits 2..3)
    unsigned short* q1 = q0 + (unsigned)R * PW_LDB;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const unsigned eo = (unsigned)(rt * 16) * PW_LDB;
      *reinterpret_cast<pm_u32x2*>(q0 + eo) = pm_u32x2{__float_as_uint(acc[k][rt][0]), __float_as_uint(acc[k][rt][1])};
      *reinterpret_cast<pm_u32x2*>(q1 + eo) = pm_u32x2{__float_as_uint(acc[k][rt][2]), __float_as_uint(acc[k][rt][3])};
    }
  }
}

// Narrow layers behind a 512-wide one (the heads: 2U / 2D outputs; the first layers' adjoints: D / D + U outputs): at
// most four output tiles, K = 512.  The generic in-place routine gives one TILE to a wave -- one wave busy for the
// policy head, four for the dynamics head -- and walks its sixteen K blocks two loads at a time: 11-13 k cycles of
// exposed round trips for 0.4 k of MFMAs per wave.  Here a wave takes one or two ROW TILES of a tile (all eight waves
// busy at four tiles), requests the tile's 32 fragments at once (128 registers: this is a layer without a ring) and
// consumes them in order as they land -- the same K order and piece order per accumulator as before: bit-identical.
// fp32 rows (with bias) to out[row][0 .. 16 n_ot), leading dimension ld floats, which may alias the input planes:
// contains the barrier between the last operand read and the first write; the caller synchronises behind it.
template <bool F16>
__device__ __forceinline__ void pw_narrow(const float* __restrict__ wf, int n_ot, const float* buf, const float* bias,
                                          float* out, int ld, int wid, int lane) {
  typedef PmPairs<2> PP;
  constexpr int NKB = 16;
  constexpr unsigned ldb = PW_LDB;
  const int g = lane >> 4, c = lane & 15;
  const int n_items = n_ot * 4;
  const int per = n_items > PM_NW ? 2 : 1;
  const int i0 = wid * per;
  const bool work = i0 < n_items;
  const int k = work ? i0 >> 2 : 0, rt0 = i0 & 3;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (work) {
    f32x4 a[NKB][2];
    const float* wp = wf + (size_t)k * NKB * 512 + lane * 4;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[kb][p] = ldg4(wp + (kb * 2 + p) * 256);
    if (bias) b4 = ldg4(bias + k * 16 + 4 * g);
    const unsigned short* lb = pw_plane_lane(buf, lane);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < per) {
          BQ<1, 2> b;
#pragma unroll
          for (int p = 0; p < 2; ++p)
            b.v[p][0] = *reinterpret_cast<const f32x4*>(lb + ((unsigned)(p * 64) + (unsigned)(rt0 + j) * 16u) * ldb + kb * 32);
          BScaled<1, F16> bs(b);
#pragma unroll
          for (int q = 0; q < PP::N; ++q) acc[j] = pm_mfma_bf<F16>(a[kb][PP::W[q]], pm_bsel<F16>(q, b, bs, 0), acc[j]);
        }
      }
    }
  }
  __syncthreads();          // every wave has read its last operand: the buffer may be overwritten
  if (work) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j < per) *reinterpret_cast<f32x4*>(out + ((rt0 + j) * 16 + c) * ld + k * 16 + 4 * g) = acc[j] + b4;
  }
}
