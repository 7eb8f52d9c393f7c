// Split-bf16 arithmetic for the hidden-width GEMMs of the latency-optimised sweep kernels.
//
// v_mfma_f32_16x16x4_f32 (exact fp32) runs at the fp32 VECTOR rate, 1/16 of the bf16 matrix rate.
// Every fp32 operand is instead written as a sum of NP bf16 pieces (round to nearest, residual,
// round again: v = p0 + p1 + p2 reproduces fp32 to ~2^-25 |v|, v = p0 + p1 to 2^-17 |v|) and the
// product a.b becomes the piece products of combined order < NP on v_mfma_f32_16x16x32_bf16
// (exact products, fp32 accumulation):
//   NP = 3:  p0q0 + (p0q1 + p1q0) + (p0q2 + p1q1 + p2q0)   6 MFMAs per K=32 block, fp32-equivalent
//   NP = 2:  p0q0 + (p0q1 + p1q0)                          3 MFMAs per K=32 block
// against 8 fp32 MFMAs of twice the issue time for the same K=32: 2.7x / 5.3x the matrix rate.
// What each costs in accuracy was measured first (tools/split_precision_study.py,
// profiles/r02_split_precision_study.txt): the FORWARD sweep needs NP = 3 (its rounding moves the
// trajectory the gradient is taken about: NP = 2 gives 1e-5 ... 1.4e-4 on the policy gradient), the
// ADJOINT sweep is linear in the incoming gradient and is fine with NP = 2 (4e-6 ... 7e-6).
//
// Layouts:
//   weights   [tile ot][K32 block kb][piece p][lane][8 bf16]: lane l holds W[ot*16 + (l&15)][kb*32 + 8*(l>>4) + e]
//             -> one global_load_dwordx4 per (tile, block, piece) and lane = one MFMA A operand
//   activations in LDS: piece planes [p][row][LDB] bf16, LDB = 16 (mod 32) elements so that the
//             ds_read_b128 of lane l (row l&15, 8 k-values at 8*(l>>4)) is conflict-free
//   MFMA D mapping as in the fp32 path: lane l ends up with features 4*(l>>4)..+3 of row l&15.
#pragma once
#include "pmbrl_dev.h"

typedef __bf16 pm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pm_u32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 pm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pm_f16x2 __attribute__((ext_vector_type(2)));

// F16 = false: bf16 pieces (fp32's exponent range: safe for gradients of any magnitude);
// F16 = true: fp16 pieces -- 11 significant bits each, so TWO pieces already carry 22 bits (fp32-equivalent
// for this path: tools/split_precision_study.py) at 4 bytes per weight; values beyond +-65504 overflow to
// inf and are reported as a failed rollout, small ones degrade gracefully (the matrix core keeps fp16
// subnormals: tools/ubench/f16_probe.hip).  Used for the forward sweep only.
template <bool F16>
__device__ __forceinline__ f32x4 pm_mfma_bf(f32x4 a, f32x4 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_f16x8, a), __builtin_bit_cast(pm_f16x8, b), c,
                                                  0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pm_bf16x8, a), __builtin_bit_cast(pm_bf16x8, b), c,
                                                   0, 0, 0);
}
// fp16 pieces, the weights' LOW piece: |w - hi| <= 2^-11 |w| is below fp16's normal range (6.1e-5) for every
// |w| < 0.125 -- all of a Xavier-initialised hidden layer -- and a subnormal low piece keeps 2^-24 ABSOLUTE, not eleven
// more bits: measured (tools/c5_precision_study.py, rows "w.lo * 2^11"; profiles/r04_c5_precision_study.txt), the trajectory
// error of the two-piece arithmetic was 2.1x the exact-fp32 one's for that reason alone, and 4x as many ReLU units
// changed sign against fp64 at the 3 x 512 shape.  So the low piece is stored as fp16((w - hi) * 2^11) -- the same
// magnitude as w, eleven good bits whenever hi has them -- and the one piece product that uses it (lo x act.hi)
// accumulates in a chain of its own that enters the result through ONE fma with 2^-11 (exact): no extra instruction
// against the sum of two chains the latency-optimised kernels ended with before (the general family, one
// accumulator per tile, scales that product's B operand instead: BScaled below).  Activations keep their plain low
// piece: its quantisation (|h| < 0.125) is an absolute 3e-8 |w| per term, below fp32's own rounding of the sum.
#define PM_F16_LO_SCALE 2048.f
#define PM_F16_LO_ISCALE (1.f / 2048.f)
// accumulator chain of piece product q of PmPairs<NP>: fp16 pieces -- chain 0 = the product with the weight's low
// piece, chain 1 = the rest; bf16 pieces -- alternating (two chains cover the dependent-MFMA latency)
template <bool F16, int NP>
__device__ __forceinline__ constexpr int pm_chain(int q) {
  return (F16 && NP == 2) ? (q == 0 ? 0 : 1) : (q & 1);
}
template <bool F16, int NP>
__device__ __forceinline__ f32x4 pm_chains_sum(f32x4 c0, f32x4 c1) {
  if constexpr (F16 && NP == 2) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(c0[i], PM_F16_LO_ISCALE, c1[i]);
    return r;
  } else {
    return c0 + c1;
  }
}
// two fp32 -> packed fp16 pair (round to nearest even) and back
__device__ __forceinline__ unsigned pm_pk_f16(float a, float b) {
  const pm_f16x2 r = __builtin_convertvector((pm_f32x2){a, b}, pm_f16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ pm_f32x2 pm_unpk_f16(unsigned p) {
  return __builtin_convertvector(__builtin_bit_cast(pm_f16x2, p), pm_f32x2);
}
// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pm_pk_bf16(float a, float b) {
  const pm_bf16x2 r = __builtin_convertvector((pm_f32x2){a, b}, pm_bf16x2);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float pm_bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float pm_bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 4 consecutive features of one row -> NP pieces of 4 packed bf16 (8 bytes each)
template <int NP, bool F16 = false>
__device__ __forceinline__ void pm_split4(f32x4 h, pm_u32x2 (&pc)[NP]) {
  float r0 = h[0], r1 = h[1], r2 = h[2], r3 = h[3];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if constexpr (F16) {
      const unsigned a = pm_pk_f16(r0, r1), b = pm_pk_f16(r2, r3);
      pc[p] = pm_u32x2{a, b};
      if (p + 1 < NP) {
        const pm_f32x2 fa = pm_unpk_f16(a), fb = pm_unpk_f16(b);
        r0 -= fa[0];
        r1 -= fa[1];
        r2 -= fb[0];
        r3 -= fb[1];
      }
    } else {
      const unsigned a = pm_pk_bf16(r0, r1), b = pm_pk_bf16(r2, r3);
      pc[p] = pm_u32x2{a, b};
      if (p + 1 < NP) {
        r0 -= pm_bf_lo(a);
        r1 -= pm_bf_hi(a);
        r2 -= pm_bf_lo(b);
        r3 -= pm_bf_hi(b);
      }
    }
  }
}
// store them into the piece planes of an activation buffer: rows R, leading dimension ldb (elements)
template <int NP, int R, bool F16 = false>
__device__ __forceinline__ void pm_store_planes(float* buf, unsigned ldb, unsigned lrow, unsigned f0, f32x4 h) {
  pm_u32x2 pc[NP];
  pm_split4<NP, F16>(h, pc);
  unsigned short* pb = reinterpret_cast<unsigned short*>(buf);
#pragma unroll
  for (int p = 0; p < NP; ++p)
    *reinterpret_cast<pm_u32x2*>(pb + ((unsigned)(p * R) + lrow) * ldb + f0) = pc[p];
}

// the same from a per-lane pointer q = plane 0, row (lane & 15), column f0: row tile rt and piece p are then
// CONSTANT offsets (p R + 16 rt) ldb -- at most 54 KB for 64 rows of 240 elements, inside the 16-bit immediate of a
// ds_write.  With the buffer's own offset in the immediate instead (the compiler's choice for the form above) the
// second plane of a 64-row buffer is out of the immediate's reach and every (row tile, piece) gets an address
// register of its own: eight of them, spilled, and reloaded behind a vmcnt(0) in every epilogue.
template <int NP, int R, bool F16 = false>
__device__ __forceinline__ void pm_store_planes_q(unsigned short* q, unsigned ldb, int rt, f32x4 h) {
  pm_u32x2 pc[NP];
  pm_split4<NP, F16>(h, pc);
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<pm_u32x2*>(q + ((unsigned)(p * R) + 16u * (unsigned)rt) * ldb) = pc[p];
}

// piece products kept, smallest contributions first
template <int NP>
struct PmPairs;
template <>
struct PmPairs<3> {
  static constexpr int N = 6;
  static constexpr int W[6] = {2, 1, 0, 1, 0, 0};   // weight piece
  static constexpr int A[6] = {0, 1, 2, 0, 1, 0};   // activation piece
};
template <>
struct PmPairs<2> {
  static constexpr int N = 3;
  static constexpr int W[3] = {1, 0, 0};
  static constexpr int A[3] = {0, 1, 0};
};

// B operands (activation pieces) of one K32 block
template <int RT, int NP>
struct BQ {
  f32x4 v[NP][RT];
};
template <int RT, int NP>
__device__ __forceinline__ void bq_load(BQ<RT, NP>& b, const unsigned short* lane_base, unsigned ldb, int kb) {
  constexpr unsigned R = 16 * RT;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      b.v[p][rt] = *reinterpret_cast<const f32x4*>(lane_base + ((unsigned)(p * R) + rt * 16u) * ldb + kb * 32);
}
// Where a routine keeps ONE accumulator per output tile (the general family: pmbrl_gsplit.h) the 2^-11 of the
// weights' scaled low piece goes onto the B operand of that one product instead of onto a chain of its own:
// act.hi * 2^-11 as packed fp16 (four v_pk_mul_f16 per K32 block and row tile, shared by every output tile of the
// wave).  Exact for |act.hi| >= 0.125; below that the product lands in fp16's subnormals and keeps 2^-25 absolute,
// i.e. <= 1.5e-8 |w| per term -- the size of the activations' own low-piece quantisation, below fp32's rounding of
// the sum.
template <int RT, bool F16>
struct BScaled {
  f32x4 v[F16 ? RT : 1];
  __device__ __forceinline__ explicit BScaled(const BQ<RT, 2>& b) {
    if constexpr (F16) {
      const _Float16 s = (_Float16)PM_F16_LO_ISCALE;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) v[rt] = __builtin_bit_cast(f32x4, __builtin_bit_cast(pm_f16x8, b.v[0][rt]) * s);
    }
  }
};
// B operand of piece product q of PmPairs<2> (q = 0: weight low piece x activation high piece)
template <bool F16, int RT>
__device__ __forceinline__ f32x4 pm_bsel(int q, const BQ<RT, 2>& b, const BScaled<RT, F16>& bs, int rt) {
  if constexpr (F16) {
    if (q == 0) return bs.v[rt];
  }
  return b.v[PmPairs<2>::A[q]][rt];
}
// this lane's base inside a plane buffer: row lane&15, k offset 8*(lane>>4)
__device__ __forceinline__ const unsigned short* pm_plane_lane(const float* buf, unsigned ldb, int lane) {
  return reinterpret_cast<const unsigned short*>(buf) + (unsigned)(lane & 15) * ldb + 8u * ((unsigned)lane >> 4);
}

// MFMAs of one stage of NB K32 blocks (NB * NP weight loads in f.a, block-major / piece-minor) with the
// activation reads software-pipelined one block ahead, across stage and tile boundaries (b0 enters
// holding block kb0 and leaves holding block kb0_next) -- the same structure as frag_compute.
template <int RT, int NB, int NP, bool F16, class FR>
__device__ __forceinline__ void frag_compute_s(const FR& f, int kb0, int kb0_next, const unsigned short* lane_base,
                                               unsigned ldb, f32x4 (&acc)[2][RT], BQ<RT, NP>& b0) {
  typedef PmPairs<NP> PP;
  if constexpr (RT >= 4) {
    // 64-row workgroups: registers are the constraint (four row tiles of B operands and accumulators per
    // wave) -- one set of B operands, read right before its 4 * N MFMAs (the SIMD's other wave covers the
    // LDS latency), and the four row tiles as the independent accumulator chains
    (void)kb0_next;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      bq_load<RT, NP>(b0, lane_base, ldb, kb0 + blk);
#pragma unroll
      for (int q = 0; q < PP::N; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[(F16 && NP == 2) ? pm_chain<F16, NP>(q) : 0][rt] =
              pm_mfma_bf<F16>(f.a[blk * NP + PP::W[q]], b0.v[PP::A[q]][rt], acc[(F16 && NP == 2) ? pm_chain<F16, NP>(q) : 0][rt]);
    }
  } else {
    BQ<RT, NP> b[2];
    b[0] = b0;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int cur = blk & 1, nxt = cur ^ 1;
      bq_load<RT, NP>(b[nxt], lane_base, ldb, blk + 1 < NB ? kb0 + blk + 1 : kb0_next);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < PP::N; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[pm_chain<F16, NP>(q)][rt] =
              pm_mfma_bf<F16>(f.a[blk * NP + PP::W[q]], b[cur].v[PP::A[q]][rt], acc[pm_chain<F16, NP>(q)][rt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    b0 = b[NB & 1];
  }
}

// Narrow head / tail on the piece planes: K32 block `wid` of the single output tile per wave (<= 8
// blocks = hidden width <= 256), weight pieces register-resident for the launch, partial tile to LDS
// in the layout head_value() reads.
template <int NP>
struct HeadWS {
  f32x4 w[NP];
};
template <int NP>
__device__ __forceinline__ void head_load_s(HeadWS<NP>& h, const float* wf, int n_kb32, int wid, int lane) {
  const int kbc = wid < n_kb32 ? wid : n_kb32 - 1;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const f32x4 v = ldg4(wf + ((size_t)(kbc * NP + p) * 64 + lane) * 4);
    h.w[p] = wid < n_kb32 ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int RT, int NP, bool F16>
__device__ __forceinline__ void head_partial_s(const HeadWS<NP>& h, int n_kb32, const float* buf, unsigned ldb,
                                               float* hpart, int wid, int lane) {
  typedef PmPairs<NP> PP;
  const int kbs = wid < n_kb32 ? wid : 0;   // absent block: zero weights, in-range read
  BQ<RT, NP> b;
  bq_load<RT, NP>(b, pm_plane_lane(buf, ldb, lane), ldb, kbs);
  f32x4 acc[2][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[0][rt] = acc[1][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < PP::N; ++q)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      acc[pm_chain<F16, NP>(q)][rt] = pm_mfma_bf<F16>(h.w[PP::W[q]], b.v[PP::A[q]][rt], acc[pm_chain<F16, NP>(q)][rt]);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    *reinterpret_cast<f32x4*>(hpart + ((size_t)(wid * RT + rt) * 64 + lane) * 4) = pm_chains_sum<F16, NP>(acc[0][rt], acc[1][rt]);
}
