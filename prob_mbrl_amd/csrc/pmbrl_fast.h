// Latency-optimised rollout kernels for the common small-state shapes
// (first-layer K <= 16, heads <= 16 outputs, <= 16 tiles per hidden layer... see
// pm_fast_ok()).  Same math and same stash layout as pmbrl_rollout.h; what
// changes is WHERE the per-step latencies go:
//   * every per-row / per-feature constant (biases, dropout bit rows, frozen
//     noise z, normalisation vectors, reward constants) is staged in LDS once per
//     launch instead of being re-read from HBM/L2 in every epilogue;
//   * the first layer of each net (K = one 16-block) and the adjoint's first GEMM
//     (K = head width) keep their weight fragments in REGISTERS for the launch;
//   * the narrow head (2U / 2D outputs) and the adjoint's narrow tail (D / D+U
//     outputs) are not GEMMs at all: they are folded into the epilogue of the
//     neighbouring hidden layer as lane-local dot products against an LDS-resident
//     matrix, reduced across the 4 lane groups by DPP shuffles and across the 4
//     waves in the next elementwise phase (2 barriers and 2 L2 round trips fewer);
//   * hidden->hidden layers stream their fragment-packed weights from L2 through a
//     double-buffered register stage that always holds the NEXT item (next tile
//     group, next layer, next net, next step) while the current one feeds the
//     MFMAs, so the weight latency is paid once per launch, not once per layer.
#pragma once
#include <type_traits>
#include "pmbrl_dev.h"
#include "pmbrl_mm.h"
#include "pmbrl_rollout.h"
#include "pmbrl_split.h"
#include "pmbrl_mm_w.h"
#include "pmbrl_xch.h"

// Streamed layers: ONE output tile per wave in flight, two accumulator chains per row tile
// (even / odd k-blocks) to cover the 40-cycle dependent-MFMA latency, and a register stage
// of CKB k-blocks per buffer (64 VGPRs) -- the arch-VGPR file is 256 per lane, so the
// stage depth, not the tile count, is where the prefetch distance comes from.
// CKB = k-blocks (of 16) per streamed chunk, a kernel template parameter chosen by the plan
// (CKB*RT*4 = 28..32 MFMAs of lookahead).  The fragment-packed weights of the streamed
// layers are zero-padded to a multiple of CKB k-blocks and the LDS activation buffers carry
// matching zero columns, so every chunk is full: the inner loops are straight-line code
// (a branchy tail defeats the scheduler and, worse, SROA: the stage then lives in scratch).
// The fast kernels run EIGHT waves per workgroup (two per SIMD): the two co-resident waves
// of a SIMD share its MFMA pipe, so one wave's LDS / weight-load waits and epilogue overlap
// the other's MFMAs.  Tile ot of a layer belongs to wave ot % 8.
#define PF_NW 8
#define PF_NT (PF_NW * 64)
#define PM_L0T 2           // max resident first-layer tiles per wave (<= 16 tiles per layer)
#define PM_HJ 16           // max fused head / tail width

// Tile balance: tile ot runs on wave ot mod 8, i.e. SIMD ot mod 4.  A layer with 4m+1 output
// tiles (13 for a 200-wide layer) would put m+1 tiles on SIMD 0 and m on the others: the phase
// then lasts as long as SIMD 0's extra tile (+25 % at m = 3).  Instead the LAST tile of such a
// layer is K-split over all 8 waves (<= 2 k-blocks each, weights LDS-resident for the launch):
// every wave adds its partial tile to LDS at the START of the phase and bumps a counter; the
// last wave, which owns at most one regular tile, waits for the 8 partials after its tile,
// sums them in fixed order and runs the tile's normal epilogue before the phase barrier.
// (not with 64-row workgroups: LDS is the constraint there)
// (split-bf16 precision: the tiles are short enough that the imbalance costs less than the K-split's
// partial / gather round trip -- every tile is a regular tile there)
__host__ __device__ constexpr bool pm_fast_ksplit(int n_ot, int RT, int prec = 0) {
  return !prec && RT < 4 && n_ot >= 5 && (n_ot % 4) == 1;
}
// LDS floats for the K-split tail weights of one sweep direction (bwd: transposed layers)
__host__ __device__ inline size_t pm_fast_tail_floats(const int* pnt, int pnl, const int* dnt, int dnl,
                                                      bool bwd, int RT, int prec = 0) {
  size_t n = 0;
  if (prec) return 0;
  for (int l = 1; l <= pnl - 2; ++l) {
    const int n_ot = bwd ? pnt[l] : pnt[l + 1], n_kb = bwd ? pnt[l + 1] : pnt[l];
    if (pm_fast_ksplit(n_ot, RT)) n += (size_t)n_kb * 256;
  }
  for (int l = 1; l <= dnl - 2; ++l) {
    const int n_ot = bwd ? dnt[l] : dnt[l + 1], n_kb = bwd ? dnt[l + 1] : dnt[l];
    if (pm_fast_ksplit(n_ot, RT)) n += (size_t)n_kb * 256;
  }
  return n;
}

__host__ __device__ inline bool pm_fast_net_ok(const int* dim, const int* nt, int nl) {
  if (nl < 2) return false;
  if (nt[0] != 1) return false;                       // first-layer K fits one 16-block
  if (dim[nl] > PM_HJ) return false;                  // fused head width
  for (int l = 1; l < nl; ++l)
    if (nt[l] > PF_NW * PM_L0T) return false;         // <= 16 tiles per hidden layer
  return true;
}

template <int CKB>
struct FragS {
  f32x4 a[CKB];
};

// Position of the weight stream, one chunk ahead of the MFMAs.  Everything the hot loop
// needs (load pointer, tile / chunk counters, the current layer's shape) lives in SGPRs and
// is advanced incrementally; the kernel-argument table is consulted only when the stream
// moves to another layer (a scalar-memory round trip per chunk was costing more than the
// MFMAs of the chunk).
// Per-layer descriptors live in the LANES of one VGPR per table and are read back with
// v_readlane (uniform lane index): a runtime layer index costs a few VALU issue slots instead
// of a scalar-memory round trip whose wait (lgkmcnt is shared with LDS) lands on the critical
// path of every phase and of every layer switch of the weight stream.
__device__ __forceinline__ int pm_rl(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ float pm_rlf(int v, int i) { return __int_as_float(__builtin_amdgcn_readlane(v, i)); }
// pointers rebuilt from integers carry no address space: stores through them would be FLAT
// (FLAT completes out of order with LDS, which degrades every LDS wait to lgkmcnt(0))
#ifndef PM_GLOBAL
#define PM_GLOBAL __attribute__((address_space(1)))
#endif
template <class T>
__device__ __forceinline__ T* pm_rlp(int v, int i) {
  const unsigned long long lo = (unsigned)__builtin_amdgcn_readlane(v, i);
  const unsigned long long hi = (unsigned)__builtin_amdgcn_readlane(v, i + 1);
  return reinterpret_cast<T*>(lo | (hi << 32));
}
__device__ __forceinline__ int pm_lo32(const void* p) { return (int)(unsigned)(reinterpret_cast<unsigned long long>(p)); }
__device__ __forceinline__ int pm_hi32(const void* p) { return (int)(unsigned)(reinterpret_cast<unsigned long long>(p) >> 32); }

// weight-stream table: lane 4*l + {0: n_ot, 1: n_kb, 2/3: wf lo/hi} of streamed layer l (<= 16)
// Device-wide barrier of a launch whose workgroups are all resident (the host checks that).  The
// rows the workgroups exchange are written with device-scope stores (pm_st_dev: write-through to
// the coherence point) and read with device-scope loads (pm_ldc<true>), so the barrier needs no
// cache maintenance: a release / acquire pair would write back and invalidate a whole L2 per
// workgroup and step (measured: 15+ us per barrier with 157 workgroups, against ~3 us).  A
// wait that does not end (it cannot, unless the residency assumption was broken) gives up after
// ~1 s and reports through the status word instead of hanging the device.
__device__ __forceinline__ void pm_st_dev(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool pm_grid_barrier(unsigned* flags, unsigned k) {
  // barrier k (1, 2, ...) of this launch: workgroup w publishes flags[w] = k once all its threads'
  // stores have completed, wave 0 polls every workgroup's flag.  Plain device-scope stores and loads:
  // no read-modify-write whose return the arriving workgroup would have to wait for, no single
  // address that 150+ workgroups serialise on.
  bool ok = true;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)      // (max, not store: a flag poisoned by a timed-out workgroup stays poisoned)
    __hip_atomic_fetch_max(flags + blockIdx.x, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    const int n = (int)gridDim.x;
    long long spins = 0;
    for (;;) {
      bool all = true;
      for (int w = (int)threadIdx.x; w < n; w += 64)
        all = all && __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= k;
      if (__all(all)) break;
      if (++spins > (1ll << 19)) {
        // give up ONCE for the whole launch: every flag goes to its maximum, so this and all later
        // barriers of all workgroups fall through (the launch finishes with the step marked failed)
        ok = false;
        for (int w = (int)threadIdx.x; w < n; w += 64)
          __hip_atomic_fetch_max(flags + w, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    // a poisoned flag also means failure for the workgroups that did not time out themselves
    if (__hip_atomic_load(flags + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu) ok = false;
  }
  __syncthreads();
  return ok;
}

// The same between the `parts` consecutive workgroups that share a moment-matching group (flags[first .. first
// + parts)): a handful of flags to poll instead of every workgroup's -- the device-wide barrier costs ~11 us at
// 250 workgroups, this one a memory round trip.  Needs the group's workgroups resident together (the host
// checks that ALL are, as for the device-wide barrier); a wait that does not end poisons the group's flags.
__device__ __forceinline__ bool pm_group_sync(unsigned* flags, int self, int first, int parts, unsigned k) {
  bool ok = true;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_max(flags + self, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    const int w = first + ((int)threadIdx.x < parts ? (int)threadIdx.x : 0);
    long long spins = 0;
    for (;;) {
      const bool here = __hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= k;
      if (__all(here)) break;
      if (++spins > (1ll << 19)) {     // ~1 s: gives up once for the group, like pm_grid_barrier
        ok = false;
        __hip_atomic_fetch_max(flags + w, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if (__hip_atomic_load(flags + self, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu) ok = false;
  }
  __syncthreads();
  return ok;
}

struct SdV {
  int n, v;
  int k;   // lane 4*l + {0: ks, 1: tw_off, 2: n_kb_real} of streamed layer l
};
template <class SC>
__device__ __forceinline__ SdV sdv_make(const StreamDesc& sd, int lane) {
  SdV r;
  r.n = SC::NS ? SC::NS : sd.n;
  r.v = 0;
  r.k = 0;
  const int l = lane >> 2, f = lane & 3;
  // lane >> 2 < 2 * PM_MAXL: all loads in bounds, issued together, selected afterwards
  const int n_ot = sd.n_ot[l], n_kb = sd.n_kb[l], ks = sd.ks[l], tw = sd.tw_off[l], kr = sd.n_kb_real[l];
  const uint64_t wf = (uint64_t)sd.wf[l];
  const int v = f == 0 ? n_ot : f == 1 ? n_kb : f == 2 ? (int)(uint32_t)wf : (int)(uint32_t)(wf >> 32);
  const int k = f == 0 ? ks : f == 1 ? tw : f == 2 ? kr : 0;
  r.v = l < sd.n ? v : 0;
  r.k = l < sd.n ? k : 0;
  return r;
}

struct Cursor {
  int li, ot, c, live, all_live;
  int n_ot, n_kb;            // of layer li (n_kb padded to the chunk size)
  const float* wp;           // next chunk to load (lane offset not included)
  // descriptor of the layer the stream enters after li, fetched (scalar loads) when li was
  // entered: by the time it is needed the loads have long landed -- a layer switch costs no
  // scalar-memory round trip on the critical path
  int nli, nn_ot, nn_kb;
  const float* nwf;
};

// Compile-time shape of the weight stream (shape-specialised kernels: every streamed layer has
// the same tile / k-block counts); zeros = read the lane table.
template <int NOT_, int NKB_, int NS_, int NOT0_ = NOT_>
struct PfStream {
  static constexpr int NOT = NOT_, NKB = NKB_, NS = NS_;   // streamed tiles, padded k-blocks, layers
  static constexpr int NOT0 = NOT0_;                       // streamed tiles of layer 0 (fewer when its first tiles
                                                           // are register-resident, resident_tile_s)
};
typedef PfStream<0, 0, 0> PfStreamAny;
#define SD_N(SC, sd) (SC::NS ? SC::NS : (sd).n)
#define SD_NOT(SC, sd, l) (SC::NOT ? ((l) == 0 ? SC::NOT0 : SC::NOT) : pm_rl((sd).v, 4 * (l)))
#define SD_NKB(SC, sd, l) (SC::NKB ? SC::NKB : pm_rl((sd).v, 4 * (l) + 1))
#define SD_WF(sd, l) pm_rlp<const float>((sd).v, 4 * (l) + 2)
template <class SC>
__device__ __forceinline__ void cur_fetch_next(const SdV& sd, Cursor& q, int wid) {
  int l = q.li;
  if (q.all_live) {
    l = (l + 1 >= SD_N(SC, sd)) ? 0 : l + 1;
  } else {
    // next layer (cyclically) in which this wave owns a tile
    for (int k = 0; k < 2 * PM_MAXL; ++k) {
      l = (l + 1 >= SD_N(SC, sd)) ? 0 : l + 1;
      if (wid < SD_NOT(SC, sd, l)) break;
    }
  }
  q.nli = l;
  q.nn_ot = SD_NOT(SC, sd, l);
  q.nn_kb = SD_NKB(SC, sd, l);
  q.nwf = SD_WF(sd, l);
}
template <class SC>
__device__ __forceinline__ void cur_switch(const SdV& sd, Cursor& q, int wid) {
  q.li = q.nli;
  q.n_ot = q.nn_ot;
  q.n_kb = q.nn_kb;
  q.ot = wid;
  q.c = 0;
  q.wp = q.nwf + (size_t)wid * (SC::NKB ? SC::NKB : q.n_kb) * 256;
  cur_fetch_next<SC>(sd, q, wid);
}
template <class SC>
__device__ __forceinline__ void cur_init(const SdV& sd, Cursor& q, int wid) {
  q.li = 0; q.ot = 0; q.c = 0; q.live = 0; q.all_live = 1; q.n_ot = 0; q.n_kb = 0; q.wp = nullptr;
  q.nli = 0; q.nn_ot = 0; q.nn_kb = 0; q.nwf = nullptr;
  int first = -1;
  for (int l = SD_N(SC, sd) - 1; l >= 0; --l) {
    if (wid < SD_NOT(SC, sd, l)) { q.live = 1; first = l; }
    else q.all_live = 0;
  }
  if (q.live) {
    // enter `first` through the same path as every later switch
    q.nli = first;
    q.nn_ot = SD_NOT(SC, sd, first);
    q.nn_kb = SD_NKB(SC, sd, first);
    q.nwf = SD_WF(sd, first);
    cur_switch<SC>(sd, q, wid);
  }
}
template <int CKB, class SC>
__device__ __forceinline__ void cur_advance(const SdV& sd, Cursor& q, int wid) {
  const int n_kb = SC::NKB ? SC::NKB : q.n_kb, n_ot = SC::NOT ? (q.li == 0 ? SC::NOT0 : SC::NOT) : q.n_ot;
  q.c += CKB;
  q.wp += (size_t)CKB * 256;
  if (q.c >= n_kb) {
    q.c = 0;
    q.ot += PF_NW;
    q.wp += (size_t)(PF_NW - 1) * n_kb * 256;
    if (q.ot >= n_ot) cur_switch<SC>(sd, q, wid);
  }
}

// The weight stream is issued with INLINE-ASM loads and waited for with explicit s_waitcnt:
// the compiler's own waitcnt insertion loses track of which stage a load belongs to across
// this loop nest (it fell back to vmcnt(0/1) right after issuing the next stage, i.e. a full
// L2 round trip per chunk, depending on unrelated code changes).  Rules that keep this safe:
//   * a stage register is "in flight" from pm_ldw to the frag_wait that names it; nothing may
//     read, copy or spill it in between (tools/check_inflight.py verifies the generated ISA);
//   * frag_wait<CKB> leaves exactly the CKB loads of the OTHER stage outstanding: it is placed
//     right after those are issued, before any store of an epilogue, so vmcnt(CKB) is exact;
//   * loads the compiler does not know about only make ITS vmcnt waits longer, never shorter.
template <int OFF>
__device__ __forceinline__ void pm_ldw(f32x4& d, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(sbase), "n"(OFF));
}
template <int CKB>
__device__ __forceinline__ void frag_load(FragS<CKB>& f, const Cursor& q, unsigned vo0, unsigned vo1) {
  const float* wp = q.wp;
  static_assert(CKB >= 1 && CKB <= 12, "stage depth");
  const unsigned vo2 = vo1 + 4096u;
  if constexpr (CKB > 8) pm_ldw<0>(f.a[8], vo2, wp);
  if constexpr (CKB > 9) pm_ldw<1024>(f.a[9], vo2, wp);
  if constexpr (CKB > 10) pm_ldw<2048>(f.a[10], vo2, wp);
  if constexpr (CKB > 11) pm_ldw<3072>(f.a[11], vo2, wp);
  if constexpr (CKB > 0) pm_ldw<0>(f.a[0], vo0, wp);
  if constexpr (CKB > 1) pm_ldw<1024>(f.a[1], vo0, wp);
  if constexpr (CKB > 2) pm_ldw<2048>(f.a[2], vo0, wp);
  if constexpr (CKB > 3) pm_ldw<3072>(f.a[3], vo0, wp);
  if constexpr (CKB > 4) pm_ldw<0>(f.a[4], vo1, wp);
  if constexpr (CKB > 5) pm_ldw<1024>(f.a[5], vo1, wp);
  if constexpr (CKB > 6) pm_ldw<2048>(f.a[6], vo1, wp);
  if constexpr (CKB > 7) pm_ldw<3072>(f.a[7], vo1, wp);
}
// everything older than the NOTHER most recent VMEM operations (the loads of the other stage,
// just issued) has landed; `f` is usable
template <int CKB, int NOTHER>
__device__ __forceinline__ void frag_wait(FragS<CKB>& f) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOTHER));
#pragma unroll
  for (int cc = 0; cc < CKB; ++cc) asm volatile("" : "+v"(f.a[cc]));
}

// B operands (LDS activations) of one k-block pair
template <int RT>
struct BPair {
  f32x4 v[2][RT];
};
template <int RT, int CKB>
__device__ __forceinline__ void bpair_load(BPair<RT>& b, const float* bp, int ld, int pair) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cc = 2 * pair + h;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      b.v[h][rt] = *reinterpret_cast<const f32x4*>(bp + rt * 16 * ld + (cc < CKB ? cc : 0) * 16);
  }
}

template <int RT, int CKB>
__device__ __forceinline__ void frag_compute(const FragS<CKB>& f, int kb0, int kb0_next, const float* lds_in,
                                             int ld, int lane, f32x4 (&acc)[2][RT], BPair<RT>& b0) {
  // Explicit one-pair-ahead software pipeline of the LDS (B operand) reads: while the 8*RT
  // MFMAs of k-block pair p issue, the reads of pair p+1 are already in flight -- ACROSS chunk
  // and tile boundaries too: b0 enters holding pair 0 of this chunk and leaves holding pair 0
  // of chunk c_next, so the LDS latency is exposed once per layer, not once per chunk.  The
  // sched_barriers pin that order (left alone, the scheduler sinks each read to just before
  // its first use and every pair pays the LDS latency).
  const float* base = lds_in + (lane & 15) * ld + 4 * (lane >> 4);
  const float* bp = base + kb0 * 16;      // this chunk starts at k-block kb0
  const float* bpn = base + kb0_next * 16;
  constexpr int NP = (CKB + 1) / 2;
  BPair<RT> b[2];
  b[0] = b0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int cur = p & 1, nxt = cur ^ 1;
    if (p + 1 < NP) bpair_load<RT, CKB>(b[nxt], bp, ld, p + 1);
    else bpair_load<RT, CKB>(b[nxt], bpn, ld, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cc = 2 * p + h;
        if (cc < CKB) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[h][rt] = mfma4(f.a[cc < CKB ? cc : 0][j], b[cur].v[h][rt][j], acc[h][rt]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  b0 = b[NP & 1];
}

// One streamed layer.  The plan pads every streamed layer to an EVEN number of chunks, so
// the two register stages keep fixed roles (fa: even chunks, fb: odd chunks) and no stage is
// ever copied: invariant on entry and exit -- fa holds (or is receiving) the chunk the
// processing order reaches next, `q` is the chunk after it.  The epilogue of tile i runs at
// the start of tile i+1, after that tile's first loads have been issued and waited for:
// at every frag_wait the only younger VMEM operations are the CKB loads just issued.
template <int RT, int CA, int CB, class SC, class Epi>
__device__ __forceinline__ void stream_layer(const SdV& sd, int li, Cursor& q, FragS<CA>& fa,
                                             FragS<CB>& fb, const float* lds_in, int ld,
                                             int wid, int lane, Epi& epi, unsigned vo0, unsigned vo1,
                                             long long* prof = nullptr) {
  // On entry the load cursor is one chunk ahead INSIDE layer li (every layer has >= 2 chunks per
  // tile), unless this wave owns no tile of it: the layer shape is already in SGPRs.
  if (!q.live || q.li != li) return;
  const int n_ot = SC::NOT ? (li == 0 ? SC::NOT0 : SC::NOT) : q.n_ot;
  const int nch2 = (SC::NKB ? SC::NKB : q.n_kb) / (CA + CB);
  int pslot = 24;
  BPair<RT> b0;
  bpair_load<RT, CA>(b0, lds_in + (lane & 15) * ld + 4 * (lane >> 4), ld, 0);
  f32x4 pacc[RT];
  typename Epi::Pre ppre[RT];
  int pot = -1;
  for (int ot = wid; ot < n_ot; ot += PF_NW) {
    if (prof && pslot < 32) prof[pslot++] = (long long)__builtin_readcyclecounter();
    typename Epi::Pre pre[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) pre[rt] = epi.pre(ot, rt);
    f32x4 acc[2][RT];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c2 = 0;
    do {   // nch2 >= 1: the body (and its waits) runs at least once per tile
      frag_load<CB>(fb, q, vo0, vo1);
      cur_advance<CB, SC>(sd, q, wid);
      frag_wait<CA, CB>(fa);
      if (c2 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) epi.landed(pre[rt], rt);
        if (pot >= 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) epi(pot, rt, pacc[rt], ppre[rt]);
        }
      }
      frag_compute<RT, CA>(fa, c2 * (CA + CB), c2 * (CA + CB) + CA, lds_in, ld, lane, acc, b0);
      frag_load<CA>(fa, q, vo0, vo1);
      cur_advance<CA, SC>(sd, q, wid);
      frag_wait<CB, CA>(fb);
      frag_compute<RT, CB>(fb, c2 * (CA + CB) + CA, (c2 + 1 < nch2) ? (c2 + 1) * (CA + CB) : 0, lds_in, ld, lane,
                           acc, b0);
    } while (++c2 < nch2);
    if (prof && pslot < 32) prof[pslot++] = (long long)__builtin_readcyclecounter();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      pacc[rt] = acc[0][rt] + acc[1][rt];
      ppre[rt] = pre[rt];
    }
    pot = ot;
  }
  if (pot >= 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) epi(pot, rt, pacc[rt], ppre[rt]);
  }
}

// The same on the bf16 piece planes (pmbrl_split.h): stages of CA / CB K32 blocks = CA*NP / CB*NP weight
// loads; the stream cursor counts loads, so every layer is padded to whole (CA + CB)-block pairs.
template <int RT, int CA, int CB, int NP, bool F16, class SC, class Epi>
__device__ __forceinline__ void stream_layer_s(const SdV& sd, int li, Cursor& q, FragS<CA * NP>& fa,
                                               FragS<CB * NP>& fb, const float* buf_in, unsigned ldb,
                                               int wid, int lane, Epi& epi, unsigned vo0, unsigned vo1,
                                               long long* prof = nullptr, int ot_base = 0) {
  // (ot_base: the stream's tile 0 of this layer is output tile ot_base -- the tiles before it are
  //  register-resident, see resident_tile_s)
  if (!q.live || q.li != li) return;
  const int n_ot = SC::NOT ? (li == 0 ? SC::NOT0 : SC::NOT) : q.n_ot;
  const int nch2 = (SC::NKB ? SC::NKB : q.n_kb) / ((CA + CB) * NP);
  int pslot = 24;
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  BQ<RT, NP> b0;
  bq_load<RT, NP>(b0, lb, ldb, 0);
  f32x4 pacc[RT];
  typename Epi::Pre ppre[RT];
  int pot = -1;
  for (int ot = wid; ot < n_ot; ot += PF_NW) {
    if (prof && pslot < 32) prof[pslot++] = (long long)__builtin_readcyclecounter();
    typename Epi::Pre pre[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) pre[rt] = epi.pre(ot + ot_base, rt);
    f32x4 acc[2][RT];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c2 = 0;
    do {
      frag_load<CB * NP>(fb, q, vo0, vo1);
      cur_advance<CB * NP, SC>(sd, q, wid);
      frag_wait<CA * NP, CB * NP>(fa);
      if (c2 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) epi.landed(pre[rt], rt);
        if (pot >= 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) epi(pot, rt, pacc[rt], ppre[rt]);
        }
      }
      frag_compute_s<RT, CA, NP, F16>(fa, c2 * (CA + CB), c2 * (CA + CB) + CA, lb, ldb, acc, b0);
      frag_load<CA * NP>(fa, q, vo0, vo1);
      cur_advance<CA * NP, SC>(sd, q, wid);
      frag_wait<CB * NP, CA * NP>(fb);
      frag_compute_s<RT, CB, NP, F16>(fb, c2 * (CA + CB) + CA, (c2 + 1 < nch2) ? (c2 + 1) * (CA + CB) : 0, lb, ldb, acc, b0);
    } while (++c2 < nch2);
    if (prof && pslot < 32) prof[pslot++] = (long long)__builtin_readcyclecounter();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      pacc[rt] = pm_chains_sum<F16, NP>(acc[0][rt], acc[1][rt]);
      ppre[rt] = pre[rt];
    }
    pot = ot + ot_base;
  }
  if (pot >= 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) epi(pot, rt, pacc[rt], ppre[rt]);
  }
}

// One output tile whose weights (NB K32 blocks x NP pieces) stay in REGISTERS for the whole launch: no
// weight load, no wait -- the sweep kernels are bound by the weight stream L2 -> CU, and the split-precision
// instantiations leave ~90 registers per lane unused, which is one tile per wave.
template <int RT, int NB, int NP, bool F16, class Epi>
__device__ __forceinline__ void resident_tile_s(const FragS<NB * NP>& w, const float* buf_in, unsigned ldb, int lane,
                                                Epi& epi, int ot) {
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  typename Epi::Pre pre[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) pre[rt] = epi.pre(ot, rt);
  BQ<RT, NP> b0;
  bq_load<RT, NP>(b0, lb, ldb, 0);
  f32x4 acc[2][RT];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag_compute_s<RT, NB, NP, F16>(w, 0, 0, lb, ldb, acc, b0);
  Epi::wait_all();      // the epilogue operands (adjoint: activation bits from L2); also lands the stream's next stage
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    epi.landed(pre[rt], rt);
    epi(ot, rt, pm_chains_sum<F16, NP>(acc[0][rt], acc[1][rt]), pre[rt]);
  }
}

// The same from LDS: one output tile per wave whose weights (NB K32 blocks x NP pieces, fragment order) were copied
// into LDS at the start of the launch -- the sweep's SECOND streamed layer in the shape-specialised 16-row plain
// instances, where a workgroup has a CU's LDS to itself and two thirds of it idle (resident_tile_s covers the first
// streamed layer with the registers that are left).  LDS reads run at four times the rate of the L2 path and a quarter
// of its latency: the layer then streams 5 of its 13 tiles like its sibling.  The last K32 block is stored for its
// first `last_lanes` lanes only (K = 208 of 224: the rest is zero padding), which is what makes eight tiles fit.
template <int RT, int NB, int NP, bool F16, class Epi>
__device__ __forceinline__ void lds_tile_s(const float* wl, int last_lanes, const float* buf_in, unsigned ldb, int lane,
                                           Epi& epi, int ot) {
  typedef PmPairs<NP> PP;
  const unsigned short* lb = pm_plane_lane(buf_in, ldb, lane);
  typename Epi::Pre pre[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) pre[rt] = epi.pre(ot, rt);
  f32x4 acc[2][RT];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[k][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool last_on = lane < last_lanes;
  const float* wp = wl + lane * 4;
  const float* wlast = wl + (size_t)(NB - 1) * NP * 256 + (last_on ? lane : 0) * 4;
  f32x4 w[NB][NP];
  auto ldw = [&](int blk) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (blk < NB - 1) {
        w[blk][p] = *reinterpret_cast<const f32x4*>(wp + (size_t)(blk * NP + p) * 256);
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wlast + (size_t)p * last_lanes * 4);
        w[blk][p] = last_on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  BQ<RT, NP> b[2];
  bq_load<RT, NP>(b[0], lb, ldb, 0);
  ldw(0);
  if (NB > 1) ldw(1);
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int cur = blk & 1, nxt = cur ^ 1;
    if (blk + 2 < NB) ldw(blk + 2);
    bq_load<RT, NP>(b[nxt], lb, ldb, blk + 1 < NB ? blk + 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < PP::N; ++q)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[pm_chain<F16, NP>(q)][rt] =
            pm_mfma_bf<F16>(w[blk][PP::W[q]], b[cur].v[PP::A[q]][rt], acc[pm_chain<F16, NP>(q)][rt]);
    __builtin_amdgcn_sched_barrier(0);
  }
  Epi::wait_all();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    epi.landed(pre[rt], rt);
    epi(ot, rt, pm_chains_sum<F16, NP>(acc[0][rt], acc[1][rt]), pre[rt]);
  }
}
// floats of one such tile in LDS
__host__ __device__ inline size_t pm_lds_tile_floats(int n_loads, int np, int last_lanes) {
  return (size_t)(n_loads - np) * 256 + (size_t)np * last_lanes * 4;
}

// Register-resident single-k-block layer (first layer forward / head adjoint backward).
template <int RT>
struct Res0 {
  f32x4 w[PM_L0T];
};
template <int RT>
__device__ __forceinline__ void res0_load(Res0<RT>& r, const float* wf, int n_ot, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < PM_L0T; ++i) {
    // unconditional load of a clamped tile, zeroed afterwards: the prologue's loads stay in one batch
    const int ot = wid + i * PF_NW, otc = ot < n_ot ? ot : n_ot - 1;
    const f32x4 v = ldg4(wf + (size_t)otc * 256 + lane * 4);
    r.w[i] = ot < n_ot ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int RT, class Epi>
struct Res0Pre {
  typename Epi::Pre p[PM_L0T][RT];
};
// epilogue operands of a resident layer: issued BEFORE the barrier that opens its phase
template <int RT, class Epi>
__device__ __forceinline__ void res0_prefetch(Res0Pre<RT, Epi>& pp, const Epi& epi, int n_ot, int wid) {
  // (four row tiles per wave, LDS operands: fetched inside res0_layer instead -- eight Pre sets held across the
  //  barrier and the MFMAs are 40 registers these instances spill, and every epilogue then waited for a scratch
  //  reload: 12 k cycles for a K = 16 layer)
  if constexpr (RT >= 4 && Epi::kLdsPre) return;
#pragma unroll
  for (int i = 0; i < PM_L0T; ++i) {
    const int ot = wid + i * PF_NW;
    const int ots = ot < n_ot ? ot : wid;     // absent tile: harmless duplicate fetch
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) pp.p[i][rt] = epi.pre(ots, rt);
  }
}
template <int RT, class Epi>
__device__ __forceinline__ void res0_layer(const Res0<RT>& r, int n_ot, const float* lds_in, int ld,
                                           int wid, int lane, Epi& epi, Res0Pre<RT, Epi>& pp) {
  const float* bbase = lds_in + (lane & 15) * ld + 4 * (lane >> 4);
  f32x4 b[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) b[rt] = *reinterpret_cast<const f32x4*>(bbase + rt * 16 * ld);
  // all resident tiles at once, branch-free (absent tiles carry zero weights): the
  // PM_L0T * RT accumulator chains interleave
  f32x4 acc[PM_L0T][RT];
#pragma unroll
  for (int i = 0; i < PM_L0T; ++i)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[i][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < PM_L0T; ++i)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[i][rt] = mfma4(r.w[i][j], b[rt][j], acc[i][rt]);
  Epi::wait_all();
#pragma unroll
  for (int i = 0; i < PM_L0T; ++i) {
    const int ot = wid + i * PF_NW;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if constexpr (RT >= 4 && Epi::kLdsPre) {
        if (ot < n_ot) epi(ot, rt, acc[i][rt], epi.pre(ot, rt));
      } else {
        epi.landed(pp.p[i][rt], rt);
        if (ot < n_ot) epi(ot, rt, acc[i][rt], pp.p[i][rt]);
      }
    }
  }
}

// Narrow head / tail (<= 16 outputs = ONE output tile, K = hidden width): the K range is
// split over the 8 waves, each wave keeps its <= 2 k-blocks of the (single-tile) weight
// fragments in registers for the whole launch, issues 4-8 MFMAs per step and leaves its
// partial tile in LDS; the consumer phase adds the 8 partials in fixed order.
//   partial layout: hpart[(wave*RT + rt)*256 + lane*4 + c]  with the MFMA D mapping
//   out j = 4*(lane>>4) + c, row = rt*16 + (lane&15)
#define PM_HKB 2
struct HeadW {
  f32x4 w[PM_HKB];
};
__device__ __forceinline__ void head_load(HeadW& h, const float* wf, int n_kb, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < PM_HKB; ++i) {
    const int kb = wid * PM_HKB + i, kbc = kb < n_kb ? kb : n_kb - 1;
    const f32x4 v = ldg4(wf + (size_t)kbc * 256 + lane * 4);
    h.w[i] = kb < n_kb ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int RT>
__device__ __forceinline__ void head_partial(const HeadW& h, int n_kb, const float* lds_in, int ld,
                                             float* hpart, int wid, int lane) {
  const float* bp = lds_in + (lane & 15) * ld + 4 * (lane >> 4);
  f32x4 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < PM_HKB; ++i) {
    const int kb = wid * PM_HKB + i;
    const int kbs = kb < n_kb ? kb : 0;   // absent blocks: zero weights, in-range dummy read
    f32x4 b[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      b[rt] = *reinterpret_cast<const f32x4*>(bp + rt * 16 * ld + kbs * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma4(h.w[i][j], b[rt][j], acc[rt]);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    *reinterpret_cast<f32x4*>(hpart + ((size_t)(wid * RT + rt) * 64 + lane) * 4) = acc[rt];
}
// value of output j for local row r: sum of the 8 wave partials (fixed order)
template <int RT>
__device__ __forceinline__ float head_value(const float* hpart, int r, int j) {
  const int rt = r >> 4, ln = ((j >> 2) << 4) + (r & 15), c = j & 3;
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < PF_NW; ++w) s += hpart[((size_t)(w * RT + rt) * 64 + ln) * 4 + c];
  return s;
}

// K-split of a layer's last output tile (pm_fast_ksplit): partial tile of this wave's <= 2
// k-blocks, weights read from LDS (tw: [n_kb][64 lanes][4]).
template <int RT>
__device__ __forceinline__ void tail_partial(const float* tw, int n_kb, const float* lds_in, int ld,
                                             float* tp, int* tcnt, int wid, int lane) {
  const float* bp = lds_in + (lane & 15) * ld + 4 * (lane >> 4);
  f32x4 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < PM_HKB; ++i) {
    const int kb = wid * PM_HKB + i;
    if (kb < n_kb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(tw + (size_t)kb * 256 + lane * 4);
      f32x4 b[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) b[rt] = *reinterpret_cast<const f32x4*>(bp + rt * 16 * ld + kb * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma4(a[j], b[rt][j], acc[rt]);
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
    *reinterpret_cast<f32x4*>(tp + ((size_t)(wid * RT + rt) * 64 + lane) * 4) = acc[rt];
  // LDS operations of a wave complete in order: the partial is visible before the count moves
  if (lane == 0) __hip_atomic_fetch_add(tcnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// last wave: wait for the 8 partials of round `round`, add them in wave order
template <int RT>
__device__ __forceinline__ void tail_gather(const float* tp, int* tcnt, int round, int lane, f32x4 (&sum)[RT]) {
  const int want = PF_NW * round;
  while (__hip_atomic_load(tcnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
    __builtin_amdgcn_s_sleep(1);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < PF_NW; ++w)
      s += *reinterpret_cast<const f32x4*>(tp + ((size_t)(w * RT + rt) * 64 + lane) * 4);
    sum[rt] = s;
  }
}

// ---------------------------------------------------------------------------
// epilogues reading bias / masks from LDS
// ---------------------------------------------------------------------------
// Both epilogues run once per output tile on the critical path of a phase (the partner wave
// of the SIMD is in ITS epilogue at the same time, so the MFMA pipe idles): they are kept to
// a few dozen VALU instructions -- 32-bit offsets from uniform bases (the stash row block is
// Rw = 16*RT, a compile-time constant), a multiply by the precomputed 1/keep.
struct PmEmpty {};
// 64-row workgroups park the heads' partial tiles (fp32 sums) in whichever activation buffer is idle
// (pm_fast_hp_alias).  On the piece planes that leaves arbitrary 16-bit patterns -- infinities and NaNs among
// them -- in the K-padding columns the layer epilogues never write; the next GEMM multiplies those by zero
// weights, and 0 x inf is not 0 (forward: the NaN pre-activations then vanish in the ReLU and silently zero
// the row's hidden units).  The epilogue of a layer's LAST tile therefore clears the padding of its rows
// (the `ot == nt - 1` branches below).
// NP = 0: the layer output goes to LDS as fp32 rows (leading dimension ld); NP > 0: as NP bf16 (F16: fp16)
// piece planes (pmbrl_split.h; leading dimension ld in 16-bit elements)
template <int RT, int NP = 0, bool F16 = false>
struct EpiFwdL {
  const float* bias;        // LDS, padded
  const uint16_t* mask;     // LDS [R][nt]
  uint8_t* abits;           // HBM [B][nt][4] slice of step t: one nibble-byte per lane group
  float inv_keep;
  float* lds_out;
  float* stash;             // HBM block or nullptr
  int ld, Rw, row0, nvalid, nt, lane;
  // F16: LDS word set when an activation leaves fp16's range; an empty member otherwise (the fp32 kernels are
  // at their scalar-register limit: not one more pointer in their epilogue descriptors)
  [[no_unique_address]] typename std::conditional<F16, int*, PmEmpty>::type ovf;
  struct Pre {
    f32x4 b;
    unsigned mw;
  };
  // operands of the epilogue of tile (ot, rt): fetched when the tile STARTS, so their latency
  // hides behind the tile's MFMAs
  __device__ __forceinline__ Pre pre(int ot, int rt) const {
    const unsigned g = (unsigned)lane >> 4;
    const unsigned lrow = rt * 16 + ((unsigned)lane & 15u);
    Pre p;
    p.b = *reinterpret_cast<const f32x4*>(bias + ot * 16 + 4 * g);
    p.mw = mask[lrow * (unsigned)nt + ot];
    return p;
  }
  __device__ __forceinline__ void landed(Pre&, int) const {}   // LDS operands: the compiler tracks them
  static __device__ __forceinline__ void wait_all() {}
  static constexpr bool kLdsPre = true;    // pre() reads LDS only: cheap enough to fetch right before the epilogue
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc, const Pre& pr) {
    constexpr unsigned RW = 16 * RT;
    const unsigned g = (unsigned)lane >> 4;
    const unsigned lrow = rt * 16 + ((unsigned)lane & 15u);
    const unsigned f0 = ot * 16 + 4 * g;
    const f32x4 b = pr.b;
    const unsigned nib = (pr.mw >> (4 * g)) & 0xFu;
    f32x4 h;
    unsigned act = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[r] + b[r];
      const bool a = ((nib >> r) & 1u) && (v > 0.f);
      h[r] = a ? v * inv_keep : 0.f;
      act |= (a ? 1u : 0u) << r;
    }
    if constexpr (NP > 0) {
      if constexpr (RT >= 4) {
        // (one per-lane pointer, constant offsets per row tile / piece: pm_store_planes_q)
        unsigned short* q = reinterpret_cast<unsigned short*>(lds_out) + ((unsigned)lane & 15u) * (unsigned)ld + f0;
        asm volatile("" : "+v"(q));      // keep it ONE register: do not fold the buffer offset into the immediates
        pm_store_planes_q<NP, 16 * RT, F16>(q, (unsigned)ld, rt, h);
        if (ot == nt - 1)
          for (unsigned c = 16; f0 + c + 16 <= (unsigned)ld; c += 16)
            pm_store_planes_q<NP, 16 * RT, F16>(q + c, (unsigned)ld, rt, f32x4{0.f, 0.f, 0.f, 0.f});
      } else {
        pm_store_planes<NP, 16 * RT, F16>(lds_out, (unsigned)ld, lrow, f0, h);
      }
    } else {
      *reinterpret_cast<f32x4*>(lds_out + lrow * (unsigned)ld + f0) = h;
    }
    if constexpr (F16) {
      // fp16 pieces: a value beyond the format's range would turn into inf - inf = NaN inside the next
      // layer and vanish in its ReLU; report it instead (h >= 0 here; the sampling phase of this step
      // turns the flag into a failed step)
      // (a WEIGHT beyond fp16's range is caught when the weights are packed: pm_pack_all, RolloutArgs::wflag)
      if (fmaxf(fmaxf(h[0], h[1]), fmaxf(h[2], h[3])) > 65504.f) *ovf = 1;
    }
#ifndef PM_EXP_NOSTASH
    if (stash) {
      PM_GLOBAL float* sp = (PM_GLOBAL float*)(stash + (unsigned)ot * (16u * RW));   // uniform
      const unsigned lo = 4 * g * RW + lrow;
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[lo + r * RW] = h[r];
    }
#endif
    if ((int)lrow < nvalid) {
      PM_GLOBAL uint8_t* ap = (PM_GLOBAL uint8_t*)(abits + (unsigned)ot * 4u);   // uniform
      ap[((unsigned)row0 + lrow) * (unsigned)nt * 4u + g] = (uint8_t)act;
    }
  }
};

template <int RT, int NP = 0>
struct EpiBwdL {
  const uint8_t* abits;     // HBM [B][nt][4] slice of step t
  float inv_keep;
  float* lds_out;
  float* stash;
  int ld, Rw, row0, nvalid, nt, lane;
  struct Pre {
    unsigned nib;
  };
  // the activation bits come from HBM / L2: fetched when the tile starts (or, for the
  // resident layers, before the barrier that opens the phase)
  // Issued as an inline-asm load (see frag_load): in flight until landed() after a wait that
  // covers it.  Rows past the end of the batch read a valid row's bits (clamped) and are
  // masked in landed().
  __device__ __forceinline__ Pre pre(int ot, int rt) const {
    const unsigned g = (unsigned)lane >> 4;
    const unsigned lrow = rt * 16 + ((unsigned)lane & 15u);
    const unsigned rowc = (int)lrow < nvalid ? lrow : (unsigned)(nvalid - 1);
    const uint8_t* ap = abits + (unsigned)ot * 4u;          // uniform
    const unsigned off = ((unsigned)row0 + rowc) * (unsigned)nt * 4u + g;
    Pre p;
    asm volatile("global_load_ubyte %0, %1, %2" : "=&v"(p.nib) : "v"(off), "s"(ap));
    return p;
  }
  __device__ __forceinline__ void landed(Pre& p, int rt) const {
    asm volatile("" : "+v"(p.nib));
    const unsigned lrow = rt * 16 + ((unsigned)lane & 15u);
    if ((int)lrow >= nvalid) p.nib = 0;
  }
  static __device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0)"); }
  static constexpr bool kLdsPre = false;   // pre() is an HBM / L2 load: issued ahead
  __device__ __forceinline__ void operator()(int ot, int rt, f32x4 acc, const Pre& pr) {
    constexpr unsigned RW = 16 * RT;
    const unsigned g = (unsigned)lane >> 4;
    const unsigned lrow = rt * 16 + ((unsigned)lane & 15u);
    const unsigned f0 = ot * 16 + 4 * g;
    const unsigned nib = pr.nib;
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = ((nib >> r) & 1u) ? acc[r] * inv_keep : 0.f;
    if constexpr (NP > 0) {
      if constexpr (RT >= 4) {
        unsigned short* q = reinterpret_cast<unsigned short*>(lds_out) + ((unsigned)lane & 15u) * (unsigned)ld + f0;
        asm volatile("" : "+v"(q));
        pm_store_planes_q<NP, 16 * RT, false>(q, (unsigned)ld, rt, h);
        if (ot == nt - 1)
          for (unsigned c = 16; f0 + c + 16 <= (unsigned)ld; c += 16)
            pm_store_planes_q<NP, 16 * RT, false>(q + c, (unsigned)ld, rt, f32x4{0.f, 0.f, 0.f, 0.f});
      } else {
        pm_store_planes<NP, 16 * RT>(lds_out, (unsigned)ld, lrow, f0, h);
      }
    } else {
      *reinterpret_cast<f32x4*>(lds_out + lrow * (unsigned)ld + f0) = h;
    }
#ifndef PM_EXP_NOSTASH
    if (stash) {
      PM_GLOBAL float* sp = (PM_GLOBAL float*)(stash + (unsigned)ot * (16u * RW));   // uniform
      const unsigned lo = 4 * g * RW + lrow;
#pragma unroll
      for (int r = 0; r < 4; ++r) sp[lo + r * RW] = h[r];
    }
#endif
  }
};

// ---------------------------------------------------------------------------
// rewards: not part of the sweep kernels
// ---------------------------------------------------------------------------
#ifdef PM_MAIN_TU   // non-template kernels: compiled into pmbrl.hip only
// All rewards of a rollout in one fully parallel pass: thread = one (t, b) row-step.
// r~ -> rt (if rewards are moment matched afterwards) or rewards; d r~/d x~ -> Jx; d r~/d a -> Ja;
// non-finite states / rewards are reported through the status word like in the sweep.
// The 256 state rows of a block go through LDS (row stride D | 1): a thread walks ITS row, so straight from HBM
// every load and every Jacobian store of a wave touched 64 cache lines (1.7 ms at D = 32, 1.6 M row-steps).  The
// Jacobian row overwrites the state row in place and leaves the same way.
// NU: compile-time bound on the action width (4: the common shapes, their path unchanged; 8 / 16: the quadratic forms of the
// action cost unrolled to that width -- an instance of their own: compiled into the one kernel their 30 KB of code sat
// between the halves of the narrow path and cost ITS launches 3 us of instruction fetch)
// KT: compile-time bound on the tip residuals k (2: the cart-pole class -- a tip position in the plane; KT: any):
// the loops over them are unrolled to KT with the entries beyond k dropped by selects, so at k = 2 an instance unrolled to
// eight carried sixteen times the loads and selects of its quadratic forms (4.6 of the launch's 10.5 us at C2 were this
// arithmetic).
template <int NU, int KT>
__global__ __launch_bounds__(256) void pm_reward_all_kernel(const RolloutArgs A) {
  extern __shared__ float rw_rows[];
  // The cache lines of the reward's constants a row-step will read -- the heads of its arrays: the scalar cache met each of
  // them as a miss of its own, one behind the other, in the middle of the arithmetic; requested here they arrive while
  // the states do.  The values are not used: the registers are held until the barrier below.
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned pf[16];
  {
    typedef __attribute__((address_space(4))) const unsigned* kup;
    const kup rp = (kup)A.rew;
    constexpr int PFO[16] = {0,
                             (int)offsetof(RewardDev, C), (int)offsetof(RewardDev, C) + 64, (int)offsetof(RewardDev, tt),
                             (int)offsetof(RewardDev, Q), (int)offsetof(RewardDev, Q) + 64, (int)offsetof(RewardDev, QQ),
                             (int)offsetof(RewardDev, QQ) + 64, (int)offsetof(RewardDev, R), (int)offsetof(RewardDev, RR),
                             (int)offsetof(RewardDev, phi_src), (int)offsetof(RewardDev, phi_mode),
                             (int)offsetof(RewardDev, d_copy), (int)offsetof(RewardDev, d_sin),
                             (int)offsetof(RewardDev, d_cos), (int)offsetof(RewardDev, w)};
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("s_load_dword %0, %1, %2" : "=s"(pf[u]) : "s"(rp), "n"(PFO[u]));
  }
#endif
  const long long n = (long long)A.H * A.B;
  const long long i0 = (long long)blockIdx.x * blockDim.x;
  const long long i = i0 + threadIdx.x;
  const int D = A.D, U = A.U;
  const int ld = D | 1;
  // (the adjoint call's failure flag, the status word's second entry: cleared here, by a launch every forward call makes)
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.status) A.status[1] = 0;
  const int nrow = (int)min((long long)blockDim.x, n - i0);
  {
    const float* src = (A.flags & PMBRL_FLAG_MM_STATES) ? A.xt + (size_t)i0 * D : A.states + ((size_t)i0 + A.B) * D;
    // (eight loads in flight per thread: one at a time -- load, wait, write -- the D = 4 block was four memory round trips
    //  in a row, a third of this launch at C2)
    const int ne = nrow * D;
    for (int e0 = threadIdx.x; e0 < ne; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + 256 * u, ne - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        if (e < ne) {
          const int r = e / D, d = e - r * D;
          rw_rows[r * ld + d] = v[u];
        }
      }
    }
  }
  // (the row's first four actions: requested here, beside the states)
  float a4[4];
  {
    const float* ac = A.actions + (size_t)min(i, n - 1) * U;
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] = ac[min(u, U - 1)];
  }
  // (wider action vectors: the NU = 8 / 16 instances)
  float av[NU];
  if constexpr (NU > 4) {
    const float* ac = A.actions + (size_t)min(i, n - 1) * U;
#pragma unroll
    for (int u = 0; u < NU; ++u) av[u] = ac[min(u, U - 1)];
  }
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int u = 0; u < 16; ++u) asm volatile("" ::"s"(pf[u]));
#endif
  if (i < n) {
  const int t = (int)(i / A.B);
  // (the reward's constants through a constant-address-space pointer: wave-uniform indices make scalar loads of them; as a
  //  generic pointer every coefficient was a vector load from global memory per thread -- a hundred of them a row-step)
  typedef __attribute__((address_space(4))) const RewardDev* rew_kptr;
#if defined(__HIP_DEVICE_COMPILE__)
  const rew_kptr rw = (rew_kptr)A.rew;
#else
  const RewardDev* rw = A.rew;
#endif
  float* xs = rw_rows + threadIdx.x * ld;
  // No per-thread array is indexed at run time (that would live in scratch): the feature map is
  // walked in gather form, x / a come from their (L1-resident) rows, only the k <= 8 tip
  // residuals sit in registers with static indices.
  const int k = rw->k, De = rw->De;
  bool ok = true;
  for (int d = 0; d < D; ++d) ok = ok && isfinite(xs[d]);
  float delta[KT];
#pragma unroll
  for (int q = 0; q < KT; ++q) delta[q] = 0.f;
  for (int j = 0; j < De; ++j) {
    const float xv = xs[rw->phi_src[j]];
    const int md = rw->phi_mode[j];
    const float ph = md == 0 ? xv : (md == 1 ? sinf(xv) : cosf(xv));
    // (the coefficients of all KT rows are loaded, the rows beyond k dropped by a select: every index is inside
    //  the array, and a load UNDER the condition is a scalar load, a wait and a branch of its own -- this kernel was a chain
    //  of ~190 of those, most of its 13 us at C2)
    float cj[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) cj[q] = rw->C[q * De + j];
#pragma unroll
    for (int q = 0; q < KT; ++q) delta[q] = q < k ? fmaf(ph, cj[q], delta[q]) : delta[q];
  }
#pragma unroll
  for (int q = 0; q < KT; ++q) {
    const float ttq = rw->tt[q];
    delta[q] -= (q < k) ? ttq : 0.f;
  }
  float cost = 0.f;
#pragma unroll
  for (int q = 0; q < KT; ++q) {
    float s = 0.f;
    float qv[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) qv[p] = rw->Q[p * k + q];      // (p k + q < KT^2 <= 64 for any k <= KT)
#pragma unroll
    for (int p = 0; p < KT; ++p) s = (p < k && q < k) ? fmaf(delta[p], qv[p], s) : s;
    cost = q < k ? fmaf(s, delta[q], cost) : cost;
  }
  // U > 4: s[q] = sum_p a_p M[p][q], p ascending, for a compile-time bound NU >= U on both indices: the rows of M loaded
  // whole (every index inside the 16 x 16 array), entries beyond U dropped by selects -- as run-time loops over U this was
  // a vector load of a_p, a scalar load of M[p][q] and a wait per term: 2 x 64 of them in a row at U = 8, half of the
  // launch's 0.5 ms at C5
  auto quad = [&](auto nu, auto Mx, float* sq) {
    constexpr int NW = decltype(nu)::value;
#pragma unroll
    for (int q = 0; q < NW; ++q) sq[q] = 0.f;
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      float row[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) row[q] = Mx[p * U + q];
#pragma unroll
      for (int q = 0; q < NW; ++q) sq[q] = (p < U && q < U) ? fmaf(av[p], row[q], sq[q]) : sq[q];
    }
  };
  auto cost_part = [&](auto nu) {
    constexpr int NW = decltype(nu)::value;
    float sq[NW];
    quad(nu, &rw->R[0], sq);
#pragma unroll
    for (int q = 0; q < NW; ++q) cost = q < U ? fmaf(sq[q], av[q], cost) : cost;
  };
  auto ja_part = [&](auto nu, float gcv) {
    constexpr int NW = decltype(nu)::value;
    float sq[NW];
    quad(nu, &rw->RR[0], sq);
#pragma unroll
    for (int q = 0; q < NW; ++q)
      if (q < U) A.Ja[(size_t)i * U + q] = gcv * sq[q];
  };
  if constexpr (NU == 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = 0.f;
      float rv4[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) rv4[p] = rw->R[p * U + q];
#pragma unroll
      for (int p = 0; p < 4; ++p) s = (p < U && q < U) ? fmaf(a4[p], rv4[p], s) : s;
      cost = q < U ? fmaf(s, a4[q], cost) : cost;
    }
  } else {
    cost_part(std::integral_constant<int, NU>{});
  }
  cost *= rw->w;
  const float rv = rw->kind == PMBRL_REWARD_EXP ? expf(-cost) : -cost;
  ok = ok && isfinite(rv);
  if (!ok) atomicMin(A.status, t);
  // adjoint with unit upstream gradient
  const float gc = (rw->kind == PMBRL_REWARD_EXP ? -rv : -1.f) * rw->w;
  float gdelta[KT];
#pragma unroll
  for (int q = 0; q < KT; ++q) {
    float s = 0.f;
    float qv[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) qv[p] = rw->QQ[p * k + q];
#pragma unroll
    for (int p = 0; p < KT; ++p) s = (p < k && q < k) ? fmaf(delta[p], qv[p], s) : s;
    gdelta[q] = gc * s;
  }
  auto gphi = [&](int j) {
    float s = 0.f;
    float cj[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) cj[q] = rw->C[q * De + j];
#pragma unroll
    for (int q = 0; q < KT; ++q) s = q < k ? fmaf(gdelta[q], cj[q], s) : s;
    return s;
  };
  for (int d = 0; d < D; ++d) {
    float g = 0.f;
    const int jc = rw->d_copy[d], js = rw->d_sin[d];
    if (jc >= 0) g += gphi(jc);
    if (js >= 0) {
      const float th = xs[d];
      g += gphi(js) * cosf(th) - gphi(rw->d_cos[d]) * sinf(th);
    }
    xs[d] = g;      // (x_d has been read: only dimension d's own sine / cosine terms use it)
  }
  if constexpr (NU == 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = 0.f;
      float rv4[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) rv4[p] = rw->RR[p * U + q];
#pragma unroll
      for (int p = 0; p < 4; ++p) s = (p < U && q < U) ? fmaf(a4[p], rv4[p], s) : s;
      if (q < U) A.Ja[(size_t)i * U + q] = gc * s;
    }
  } else {
    ja_part(std::integral_constant<int, NU>{}, gc);
  }
  if (A.flags & PMBRL_FLAG_MM_REWARDS) A.rt[i] = rv;
  else A.rewards[i] = rv;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nrow * D; e += blockDim.x) {
    const int r = e / D, d = e - r * D;
    A.Jx[(size_t)i0 * D + e] = rw_rows[r * ld + d];
  }
}

// moment matching of the rewards for all (t, group) at once (one wave each), and its adjoint
// A LARGE group (mm_groups=None: one group of 2 500 rows) first brings its rows -- rewards, noise, in the adjoint the
// upstream gradient -- into LDS, sixteen loads in flight per lane, and runs the same routine on them there: its sums walk
// the rows one dependent load at a time, three to five passes of 39 memory round trips on one wave (26 / 34 us per launch
// at 2 500 rows).  Same arithmetic in the same order: the results are the unstaged form's bit for bit.
#define PM_MMR_STAGE_MIN 128           // rows from which a group is staged
#define PM_MMR_STAGE_MAX 4096          // ... and up to which its three arrays fit (48 KB beside the scratch)
__host__ __device__ inline size_t pm_mmr_lds_bytes(int M, int arrays) {
  const size_t scr = pm_mm_scratch_doubles(1) * sizeof(double);
  return (M >= PM_MMR_STAGE_MIN && M <= PM_MMR_STAGE_MAX) ? scr + (size_t)arrays * M * sizeof(float) : scr;
}
// rows src[idx(i)] for i < M into dst[i]
template <class IDX>
__device__ __forceinline__ void pm_mmr_stage(float* dst, const float* src, int M, int lane, IDX idx) {
  for (int i0 = lane; i0 < M; i0 += 16 * 64) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[idx(min(i0 + 64 * u, M - 1))];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + 64 * u < M) dst[i0 + 64 * u] = v[u];
  }
}
// A SMALL group (<= 64 rows: lane l holds row l) in registers: what pm_mm_fwd / pm_mm_bwd do at d = 1 -- the same sums
// in the same order (a lane's one row, the butterfly over the wave), the same formulas -- without their passes over
// memory: at 25 rows every pass of the general routine was a global load and a wait, three to five in a row per group
// (6 + 8 us per iteration at C3 for 4 000 groups).  Returns false on a non-positive pivot (pm_mm_chol's rule).
struct Mm1Fac { double mean, zmean, zistd, L, invd; };
__device__ __forceinline__ bool pm_mm1_reg_factor(float sv, float zv, bool in, int M, Mm1Fac& f) {
  const double inv_m = 1.0 / (double)M, inv_m1 = 1.0 / (double)(M - 1);
  double m = 0.0, zm = 0.0, zz = 0.0;
  if (in) {
    m += (double)sv;
    const double zd = (double)zv;
    zm += zd;
    zz += zd * zd;
  }
  m = pm_seg_sum(m, 64);
  zm = pm_seg_sum(zm, 64);
  zz = pm_seg_sum(zz, 64);
  m *= inv_m;
  zm *= inv_m;
  f.mean = m;
  f.zmean = zm;
  f.zistd = pm_rsqrt((zz - (double)M * zm * zm) * inv_m1);
  double acc = 0.0;
  if (in) acc += ((double)sv - m) * ((double)sv - m);
  acc = pm_seg_sum(acc, 64);
  const double c0 = acc * inv_m1 + 1e-12;
  double piv = c0;
  bool ok = true;
  if (!(piv > 6e-8 * c0)) {
    ok = false;
    piv = 1.0;
  }
  const double rs = pm_rsqrt(piv);
  f.L = piv * rs;
  f.invd = rs;
  return ok;
}
__global__ void pm_mm_rewards_fwd_kernel(const RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) double mmscr_r[];
  const int t = blockIdx.x / A.G, gi = blockIdx.x - t * A.G, lane = threadIdx.x;
  const int r0 = gi * A.M;
  const float* s = A.rt + (size_t)t * A.B + r0;
  const float* z = pm_zbase(A.zrr, 1, t, A.Bg, A.flags);
  int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags), Bg = A.Bg;
  if (A.M <= 64 && !(A.flags & PMBRL_FLAG_INFER_NS)) {
    const bool in = lane < A.M;
    const int li = min(lane, A.M - 1);
    const float sv = s[li], zv = z[(size_t)pm_zidx(zrow0, li, Bg)];
    Mm1Fac f;
    const bool ok = pm_mm1_reg_factor(sv, zv, in, A.M, f);
    if (in) {
      double acc = f.mean;
      acc += ((double)zv - f.zmean) * f.zistd * f.L;
      A.rewards[(size_t)t * A.B + r0 + lane] = (float)acc;
    }
    if (!ok && lane == 0) atomicMin(A.status, t);
    return;
  }
  if (A.M >= PM_MMR_STAGE_MIN && A.M <= PM_MMR_STAGE_MAX) {
    float* ls = reinterpret_cast<float*>(mmscr_r + pm_mm_scratch_doubles(1));
    float* lz = ls + A.M;
    pm_mmr_stage(ls, s, A.M, lane, [](int i) { return i; });
    pm_mmr_stage(lz, z, A.M, lane, [=](int i) { return (size_t)pm_zidx(zrow0, i, Bg); });
    pm_wave_sync();
    s = ls; z = lz; zrow0 = 0; Bg = 0;
  }
  const bool ok = pm_mm_fwd(s, 1, A.M, 1, z, 1, zrow0, Bg, (A.flags & PMBRL_FLAG_INFER_NS) != 0,
                            A.rewards + (size_t)t * A.B + r0, 1, mmscr_r, lane);
  if (!ok && lane == 0) atomicMin(A.status, t);
}
__global__ void pm_mm_rewards_bwd_kernel(const RolloutArgs A, float* gr_tilde) {
  extern __shared__ __attribute__((aligned(16))) double mmscr_r[];
  const int t = blockIdx.x / A.G, gi = blockIdx.x - t * A.G, lane = threadIdx.x;
  const int r0 = gi * A.M;
  if (A.nvalid && t >= *A.nvalid) return;   // a step the forward sweep did not complete
  const float* s = A.rt + (size_t)t * A.B + r0;
  const float* z = pm_zbase(A.zrr, 1, t, A.Bg, A.flags);
  const float* g = A.grad_rewards + (size_t)t * A.B + r0;
  int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags), Bg = A.Bg;
  if (A.M <= 64 && !(A.flags & PMBRL_FLAG_INFER_NS)) {
    const bool in = lane < A.M;
    const int li = min(lane, A.M - 1);
    const float sv = s[li], zv = z[(size_t)pm_zidx(zrow0, li, Bg)], gv = g[li];
    Mm1Fac f;
    (void)pm_mm1_reg_factor(sv, zv, in, A.M, f);
    const double inv_m = 1.0 / (double)A.M, inv_m1 = 1.0 / (double)(A.M - 1);
    // mbar = sum_r g ;  Lbar = g^T zhat
    double a = 0.0, lb = 0.0;
    if (in) {
      a += (double)gv;
      lb += (double)gv * (((double)zv - f.zmean) * f.zistd);
    }
    const double mbar = pm_seg_sum(a, 64);
    lb = pm_seg_sum(lb, 64);
    // pm_mm_bwd_solve at d = 1: Phi = L Lbar / 2, X = Phi / L, Sbar = X / L, P = 2 Sbar / (M - 1)
    double phi = 0.0;
    phi += f.L * lb;
    phi *= 0.5;
    const double xs = phi * f.invd;
    const double sb = xs * f.invd;
    const double P = (sb + sb) * inv_m1;
    if (in) {
      double acc = mbar * inv_m;
      acc += ((double)sv - f.mean) * P;
      gr_tilde[(size_t)t * A.B + r0 + lane] = (float)acc;
    }
    return;
  }
  if (A.M >= PM_MMR_STAGE_MIN && A.M <= PM_MMR_STAGE_MAX) {
    float* ls = reinterpret_cast<float*>(mmscr_r + pm_mm_scratch_doubles(1));
    float* lz = ls + A.M;
    float* lg = lz + A.M;
    pm_mmr_stage(ls, s, A.M, lane, [](int i) { return i; });
    pm_mmr_stage(lz, z, A.M, lane, [=](int i) { return (size_t)pm_zidx(zrow0, i, Bg); });
    pm_mmr_stage(lg, g, A.M, lane, [](int i) { return i; });
    pm_wave_sync();
    s = ls; z = lz; g = lg; zrow0 = 0; Bg = 0;
  }
  pm_mm_bwd(s, 1, A.M, 1, z, 1, zrow0, Bg, (A.flags & PMBRL_FLAG_INFER_NS) != 0, g, 1,
            gr_tilde + (size_t)t * A.B + r0, 1, mmscr_r, lane);
}

#endif   // PM_MAIN_TU

// ---------------------------------------------------------------------------
// LDS map of the fast kernels
// ---------------------------------------------------------------------------
struct FastLds {
  float *bufA, *bufB, *xa, *xb, *av, *gad, *rr, *gr;
  int hp_off;                    // head / tail partial tiles [PF_NW][RT][64][4]: offset from bufA, or -1 when
                                 // they live in whichever activation buffer is idle (pm_fast_hp_alias)
  float* base;
  float *zp, *zd, *mx, *iSx, *my, *Sy, *lSy, *psc, *pbi;
  float *jx;                     // backward: dL/dx~ rows [R][16]
  float *stg;                    // backward: staged per-row inputs of one step [R][1+2D+3U]
  float *zs;                     // in-kernel moment matching: this step's noise rows [R][D]
  float *xin;                    // split precision: fp32 input tile of the resident first layers [R][PM_XIN_LD]
  int *ovf;                      // split precision, fp16 pieces: range-overflow flag of the launch
  float *tp;                     // K-split last tile: partial tiles [PF_NW][RT][64][4]
  int *tcnt;                     //   ... and the arrival counter
  float *tw;                     //   ... and the LDS-resident weight k-blocks of those tiles
  double* mm;
};

// The partial head tiles are written while one activation buffer (the head's input) is being
// read and the other is dead: when that buffer is large enough they live there and cost no LDS.
__host__ __device__ inline bool pm_fast_hp_alias(int R, int LD, int RT) {
  return RT >= 4 && (size_t)R * LD >= (size_t)PF_NW * RT * 256;   // only where LDS is the constraint (64-row workgroups)
}

#define PM_XIN_LD 24             // = 8 (mod 16): conflict-free ds_read_b128 of the MFMA B operand
// LDS of the statistics exchange of split groups (behind wave 0's moment-matching scratch): the reference point (8),
// the z standardisation [zm | zi] of every step of the launch, and two staging sets, alternating between steps, of
// what wave 1 fetches a step ahead while wave 0 works -- the part's noise rows (forward sweep); scratch with the
// factor, the part's pre-mm rows and noise rows (adjoint).  A multiple of 4 floats.
__host__ __device__ inline size_t pm_fast_xch_floats(int R, int d, int steps) {
  const size_t n = 8 + 2 * (size_t)steps * 2 * d + 2 * (2 * pm_mm_scratch_doubles(d) + 2 * (size_t)R * d);
  return (n + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t pm_fast_lds_floats(int R, int LD, int D, int U, int RT,
                                                     const int* pnt, int pnl, const int* dnt,
                                                     int dnl, int mm_d, int prec = 0, int mm_waves = PF_NW,
                                                     int mm_group_rows = 0, int mm_steps = 0) {
  // (mm_waves: waves that do in-kernel moment matching at once = whole groups per workgroup, at most PF_NW;
  //  mm_group_rows: rows of a group split over workgroups -- four row blocks of the whole group behind L.mm)
  size_t n = 2 * (size_t)R * LD + 2 * (size_t)R * D + (size_t)R * U + (size_t)R * 16 + 2 * (size_t)R;
  if (!pm_fast_hp_alias(R, LD, RT)) n += (size_t)PF_NW * RT * 256;
  for (int l = 0; l < pnl; ++l) n += (size_t)pnt[l + 1] * 16;
  for (int l = 0; l < dnl; ++l) n += (size_t)dnt[l + 1] * 16;
  for (int l = 0; l < pnl - 1; ++l) n += ((size_t)R * pnt[l + 1] + 1) / 2;
  for (int l = 0; l < dnl - 1; ++l) n += ((size_t)R * dnt[l + 1] + 1) / 2;
  n += (size_t)R * U + (size_t)R * D;                 // zp, zd
  n += 2 * (size_t)(D + U) + 3 * (size_t)D + 2 * (size_t)U;
  n = (n + 3) & ~(size_t)3;
  n += (size_t)R * 16 + (size_t)R * (1 + 2 * D + 3 * U) + (size_t)R * D;   // jx, stg, zs
  n = (n + 3) & ~(size_t)3;
  if (prec) n += (size_t)R * PM_XIN_LD + 4;                                 // xin, ovf
  {
    const size_t tf = pm_fast_tail_floats(pnt, pnl, dnt, dnl, false, RT, prec), tb = pm_fast_tail_floats(pnt, pnl, dnt, dnl, true, RT, prec);
    const size_t tw = tf > tb ? tf : tb;
    if (tw) n += (size_t)PF_NW * RT * 256 + 4 + tw;   // tp, tcnt, tw
  }
  n += 2 * (size_t)mm_waves * pm_mm_scratch_doubles(mm_d);
  // split groups: what the statistics exchange keeps (pm_fast_xch_floats), then -- LAST, so that the LDS-resident weight
  // tiles of the instances that never exchange rows can take its place (pm_fast_rows_area_off) -- four row blocks of
  // the whole group for the rows + flags form
  if (mm_group_rows) n += pm_fast_xch_floats(R, mm_d, mm_steps) + 4 * (size_t)mm_group_rows * mm_d;
  return n;
}
// where the row blocks of the rows + flags form start (floats from the LDS base)
__host__ __device__ inline size_t pm_fast_rows_area_off(int R, int LD, int D, int U, int RT, const int* pnt, int pnl,
                                                        const int* dnt, int dnl, int mm_d, int prec, int mm_waves,
                                                        int mm_group_rows, int mm_steps) {
  return pm_fast_lds_floats(R, LD, D, U, RT, pnt, pnl, dnt, dnl, mm_d, prec, mm_waves, mm_group_rows, mm_steps) -
         4 * (size_t)mm_group_rows * mm_d;
}

// (pnt / dnt: 16-wide tile counts of layer inputs / outputs, nl + 1 entries each)
__device__ __forceinline__ FastLds pm_fast_carve(float* base, int R, int LD, int D, int U, int RT,
                                                 const int* pnt, int pnl, const int* dnt, int dnl, int prec = 0) {
  FastLds m;
  float* p = base;
  m.bufA = p; p += (size_t)R * LD;
  m.bufB = p; p += (size_t)R * LD;
  m.xa = p; p += (size_t)R * D;
  m.xb = p; p += (size_t)R * D;
  m.av = p; p += (size_t)R * U;
  m.gad = p; p += (size_t)R * 16;
  m.rr = p; p += R;
  m.gr = p; p += R;
  m.hp_off = -1;
  if (!pm_fast_hp_alias(R, LD, RT)) {
    m.hp_off = (int)(p - base);
    p += (size_t)PF_NW * RT * 256;
  }
  m.base = base;
  for (int l = 0; l < pnl; ++l) p += (size_t)pnt[l + 1] * 16;
  for (int l = 0; l < dnl; ++l) p += (size_t)dnt[l + 1] * 16;
  for (int l = 0; l < pnl - 1; ++l) p += ((size_t)R * pnt[l + 1] + 1) / 2;
  for (int l = 0; l < dnl - 1; ++l) p += ((size_t)R * dnt[l + 1] + 1) / 2;
  m.zp = p; p += (size_t)R * U;
  m.zd = p; p += (size_t)R * D;
  m.mx = p; p += D + U;
  m.iSx = p; p += D + U;
  m.my = p; p += D;
  m.Sy = p; p += D;
  m.lSy = p; p += D;
  m.psc = p; p += U;
  m.pbi = p; p += U;
  size_t n = ((size_t)(p - base) + 3) & ~(size_t)3;
  p = base + n;
  m.jx = p; p += (size_t)R * 16;
  m.stg = p; p += (size_t)R * (1 + 2 * D + 3 * U);
  m.zs = p; p += (size_t)R * D;
  n = ((size_t)(p - base) + 3) & ~(size_t)3;
  p = base + n;
  m.xin = p;
  m.ovf = reinterpret_cast<int*>(p + (size_t)R * PM_XIN_LD);
  if (prec) p += (size_t)R * PM_XIN_LD + 4;
  {
    const size_t tf = pm_fast_tail_floats(pnt, pnl, dnt, dnl, false, RT, prec), tb = pm_fast_tail_floats(pnt, pnl, dnt, dnl, true, RT, prec);
    const size_t tw = tf > tb ? tf : tb;
    m.tp = p;
    m.tcnt = reinterpret_cast<int*>(p + (size_t)PF_NW * RT * 256);
    m.tw = p + (size_t)PF_NW * RT * 256 + 4;
    if (tw) p += (size_t)PF_NW * RT * 256 + 4 + tw;
  }
  m.mm = reinterpret_cast<double*>(p);
  return m;
}
__device__ __forceinline__ FastLds pm_fast_carve(float* base, int R, int LD, int D, int U, int RT,
                                                 const NetDev& P, const NetDev& F, int prec = 0) {
  return pm_fast_carve(base, R, LD, D, U, RT, P.nt, P.nl, F.nt, F.nl, prec);
}
// shape-specialised kernels: the whole carve-up from compile-time constants (-> immediate
// LDS offsets); layer widths 1 | NT ... NT | 1 tiles
template <class SH, int RT, int PR = 0>
__device__ __forceinline__ FastLds pm_fast_carve_shaped(float* base, const NetDev& P, const NetDev& F,
                                                        int R, int LD, int D, int U) {
  if constexpr (SH::NT != 0 && SH::NL != 0) {
    int nt[PM_MAXL + 1];
#pragma unroll
    for (int l = 0; l <= PM_MAXL; ++l) nt[l] = (l == 0 || l >= SH::NL) ? 1 : SH::NT;
    return pm_fast_carve(base, R, LD, D, U, RT, nt, SH::NL, nt, SH::NL, PR);
  } else {
    return pm_fast_carve(base, R, LD, D, U, RT, P, F, PR);
  }
}

// per-layer LDS regions: offsets (in floats from the LDS base) live in the kernel
// arguments (FastOff, filled on the host with the same walk as pm_fast_carve) so that a
// runtime layer index becomes a scalar load, not a private-memory array access
#define PBIAS(l) (L.base + A.fo.pbias[l])
#define DBIAS(l) (L.base + A.fo.dbias[l])
#define PMASK(l) (reinterpret_cast<uint16_t*>(L.base + A.fo.pmask[l]))
#define DMASK(l) (reinterpret_cast<uint16_t*>(L.base + A.fo.dmask[l]))

__device__ inline void pm_fast_preload_tails(const StreamDesc& sd, const FastLds& L, int tid);
// stage launch-invariant data in LDS
template <int RT>
__device__ inline void pm_fast_preload(const RolloutArgs& A, const FastLds& L, int row0, int nvalid,
                                       int tid) {
  constexpr int R = 16 * RT;
  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  const int D = A.D, U = A.U;
  for (int l = 0; l < P.nl; ++l)
    for (int i = tid; i < P.nt[l + 1] * 16; i += PF_NT) PBIAS(l)[i] = P.bias[l][i];
  for (int l = 0; l < F.nl; ++l)
    for (int i = tid; i < F.nt[l + 1] * 16; i += PF_NT) DBIAS(l)[i] = F.bias[l][i];
  for (int l = 0; l < P.nl - 1; ++l) {
    const int nt = P.nt[l + 1];
    for (int i = tid; i < R * nt; i += PF_NT) {
      const int r = i / nt;
      PMASK(l)[i] = (r < nvalid) ? P.mask[l][(size_t)(row0 + r) * nt + (i - r * nt)] : (uint16_t)0;
    }
  }
  for (int l = 0; l < F.nl - 1; ++l) {
    const int nt = F.nt[l + 1];
    for (int i = tid; i < R * nt; i += PF_NT) {
      const int r = i / nt;
      DMASK(l)[i] = (r < nvalid) ? F.mask[l][(size_t)(row0 + r) * nt + (i - r * nt)] : (uint16_t)0;
    }
  }
  for (int i = tid; i < R * U; i += PF_NT)
    L.zp[i] = (i / U < nvalid && A.zpol_ss == 0) ? A.zpol[(size_t)row0 * U + i] : 0.f;
  for (int i = tid; i < R * D; i += PF_NT)
    L.zd[i] = (i / D < nvalid && A.zdyn_ss == 0) ? A.zdyn[(size_t)row0 * D + i] : 0.f;
  for (int i = tid; i < D + U; i += PF_NT) {
    L.mx[i] = A.mx[i];
    L.iSx[i] = A.iSx[i];
  }
  for (int i = tid; i < D; i += PF_NT) {
    L.my[i] = A.my[i];
    L.Sy[i] = A.Sy[i];
    L.lSy[i] = logf(A.Sy[i]);
  }
  for (int i = tid; i < U; i += PF_NT) {
    L.psc[i] = A.pscale[i];
    L.pbi[i] = A.pbias[i];
  }
}

// The same for a shape-specialised kernel (uniform hidden width: NT tiles, NL layers per net): every
// trip count is a compile-time constant, every load is unconditional (clamped index), so ALL the
// prologue's global loads are in flight together and are waited for once.  The general form above is
// a chain of ~40 dependent memory round trips (one per short loop and conditional load): 20 us of a
// sweep launch, paid once per iteration in the single-launch modes and once per STEP when a
// moment-matching group spans workgroups.
template <int RT, class SH, int PR = 0>
__device__ __forceinline__ void pm_fast_preload_shaped(const RolloutArgs& A, const FastLds& L, const StreamDesc& sd,
                                                       int row0, int nvalid, int tid) {
  constexpr int R = 16 * RT, NL = SH::NL, NT = SH::NT, D = SH::D, U = SH::U;
  constexpr int NS = 2 * (NL - 2);                                   // streamed layers
  constexpr int IT_B = (NT * 16 + PF_NT - 1) / PF_NT, IT_M = (R * NT + PF_NT - 1) / PF_NT;
  constexpr int IT_T = (NT * 64 + PF_NT - 1) / PF_NT, IT_Z = (R * D + PF_NT - 1) / PF_NT;
  static_assert(R * U <= PF_NT && D + U <= PF_NT, "one item per thread");
  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  const int nv1 = nvalid - 1;
  float pb[NL][IT_B], db[NL][IT_B], zd[IT_Z], zp, c0, c1, c2, c3, c4, c5;
  uint16_t pm[NL - 1][IT_M], dm[NL - 1][IT_M];
  f32x4 tw[NS][IT_T];
  // ---- loads (all unconditional)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int n = (l < NL - 1 ? NT : 1) * 16;
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int i = min(tid + it * PF_NT, n - 1);
      pb[l][it] = P.bias[l][i];
      db[l][it] = F.bias[l][i];
    }
  }
#pragma unroll
  for (int l = 0; l < NL - 1; ++l)
#pragma unroll
    for (int it = 0; it < IT_M; ++it) {
      const int i = min(tid + it * PF_NT, R * NT - 1);
      const int r = i / NT, c = i - r * NT;
      const size_t src = (size_t)(row0 + min(r, nv1)) * NT + c;
      pm[l][it] = P.mask[l][src];
      dm[l][it] = F.mask[l][src];
    }
  if constexpr (!PR) {
#pragma unroll
  for (int l = 0; l < NS; ++l)
#pragma unroll
    for (int it = 0; it < IT_T; ++it) {
      const int i = min(tid + it * PF_NT, NT * 64 - 1);
      // the tile after the streamed ones (clamped read when this layer has no K-split tile: not stored)
      tw[l][it] = ldg4(sd.wf[l] + (size_t)sd.n_ot[l] * sd.n_kb[l] * 256 + (size_t)i * 4 * (sd.ks[l] ? 1 : 0));
    }
  }
  {
    const int i = min(tid, R * U - 1);
    zp = A.zpol[(size_t)row0 * U + min(i, nvalid * U - 1)];
  }
#pragma unroll
  for (int it = 0; it < IT_Z; ++it) {
    const int i = min(tid + it * PF_NT, R * D - 1);
    zd[it] = A.zdyn[(size_t)row0 * D + min(i, nvalid * D - 1)];
  }
  {
    const int i = min(tid, D + U - 1), j = min(tid, D - 1), k = min(tid, U - 1);
    c0 = A.mx[i]; c1 = A.iSx[i]; c2 = A.my[j]; c3 = A.Sy[j]; c4 = A.pscale[k]; c5 = A.pbias[k];
  }
  // ---- stores
  if (!PR && tid == 0) *L.tcnt = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int n = (l < NL - 1 ? NT : 1) * 16;
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int i = tid + it * PF_NT;
      if (i < n) {
        PBIAS(l)[i] = pb[l][it];
        DBIAS(l)[i] = db[l][it];
      }
    }
  }
#pragma unroll
  for (int l = 0; l < NL - 1; ++l)
#pragma unroll
    for (int it = 0; it < IT_M; ++it) {
      const int i = tid + it * PF_NT;
      if (i < R * NT) {
        const bool live = i / NT < nvalid;
        PMASK(l)[i] = live ? pm[l][it] : (uint16_t)0;
        DMASK(l)[i] = live ? dm[l][it] : (uint16_t)0;
      }
    }
  if constexpr (!PR) {
#pragma unroll
  for (int l = 0; l < NS; ++l)
#pragma unroll
    for (int it = 0; it < IT_T; ++it) {
      const int i = tid + it * PF_NT;
      if (sd.ks[l] && i < NT * 64) *reinterpret_cast<f32x4*>(L.tw + sd.tw_off[l] + (size_t)i * 4) = tw[l][it];
    }
  }
  if (tid < R * U) L.zp[tid] = (tid / U < nvalid && A.zpol_ss == 0) ? zp : 0.f;
#pragma unroll
  for (int it = 0; it < IT_Z; ++it) {
    const int i = tid + it * PF_NT;
    if (i < R * D) L.zd[i] = (i / D < nvalid && A.zdyn_ss == 0) ? zd[it] : 0.f;
  }
  if (tid < D + U) { L.mx[tid] = c0; L.iSx[tid] = c1; }
  if (tid < D) { L.my[tid] = c2; L.Sy[tid] = c3; L.lSy[tid] = logf(c3); }
  if (tid < U) { L.psc[tid] = c4; L.pbi[tid] = c5; }
}

// partial head tiles: fixed region, or the idle activation buffer (Y)
#define PM_HP() (L.bufA + ((RT < 4 || L.hp_off >= 0) ? L.hp_off : (xsel ^ 1) * (R * LD)))   // aliasing: RT >= 4 only
// one streamed layer (index SI of the stream table) with epilogue ES and, when the layer's last
// tile is K-split, that tile's partial / gather / epilogue around it
#define PM_STREAM_LAYER(SI, ES, HW_UNUSED, PROF)                                                        \
  if constexpr (PR != 0) {                                                                              \
    const int rb_ = (RESK && (SI) == 0) ? res_tiles : (LDSK && (SI) == 1) ? PF_NW : 0;                  \
    if constexpr (RESK) {                                                                               \
      if ((SI) == 0 && rb_) resident_tile_s<RT, CA + CB, NP, F16>(wres, X, (unsigned)LDB, lane, (ES), wid); \
    }                                                                                                   \
    if constexpr (LDSK) {                                                                               \
      if ((SI) == 1) lds_tile_s<RT, CA + CB, NP, F16>(wlds, A.lds_last_lanes, X, (unsigned)LDB, lane, (ES), wid); \
    }                                                                                                   \
    stream_layer_s<RT, CA, CB, NP, F16, SC>(sd, (SI), q, fa, fb, X, (unsigned)LDB, wid, lane, (ES), vo0, vo1, (PROF), rb_); \
  } else {                                                                                              \
    const int ks_ = SKS_KNOWN ? 1 : (SC::NOT ? 0 : pm_rl(sd.k, 4 * (SI)));                                                            \
    typename std::remove_reference<decltype(ES)>::type::Pre tpre_[RT];                                  \
    const int ot_last_ = SD_NOT(SC, sd, (SI));                                                          \
    if (ks_) {                                                                                          \
      tail_partial<RT>(L.tw + pm_rl(sd.k, 4 * (SI) + 1), (SH::NT ? SH::NT : pm_rl(sd.k, 4 * (SI) + 2)), X, LD, L.tp, L.tcnt, \
                       wid, lane);                                                                      \
      ++tround;                                                                                         \
    }                                                                                                   \
    stream_layer<RT, CA, CB, SC>(sd, (SI), q, fa, fb, X, LD, wid, lane, (ES), vo0, vo1, (PROF));           \
    if (ks_ && wid == PF_NW - 1) {                                                                      \
      /* epilogue operands fetched here, in the block that waits for them (their latency hides behind */ \
      /* the wait for the partials)                                                                   */ \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) tpre_[rt] = (ES).pre(ot_last_, rt);            \
      f32x4 tsum_[RT];                                                                                  \
      tail_gather<RT>(L.tp, L.tcnt, tround, lane, tsum_);                                               \
      std::remove_reference<decltype(ES)>::type::wait_all();                                            \
      _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                              \
        (ES).landed(tpre_[rt], rt);                                                                     \
        (ES)(ot_last_, rt, tsum_[rt], tpre_[rt]);                                                       \
      }                                                                                                 \
    }                                                                                                   \
  }
#define PM_SWAP_XY() { xsel ^= 1; X = L.bufA + xsel * (R * LD); Y = L.bufA + (xsel ^ 1) * (R * LD); }
// fp32 input tile of a resident (K <= 16) layer: the activation buffer itself, or -- split precision, where
// the activation buffers hold bf16 piece planes -- the separate tile L.xin
#define PM_XIN(Xbuf) (PR ? L.xin : (Xbuf))
#define PM_XLD (PR ? PM_XIN_LD : LD)

// K-split tails: weight k-blocks of the last tile of every such layer -> LDS, counter -> 0
__device__ inline void pm_fast_preload_tails(const StreamDesc& sd, const FastLds& L, int tid) {
  if (tid == 0) *L.tcnt = 0;
  for (int i = 0; i < sd.n; ++i) {
    if (!sd.ks[i]) continue;
    const float* src = sd.wf[i] + (size_t)sd.n_ot[i] * sd.n_kb[i] * 256;   // the tile after the streamed ones
    float* dst = L.tw + sd.tw_off[i];
    for (int e = tid; e < sd.n_kb_real[i] * 256; e += PF_NT) dst[e] = src[e];
  }
}

// ===========================================================================
// forward (fast)
// ===========================================================================
// Kernel variants (template parameter VAR).  The sweep kernels are one long register- and
// SGPR-bound body: code that a launch never executes still costs every phase registers, so the
// rarely used paths are compiled only into the variants that need them.
//   LEAN  what mc_pilco's fused iteration and bench.py launch: the whole horizon in one launch, no
//         moment matching of states, frozen output noise, no external state / action gradients, no action-gradient
//         norms, no cycle stamps
//   EXT   + per-step output noise (resample_*_noise), grad_states / grad_actions inputs,
//         action_grad_norms output, cycle stamps (pmbrl_plan_set_prof)
//   MM    EXT + the in-kernel moment matching of states (mm_mode 1: whole groups per workgroup,
//         every step; mm_mode 3: groups spanning workgroups, in the prologue of per-step launches)
// Shape specialisation (template parameter SH): the state / action widths, the activation
// leading dimension and the layer count as compile-time constants (0 = read from the
// arguments).  The LDS carve-up, every row / column index and the layer loops then fold into
// immediates: 9 % on the C2 sweep.  Instantiated for the shapes of the shipped
// configurations (PM_FAST_SHAPED_CASES in pmbrl.hip); anything else runs the general kernels.
//   NT = 16-wide tiles of every hidden layer (both nets), which also fixes the weight stream
template <int D_, int U_, int LD_, int NL_, int NT_>
struct PfShape {
  static constexpr int D = D_, U = U_, LD = LD_, NL = NL_, NT = NT_;
  static constexpr bool TREE = false;
};
// ... of a launch whose moment-matching groups are split over MORE than PM_XCH_FLAT workgroups: the two-level sum
// exchange (pm_xch_get_tree) keeps 64 registers of granules in flight, which the instances that never use it should
// not pay for (inlined under a run-time test it cost the cart-pole instance 20 spilled registers on every path)
template <int D_, int U_, int LD_, int NL_, int NT_>
struct PfShapeTree : PfShape<D_, U_, LD_, NL_, NT_> {
  static constexpr bool TREE = true;
};
typedef PfShape<0, 0, 0, 0, 0> PfShapeAny;

// (PF_VAR_* are defined in pmbrl_dev.h)
#define PF_MARK(slot)                                                                              \
  do {                                                                                             \
    if (EXT && A.prof && wg == 0 && tid == 0)                                                      \
      A.prof[(size_t)t * 32 + (slot)] = (long long)__builtin_readcyclecounter();                   \
  } while (0)

// PR: precision of the hidden-width GEMMs -- 0: exact fp32 MFMA; 1: split bf16 (pmbrl_split.h; the forward
// sweep uses three pieces, the fp32-equivalent form).  CA / CB then count K32 blocks per stage.
template <int RT, int CA, int CB, int VAR, class SH = PfShapeAny, int PR = 0>
__global__ __launch_bounds__(PF_NT, 2) void pm_rollout_fwd_fast(const RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool MMG = VAR == PF_VAR_MMG, MM = VAR == PF_VAR_MM || MMG, EXT = VAR != PF_VAR_LEAN;
  // PR = 1: three bf16 pieces; PR = 2: two fp16 pieces (4 bytes per weight on the stream instead of 6)
  constexpr int NP = PR == 1 ? 3 : PR == 2 ? 2 : 0;
  constexpr bool F16 = PR == 2;
  constexpr int FA = PR ? CA * NP : CA, FB = PR ? CB * NP : CB;     // weight loads per stage
  // weight stream of a shape-specialised kernel: every streamed layer is NT x NT tiles
  constexpr bool SKS_KNOWN = SH::NT && pm_fast_ksplit(SH::NT, RT, PR);
  constexpr int NKBc = !SH::NT ? 0
                       : PR ? ((SH::NT + 1) / 2 + CA + CB - 1) / (CA + CB) * (CA + CB) * NP     // loads per tile
                            : (SH::NT + CA + CB - 1) / (CA + CB) * (CA + CB);
  // register-resident first tiles (resident_tile_s): compiled into the 16-row plain variants whose stage pair
  // is a whole tile; the tile counts of the stream then differ between layers (read from the table)
  constexpr bool RESK = PR != 0 && NP == 2 && RT == 1 && CA + CB == 7 && (VAR == PF_VAR_LEAN || VAR == PF_VAR_EXT);
  // ... and LDS-resident first tiles of the second streamed layer (lds_tile_s): two streamed layers, compile-time shape
  // (also where moment matching runs inside a 16-row sweep: no registers for resident tiles there, but the LDS is as idle)
  // (fp16-piece build only: the bf16-piece build's activation planes leave no room for eight tiles)
  constexpr bool LDSK = PR == 2 && NP == 2 && RT == 1 && CA + CB == 7 && SH::NT > 8 && SH::NL == 3 &&
                        (VAR == PF_VAR_LEAN || VAR == PF_VAR_EXT || VAR == PF_VAR_MM);
  typedef PfStream<(SH::NT ? SH::NT - (SKS_KNOWN ? 1 : 0) : 0) - (LDSK ? 8 : 0), NKBc,
                   (SH::NT && SH::NL) ? 2 * (SH::NL - 2) : 0,
                   (SH::NT ? SH::NT - (SKS_KNOWN ? 1 : 0) : 0) - ((RESK && SH::NT > 8) ? 8 : 0)> SC;
  constexpr int LDBc = (PR && SH::NT) ? NKBc / (PR ? NP : 1) * 32 + 16 : 0;
  const int LDB = PR ? (LDBc ? LDBc : A.LDB) : 0;     // leading dimension of the piece planes (bf16 elements)
  const int ELD = PR ? LDB : (SH::LD ? SH::LD : A.LD);   // ... of whatever the hidden-layer epilogues write
  // LEAN: whole horizon in one launch, no moment matching of states anywhere
  const int T0 = EXT ? A.t0 : 0, T1 = EXT ? A.t1 : A.H;
  if (EXT && A.prof && blockIdx.x == 0 && threadIdx.x == 0)      // kernel entry (per-step launches: prologue cost)
    A.prof[(size_t)T0 * 32 + 30] = (long long)__builtin_readcyclecounter();
  const bool mm_states = EXT && (A.flags & PMBRL_FLAG_MM_STATES);
  constexpr int R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x + A.wg0;
  const int rows_per_wg = MM ? A.rows_per_wg : 16 * RT;   // whole groups under MM, else full tiles
  // (a group split over mm_parts workgroups: part p owns rows [p * rows_per_wg, ...) of its group, the last
  //  part what is left of the group's M rows)
  const int mmp_n = (VAR == PF_VAR_MM && A.mm_mode == 1) ? A.mm_parts : 1;
  const int row0 = mmp_n > 1 ? (wg / mmp_n) * A.M + (wg % mmp_n) * rows_per_wg : wg * rows_per_wg;
  const int nvalid = mmp_n > 1 ? min(rows_per_wg, A.M - (wg % mmp_n) * rows_per_wg) : min(rows_per_wg, A.B - row0);
  const int D = SH::D ? SH::D : A.D, U = SH::U ? SH::U : A.U, LD = SH::LD ? SH::LD : A.LD, B = A.B;
  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  FastLds L = pm_fast_carve_shaped<SH, RT, PR>(smem, P, F, R, LD, D, U);
  float* xa = L.xa;
  float* xb = L.xb;

  // mm_mode 3 (a moment-matching group spans several workgroups; one launch per step): the
  // moment matching of the PREVIOUS step's sampled states is done here, by every workgroup for
  // its own rows -- the group statistics are recomputed per workgroup from all the group's rows
  // in HBM (the previous launch wrote them), 8 waves sharing the row sums.  No separate kernel
  // launch per step, and the work hides behind the rest of the prologue's memory traffic.
  bool x_ready = false;
  const bool mm_gs = MMG && mm_states;   // one launch, device-wide barrier per step
  auto mm_span_fwd = [&](int tp, auto coh) {
    constexpr bool COH = decltype(coh)::value;   // the rows come from other workgroups of this launch
    const float* zmm = pm_zbase(A.zmm, D, tp, A.Bg, A.flags);
    double* part = reinterpret_cast<double*>(L.bufA);          // the activation buffers are still free
    const int g_lo = row0 / A.M, g_hi = (row0 + nvalid - 1) / A.M;
    for (int gi = g_lo; gi <= g_hi; ++gi) {
      const int gr0 = gi * A.M;
      const float* sp = A.xt + ((size_t)tp * B + gr0) * D;
      const int zrow0 = pm_zrow0(tp, A.row_off + gr0, A.flags);
      const int m_lo = max(row0, gr0) - gr0, m_hi = min(row0 + nvalid, gr0 + A.M) - gr0;
      const int o_lo = wid == 0 ? m_lo : 0, o_hi = wid == 0 ? m_hi : 0;   // wave 0 writes this workgroup's rows
      bool ok = true;
#define PM_CALL(DD) ok = pm_mm_fwd_rows<DD, COH>(sp, D, A.M, zmm, D, zrow0, A.Bg, xa, D, o_lo, o_hi, row0 - gr0, L.mm, \
                                            part, PF_NW, wid, lane)
      switch (D) {
        case 4: PM_CALL(4); break;
        case 5: PM_CALL(5); break;
        case 6: PM_CALL(6); break;
        default: ok = false; break;      // (the host only selects mm_mode 3 for these widths)
      }
#undef PM_CALL
      if (!ok && tid == 0) atomicMin(A.status, tp);
      __syncthreads();
    }
    for (int i = nvalid * D + tid; i < R * D; i += PF_NT) xa[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < nvalid * D; i += PF_NT) A.states[((size_t)(tp + 1) * B + row0) * D + i] = xa[i];
  };
  if (VAR == PF_VAR_MM && A.mm_mode == 3 && mm_states && T0 > 0) {
    mm_span_fwd(T0 - 1, std::false_type{});
    x_ready = true;
  }

  if constexpr (SH::NL != 0 && SH::NT != 0) {
    pm_fast_preload_shaped<RT, SH, PR>(A, L, A.sd_fwd, row0, nvalid, tid);
  } else {
    pm_fast_preload<RT>(A, L, row0, nvalid, tid);
    if constexpr (!PR) pm_fast_preload_tails(A.sd_fwd, L, tid);
  }
  int tround = 0;   // K-split rounds so far (the arrival counter is monotonic)
  if (F16 && tid == 0) *L.ovf = 0;
  for (int i = tid; i < 2 * R * LD; i += PF_NT) L.bufA[i] = 0.f;   // bufA, bufB: zero K padding
  if (!x_ready) {
    const float* src = (T0 == 0) ? A.x0 : A.states + (size_t)T0 * B * D;
    for (int i = tid; i < R * D; i += PF_NT) {
      const int r = i / D, d = i - r * D;
      const float v = (r < nvalid) ? src[(size_t)(row0 + r) * D + d] : 0.f;
      xa[i] = v;
      if (T0 == 0 && r < nvalid) A.states[(size_t)(row0 + r) * D + d] = v;
    }
  }
  // register-resident first layers and head slices
  HeadW hwp, hwd;
  HeadWS<PR ? NP : 1> hsp, hsd;
  if constexpr (PR != 0) {
    head_load_s(hsp, P.wf[P.nl - 1], (P.nt[P.nl - 1] + 1) / 2, wid, lane);
    head_load_s(hsd, F.wf[F.nl - 1], (F.nt[F.nl - 1] + 1) / 2, wid, lane);
  } else {
    head_load(hwp, P.wf[P.nl - 1], P.nt[P.nl - 1], wid, lane);
    head_load(hwd, F.wf[F.nl - 1], F.nt[F.nl - 1], wid, lane);
  }
  Res0<RT> w0p, w0d;
  res0_load<RT>(w0p, P.wf[0], P.nt[1], wid, lane);
  res0_load<RT>(w0d, F.wf[0], F.nt[1], wid, lane);
  // weight stream over the hidden->hidden layers of both nets
  const SdV sd = sdv_make<SC>(A.sd_fwd, lane);   // policy hidden layers, then dynamics hidden layers
  const int n_pol_stream = P.nl - 2;
  Cursor q;
  cur_init<SC>(sd, q, wid);
  FragS<FA> fa;
  FragS<FB> fb;
  const unsigned vo0 = (unsigned)lane * 16u, vo1 = vo0 + 4096u;   // byte offsets of this lane in a chunk
  if (q.live) {
    frag_load<FA>(fa, q, vo0, vo1);
    cur_advance<FA, SC>(sd, q, wid);
  }
  // register-resident tile of this wave: output tile `wid` of the sweep's first streamed layer
  // (shape-specialised instantiations: always on, the stream's tile counts are compile-time constants)
  // fp16 pieces: a weight beyond +-65504 (or non-finite) was seen while packing this launch's weights
  if (F16 && tid == 0 && A.wflag && *A.wflag == A.wgen) atomicMin(A.status, A.t0);
  const int res_tiles = !RESK ? 0 : (SH::NT > 8 ? 8 : A.res_tiles);
  FragS<RESK ? (CA + CB) * NP : 1> wres;
  if constexpr (RESK) {
    if (res_tiles) {
      const float* wp = A.res_w + (size_t)wid * ((CA + CB) * NP) * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < (CA + CB) * NP; ++i) wres.a[i] = ldg4(wp + (size_t)i * 256);
    }
  }
  // this wave's tile of the second streamed layer -> LDS (read by this wave only: no barrier)
  const float* wlds = nullptr;
  if constexpr (LDSK) {
    constexpr int NLD = (CA + CB) * NP;
    float* dst = smem + A.wlds_off + (size_t)wid * pm_lds_tile_floats(NLD, NP, A.lds_last_lanes);
    const float* src = A.lds_w + (size_t)wid * NLD * 256;
    for (int i = 0; i < NLD - NP; ++i)
      *reinterpret_cast<f32x4*>(dst + (size_t)i * 256 + lane * 4) = ldg4(src + (size_t)i * 256 + lane * 4);
    if (lane < A.lds_last_lanes) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
        *reinterpret_cast<f32x4*>(dst + (size_t)(NLD - NP) * 256 + (size_t)pc * A.lds_last_lanes * 4 + lane * 4) =
            ldg4(src + (size_t)(NLD - NP + pc) * 256 + lane * 4);
    }
    wlds = dst;
  }
  const float max_std_pol = expf(A.mls_pol), max_std_dyn = expf(A.mls_dyn);
  const bool mm_in = VAR == PF_VAR_MM && A.mm_mode == 1 && mm_states;
  // ... with the group split over mm_parts workgroups: first workgroup / first row of the group, and the LDS
  // blocks [M][D] of the whole group's rows (sampled rows, noise rows, result) behind wave 0's scratch
  const bool mm_pair = mm_in && A.mm_parts > 1;
  // ... exchanging sums instead of rows (pm_xch_sum) where the one-wave routines of compile-time width apply
  // (the instances of compile-time width take no other form of a split group: the host launches the generic
  //  instance when the exchange is switched off)
  constexpr bool XW = SH::D >= 1 && SH::D <= 6;
  const bool mm_xch = mm_pair && XW && (A.xch != nullptr);
  const int mmp_first = mm_pair ? (wg / A.mm_parts) * A.mm_parts : 0;
  const int mmp_g0 = mm_pair ? (wg / A.mm_parts) * A.M : 0;
  // LDS behind wave 0's scratch: the statistics exchange's (mmx), then the row blocks of the rows + flags form (mmg)
  float* const mmx = reinterpret_cast<float*>(L.mm + pm_mm_scratch_doubles(D));
  float* const mmg = mmx + (mm_pair ? pm_fast_xch_floats(R, D, A.H) : 0);
  // (statistics exchange: the first reference point is the group's first row, which every part can read)
  if (mm_xch && tid < D) mmx[tid] = ((T0 == 0) ? A.x0 : A.states + (size_t)T0 * B * D)[(size_t)mmp_g0 * D + tid];
  // ... and the standardisation of the group's noise rows, mean and 1 / std per column, for every step of the launch
  // (the noise is an input: nothing of it waits for the recursion)
  double* const mmzt = reinterpret_cast<double*>(mmx + 8);      // [T1 - T0][zm (D) | zi (D)]
  // this part's noise rows, two sets [R][D] (step t in set t & 1): fetched a step ahead by wave 1 while wave 0 works
  float* const mmzs = reinterpret_cast<float*>(mmzt + (size_t)A.H * 2 * D);
  if (mm_xch) {
    {
      const float* zb = pm_zbase(A.zmm, D, T0, A.Bg, A.flags);
      const int z0 = pm_zrow0(T0, A.row_off + mmp_g0, A.flags);
      for (int e = tid; e < nvalid * D; e += PF_NT) {
        const int r = e / D, d = e - r * D;
        mmzs[(T0 & 1) * R * D + e] = zb[(size_t)pm_zidx(z0, row0 - mmp_g0 + r, A.Bg) * D + d];
      }
    }
    const double dM = (double)A.M, inv_m = 1.0 / dM, inv_m1 = 1.0 / (double)(A.M - 1);
    if (A.mm_ztab) {      // (worked out once for the launch by pm_mm_ztable_kernel: a large group)
      const double* zt = A.mm_ztab + ((size_t)T0 * A.mmfac_groups + mmp_g0 / A.M) * 2 * D;
      for (int e = tid; e < (T1 - T0) * 2 * D; e += PF_NT) mmzt[e] = zt[(size_t)(e / (2 * D)) * A.mmfac_groups * 2 * D + e % (2 * D)];
    } else
    for (int tz = T0 + wid; tz < T1; tz += PF_NW) {
      const float* zb = pm_zbase(A.zmm, D, tz, A.Bg, A.flags);
      const int z0 = pm_zrow0(tz, A.row_off + mmp_g0, A.flags);
      for (int j = 0; j < D; ++j) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = lane; r < A.M; r += 64) {
          const double zv = (double)zb[(size_t)pm_zidx(z0, r, A.Bg) * D + j];
          s1 += zv;
          s2 += zv * zv;
        }
        const double sm = pm_seg_sum(s1, 64), sq = pm_seg_sum(s2, 64);
        const double zm = sm * inv_m;
        if (lane == 0) {
          mmzt[(size_t)(tz - T0) * 2 * D + j] = zm;
          mmzt[(size_t)(tz - T0) * 2 * D + D + j] = pm_rsqrt((sq - dM * zm * zm) * inv_m1);
        }
      }
    }
  }
  // Phase pattern: everything a phase needs that does NOT depend on the previous phase's LDS
  // output (epilogue descriptors = scalar loads from the kernel arguments, epilogue operands)
  // is issued BEFORE the barrier that opens the phase, so those latencies overlap the barrier
  // wait.  The barrier that closes a step is the one that opens the next step's first phase.
  // net tables: lane 8*l + {0: nt[l+1], 1: bias offset, 2: mask offset, 3/4: abits, 5: 1/keep, 6/7: stash}
  int vdp = 0, vdd = 0;
  {
    // every table entry is loaded unconditionally (lane >> 3 < PM_MAXL keeps the indices in the
    // argument arrays) and selected afterwards: one round trip to the kernel arguments, not one per field
    const int l = lane >> 3, f = lane & 7, l1 = l + 1 < PM_MAXL ? l + 1 : PM_MAXL - 1;
    const int p_nt = P.nt[l + 1], p_b = A.fo.pbias[l], p_m = A.fo.pmask[l], p_ik = __float_as_int(P.inv_keep[l]);
    const int d_nt = F.nt[l + 1], d_b = A.fo.dbias[l], d_m = A.fo.dmask[l], d_ik = __float_as_int(F.inv_keep[l]);
    const uint64_t p_ab = (uint64_t)P.abits[l], d_ab = (uint64_t)F.abits[l], p_at = (uint64_t)A.actT[l1];
    const bool phid = l < P.nl - 1, dhid = l < F.nl - 1;
    const int p64 = f == 3 ? (int)(uint32_t)p_ab : f == 4 ? (int)(uint32_t)(p_ab >> 32)
                  : f == 6 ? (int)(uint32_t)p_at : (int)(uint32_t)(p_at >> 32);
    const int pv = f == 0 ? p_nt : f == 1 ? p_b : f == 5 ? p_ik : (phid ? (f == 2 ? p_m : p64) : 0);
    vdp = l < P.nl ? pv : 0;
    const int d64 = f == 3 ? (int)(uint32_t)d_ab : (int)(uint32_t)(d_ab >> 32);
    const int dv = f == 0 ? d_nt : f == 1 ? d_b : f == 5 ? d_ik : f >= 6 ? 0 : (dhid ? (f == 2 ? d_m : d64) : 0);
    vdd = l < F.nl ? dv : 0;
  }
  const auto ovf_init = [&] {
    if constexpr (F16) return L.ovf;
    else return PmEmpty{};
  }();
  auto pol_epi = [&](int l, int t, size_t blk, float* out) {
    const int nt = SH::NT ? SH::NT : pm_rl(vdp, 8 * l);
    return EpiFwdL<RT, NP, F16>{L.base + pm_rl(vdp, 8 * l + 1), reinterpret_cast<const uint16_t*>(L.base + pm_rl(vdp, 8 * l + 2)),
                           pm_rlp<uint8_t>(vdp, 8 * l + 3) + (size_t)t * B * nt * 4, pm_rlf(vdp, 8 * l + 5), out,
                           pm_rlp<float>(vdp, 8 * l + 6) + blk * (size_t)nt * 16 * R, ELD, R, row0, nvalid, nt, lane,
                           ovf_init};
  };
  auto dyn_epi = [&](int l, int t, float* out) {
    const int nt = SH::NT ? SH::NT : pm_rl(vdd, 8 * l);
    return EpiFwdL<RT, NP, F16>{L.base + pm_rl(vdd, 8 * l + 1), reinterpret_cast<const uint16_t*>(L.base + pm_rl(vdd, 8 * l + 2)),
                           pm_rlp<uint8_t>(vdd, 8 * l + 3) + (size_t)t * B * nt * 4, pm_rlf(vdd, 8 * l + 5), out,
                           nullptr, ELD, R, row0, nvalid, nt, lane, ovf_init};
  };
  const float* pol_head_bias = L.base + pm_rl(vdp, 8 * (P.nl - 1) + 1);
  const float* dyn_head_bias = L.base + pm_rl(vdd, 8 * (F.nl - 1) + 1);
  const int pol_head_kb = PR ? (P.nt[P.nl - 1] + 1) / 2 : P.nt[P.nl - 1];   // K blocks of the heads (K32 / K16)
  const int dyn_head_kb = PR ? (F.nt[F.nl - 1] + 1) / 2 : F.nt[F.nl - 1];
  const int pnl = SH::NL ? SH::NL : P.nl, fnl = SH::NL ? SH::NL : F.nl;

  bool fed = false;   // the previous step's sampling phase already wrote this step's policy input
  for (int t = T0; t < T1; ++t) {
    const size_t blk = (size_t)t * A.nwg + wg;
    int xsel = 0;           // X = bufA + xsel*R*LD (bufB directly follows bufA)
    float* X = L.bufA;
    float* Y = L.bufB;
    PF_MARK(0);
    if (!fed) {
      // dynamics-state rows -> policy input tile (+ dW stash); later steps of the plain path
      // get this written by the previous step's sampling phase
      __syncthreads();
      float* st = A.actT[0] + blk * (size_t)16 * (16 * RT);
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int k = i / R, r = i - k * R;
        const float v = (k < D) ? xa[r * D + k] : 0.f;
        PM_XIN(X)[r * PM_XLD + k] = v;
        st[(size_t)k * (16 * RT) + r] = v;
      }
      if (mm_in) {
        // this step's moment-matching noise rows -> LDS, a whole step before they are needed (the
        // statistics loops of pm_mm_* would otherwise chase them through HBM one at a time)
        const float* zb = pm_zbase(A.zmm, D, t, A.Bg, A.flags);
        if (mm_xch) {      // (staged a step ahead, next to the moment matching)
        } else if (mm_pair) {     // the whole group's noise rows
          const int z0 = pm_zrow0(t, A.row_off + mmp_g0, A.flags);
          for (int i = tid; i < A.M * D; i += PF_NT) {
            const int r = i / D, d = i - r * D;
            mmg[A.M * D + i] = zb[(size_t)pm_zidx(z0, r, A.Bg) * D + d];
          }
        } else {
        const int z0 = pm_zrow0(t, A.row_off + row0, A.flags);
        for (int i = tid; i < nvalid * D; i += PF_NT) {
          const int r = i / D, d = i - r * D;
          L.zs[i] = zb[(size_t)pm_zidx(z0, r, A.Bg) * D + d];
        }
        }
      }
    }
    // ---- policy: first layer (resident), hidden layers (streamed), head K-split over the waves
    EpiFwdL<RT, NP, F16> es{};
    {
      EpiFwdL<RT, NP, F16> e = pol_epi(0, t, blk, Y);
      Res0Pre<RT, EpiFwdL<RT, NP, F16>> pp;
      res0_prefetch<RT>(pp, e, e.nt, wid);
      __syncthreads();
      PF_MARK(1);
      res0_layer<RT>(w0p, e.nt, PM_XIN(X), PM_XLD, wid, lane, e, pp);
      if (pnl > 2) es = pol_epi(1, t, blk, X);     // layer 1 writes the buffer that is X now
    }
    __syncthreads();
    PM_SWAP_XY();
    PF_MARK(2);
    for (int l = 1; l < pnl - 1; ++l) {
      PM_STREAM_LAYER(l - 1, es, 0, (EXT && A.prof && wg == 0 && tid == 0) ? A.prof + (size_t)t * 32 : nullptr);
      if (l + 1 < pnl - 1) es = pol_epi(l + 1, t, blk, X);
      __syncthreads();
      PM_SWAP_XY();
      PF_MARK(2 + l);
    }
    if constexpr (PR != 0) head_partial_s<RT, NP, F16>(hsp, pol_head_kb, X, (unsigned)LDB, PM_HP(), wid, lane);
    else head_partial<RT>(hwp, pol_head_kb, X, LD, PM_HP(), wid, lane);
    __syncthreads();
    PF_MARK(10);
    // ---- squash + dynamics input
    {
      const float* hb = pol_head_bias;
      const float* hp = PM_HP();
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, k = i & 15;
        float v = 0.f;
        if (k < D) {
          v = (xa[r * D + k] - L.mx[k]) * L.iSx[k];
        } else if (k < D + U) {
          const int j = k - D;
          const float mu = hb[j] + head_value<RT>(hp, r, j);
          const float ls = hb[U + j] + head_value<RT>(hp, r, U + j);
          float z = L.zp[r * U + j];
          asm volatile("" : "+v"(z));   // keep the LDS load a load (no select of LDS / HBM addresses -> FLAT)
          if (EXT && A.zpol_ss != 0 && r < nvalid) z = A.zpol[(size_t)t * A.zpol_ss + (size_t)(row0 + r) * U + j];
          // lc = c - softplus(c - ls)  =>  e = exp(lc) = exp(c) sigmoid(ls - c),
          // d lc / d ls = sigmoid(c - ls) = 1 - sigmoid(ls - c): one exp instead of four
          const float sg = 1.f / (1.f + expf(A.mls_pol - ls));
          const float e = max_std_pol * sg;
          const float u = mu + z * e;
          const float a = L.psc[j] * tanhf(u) + L.pbi[j];
          L.av[r * U + j] = a;
          if (r < nvalid) {
            const size_t o = ((size_t)t * B + row0 + r) * U + j;
            A.actions[o] = a;
            A.Tp[o] = z * e * (1.f - sg);
          }
          v = (a - L.mx[k]) * L.iSx[k];
        }
        PM_XIN(X)[r * PM_XLD + k] = v;
      }
    }
    // ---- dynamics
    {
      EpiFwdL<RT, NP, F16> e = dyn_epi(0, t, Y);
      Res0Pre<RT, EpiFwdL<RT, NP, F16>> pp;
      res0_prefetch<RT>(pp, e, e.nt, wid);
      __syncthreads();
      PF_MARK(11);
      res0_layer<RT>(w0d, e.nt, PM_XIN(X), PM_XLD, wid, lane, e, pp);
      if (fnl > 2) es = dyn_epi(1, t, X);
    }
    __syncthreads();
    PM_SWAP_XY();
    PF_MARK(12);
    for (int l = 1; l < fnl - 1; ++l) {
      PM_STREAM_LAYER(n_pol_stream + l - 1, es, 0, nullptr);
      if (l + 1 < fnl - 1) es = dyn_epi(l + 1, t, X);
      __syncthreads();
      PM_SWAP_XY();
      PF_MARK(12 + l);
    }
    if constexpr (PR != 0) head_partial_s<RT, NP, F16>(hsd, dyn_head_kb, X, (unsigned)LDB, PM_HP(), wid, lane);
    else head_partial<RT>(hwd, dyn_head_kb, X, LD, PM_HP(), wid, lane);
    __syncthreads();
    PF_MARK(20);
    // ---- sample next state; on the plain path also the next step's policy input tile
    {
      const float* hb = dyn_head_bias;
      const float* hp = PM_HP();
      // (not when the partial tiles sit in bufA: the feed below writes there)
      const bool feed = !mm_in && !mm_gs && (t + 1 < T1) && !(RT >= 4 && L.hp_off < 0 && xsel == 1);
      fed = feed;
      float* stn = A.actT[0] + (blk + A.nwg) * (size_t)16 * (16 * RT);
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, d = i & 15;
        float xn = 0.f;
        if (d < D) {
          const int o_l = r * D + d;
          const float mu = hb[d] + head_value<RT>(hp, r, d);
          const float ls = hb[D + d] + head_value<RT>(hp, r, D + d);
          float z = L.zd[o_l];
          asm volatile("" : "+v"(z));
          if (EXT && A.zdyn_ss != 0 && r < nvalid) z = A.zdyn[(size_t)t * A.zdyn_ss + (size_t)(row0 + r) * D + d];
          const float sg = 1.f / (1.f + expf(A.mls_dyn - ls));
          const float e = max_std_dyn * L.Sy[d] * sg;
          xn = xa[o_l] + (mu * L.Sy[d] + L.my[d] + z * e);
          xb[o_l] = xn;
          if (r < nvalid) {
            const size_t o = ((size_t)t * B + row0 + r) * D + d;
            A.Td[o] = z * e * (1.f - sg);
            if (mm_gs || (mm_pair && !mm_xch)) pm_st_dev(A.xt + o, xn);   // read by other workgroups after the barrier
            else if (mm_states) A.xt[o] = xn;
            else A.states[o + (size_t)B * D] = xn;
          }
        }
        if (feed) {
          PM_XIN(L.bufA)[r * PM_XLD + d] = xn;     // free: the heads were read before the barrier
          stn[(size_t)d * (16 * RT) + r] = xn;
        }
      }
    }
    if (F16 && tid == 0 && *L.ovf) atomicMin(A.status, t);   // an activation left fp16's range in this step (or earlier)
    PF_MARK(21);
    // The reward is NOT evaluated here: r~[t,b] depends only on the stored (x~, a), never feeds
    // the state recursion, and is computed for all (t, b) at once by pm_reward_all_kernel
    // after the sweep (so is the moment matching of rewards).  Only the moment matching of
    // STATES is part of the recursion.
    if (mm_xch) {
      // group split over workgroups, compile-time width: wave 0 sums the Gram tile over ITS rows (the other parts'
      // places hold the reference point: they add nothing), adds the other parts' sums (pm_xch_sum) and goes on as
      // if it had seen the whole group -- no row exchange, no flag barrier, the same bits in every part
      constexpr int DDc = (SH::D >= 1 && SH::D <= 6) ? SH::D : 1;
      float* const mmc = mmx;      // the reference point
      __syncthreads();                            // this step's sampled rows (xb) are complete
      PF_MARK(28);
      if (wid == 0) {
        const int me = wg - mmp_first;
        const unsigned k = (unsigned)(t - T0 + 1);
        const double refl = (double)mmc[(lane & 15) < D ? (lane & 15) : 0];
        // sums over this part's rows, relative to the reference point every part uses
        pm_f64x4 G = pm_mm_gram_lds<DDc, 1, 4 * RT>(xb, xb, nvalid, lane, refl);
        double v[2] = {G[0], G[1]};
        PF_MARK(29);
        pm_xch_put<2>(A.xch, mmp_first, me, k, v, lane);
        // while they travel: the z standardisation of the whole group (prologue)
        MMW<DDc> q;
        {
          const double* zt = mmzt + (size_t)(t - T0) * 2 * D;
#pragma unroll
          for (int j = 0; j < DDc; ++j) {
            q.zm[j] = zt[j];
            q.zi[j] = zt[DDc + j];
          }
        }
        bool xok;
        if constexpr (SH::TREE) xok = pm_xch_get_tree<2>(A.xch, A.nwg, mmp_first, A.mm_parts, A.mm_fan, me, k, v, lane);
        else xok = pm_xch_get<2>(A.xch, mmp_first, A.mm_parts, me, k, v, lane);
        PF_MARK(31);
        G[0] = v[0];
        G[1] = v[1];
        const bool ok = pm_mmw_factor<DDc, false>(G, A.M, q) && xok;
        if (!ok && lane == 0) atomicMin(A.status, t);
        double mean[DDc];
#pragma unroll
        for (int j = 0; j < DDc; ++j) mean[j] = q.mean[j] + pm_rl64(refl, j);
        // this part's rows of the result; the next step's reference point; the factor for the adjoint sweep
        if (lane < nvalid) {
          const float* zr = mmzs + (t & 1) * R * D + lane * D;
          double zh[DDc];
#pragma unroll
          for (int c = 0; c < DDc; ++c) zh[c] = ((double)zr[c] - q.zm[c]) * q.zi[c];
          float* so = A.states + ((size_t)(t + 1) * B + row0 + lane) * D;
#pragma unroll
          for (int j = 0; j < DDc; ++j) {
            double acc = mean[j];
#pragma unroll
            for (int c = 0; c <= j; ++c) acc += zh[c] * q.L[j][c];
            xa[lane * D + j] = (float)acc;
            so[j] = (float)acc;
          }
        }
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < DDc; ++j) mmc[j] = (float)mean[j];
          if (me == 0) {
            // [mean | zmean | zistd | (mbar) | invd | L row-major, lower triangle]
            double* fac = A.mmfac + ((size_t)t * A.mmfac_groups + mmp_g0 / A.M) * pm_mm_fac_doubles(D);
#pragma unroll
            for (int j = 0; j < DDc; ++j) {
              fac[j] = mean[j];
              fac[DDc + j] = q.zm[j];
              fac[2 * DDc + j] = q.zi[j];
              fac[4 * DDc + j] = q.invd[j];
#pragma unroll
              for (int c = 0; c <= j; ++c) fac[5 * DDc + j * DDc + c] = q.L[j][c];
            }
          }
        }
      }
      else if (wid == 1 && t + 1 < T1) {
        // (idle otherwise) the next step's noise rows of this part
        const float* zb = pm_zbase(A.zmm, D, t + 1, A.Bg, A.flags);
        const int z0 = pm_zrow0(t + 1, A.row_off + mmp_g0, A.flags);
        for (int e = lane; e < nvalid * D; e += 64) {
          const int r = e / D, d = e - r * D;
          mmzs[((t + 1) & 1) * R * D + e] = zb[(size_t)pm_zidx(z0, row0 - mmp_g0 + r, A.Bg) * D + d];
        }
      }
      PF_MARK(30);
      // (no barrier: the next step's first phase opens with one)
    } else if (!XW && mm_pair) {
      // every workgroup of the group has its sampled rows of this step in A.xt once the group has met; each of
      // them then matches the moments of the WHOLE group (redundantly: the d x d chain is serial anyway) and
      // keeps its own rows
      if (!pm_group_sync(A.gsync, wg, mmp_first, A.mm_parts, (unsigned)(t - T0 + 1)) && tid == 0) atomicMin(A.status, t);
      PF_MARK(28);
      const float* xg = A.xt + ((size_t)t * B + mmp_g0) * D;
      for (int i = tid; i < A.M * D; i += PF_NT) mmg[i] = pm_ldc<true>(xg + i);
      __syncthreads();
      PF_MARK(29);
      if (wid == 0) {
        bool ok = false;
        double* fac = wg == mmp_first ? A.mmfac + ((size_t)t * A.mmfac_groups + mmp_g0 / A.M) * pm_mm_fac_doubles(D) : nullptr;
        // (the compile-time-width routines keep one row per lane: groups of up to 64 rows)
        if (SH::D >= 1 && SH::D <= 6 && A.M <= 64)
          ok = pm_mm_fwd_w<(SH::D >= 1 && SH::D <= 6) ? SH::D : 1>(mmg, D, A.M, mmg + A.M * D, D, mmg + 2 * A.M * D, D, lane, fac);
        else
          ok = pm_mm_fwd(mmg, D, A.M, D, mmg + A.M * D, D, 0, 0, false, mmg + 2 * A.M * D, D, L.mm, lane, fac);
        if (!ok && lane == 0) atomicMin(A.status, t);
      }
      __syncthreads();
      PF_MARK(30);
      const float* res = mmg + 2 * A.M * D + (row0 - mmp_g0) * D;
      for (int i = tid; i < nvalid * D; i += PF_NT) {
        xa[i] = res[i];
        A.states[((size_t)(t + 1) * B + row0) * D + i] = res[i];
      }
    } else if (mm_in) {
      __syncthreads();
      const int gpw = rows_per_wg / A.M;
      for (int gi = wid; gi < gpw; gi += PF_NW) {
        const int lr0 = gi * A.M;
        if (lr0 >= nvalid) break;
        double* scr = L.mm + (size_t)wid * pm_mm_scratch_doubles(D);
        bool ok;
        if constexpr (SH::D >= 1 && SH::D <= 6) {
          // compile-time width: Gram tile on the fp64 matrix core, d x d algebra in registers (pmbrl_mm_w.h)
          ok = pm_mm_fwd_w<SH::D ? SH::D : 1>(xb + lr0 * D, D, A.M, L.zs + lr0 * D, D, xa + lr0 * D, D, lane,
                                              A.mmfac + ((size_t)t * A.mmfac_groups + (row0 + lr0) / A.M) * pm_mm_fac_doubles(D));
        } else {
          ok = pm_mm_fwd(xb + lr0 * D, D, A.M, D, L.zs + lr0 * D, D, 0, 0, false,
                         xa + lr0 * D, D, scr, lane,
                         A.mmfac + ((size_t)t * A.mmfac_groups + (row0 + lr0) / A.M) * pm_mm_fac_doubles(D));
        }
        if (!ok && lane == 0) atomicMin(A.status, t);
      }
      __syncthreads();
      for (int i = tid; i < nvalid * D; i += PF_NT)
        A.states[((size_t)(t + 1) * B + row0) * D + i] = xa[i];
    } else if (mm_gs) {
      // every workgroup's sampled rows of this step are in A.xt once all have passed the barrier
      if (!pm_grid_barrier(A.gsync, (unsigned)(t - T0 + 1)) && tid == 0) atomicMin(A.status, t);
      PF_MARK(28);
      mm_span_fwd(t, std::true_type{});
      PF_MARK(29);
      // the row sums went through the activation buffers: restore their zero K padding
      for (int i = tid; i < 2 * R * LD; i += PF_NT) L.bufA[i] = 0.f;
    } else {
      float* tmp = xa; xa = xb; xb = tmp;
    }
    PF_MARK(22);
    // (no closing barrier: the next step's first phase opens with one)
  }
}

// ===========================================================================
// backward sweep (fast)
// ===========================================================================
// PR = 1: split bf16 with TWO pieces -- the adjoint is linear in the incoming gradient, its rounding does
// not feed back into the trajectory (tools/split_precision_study.py: 4e-6 ... 7e-6 on the policy gradient)
template <int RT, int CA, int CB, int VAR, class SH = PfShapeAny, int PR = 0>
__global__ __launch_bounds__(PF_NT, 2) void pm_rollout_bwd_fast(const RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool MMG = VAR == PF_VAR_MMG, MM = VAR == PF_VAR_MM || MMG, EXT = VAR != PF_VAR_LEAN;
  constexpr int NP = PR ? 2 : 0;
  constexpr bool F16 = false;     // gradients of any magnitude: bf16 pieces
  constexpr int FA = PR ? CA * NP : CA, FB = PR ? CB * NP : CB;     // weight loads per stage
  // weight stream of a shape-specialised kernel: every streamed layer is NT x NT tiles
  constexpr bool SKS_KNOWN = SH::NT && pm_fast_ksplit(SH::NT, RT, PR);
  constexpr int NKB32c = ((SH::NT + 1) / 2 + CA + CB - 1) / (CA + CB) * (CA + CB);   // padded K32 blocks per tile
  constexpr int NKBc = !SH::NT ? 0 : PR ? NKB32c * NP : (SH::NT + CA + CB - 1) / (CA + CB) * (CA + CB);
  constexpr bool RESK = PR != 0 && NP == 2 && RT == 1 && CA + CB == 7 && (VAR == PF_VAR_LEAN || VAR == PF_VAR_EXT);
  // ... and LDS-resident first tiles of the second streamed layer (lds_tile_s): two streamed layers, compile-time shape
  // (also where moment matching runs inside a 16-row sweep: no registers for resident tiles there, but the LDS is as idle)
  // (fp16-piece build only: the bf16-piece build's activation planes leave no room for eight tiles)
  constexpr bool LDSK = PR == 2 && NP == 2 && RT == 1 && CA + CB == 7 && SH::NT > 8 && SH::NL == 3 &&
                        (VAR == PF_VAR_LEAN || VAR == PF_VAR_EXT || VAR == PF_VAR_MM);
  typedef PfStream<(SH::NT ? SH::NT - (SKS_KNOWN ? 1 : 0) : 0) - (LDSK ? 8 : 0), NKBc,
                   (SH::NT && SH::NL) ? 2 * (SH::NL - 2) : 0,
                   (SH::NT ? SH::NT - (SKS_KNOWN ? 1 : 0) : 0) - ((RESK && SH::NT > 8) ? 8 : 0)> SC;
  constexpr int LDBc = (PR && SH::NT) ? NKB32c * 32 + 16 : 0;
  const int LDB = PR ? (LDBc ? LDBc : A.LDB) : 0;     // leading dimension of the piece planes (bf16 elements)
  const int ELD = PR ? LDB : (SH::LD ? SH::LD : A.LD);
  // LEAN: whole horizon in one launch, no moment matching of states anywhere
  // Truncated horizon (utils/rollout.py:154-157): the sweep covers only the steps the forward sweep
  // completed, read from its status word on the device (no host round trip in between).
  const int T0 = EXT ? A.t0 : 0;
  int T1 = EXT ? A.t1 : A.H;
  if (A.nvalid) T1 = min(T1, __builtin_amdgcn_readfirstlane(*A.nvalid));
  if (EXT && A.prof && blockIdx.x == 0 && threadIdx.x == 0)      // kernel entry (per-step launches: prologue cost)
    A.prof[(size_t)T0 * 32 + 30] = (long long)__builtin_readcyclecounter();
  const bool mm_states = EXT && (A.flags & PMBRL_FLAG_MM_STATES);
  constexpr int R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x + A.wg0;
  const int rows_per_wg = MM ? A.rows_per_wg : 16 * RT;   // whole groups under MM, else full tiles
  // (a group split over mm_parts workgroups: part p owns rows [p * rows_per_wg, ...) of its group, the last
  //  part what is left of the group's M rows)
  const int mmp_n = (VAR == PF_VAR_MM && A.mm_mode == 1) ? A.mm_parts : 1;
  const int row0 = mmp_n > 1 ? (wg / mmp_n) * A.M + (wg % mmp_n) * rows_per_wg : wg * rows_per_wg;
  const int nvalid = mmp_n > 1 ? min(rows_per_wg, A.M - (wg % mmp_n) * rows_per_wg) : min(rows_per_wg, A.B - row0);
  const int D = SH::D ? SH::D : A.D, U = SH::U ? SH::U : A.U, LD = SH::LD ? SH::LD : A.LD, B = A.B;
  const NetDev& P = A.pol;
  const NetDev& F = A.dyn;
  if (T1 <= T0) {
    // nothing to sweep (a step of a per-step launch sequence past the valid horizon, or no valid step
    // at all): hand the incoming gradient on unchanged
    for (int i = tid; i < nvalid * D; i += PF_NT) {
      const size_t o = (size_t)row0 * D + i;
      float v = 0.f;
      if (EXT && A.gx_from_carry) v = A.gx_carry[o];
      else if (EXT && A.grad_states) v = A.grad_states[(size_t)max(T1, 0) * B * D + o];   // (T1: the truncated horizon's end)
      if (EXT && A.gx_carry && !(MMG && mm_states)) (A.gx_carry_out ? A.gx_carry_out : A.gx_carry)[o] = v;
      if (T0 == 0 && A.grad_x0) A.grad_x0[o] = v;
    }
    return;
  }
  FastLds L = pm_fast_carve_shaped<SH, RT, PR>(smem, P, F, R, LD, D, U);
  float* gx = L.xa;     // dL/dx_{t+1} on entry of a step, dL/dx_t on exit
  float* gxt = L.xb;    // moment-matching adjoint of gx (in-kernel mm only)
  float* gxn = L.jx;    // dL/dx~ incl. the reward term, then + dynamics-input term
  const bool mms = mm_states;

  // mm_mode 3: adjoint of the moment matching that produced x_{t+1} (t = this launch's step), per
  // workgroup for its own rows, from the whole group's carried gradient -- see the forward kernel
  bool g_ready = false;
  const bool mm_gs = MMG && mms;   // one launch, device-wide barrier per step
  auto mm_span_bwd = [&](int tp, const float* carry, auto coh) {
    constexpr bool COH = decltype(coh)::value;   // the carried gradient comes from other workgroups of this launch
    const float* zmm = pm_zbase(A.zmm, D, tp, A.Bg, A.flags);
    double* part = reinterpret_cast<double*>(L.bufA);
    const int g_lo = row0 / A.M, g_hi = (row0 + nvalid - 1) / A.M;
    for (int gi = g_lo; gi <= g_hi; ++gi) {
      const int gr0 = gi * A.M;
      const float* sp = A.xt + ((size_t)tp * B + gr0) * D;
      const float* gin = carry + (size_t)gr0 * D;
      const int zrow0 = pm_zrow0(tp, A.row_off + gr0, A.flags);
      const int m_lo = max(row0, gr0) - gr0, m_hi = min(row0 + nvalid, gr0 + A.M) - gr0;
      const int o_lo = wid == 0 ? m_lo : 0, o_hi = wid == 0 ? m_hi : 0;
#define PM_CALL(DD) pm_mm_bwd_rows<DD, COH>(sp, D, A.M, zmm, D, zrow0, A.Bg, gin, D, gx, D, o_lo, o_hi, row0 - gr0, L.mm, \
                                       part, PF_NW, wid, lane)
      switch (D) {
        case 4: PM_CALL(4); break;
        case 5: PM_CALL(5); break;
        case 6: PM_CALL(6); break;
        default: break;
      }
#undef PM_CALL
      __syncthreads();
    }
    for (int i = nvalid * D + tid; i < R * D; i += PF_NT) gx[i] = 0.f;
    __syncthreads();
  };
  if (VAR == PF_VAR_MM && A.mm_mode == 3 && mms) {
    mm_span_bwd(T0, A.gx_carry, std::false_type{});
    g_ready = true;
  }

  if constexpr (SH::NL != 0 && SH::NT != 0) {
    pm_fast_preload_shaped<RT, SH, PR>(A, L, A.sd_bwd, row0, nvalid, tid);
  } else {
    pm_fast_preload<RT>(A, L, row0, nvalid, tid);
    if constexpr (!PR) pm_fast_preload_tails(A.sd_bwd, L, tid);
  }
  int tround = 0;   // K-split rounds so far (the arrival counter is monotonic)
  for (int i = tid; i < 2 * R * LD; i += PF_NT) L.bufA[i] = 0.f;   // bufA, bufB: zero K padding
  if (!g_ready)
  for (int i = tid; i < R * D; i += PF_NT) {
    const int r = i / D, d = i - r * D;
    float v = 0.f;
    if (r < nvalid) {
      if (EXT && A.gx_from_carry) v = A.gx_carry[(size_t)(row0 + r) * D + d];
      else if (EXT && A.grad_states) v = A.grad_states[((size_t)T1 * B + row0 + r) * D + d];
    }
    gx[i] = v;
  }
  // resident tail slices (grad wrt the nets' inputs) and head-adjoint weights
  HeadW twd, twp;
  HeadWS<PR ? NP : 1> tsd, tsp;
  if constexpr (PR != 0) {
    head_load_s(tsd, F.wb[0], (F.nt[1] + 1) / 2, wid, lane);
    head_load_s(tsp, P.wb[0], (P.nt[1] + 1) / 2, wid, lane);
  } else {
    head_load(twd, F.wb[0], F.nt[1], wid, lane);
    head_load(twp, P.wb[0], P.nt[1], wid, lane);
  }
  Res0<RT> whd, whp;
  res0_load<RT>(whd, F.wb[F.nl - 1], F.nt[F.nl - 1], wid, lane);
  res0_load<RT>(whp, P.wb[P.nl - 1], P.nt[P.nl - 1], wid, lane);
  // stream: dynamics hidden layers (reverse), then policy hidden layers (reverse)
  const SdV sd = sdv_make<SC>(A.sd_bwd, lane);   // dynamics hidden layers (reverse), then policy (reverse)
  const int n_dyn_stream = F.nl - 2;
  Cursor q;
  cur_init<SC>(sd, q, wid);
  FragS<FA> fa;
  FragS<FB> fb;
  const unsigned vo0 = (unsigned)lane * 16u, vo1 = vo0 + 4096u;   // byte offsets of this lane in a chunk
  if (q.live) {
    frag_load<FA>(fa, q, vo0, vo1);
    cur_advance<FA, SC>(sd, q, wid);
  }
  // register-resident tile of this wave: output tile `wid` of the sweep's first streamed layer
  // (shape-specialised instantiations: always on, the stream's tile counts are compile-time constants)
  const int res_tiles = !RESK ? 0 : (SH::NT > 8 ? 8 : A.res_tiles);
  FragS<RESK ? (CA + CB) * NP : 1> wres;
  if constexpr (RESK) {
    if (res_tiles) {
      const float* wp = A.res_w + (size_t)wid * ((CA + CB) * NP) * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < (CA + CB) * NP; ++i) wres.a[i] = ldg4(wp + (size_t)i * 256);
    }
  }
  // this wave's tile of the second streamed layer -> LDS (read by this wave only: no barrier)
  const float* wlds = nullptr;
  if constexpr (LDSK) {
    constexpr int NLD = (CA + CB) * NP;
    float* dst = smem + A.wlds_off + (size_t)wid * pm_lds_tile_floats(NLD, NP, A.lds_last_lanes);
    const float* src = A.lds_w + (size_t)wid * NLD * 256;
    for (int i = 0; i < NLD - NP; ++i)
      *reinterpret_cast<f32x4*>(dst + (size_t)i * 256 + lane * 4) = ldg4(src + (size_t)i * 256 + lane * 4);
    if (lane < A.lds_last_lanes) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
        *reinterpret_cast<f32x4*>(dst + (size_t)(NLD - NP) * 256 + (size_t)pc * A.lds_last_lanes * 4 + lane * 4) =
            ldg4(src + (size_t)(NLD - NP + pc) * 256 + lane * 4);
    }
    wlds = dst;
  }
  __syncthreads();

  // Per-row inputs of one step, staged in LDS: [gr | Jx(D) | Ja(U) | Td(D) | Tp(U) | a(U)].
  // The values of step t-1 are fetched into registers at the START of step t and parked in
  // LDS in the middle of it (after phase B, the last reader of step t's values), so their HBM
  // latency is covered by half a step of GEMM work.  Each thread's source pointer / step
  // stride is fixed for the launch.
  const int S = 1 + 2 * D + 3 * U;
  constexpr int PFV = (R * (1 + 2 * 16 + 3 * 8) + PF_NT - 1) / PF_NT;   // staged values per thread (bound)
  const float* pf_base[PFV];
  unsigned pf_str[PFV];
#pragma unroll
  for (int u = 0; u < PFV; ++u) {
    const int idx = tid + u * PF_NT;
    const int r = idx / S, c = idx - r * S;
    pf_base[u] = nullptr;
    pf_str[u] = 0;
    if (idx < R * S && r < nvalid) {
      const size_t row = (size_t)row0 + r;
      if (c == 0) { pf_base[u] = A.grad_rewards + row; pf_str[u] = B; }
      else if (c < 1 + D) { pf_base[u] = A.Jx + row * D + (c - 1); pf_str[u] = B * D; }
      else if (c < 1 + D + U) { pf_base[u] = A.Ja + row * U + (c - 1 - D); pf_str[u] = B * U; }
      else if (c < 1 + 2 * D + U) { pf_base[u] = A.Td + row * D + (c - 1 - D - U); pf_str[u] = B * D; }
      else if (c < 1 + 2 * D + 2 * U) { pf_base[u] = A.Tp + row * U + (c - 1 - 2 * D - U); pf_str[u] = B * U; }
      else { pf_base[u] = A.actions + row * U + (c - 1 - 2 * D - 2 * U); pf_str[u] = B * U; }
    }
  }
#pragma unroll
  for (int u = 0; u < PFV; ++u) {
    const int idx = tid + u * PF_NT;
    if (idx < R * S) L.stg[idx] = pf_base[u] ? pf_base[u][(size_t)(T1 - 1) * pf_str[u]] : 0.f;
  }
  __syncthreads();

  // Phase A of a step, element (r, k) of the 16-wide dynamics-head adjoint input:
  //   g = g0 + gr~ Jx ;  k < D: gxn = g, X = g*Sy ;  D <= k < 2D: X = g*Td ;  gad = gr~ Ja
  // g0 = dL/dx_{t+1} (after the moment-matching adjoint if that runs in the kernel).
  const float* stg = L.stg;
  auto phase_a = [&](int r, int k, float g0, float* gxn_dst) {
    float v = 0.f;
    if (r < nvalid && k < 2 * D) {
      const int d = k < D ? k : k - D;
      const float grv = stg[r * S];   // dL/dr~ (already through the reward mm adjoint)
      const float g = g0 + grv * stg[r * S + 1 + d];
      if (k < D) {
        gxn_dst[r * D + d] = g;
        v = g * L.Sy[d];
      } else {
        v = g * stg[r * S + 1 + D + U + d];
      }
    }
    PM_XIN(L.bufA)[r * PM_XLD + k] = v;
    if (k < U) L.gad[r * 16 + k] = (r < nvalid) ? stg[r * S] * stg[r * S + 1 + D + k] : 0.f;
  };
  const bool mm_in = VAR == PF_VAR_MM && (A.mm_mode == 1 && mms);
  // group split over mm_parts workgroups (see the forward sweep): LDS blocks [M][D] of the whole group's
  // pre-mm rows, noise rows, incoming gradient and result behind wave 0's scratch
  const bool mm_pair = mm_in && A.mm_parts > 1;
  // ... exchanging sums instead of rows (pm_xch_sum) where the one-wave routines of compile-time width apply
  // (the instances of compile-time width take no other form of a split group: the host launches the generic
  //  instance when the exchange is switched off)
  constexpr bool XW = SH::D >= 1 && SH::D <= 6;
  const bool mm_xch = mm_pair && XW && (A.xch != nullptr);
  const int mmp_first = mm_pair ? (wg / A.mm_parts) * A.mm_parts : 0;
  const int mmp_g0 = mm_pair ? (wg / A.mm_parts) * A.M : 0;
  float* const mmx = reinterpret_cast<float*>(L.mm + pm_mm_scratch_doubles(D));      // (layout: see the forward sweep)
  float* const mmg = mmx + (mm_pair ? pm_fast_xch_floats(R, D, A.H) : 0);
  // statistics exchange: two staging sets (step t in set t & 1) of [scratch with the forward sweep's factor | this
  // part's pre-mm rows [R][D] | its noise rows [R][D]], fetched a step ahead by wave 1 while wave 0 works
  const size_t mmst_set = 2 * pm_mm_scratch_doubles(D) + 2 * (size_t)R * D;      // floats
  float* const mmst = mmx + 8 + 2 * (size_t)A.H * 2 * D;
  auto mm_stage = [&](int ts, int i0, int istep) {
    float* set = mmst + (size_t)(ts & 1) * mmst_set;
    double* fdst = reinterpret_cast<double*>(set);
    const double* fac = A.mmfac + ((size_t)ts * A.mmfac_groups + mmp_g0 / A.M) * pm_mm_fac_doubles(D);
    for (int e = i0; e < (int)pm_mm_fac_doubles(D); e += istep) fdst[e] = fac[e];
    float* xdst = set + 2 * pm_mm_scratch_doubles(D);
    float* zdst = xdst + R * D;
    const float* xsrc = A.xt + ((size_t)ts * B + row0) * D;
    const float* zb = pm_zbase(A.zmm, D, ts, A.Bg, A.flags);
    const int z0 = pm_zrow0(ts, A.row_off + mmp_g0, A.flags);
    for (int e = i0; e < nvalid * D; e += istep) {
      const int r = e / D, d = e - r * D;
      xdst[e] = xsrc[e];
      zdst[e] = zb[(size_t)pm_zidx(z0, row0 - mmp_g0 + r, A.Bg) * D + d];
    }
  };
  if (mm_xch && T1 > T0) mm_stage(T1 - 1, tid, PF_NT);
  int gsel = 0;   // plain path: gxn alternates between L.jx and L.xb
  const int xb_off = (int)(L.xb - L.jx);

  // net tables: lane 8*idx + {0: nt[idx+1], 1/2: abits, 3: 1/keep, 4/5: G stash} of hidden layer idx
  int vdp = 0, vdd = 0;
  {
    // loaded unconditionally, selected afterwards (one round trip to the kernel arguments)
    const int l = lane >> 3, f = lane & 7;
    const int p_nt = P.nt[l + 1], p_ik = __float_as_int(P.inv_keep[l]);
    const int d_nt = F.nt[l + 1], d_ik = __float_as_int(F.inv_keep[l]);
    const uint64_t p_ab = (uint64_t)P.abits[l], d_ab = (uint64_t)F.abits[l], p_g = (uint64_t)A.gT[l];
    const int pv = f == 0 ? p_nt : f == 1 ? (int)(uint32_t)p_ab : f == 2 ? (int)(uint32_t)(p_ab >> 32) : f == 3 ? p_ik
                 : f == 4 ? (int)(uint32_t)p_g : f == 5 ? (int)(uint32_t)(p_g >> 32) : 0;
    const int dv = f == 0 ? d_nt : f == 1 ? (int)(uint32_t)d_ab : f == 2 ? (int)(uint32_t)(d_ab >> 32) : f == 3 ? d_ik : 0;
    vdp = l < P.nl - 1 ? pv : 0;
    vdd = l < F.nl - 1 ? dv : 0;
  }
  // epilogue of the adjoint GEMM whose output carries the activation pattern of layer idx
  auto dyn_epi = [&](int idx, int t, float* out) {
    const int nt = SH::NT ? SH::NT : pm_rl(vdd, 8 * idx);
    return EpiBwdL<RT, NP>{pm_rlp<const uint8_t>(vdd, 8 * idx + 1) + (size_t)t * B * nt * 4, pm_rlf(vdd, 8 * idx + 3),
                           out, nullptr, ELD, R, row0, nvalid, nt, lane};
  };
  auto pol_epi = [&](int idx, int t, size_t blk, float* out) {
    const int nt = SH::NT ? SH::NT : pm_rl(vdp, 8 * idx);
    return EpiBwdL<RT, NP>{pm_rlp<const uint8_t>(vdp, 8 * idx + 1) + (size_t)t * B * nt * 4, pm_rlf(vdp, 8 * idx + 3),
                           out, pm_rlp<float>(vdp, 8 * idx + 4) + blk * (size_t)nt * 16 * R, ELD, R, row0, nvalid, nt, lane};
  };
  const int pnl = SH::NL ? SH::NL : P.nl, fnl = SH::NL ? SH::NL : F.nl;
  const int dyn_tail_kb = PR ? (F.nt[1] + 1) / 2 : F.nt[1], pol_tail_kb = PR ? (P.nt[1] + 1) / 2 : P.nt[1];
  float* const gT_head = A.gT[P.nl - 1];

  // Same phase pattern as the forward sweep: descriptors and the epilogue's activation bits
  // (HBM / L2 reads) are issued before the barrier that opens a phase.
  bool pa_done = false;   // the previous step's last phase already ran this step's phase A
  // in-kernel moment matching: the pre-mm rows and the noise rows of step t - 1 are fetched during step t
  // (one value per thread: R * D <= PF_NT) -- their HBM round trip used to open the step's first phase
  // (not at four row tiles per wave: those instances are out of registers and the two values live across
  //  the step cost more in spills than the round trip -- C4 adjoint 2.00 -> 2.06 ms when tried)
  constexpr bool MMPF = VAR == PF_VAR_MM && RT <= 2 && (16 * RT) * (SH::D ? SH::D : 64) <= PF_NT;
  float mmx_cur = 0.f, mmz_cur = 0.f;
  bool mm_have = false;
  for (int t = T1 - 1; t >= T0; --t) {
    const size_t blk = (size_t)t * A.nwg + wg;
    int xsel = 0;           // X = bufA + xsel*R*LD (bufB directly follows bufA)
    float* X = L.bufA;
    float* Y = L.bufB;
    PF_MARK(0);
    float pfv[PFV];
#pragma unroll
    for (int u = 0; u < PFV; ++u)
      pfv[u] = (t > T0 && pf_base[u]) ? pf_base[u][(size_t)(t - 1) * pf_str[u]] : 0.f;
    if (mm_gs) {
      // dL/dx_{t+1} of every row of the group is needed: own rows -> HBM, meet the other workgroups,
      // then the adjoint of the moment matching that produced x_{t+1}.  The two carry buffers
      // alternate (a workgroup may write step t-1's rows while another still reads step t's).
      float* carry = ((T1 - 1 - t) & 1) ? A.gx_carry_out : A.gx_carry;
      __syncthreads();
      for (int i = tid; i < nvalid * D; i += PF_NT) pm_st_dev(carry + (size_t)row0 * D + i, gx[i]);
      if (!pm_grid_barrier(A.gsync, (unsigned)(T1 - t)) && tid == 0 && A.status) atomicMax(A.status, 1);
      PF_MARK(28);
      mm_span_bwd(t, carry, std::true_type{});
      PF_MARK(29);
      for (int i = tid; i < 2 * R * LD; i += PF_NT) L.bufA[i] = 0.f;   // restore the zero K padding
    }
    if (mm_xch) {
      // statistics exchange (see the forward sweep): wave 0 sums g^T [z | 1] over this part's rows, adds the other
      // parts' sums, and finishes the adjoint of the whole group's moment matching with the forward sweep's factor
      // for its own rows.  What comes from HBM (the factor, this part's pre-mm rows and noise rows) was staged during
      // the previous step's moment matching by wave 1, which has nothing else to do then.
      constexpr int DDc = (SH::D >= 1 && SH::D <= 6) ? SH::D : 1;
      float* const set = mmst + (size_t)(t & 1) * mmst_set;
      float* const xst = set + 2 * pm_mm_scratch_doubles(D);      // this part's pre-mm rows [nvalid][D]
      float* const zst = xst + R * D;                             // its noise rows of step t
      __syncthreads();          // dL/dx_{t+1} of this part's rows (gx) is complete
      PF_MARK(28);
      if (wid == 1 && t > T0) mm_stage(t - 1, lane, 64);      // (idle otherwise) what step t - 1 will need from HBM
      if (wid == 0) {
        const int me = wg - mmp_first;
        const unsigned k = (unsigned)(T1 - t);
        pm_f64x4 H = pm_mm_gram_h_lds<DDc, 4 * RT>(gx, zst, nvalid, lane, 0.0, 1.0);     // raw z: standardised below
        double v[2] = {H[0], H[1]};
        PF_MARK(30);
        pm_xch_put<2>(A.xch, mmp_first, me, k, v, lane);
        const MMScratch q = pm_mm_carve(reinterpret_cast<double*>(set), DDc);      // (the factor is in place)
        bool xok;
        if constexpr (SH::TREE) xok = pm_xch_get_tree<2>(A.xch, A.nwg, mmp_first, A.mm_parts, A.mm_fan, me, k, v, lane);
        else xok = pm_xch_get<2>(A.xch, mmp_first, A.mm_parts, me, k, v, lane);
        PF_MARK(31);
        if (!xok && lane == 0 && A.status) atomicMax(A.status, 1);
        // Lbar = tril((g^T z - mbar zm^T) diag(zi)), mbar = g^T 1
        {
          const int gq = lane >> 4, c = lane & 15, cc = c < DDc ? c : 0;
          double mb[DDc];
#pragma unroll
          for (int i = 0; i < DDc; ++i) mb[i] = pm_rl64(v[i >> 2], ((i & 3) << 4) | DDc);
          const double zmc = q.zmean[cc], zsc = q.zistd[cc];
#pragma unroll
          for (int r = 0; r < (DDc + 3) / 4; ++r) {
            const int i = gq + 4 * r;
            double mi = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (4 * r + u < DDc) mi = (gq == u) ? mb[4 * r + u] : mi;
            if (i < DDc && c < DDc) q.P[i * DDc + c] = (c <= i) ? (v[r] - zmc * mi) * zsc : 0.0;
          }
          if (lane < DDc) q.mbar[lane] = pm_sel(mb, lane);
        }
        pm_wave_sync();
        pm_mm_bwd_l_tail<DDc>(q, lane, A.M);
        const double inv_m = 1.0 / (double)A.M;
        for (int e = lane; e < R * DDc; e += 64) {      // (DDc = D here: divisions by a constant)
          const int r = e / DDc, j = e - r * DDc;
          double acc = 0.0;
          if (e < nvalid * DDc) {
            acc = q.mbar[j] * inv_m;
#pragma unroll
            for (int c = 0; c < DDc; ++c) acc += ((double)xst[r * DDc + c] - q.mean[c]) * q.P[c * DDc + j];
          }
          gxt[e] = (float)acc;
        }
      }
      PF_MARK(29);
      // (the barrier that opens phase A follows)
    } else if (!XW && mm_pair) {
      // dL/dx_{t+1} of the whole group is needed: own rows -> HBM (two buffers alternate: a partner may already
      // write step t-1's rows while this workgroup still reads step t's), meet the group, then the adjoint of
      // the moment matching of the WHOLE group on wave 0 of every workgroup of it; each keeps its own rows
      float* carry = ((T1 - 1 - t) & 1) ? A.gx_carry_out : A.gx_carry;
      __syncthreads();
      for (int i = tid; i < nvalid * D; i += PF_NT) pm_st_dev(carry + (size_t)row0 * D + i, gx[i]);
      // (the group's pre-mm rows, noise rows and the forward sweep's factor: on their way while the rows above go out)
      {
        const float* xsrc = A.xt + ((size_t)t * B + mmp_g0) * D;
        const float* zb = pm_zbase(A.zmm, D, t, A.Bg, A.flags);
        const int z0 = pm_zrow0(t, A.row_off + mmp_g0, A.flags);
        for (int i = tid; i < A.M * D; i += PF_NT) {
          const int r = i / D, d = i - r * D;
          mmg[i] = xsrc[i];
          mmg[A.M * D + i] = zb[(size_t)pm_zidx(z0, r, A.Bg) * D + d];
        }
        const double* fac = A.mmfac + ((size_t)t * A.mmfac_groups + mmp_g0 / A.M) * pm_mm_fac_doubles(D);
        if (SH::D >= 1 && SH::D <= 6 && A.M <= 64 && wid == 0)
          for (int e = lane; e < (int)pm_mm_fac_doubles(D); e += 64) L.mm[e] = fac[e];
      }
      if (!pm_group_sync(A.gsync, wg, mmp_first, A.mm_parts, (unsigned)(T1 - t)) && tid == 0 && A.status) atomicMax(A.status, 1);
      for (int i = tid; i < A.M * D; i += PF_NT) mmg[2 * A.M * D + i] = pm_ldc<true>(carry + (size_t)mmp_g0 * D + i);
      __syncthreads();
      if (wid == 0) {
        if (SH::D >= 1 && SH::D <= 6 && A.M <= 64)
          pm_mm_bwd_l<(SH::D >= 1 && SH::D <= 6) ? SH::D : 1>(mmg, D, A.M, mmg + A.M * D, D, mmg + 2 * A.M * D, D,
                                                                mmg + 3 * A.M * D, D, L.mm, lane, nullptr, true);
        else
          pm_mm_bwd(mmg, D, A.M, D, mmg + A.M * D, D, 0, 0, false, mmg + 2 * A.M * D, D, mmg + 3 * A.M * D, D, L.mm, lane,
                    A.mmfac + ((size_t)t * A.mmfac_groups + mmp_g0 / A.M) * pm_mm_fac_doubles(D));
      }
      __syncthreads();
      for (int i = tid; i < R * D; i += PF_NT)
        gxt[i] = i < nvalid * D ? mmg[3 * A.M * D + (row0 - mmp_g0) * D + i] : 0.f;
    } else if (mm_in) {
      // adjoint of the in-kernel moment matching of states (needs the pre-mm rows)
      // (scratch rows: the idle activation buffer, or -- split precision, where a stray fp32 row would
      //  land in the piece planes' zero padding -- the input tile, which is free until phase A)
      const float* xsrc = A.xt + (size_t)t * B * D;
      float* const xrows = PM_XIN(Y);
      const int xrows_ld = PM_XLD;
      if (MMPF && mm_have) {
        if (tid < R * D) {
          const int r = tid / D, d = tid - r * D;
          xrows[r * xrows_ld + d] = mmx_cur;
          if (r < nvalid) L.zs[tid] = mmz_cur;
        }
      } else {
        for (int i = tid; i < R * D; i += PF_NT) {
          const int r = i / D, d = i - r * D;
          xrows[r * xrows_ld + d] = (r < nvalid) ? xsrc[(size_t)(row0 + r) * D + d] : 0.f;
        }
        const float* zb = pm_zbase(A.zmm, D, t, A.Bg, A.flags);
        const int z0 = pm_zrow0(t, A.row_off + row0, A.flags);
        for (int i = tid; i < nvalid * D; i += PF_NT) {
          const int r = i / D, d = i - r * D;
          L.zs[i] = zb[(size_t)pm_zidx(z0, r, A.Bg) * D + d];
        }
      }
      if constexpr (MMPF) {
        // step t - 1's rows: in flight until the next iteration
        mm_have = t > T0;
        mmx_cur = mmz_cur = 0.f;
        if (mm_have && tid < R * D) {
          const int r = tid / D, d = tid - r * D;
          if (r < nvalid) {
            mmx_cur = A.xt[((size_t)(t - 1) * B + row0 + r) * D + d];
            const float* zb1 = pm_zbase(A.zmm, D, t - 1, A.Bg, A.flags);
            mmz_cur = zb1[(size_t)pm_zidx(pm_zrow0(t - 1, A.row_off + row0, A.flags), r, A.Bg) * D + d];
          }
        }
      }
      const int gpw = rows_per_wg / A.M;
      // this wave's group: statistics and factor of the forward sweep into the scratch, in flight with the rows
      constexpr bool MML = SH::D >= 1 && SH::D <= 6;
      if (MML && wid < gpw && wid * A.M < nvalid) {
        const double* fac = A.mmfac + ((size_t)t * A.mmfac_groups + (row0 + wid * A.M) / A.M) * pm_mm_fac_doubles(D);
        double* scr = L.mm + (size_t)wid * pm_mm_scratch_doubles(D);
        for (int e = lane; e < (int)pm_mm_fac_doubles(D); e += 64) scr[e] = fac[e];
      }
      __syncthreads();
      for (int gi = wid; gi < gpw; gi += PF_NW) {
        const int lr0 = gi * A.M;
        if (lr0 >= nvalid) break;
        double* scr = L.mm + (size_t)wid * pm_mm_scratch_doubles(D);
        if constexpr (MML) {
          pm_mm_bwd_l<SH::D ? SH::D : 1>(xrows + lr0 * xrows_ld, xrows_ld, A.M, L.zs + lr0 * D, D, gx + lr0 * D, D,
                                         gxt + lr0 * D, D, scr, lane,
                                         A.mmfac + ((size_t)t * A.mmfac_groups + (row0 + lr0) / A.M) * pm_mm_fac_doubles(D),
                                         gi == wid && gi < PF_NW);
        } else if constexpr (false) {
          // (the register form of the adjoint, pm_mm_bwd_w, was measured slower than the LDS form with the
          //  factor handed over from the forward sweep: 8.0 k vs 7.5 k cycles at d = 4 -- two Gram tiles and
          //  ~320 dependent fp64 operations on one wave; kept for reference in pmbrl_mm_w.h)
          pm_mm_bwd_w<SH::D ? SH::D : 1>(xrows + lr0 * xrows_ld, xrows_ld, A.M, L.zs + lr0 * D, D, gx + lr0 * D, D,
                                         gxt + lr0 * D, D, lane);
        } else {
          pm_mm_bwd(xrows + lr0 * xrows_ld, xrows_ld, A.M, D, L.zs + lr0 * D, D, 0, 0, false, gx + lr0 * D, D,
                    gxt + lr0 * D, D, scr, lane,
                    A.mmfac + ((size_t)t * A.mmfac_groups + (row0 + lr0) / A.M) * pm_mm_fac_doubles(D));
        }
      }
    }
    PF_MARK(1);
    // ---- reward adjoint from the stashed Jacobian, fused with the dynamics head adjoint input.
    //      On the plain path the previous step's last phase has already done this.
    if (!pa_done) {
      __syncthreads();
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, k = i & 15;
        const int d = k < D ? k : k - D;
        const float g0 = (k < 2 * D) ? (mm_in ? gxt[r * D + d] : gx[r * D + d]) : 0.f;
        phase_a(r, k, g0, gxn);
      }
    }
    // ---- dynamics trunk (dX only); tail (grad wrt [x|a]) K-split over the waves
    EpiBwdL<RT, NP> es{};
    {
      EpiBwdL<RT, NP> e = dyn_epi(fnl - 2, t, Y);
      Res0Pre<RT, EpiBwdL<RT, NP>> pp;
      res0_prefetch<RT>(pp, e, e.nt, wid);
      __syncthreads();
      PF_MARK(3);
      res0_layer<RT>(whd, e.nt, PM_XIN(X), PM_XLD, wid, lane, e, pp);
      if (fnl > 2) es = dyn_epi(fnl - 3, t, X);
    }
    __syncthreads();
    PM_SWAP_XY();
    PF_MARK(4);
    for (int l = fnl - 2, si = 0; l >= 1; --l, ++si) {
      PM_STREAM_LAYER(si, es, 0, nullptr);
      if (l - 1 >= 1) es = dyn_epi(l - 2, t, X);
      __syncthreads();
      PM_SWAP_XY();
      PF_MARK(4 + l);
    }
    if constexpr (PR != 0) head_partial_s<RT, NP, F16>(tsd, dyn_tail_kb, X, (unsigned)LDB, PM_HP(), wid, lane);
    else head_partial<RT>(twd, dyn_tail_kb, X, LD, PM_HP(), wid, lane);
    __syncthreads();
    PF_MARK(12);
    // ---- phase B: tail result; state part -> gxn, action part -> policy head adjoint
    {
      float* gst = gT_head + blk * (size_t)16 * (16 * RT);
      const float* hp = PM_HP();
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, k = i & 15;
        float tail = 0.f;
        if (k < D + U) tail = head_value<RT>(hp, r, k) * L.iSx[k];
        if (k < D) gxn[r * D + k] += tail;
        if (k >= D && k < D + U) {
          const int j = k - D;
          float go_mu = 0.f, go_ls = 0.f;
          if (r < nvalid) {
            float ga = L.gad[r * 16 + j] + tail;
            if (EXT && A.grad_actions) ga += A.grad_actions[((size_t)t * B + row0 + r) * U + j];
            L.gad[r * 16 + j] = ga;
            const float sc = L.psc[j];
            const float th = (stg[r * S + 1 + 2 * D + 2 * U + j] - L.pbi[j]) / sc;
            const float gu = ga * sc * (1.f - th * th);
            go_mu = gu;
            go_ls = gu * stg[r * S + 1 + 2 * D + U + j];
          }
          PM_XIN(X)[r * PM_XLD + j] = go_mu;
          PM_XIN(X)[r * PM_XLD + U + j] = go_ls;
          gst[(size_t)j * (16 * RT) + r] = go_mu;
          gst[(size_t)(U + j) * (16 * RT) + r] = go_ls;
        }
      }
      // zero the K padding of the head-gradient block (columns 2U..15)
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, k = i & 15;
        if (k >= 2 * U) {
          PM_XIN(X)[r * PM_XLD + k] = 0.f;
          gst[(size_t)k * (16 * RT) + r] = 0.f;
        }
      }
    }
    // ---- policy trunk: dX chain + G stash; tail (grad wrt x) K-split over the waves
    {
      EpiBwdL<RT, NP> e = pol_epi(pnl - 2, t, blk, Y);
      Res0Pre<RT, EpiBwdL<RT, NP>> pp;
      res0_prefetch<RT>(pp, e, e.nt, wid);
      __syncthreads();
      PF_MARK(13);
      // park the prefetched inputs of step t-1 (phase B was the last reader of step t's)
#pragma unroll
      for (int u = 0; u < PFV; ++u) {
        const int i = tid + u * PF_NT;
        if (i < R * S) L.stg[i] = pfv[u];
      }
      if (EXT && A.agn) {
        for (int r = tid; r < nvalid; r += PF_NT) {
          float s2 = 0.f;
          for (int j = 0; j < U; ++j) s2 = fmaf(L.gad[r * 16 + j], L.gad[r * 16 + j], s2);
          A.agn[(size_t)t * B + row0 + r] = sqrtf(s2);
        }
      }
      res0_layer<RT>(whp, e.nt, PM_XIN(X), PM_XLD, wid, lane, e, pp);
      if (pnl > 2) es = pol_epi(pnl - 3, t, blk, X);
    }
    __syncthreads();
    PM_SWAP_XY();
    PF_MARK(14);
    for (int l = pnl - 2, si = 0; l >= 1; --l, ++si) {
      PM_STREAM_LAYER(n_dyn_stream + si, es, 0, nullptr);
      if (l - 1 >= 1) es = pol_epi(l - 2, t, blk, X);
      __syncthreads();
      PM_SWAP_XY();
      PF_MARK(14 + l);
    }
    if constexpr (PR != 0) head_partial_s<RT, NP, F16>(tsp, pol_tail_kb, X, (unsigned)LDB, PM_HP(), wid, lane);
    else head_partial<RT>(twp, pol_tail_kb, X, LD, PM_HP(), wid, lane);
    __syncthreads();
    PF_MARK(22);
    // ---- dL/dx_t = gxn + policy tail (+ external state gradient); on the plain path the same
    //      threads go straight on to phase A of step t-1 (its inputs were parked mid-step)
    {
      const float* hp = PM_HP();
      // phase A of step t-1 writes bufA: not while the partial tiles sit there
      const bool do_pa = !mm_in && !mm_gs && t > T0 && !(RT >= 4 && L.hp_off < 0 && xsel == 1);
      pa_done = do_pa;
      if (do_pa) gsel ^= 1;
      float* gxn_next = L.jx + (gsel ? xb_off : 0);
      for (int i = tid; i < R * 16; i += PF_NT) {
        const int r = i >> 4, k = i & 15;
        const int d = k < D ? k : k - D;
        float v = 0.f;
        if (k < 2 * D) {
          v = gxn[r * D + d] + head_value<RT>(hp, r, d);
          if (EXT && A.grad_states && r < nvalid) v += A.grad_states[((size_t)t * B + row0 + r) * D + d];
          if (k < D) gx[r * D + d] = v;
        }
        if (do_pa) phase_a(r, k, v, gxn_next);
      }
      gxn = gxn_next;
    }
    PF_MARK(23);
    // (no closing barrier: the next step's first phase opens with one)
  }
  // nothing of the weight stream is outstanding here (its last chunk was consumed by the last
  // layer); stated for tools/check_inflight.py, which cannot know the loop ran at least once
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < nvalid * D; i += PF_NT) {
    const size_t o = (size_t)row0 * D + i;
    if (EXT && A.gx_carry && !mm_gs) (A.gx_carry_out ? A.gx_carry_out : A.gx_carry)[o] = gx[i];
    if (T0 == 0 && A.grad_x0) A.grad_x0[o] = gx[i];
  }
}
