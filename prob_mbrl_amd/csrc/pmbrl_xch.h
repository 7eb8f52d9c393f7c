// Statistics exchange between the workgroups that share a moment-matching group (data-tagged 8-byte granules; no
// flag, no barrier, no row exchange).  Used by the latency-optimised sweeps (pmbrl_fast.h) and by the register-resident
// family's moment-matching instances (pmbrl_reg.h).
#pragma once
#include <hip/hip_runtime.h>
// (polling with workgroup-scope loads -- sc0: meant to meet a same-XCD partner's stores in the shared L2 -- was tried once the
//  parts of a group were placed on one XCD: the CU's L1 serves the poller its own stale line, the wait never ends.  Agent
//  scope stays; the placement alone is worth 2-4 %.)
#ifndef PM_XCH_POLL_SCOPE
#define PM_XCH_POLL_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#ifndef PM_GLOBAL
#define PM_GLOBAL __attribute__((address_space(1)))
#endif

// Statistics of a moment-matching group that is split over `parts` workgroups: every part computes the sums over its
// OWN rows (relative to a reference point all parts share), and wave 0 of each adds up the parts' sums -- NV doubles
// per lane, element-wise, in part order (so every part ends with the same bits).  The doubles travel as data-tagged
// 8-byte granules {tag = k, 32 bits of the value} written by one device-scope store each and polled by the reader:
// no flag, no barrier, no row exchange (cdna_hip_programming.md, Guideline 16, form R2).  xb: [nwg][2][NV][2][64]
// granules, zeroed before every launch (tag 0 = nothing yet); the two sets alternate between steps -- a part
// writes its step k + 2 only after it read everybody's k + 1, which was written after that part had read this k.
// A wait that does not end (a partner that is not resident -- the host checks that all are) gives up after ~1 s.
#define PM_XCH_WG_WORDS(NV) (2 * (NV) * 2 * 64)
// More than PM_XCH_FLAT parts (one moment-matching group over the whole batch: 157 parts at 2 500 rows): TWO LEVELS.
// Every `fan` consecutive parts have a collector (the first of them), which adds up their sums in part order and
// publishes the result in a slot of its own (slot nwg + first + c: the buffer holds 2 x nwg slots); every part then
// adds up the collectors' slots in collector order -- the same bits everywhere again, two hops of at most `fan` slots
// each instead of one walk over every part's slot (157 x 2 KB per part and step).  Slots are polled in batches of 8
// (all their granules requested before the first is looked at: one memory round trip a batch, not one a slot).
// The alternation argument above holds level by level.
#define PM_XCH_FLAT 8
// publish this part's NV doubles per lane for step k (nothing is waited for: the stores are on their way)
template <int NV>
__device__ __forceinline__ void pm_xch_put(unsigned long long* xb, int first, int me, unsigned k, const double (&v)[NV],
                                           int lane) {
  typedef PM_GLOBAL unsigned long long gu64;
  const unsigned long long tag = (unsigned long long)k << 32;
  gu64* mine = (gu64*)xb + (size_t)(first + me) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64) + lane;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
    __hip_atomic_store(mine + (2 * i) * 64, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + (2 * i + 1) * 64, tag | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// v <- the sum over all parts, in part order (v holds this part's own contribution on entry)
template <int NV>
__device__ __forceinline__ bool pm_xch_get(const unsigned long long* xb, int first, int parts, int me, unsigned k,
                                           double (&v)[NV], int lane) {
  typedef PM_GLOBAL unsigned long long gu64;
  double tot[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) tot[i] = 0.0;
  bool ok = true;
  for (int q = 0; q < parts; ++q) {
    if (q == me) {
#pragma unroll
      for (int i = 0; i < NV; ++i) tot[i] += v[i];
      continue;
    }
    const gu64* theirs = (const gu64*)xb + (size_t)(first + q) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64) + lane;
    unsigned long long g[2 * NV];
    for (int spins = 0;;) {
      bool here = true;
#pragma unroll
      for (int i = 0; i < 2 * NV; ++i) {
        g[i] = __hip_atomic_load(theirs + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        here = here && (g[i] >> 32) == (unsigned long long)k;
      }
      if (__all(here)) break;
      if (++spins > (1 << 19)) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) break;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      tot[i] += __longlong_as_double((long long)(((g[2 * i] & 0xffffffffull) << 32) | (g[2 * i + 1] & 0xffffffffull)));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = tot[i];
  return ok;
}


// Two parts: the one partner's granules, the sum in part order (pm_xch_get's arithmetic without its loop over the parts:
// a poll is 2 NV loads and as many compares, and what stands between the arrival of the partner's sums and their use is
// one such round)
template <int NV>
__device__ __forceinline__ bool pm_xch_get_pair(const unsigned long long* xb, int first, int me, unsigned k, double (&v)[NV], int lane) {
  typedef PM_GLOBAL unsigned long long gu64;
  const gu64* theirs = (const gu64*)xb + (size_t)(first + (me ^ 1)) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64) + lane;
  unsigned long long g[2 * NV];
  for (int spins = 0;;) {
    unsigned tags = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 2 * NV; ++i) {
      g[i] = __hip_atomic_load(theirs + i * 64, __ATOMIC_RELAXED, PM_XCH_POLL_SCOPE);
      tags &= (unsigned)(g[i] >> 32) ^ ~k;      // all ones where the tag is k
    }
    if (__all(tags == 0xffffffffu)) break;
    if (++spins > (1 << 19)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double t = __longlong_as_double((long long)(((g[2 * i] & 0xffffffffull) << 32) | (g[2 * i + 1] & 0xffffffffull)));
    v[i] = me == 0 ? v[i] + t : t + v[i];
  }
  return true;
}

// The same with the partners' granules requested TOGETHER (batches of NB slots): pm_xch_get walks the parts one memory
// round trip after the other -- three of them for a group in four parts, ~0.7 k cycles each even when everything has long
// arrived.  This part's own contribution comes from its registers (v on entry); the sum is taken in part order, so every
// part ends with the same bits.
template <int NV, int NB>
__device__ __forceinline__ bool pm_xch_get_all(const unsigned long long* xb, int first, int parts, int me, unsigned k,
                                               double (&v)[NV], int lane) {
  typedef PM_GLOBAL unsigned long long gu64;
  double tot[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) tot[i] = 0.0;
  for (int q0 = 0; q0 < parts; q0 += NB) {
    unsigned long long g[NB][2 * NV];
    for (int spins = 0;;) {
      bool here = true;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int q = q0 + b < parts ? q0 + b : q0;      // (a short last batch asks for its first slot again)
        const gu64* theirs = (const gu64*)xb + (size_t)(first + q) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) g[b][i] = __hip_atomic_load(theirs + i * 64, __ATOMIC_RELAXED, PM_XCH_POLL_SCOPE);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool mine = q0 + b == me || q0 + b >= parts;
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) here = here && (mine || (g[b][i] >> 32) == (unsigned long long)k);
      }
      if (__all(here)) break;
      if (++spins > (1 << 19)) return false;
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (q0 + b < parts) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const double theirs = __longlong_as_double((long long)(((g[b][2 * i] & 0xffffffffull) << 32) | (g[b][2 * i + 1] & 0xffffffffull)));
          tot[i] += q0 + b == me ? v[i] : theirs;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = tot[i];
  return true;
}

// tot += the sums in slots [slot0, slot0 + n), in slot order; batches of 8 slots in flight
// PRE: wait for ONE granule (lane 0's first word of the batch's last slot: every lane asks for the same 8 bytes) before the
// batch is asked for -- with hundreds of waves polling the same dozen slots (one group over the batch: every part reads
// every collector's slot) the full-batch polls, 4 NV NB x 512 bytes each, are a flood of uncached reads that the stores they
// wait for queue behind (MI355X_MICROARCH.md: "255 pollers cut chip bandwidth 37-71 %"; poll one word, then read)
template <int NV, int NB = 8, bool PRE = false>
__device__ __forceinline__ bool pm_xch_add_slots(const unsigned long long* xb, int slot0, int n, unsigned k, double (&tot)[NV],
                                                 int lane) {
  typedef PM_GLOBAL unsigned long long gu64;
  for (int q0 = 0; q0 < n; q0 += NB) {
    const int nb = n - q0 < NB ? n - q0 : NB;
    if constexpr (PRE) {
      const gu64* one = (const gu64*)xb + (size_t)(slot0 + q0 + nb - 1) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64);
      for (int spins = 0;;) {
        const unsigned long long g1 = __hip_atomic_load(one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((g1 >> 32) == (unsigned long long)k) break;
        if (++spins > (1 << 19)) return false;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    unsigned long long g[NB][2 * NV];
    for (int spins = 0;;) {
      bool here = true;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int q = q0 + (b < nb ? b : 0);        // (a short last batch asks for its first slot again)
        const gu64* theirs = (const gu64*)xb + (size_t)(slot0 + q) * PM_XCH_WG_WORDS(NV) + (size_t)(k & 1u) * (NV * 2 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) g[b][i] = __hip_atomic_load(theirs + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) here = here && (g[b][i] >> 32) == (unsigned long long)k;
      if (__all(here)) break;
      if (++spins > (1 << 19)) return false;
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (b < nb) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          tot[i] += __longlong_as_double((long long)(((g[b][2 * i] & 0xffffffffull) << 32) | (g[b][2 * i + 1] & 0xffffffffull)));
      }
  }
  return true;
}
// v <- the sum over all parts, two levels (v holds this part's own contribution on entry, already published by pm_xch_put)
// (NB: slots in flight per batch -- 4 NV NB registers of granules)
template <int NV, int NB = 8>
__device__ __forceinline__ bool pm_xch_get_tree(unsigned long long* xb, int nwg, int first, int parts, int fan, int me, unsigned k,
                                                double (&v)[NV], int lane) {
  const int c = me / fan, c0 = c * fan, nc = (parts + fan - 1) / fan;
  bool ok = true;
  if (me == c0) {
    double tot[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] = 0.0;
    ok = pm_xch_add_slots<NV, NB>(xb, first + c0, (parts - c0 < fan ? parts - c0 : fan), k, tot, lane);
    pm_xch_put<NV>(xb, nwg + first, c, k, tot, lane);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
  ok = pm_xch_add_slots<NV, NB>(xb, nwg + first, nc, k, v, lane) && ok;
  return ok;
}

// ---------------------------------------------------------------------------
// The two-level exchange with ALL waves of the workgroup polling (the register-resident family, round 6).  One wave walks a
// level's <= 13 slots in batches of NB = 4 (what its registers hold): four memory round trips per level, eight per step --
// 14 k cycles of a 27 k-cycle step at 157 parts (profiles/r06_single_group.txt).  The workgroup's other waves idle during
// the chain: wave w (1 .. 3) takes the slots [w NB, (w + 1) NB) of a level in ONE batch, leaves their sum (slot order) in
// LDS behind a tag, and the chain's wave adds  (its own batch) + p1 + p2 + p3  -- the same association in every part, so
// every part still ends with the same bits.  One round trip per level; <= 4 NB = 16 slots per level (256 parts).
// hp: LDS [3 helper waves][NV][64] doubles, shared by the two levels -- a collector's helper writes its level-2 sum only
// once the chain's wave has acknowledged reading the level-1 sums (tags[6]); tags: LDS [2 levels][3 waves] + the
// acknowledgement, 7 words, zero at launch (a tag is a step's k > 0).
// ---------------------------------------------------------------------------
#define PM_XCH_HELP_DOUBLES(NV) (3 * (NV) * 64)
// (LDS pointers carry their address space: through generic pointers the tags and the sums were FLAT accesses -- counted by
//  vmcnt AND lgkmcnt, out of order with the LDS traffic around them)
#define PM_LDS __attribute__((address_space(3)))
typedef PM_LDS double* pm_lds_d;
typedef PM_LDS const double* pm_lds_cd;
typedef PM_LDS volatile unsigned* pm_lds_vu;
typedef PM_LDS volatile const unsigned* pm_lds_cvu;
__device__ __forceinline__ bool pm_xch_lds_wait(pm_lds_cvu tag, unsigned k) {
  for (int spins = 0;;) {
    if (*tag == k) return true;
    if (++spins > (1 << 19)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}
template <int NV, int NB>
__device__ __forceinline__ bool pm_xch_help_one(const unsigned long long* xb, int slot0, int n, int w, unsigned k, pm_lds_d part,
                                                pm_lds_vu tag, pm_lds_cvu ack, int lane) {
  const int lo = w * NB;
  if (lo >= n) return true;
  double tot[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) tot[i] = 0.0;
  bool ok = pm_xch_add_slots<NV, NB, true>(xb, slot0 + lo, n - lo < NB ? n - lo : NB, k, tot, lane);
  if (ack) ok = pm_xch_lds_wait(ack, k) && ok;
#pragma unroll
  for (int i = 0; i < NV; ++i) part[i * 64 + lane] = tot[i];
  // (a wave's LDS operations complete in order: the tag is written once the sums are in LDS)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  *tag = k;
  return ok;
}
// helper wave w (1 .. 3) of part `me`
template <int NV, int NB>
__device__ __forceinline__ bool pm_xch_tree_help(const unsigned long long* xb, int nwg, int first, int parts, int fan, int me,
                                                 unsigned k, int w, double* hp_, volatile unsigned* tags_, int lane) {
  const pm_lds_d hp = (pm_lds_d)hp_;
  const pm_lds_vu tags = (pm_lds_vu)tags_;
  const int c = me / fan, c0 = c * fan, nc = (parts + fan - 1) / fan;
  const pm_lds_d part = hp + (w - 1) * NV * 64;
  bool ok = true;
  const int n1 = parts - c0 < fan ? parts - c0 : fan;
  const bool wrote1 = me == c0 && w * NB < n1;
  if (me == c0) ok = pm_xch_help_one<NV, NB>(xb, first + c0, n1, w, k, part, tags + (w - 1), (pm_lds_cvu)nullptr, lane);
  ok = pm_xch_help_one<NV, NB>(xb, nwg + first, nc, w, k, part, tags + 3 + (w - 1), wrote1 ? (pm_lds_cvu)(tags + 6) : (pm_lds_cvu)nullptr, lane) && ok;
  return ok;
}
// the chain's wave: tot <- (slots 0 .. NB - 1) + the helpers' partial sums, in wave order
template <int NV, int NB>
__device__ __forceinline__ bool pm_xch_sum_helped(const unsigned long long* xb, int slot0, int n, unsigned k, double (&tot)[NV],
                                                  pm_lds_cd hp, pm_lds_cvu tags, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) tot[i] = 0.0;
  const bool ok = pm_xch_add_slots<NV, NB, true>(xb, slot0, n < NB ? n : NB, k, tot, lane);
  for (int w = 1; w < 4; ++w) {
    if (w * NB >= n) break;
    if (!pm_xch_lds_wait(tags + (w - 1), k)) return false;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] += hp[(w - 1) * NV * 64 + i * 64 + lane];
  }
  return ok;
}
// v <- the sum over all parts (this part's own contribution already published by pm_xch_put)
template <int NV, int NB>
__device__ __forceinline__ bool pm_xch_get_tree_helped(unsigned long long* xb, int nwg, int first, int parts, int fan, int me,
                                                       unsigned k, double (&v)[NV], const double* hp_, volatile unsigned* tags_,
                                                       int lane) {
  const pm_lds_cd hp = (pm_lds_cd)hp_;
  const pm_lds_vu tags = (pm_lds_vu)tags_;
  const int c = me / fan, c0 = c * fan, nc = (parts + fan - 1) / fan;
  bool ok = true;
  if (me == c0) {
    double tot[NV];
    ok = pm_xch_sum_helped<NV, NB>(xb, first + c0, parts - c0 < fan ? parts - c0 : fan, k, tot, hp, tags, lane);
    // (the level-1 sums are in registers -- the additions above waited for them: their buffer is the helpers' again)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tags[6] = k;
    pm_xch_put<NV>(xb, nwg + first, c, k, tot, lane);
  }
  ok = pm_xch_sum_helped<NV, NB>(xb, nwg + first, nc, k, v, hp, tags + 3, lane) && ok;
  return ok;
}
