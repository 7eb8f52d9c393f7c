// One-shot peer-to-peer all-reduce for LATENCY-bound messages: the per-step fp64 statistics of
// moment-matching groups spread over ranks (a few KB, 2 H + 2 times per iteration) and the flat policy gradient
// (163 KiB at the cart-pole shape, once per iteration).  A ring all-reduce pays 2 (N - 1) hops of latency for such a
// message; here every rank WRITES its contribution straight into a slot of every peer's buffer (xGMI stores through
// IPC-mapped pointers), raises a flag per peer, waits for its own N flags and adds the N slots in rank order -- one
// hop, the same bits on every rank, one kernel per rank, no host round trip (capturable: the generation number is a
// kernel argument, so a captured graph must be re-captured -- or the eager path used -- if calls are added).
//
// Memory: every rank owns one UNCACHED device allocation (flags + two slot sets that alternate between calls),
// exported with hipIpcGetMemHandle; the launcher carries the 64-byte handles to the other ranks (torch.distributed,
// MPI, a file -- anything), which map them with hipIpcOpenMemHandle.  Uncached so that a peer's stores are seen
// by the owner's loads whatever XCD's L2 either runs behind.  Why two slot sets suffice: a rank starts call g + 2
// (which reuses the set of call g) only after it finished call g + 1, i.e. after it saw every peer's flag of call g + 1,
// which a peer raises only after its own call g kernel -- the last reader of this rank's call-g data there -- ended.
// Waits are bounded (tens of seconds, PMBRL_P2P_TIMEOUT_SPINS): a peer that never arrives sets the error word instead
// of hanging the device; the callers read that word at their per-iteration host sync and agree on it across ranks
// (distributed.p2p_check) -- a timed-out exchange leaves the buffer un-reduced, which must never be used silently.
#include <cstdlib>
#include <cstring>
#include "pmbrl_host.h"

#define PM_P2P_MAX_RANKS 16
#define PM_P2P_BLOCKS 16          // chunks of a message: one workgroup and one flag each
#define PM_P2P_THREADS 256

struct pmbrl_p2p {
  int rank, nranks, device;
  size_t cap;                       // bytes per (set, source rank) slot
  char* region[PM_P2P_MAX_RANKS];   // mapped base of every rank's allocation (region[rank]: this rank's own)
  bool opened[PM_P2P_MAX_RANKS];
  unsigned gen;                     // calls so far
  int* err_d;                       // device word: set when a wait timed out
  long long max_spins;              // polls of one flag before a wait gives up (PMBRL_P2P_TIMEOUT_SPINS)
};

// layout of a region: flags [MAX_RANKS][BLOCKS] unsigned (padded to 4 KB), then slots [2][MAX_RANKS][cap]
__host__ __device__ static inline size_t p2p_flags_bytes() { return 4096; }
static inline size_t p2p_region_bytes(size_t cap) { return p2p_flags_bytes() + (size_t)2 * PM_P2P_MAX_RANKS * cap; }

struct P2PArgs {
  char* region[PM_P2P_MAX_RANKS];
  int rank, nranks;
  unsigned gen;
  size_t cap;
  long long n;
  int* err;
  long long max_spins;
};

template <typename T>
__global__ __launch_bounds__(PM_P2P_THREADS) void pm_p2p_allreduce_kernel(const P2PArgs A, T* __restrict__ buf) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long per = (A.n + PM_P2P_BLOCKS - 1) / PM_P2P_BLOCKS;
  const long long lo = (long long)b * per, hi = min(A.n, lo + per);
  const int set = (int)(A.gen & 1u);
  const size_t slot_off = p2p_flags_bytes() + ((size_t)set * PM_P2P_MAX_RANKS + A.rank) * A.cap;
  // 1. this rank's chunk into its slot on every rank (own one last), then one flag per peer
  for (int k = 1; k <= A.nranks; ++k) {
    const int q = (A.rank + k) % A.nranks;
    T* dst = reinterpret_cast<T*>(A.region[q] + slot_off);
    for (long long i = lo + tid; i < hi; i += PM_P2P_THREADS) __builtin_nontemporal_store(buf[i], dst + i);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < A.nranks) {
    unsigned* f = reinterpret_cast<unsigned*>(A.region[tid]) + (size_t)A.rank * PM_P2P_BLOCKS + b;
    __hip_atomic_store(f, A.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. every rank's chunk has arrived here
  __shared__ int ok_s;
  if (tid == 0) ok_s = 1;
  __syncthreads();
  if (tid < A.nranks) {
    const unsigned* f = reinterpret_cast<const unsigned*>(A.region[A.rank]) + (size_t)tid * PM_P2P_BLOCKS + b;
    long long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < A.gen) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > A.max_spins) {     // (default: tens of seconds -- a late peer is waited for, a dead one is not)
        ok_s = 0;
        atomicExch(A.err, 1);
        break;
      }
    }
  }
  __syncthreads();
  if (!ok_s) return;
  // 3. the sum in rank order: the same bits on every rank
  const char* base = A.region[A.rank] + p2p_flags_bytes() + (size_t)set * PM_P2P_MAX_RANKS * A.cap;
  for (long long i = lo + tid; i < hi; i += PM_P2P_THREADS) {
    T s = (T)0;
    for (int q = 0; q < A.nranks; ++q)
      s += __builtin_nontemporal_load(reinterpret_cast<const T*>(base + (size_t)q * A.cap) + i);
    buf[i] = s;
  }
}

extern "C" int pmbrl_p2p_create(int32_t rank, int32_t nranks, int32_t device, int64_t max_bytes, pmbrl_p2p** out) {
  if (!out || nranks < 1 || nranks > PM_P2P_MAX_RANKS || rank < 0 || rank >= nranks || max_bytes < 8)
    return pm_fail(-1, "pmbrl_p2p_create: bad argument");
  HIPCHK(hipSetDevice(device));
  pmbrl_p2p* p = new pmbrl_p2p();
  memset(p, 0, sizeof(*p));
  p->rank = rank; p->nranks = nranks; p->device = device;
  p->cap = ((size_t)max_bytes + 255) / 256 * 256;
  // A rank can be seconds late (a plan rebuilt for a precision retry, a host callback, first-call skew): the wait
  // must outlast that -- what it must not outlast is a peer that died.  ~1.5 us per poll: 20 M polls ~ 30 s.
  p->max_spins = 20000000ll;
  if (const char* e = getenv("PMBRL_P2P_TIMEOUT_SPINS")) {
    const long long v = atoll(e);
    if (v > 0) p->max_spins = v;
  }
  void* mem = nullptr;
  // uncached: remote stores and local loads of the same bytes meet in memory, not in some XCD's L2
  if (hipExtMallocWithFlags(&mem, p2p_region_bytes(p->cap), hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    delete p;
    return pm_fail(-3, "pmbrl_p2p_create: hipExtMallocWithFlags(hipDeviceMallocUncached) failed");
  }
  p->region[rank] = static_cast<char*>(mem);
  p->opened[rank] = false;
  HIPCHK(hipMemset(mem, 0, p2p_flags_bytes()));
  HIPCHK(hipMalloc(&p->err_d, sizeof(int)));
  HIPCHK(hipMemset(p->err_d, 0, sizeof(int)));
  HIPCHK(hipDeviceSynchronize());
  *out = p;
  return 0;
}

extern "C" int pmbrl_p2p_handle(pmbrl_p2p* p, void* handle_out) {
  if (!p || !handle_out) return pm_fail(-1, "null argument");
  hipIpcMemHandle_t h;
  HIPCHK(hipIpcGetMemHandle(&h, p->region[p->rank]));
  static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

extern "C" int pmbrl_p2p_open(pmbrl_p2p* p, int32_t peer, const void* handle) {
  if (!p || !handle || peer < 0 || peer >= p->nranks) return pm_fail(-1, "bad argument");
  if (peer == p->rank) return 0;
  HIPCHK(hipSetDevice(p->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* ptr = nullptr;
  HIPCHK(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
  p->region[peer] = static_cast<char*>(ptr);
  p->opened[peer] = true;
  return 0;
}

template <typename T>
static int p2p_allreduce(pmbrl_p2p* p, void* stream, T* buf_d, int64_t n) {
  if (!p || !buf_d || n < 0) return pm_fail(-1, "bad argument");
  if ((size_t)n * sizeof(T) > p->cap) return pm_fail(-2, "pmbrl_p2p: message larger than the slots (max_bytes of pmbrl_p2p_create)");
  for (int q = 0; q < p->nranks; ++q)
    if (!p->region[q]) return pm_fail(-3, "pmbrl_p2p: a peer's buffer has not been opened");
  if (n == 0) return 0;
  {
    // not capturable: the generation the flags are compared with is a kernel ARGUMENT -- a replayed recording would find
    // last time's flags already raised and add up stale slots
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return pm_fail(-4, "pmbrl_p2p: the peer-to-peer all-reduce cannot be recorded into a hipGraph (its generation counter is a kernel argument)");
  }
  P2PArgs A;
  for (int q = 0; q < PM_P2P_MAX_RANKS; ++q) A.region[q] = q < p->nranks ? p->region[q] : nullptr;
  A.rank = p->rank; A.nranks = p->nranks; A.cap = p->cap; A.n = n; A.err = p->err_d;
  A.max_spins = p->max_spins;
  A.gen = ++p->gen;
  hipLaunchKernelGGL(pm_p2p_allreduce_kernel<T>, dim3(PM_P2P_BLOCKS), dim3(PM_P2P_THREADS), 0, (hipStream_t)stream, A, buf_d);
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int pmbrl_p2p_allreduce_f32(pmbrl_p2p* p, void* stream, float* buf_d, int64_t n) {
  return p2p_allreduce<float>(p, stream, buf_d, n);
}
extern "C" int pmbrl_p2p_allreduce_f64(pmbrl_p2p* p, void* stream, double* buf_d, int64_t n) {
  return p2p_allreduce<double>(p, stream, buf_d, n);
}
// pmbrl_collective_fn (include/pmbrl.h) over a pmbrl_p2p: what pmbrl_plan_set_p2p attaches
static int pm_coll_p2p(void* ctx, void* stream, double* buf_d, int64_t n) {
  return p2p_allreduce<double>(static_cast<pmbrl_p2p*>(ctx), stream, buf_d, n);
}
extern "C" int pmbrl_plan_set_p2p(pmbrl_plan* plan, pmbrl_p2p* p) {
  if (!plan || !p) return pm_fail(-1, "null argument");
  return pmbrl_plan_set_collective(plan, pm_coll_p2p, p);
}

// host sync: 1 if a wait timed out since the last call (a peer never arrived), 0 otherwise
extern "C" int pmbrl_p2p_error(pmbrl_p2p* p, int32_t* err_out) {
  if (!p || !err_out) return pm_fail(-1, "null argument");
  int e = 0;
  HIPCHK(hipMemcpy(&e, p->err_d, sizeof(int), hipMemcpyDeviceToHost));
  if (e) HIPCHK(hipMemset(p->err_d, 0, sizeof(int)));
  *err_out = e;
  return 0;
}

extern "C" void pmbrl_p2p_destroy(pmbrl_p2p* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < p->nranks; ++q)
    if (q != p->rank && p->opened[q] && p->region[q]) (void)hipIpcCloseMemHandle(p->region[q]);
  if (p->region[p->rank]) (void)hipFree(p->region[p->rank]);
  if (p->err_d) (void)hipFree(p->err_d);
  delete p;
}
