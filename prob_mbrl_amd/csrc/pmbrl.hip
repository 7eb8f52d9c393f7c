// libpmbrl_hip.so -- C ABI (include/pmbrl.h) over the gfx950 kernels.
// Host side: shape validation, tiling choice, workspace carve-up, launches.
#define PM_MAIN_TU
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "pmbrl_host.h"
#include "pmbrl_mm.h"
#include "pmbrl_rollout.h"
#include "pmbrl_mmx.h"
#include "pmbrl_mm_wide.h"
#include "pmbrl_fast.h"
#include "pmbrl_dw.h"
#include "pmbrl_mlp.h"
#include "pmbrl_bnn.h"

static thread_local std::string g_err;
int pm_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int fail(int code, const std::string& msg) { return pm_fail(code, msg); }

extern "C" const char* pmbrl_last_error(void) { return g_err.c_str(); }
extern "C" int pmbrl_version(void) { return 6; }

// ---------------------------------------------------------------------------
// hipGraph entry points: record the library calls queued on a stream between begin and end, replay them with one
// launch.  What it is for: the forms of the sweeps that are MANY launches per call (one per step: moment-matching
// groups beyond a workgroup, wide states; pipelined adjoint) -- a plain iteration is a dozen launches the queue
// already hides.  Everything the library queues is capturable (kernel launches, memsets, the event hand-offs of the
// second stream); a host-side collective attached with pmbrl_plan_set_collective is not.  The status word, the step
// counter of the guarded Adam and every pointer are device-side, so a replay is a whole new iteration as long as the
// caller's buffers stay where they were.  (The fp16 weight-range flag of a replayed call carries the generation number
// of the capture: exact unless an eager call on the same plan reported an overflow in between.)
// ---------------------------------------------------------------------------
struct pmbrl_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};
extern "C" int pmbrl_graph_capture_begin(void* stream) {
  if (!stream) return fail(-1, "graph capture needs a non-default stream");
  HIPCHK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
  return 0;
}
extern "C" int pmbrl_graph_capture_end(void* stream, pmbrl_graph** out) {
  if (!stream || !out) return fail(-1, "null argument");
  hipGraph_t g = nullptr;
  HIPCHK(hipStreamEndCapture((hipStream_t)stream, &g));
  if (!g) return fail(-3, "stream capture ended without a graph (a call inside it invalidated the capture)");
  pmbrl_graph* r = new pmbrl_graph();
  r->graph = g;
  r->exec = nullptr;
  hipError_t e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    delete r;
    return fail(-100 - (int)e, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  }
  *out = r;
  return 0;
}
extern "C" int pmbrl_graph_launch(pmbrl_graph* g, void* stream) {
  if (!g || !g->exec) return fail(-1, "null argument");
  HIPCHK(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return 0;
}
extern "C" int pmbrl_graph_num_nodes(pmbrl_graph* g, int64_t* n_out) {
  if (!g || !n_out) return fail(-1, "null argument");
  size_t n = 0;
  HIPCHK(hipGraphGetNodes(g->graph, nullptr, &n));
  *n_out = (int64_t)n;
  return 0;
}
extern "C" void pmbrl_graph_destroy(pmbrl_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}

// hash of the kernel sources this library was built from (Makefile: pmbrl_build_id.inc)
extern "C" const char* pmbrl_build_id(void) {
  return
#include "pmbrl_build_id.inc"
      ;
}

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
// fragment packing: dst[((ot*n_kb + kb)*64 + lane)*4 + j] = W[ot*16 + (lane&15)][kb*16 + 4*(lane>>4) + j]
// (transpose=1: the roles of the two indices of W[O][K] are swapped)
__global__ void pm_pack_frag(const float* __restrict__ W, int O, int K, int transpose, int kb_mult,
                             float* __restrict__ dst) {
  const int n_out = transpose ? K : O, n_in = transpose ? O : K;
  const int n_ot = (n_out + 15) / 16;
  const int n_kb = ((n_in + 15) / 16 + kb_mult - 1) / kb_mult * kb_mult;   // zero-padded k-blocks
  const size_t total = (size_t)n_ot * n_kb * 256;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63;
    const size_t tb = i >> 8;
    const int kb = tb % n_kb, ot = tb / n_kb;
    const int o = ot * 16 + (lane & 15);
    const int k = kb * 16 + 4 * (lane >> 4) + j;
    float v = 0.f;
    if (o < n_out && k < n_in) v = transpose ? W[(size_t)k * K + o] : W[(size_t)o * K + k];
    dst[i] = v;
  }
}
__global__ void pm_pack_bias(const float* __restrict__ b, int O, int O16, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < O16) dst[i] = i < O ? b[i] : 0.f;
}
__global__ void pm_pack_mask_kernel(const float* __restrict__ m, int B, int h, int ld,
                                    uint16_t* __restrict__ bits) {
  const int nt = (h + 15) / 16;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nt) return;
  const int r = i / nt, t = i - r * nt;
  unsigned w = 0;
  for (int j = 0; j < 16; ++j) {
    const int f = t * 16 + j;
    if (f < h && m[(size_t)r * ld + f] != 0.f) w |= 1u << j;
  }
  bits[i] = (uint16_t)w;
}
__global__ void pm_set_int(int* p, int v) { *p = v; }

// ---------------------------------------------------------------------------
// Dropout masks drawn on the device (pmbrl_draw_masks): bit rows straight from a counter-based generator.
// Philox4x32-10 (Salmon et al., SC'11) keyed by the caller's seed; counter = (row, 16-unit tile, quad + 4 * stream,
// offset): a draw depends on (seed, offset, row, unit) alone, not on the launch shape.  Stream 0: the uniform noise u;
// stream 1: the uniform the hard Bernoulli sample is thresholded with.
// ---------------------------------------------------------------------------
struct DrawArgs {
  int kind;              // 0: Bernoulli(keep) -- BDropout.update_noise, models/modules.py:40-44;  1: concrete -- :95-118
  unsigned long long seed, offset;
  const float* param;    // kind 0: keep probability; kind 1: logit_p  (param_len 1 or h)
  int param_len;
  float temp;
  int rows, h;
  const float *u_in, *v_in;     // [rows][h] uniforms to use instead of the generator's (nullptr: generate)
  uint16_t* bits;        // [rows][ceil(h/16)]
  int aux_row0, aux_rows;       // rows whose float values are also written (a module keeps its LAST draw as state)
  float *u_out, *hard_out, *probs_out;   // [aux_rows][h] each, or nullptr
};
__global__ __launch_bounds__(256) void pm_draw_masks_kernel(const DrawArgs A) {
  const int nt = (A.h + 15) / 16;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)A.rows * nt) return;
  const int r = (int)(i / nt), t = (int)(i - (long long)r * nt);
  const unsigned k0 = (unsigned)A.seed, k1 = (unsigned)(A.seed >> 32);
  const unsigned o0 = (unsigned)A.offset, o1 = (unsigned)(A.offset >> 32);
  unsigned w = 0;
  const bool aux = r >= A.aux_row0 && r < A.aux_row0 + A.aux_rows;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned ru[4], rv[4];
    if (!A.u_in) pm_philox((unsigned)r, (unsigned)t, (unsigned)q ^ (o1 << 3), o0, k0, k1, ru);
    if (A.kind == 1 && !A.v_in) pm_philox((unsigned)r, (unsigned)t, (unsigned)(4 + q) ^ (o1 << 3), o0, k0, k1, rv);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int f = t * 16 + q * 4 + e;
      if (f >= A.h) continue;
      const size_t idx = (size_t)r * A.h + f;
      // 24-bit uniforms in [0, 1) (what torch.rand gives a float tensor)
      const float u = A.u_in ? A.u_in[idx] : (float)(ru[e] >> 8) * (1.0f / 16777216.0f);
      const float pr = A.param[A.param_len > 1 ? f : 0];
      float probs;
      bool bit;
      if (A.kind == 0) {
        probs = pr;
        bit = u < pr;                                    // torch.bernoulli(p): 1 with probability p
      } else {
        // models/modules.py:102-114: probs = sigmoid((logit_p + log((u + 1e-7) / (1 - (u - 1e-7)))) / temp)
        const float cp = pr + logf((u + 1e-7f) / (1.f - (u - 1e-7f)));
        probs = 1.f / (1.f + expf(-cp / A.temp));
        const float v = A.v_in ? A.v_in[idx] : (float)(rv[e] >> 8) * (1.0f / 16777216.0f);
        bit = v < probs;
      }
      if (bit) w |= 1u << (q * 4 + e);
      if (aux) {
        const size_t o = (size_t)(r - A.aux_row0) * A.h + f;
        if (A.u_out) A.u_out[o] = u;
        if (A.hard_out) A.hard_out[o] = bit ? 1.f : 0.f;
        if (A.probs_out) A.probs_out[o] = probs;
      }
    }
  }
  A.bits[i] = (uint16_t)w;
}

// Small reductions: per-block partial sums in a per-device scratch, finished in FIXED order
// (bit-reproducible).  The scratch is shared by all calls on a device: calls are expected to
// be stream-ordered (one optimisation loop per device), like the rest of a plan's work.
// the reward launch's instance for an action width (pmbrl_fast.h)
typedef void (*pm_reward_kernel_t)(const RolloutArgs);
static inline pm_reward_kernel_t pm_reward_kernel_for(int U, int k) {
  if (k <= 2) return U <= 4 ? pm_reward_all_kernel<4, 2> : (U <= 8 ? pm_reward_all_kernel<8, 2> : pm_reward_all_kernel<16, 2>);
  return U <= 4 ? pm_reward_all_kernel<4, PMBRL_MAX_TIP>
                : (U <= 8 ? pm_reward_all_kernel<8, PMBRL_MAX_TIP> : pm_reward_all_kernel<16, PMBRL_MAX_TIP>);
}
static inline int pm_norm_blocks(long long n) { return (int)std::max<long long>(1, std::min<long long>(PM_NORM_MAXB, (n + 255) / 256)); }
static int clip_adam_guarded_impl(void* stream, float* params_d, float* grads_d, float* exp_avg_d, float* exp_avg_sq_d,
                                  int64_t n, int64_t* step_d, double lr, double beta1, double beta2, double eps,
                                  double max_norm, float* norm_out_d, const int32_t* status_d, int32_t expect, bool norm_done,
                                  float* loss_out_d = nullptr, int n_loss_part = 0);
#define PM_RED_MAXB 128
__device__ double g_red_part[2][PM_RED_MAXB];
__device__ unsigned g_red_count;

__global__ __launch_bounds__(256) void pm_weighted_sum_kernel(const float* __restrict__ a,
                                                              const float* __restrict__ w,
                                                              long long n, float* __restrict__ out,
                                                              const int* __restrict__ nvalid, long long n_per_step) {
  __shared__ double sm[256];
  __shared__ unsigned ticket;
  if (nvalid) n = min(n, (long long)max(0, *nvalid) * n_per_step);   // truncated horizon
  double s = 0.0;
  // (eight elements' loads in flight at a time, added in the order a one-by-one loop adds them: as that loop -- load, wait,
  //  convert, add -- a thread's eight elements at C2 were eight memory round trips in a row, most of this launch)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 8 * stride) {
    float av[8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      // (beyond n: the last element's address, its value dropped below -- a load under a condition is waited for where
      //  the branches join, one at a time again)
      const long long j = min(i + u * stride, n - 1);
      av[u] = a[j];
      wv[u] = w[j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += i + u * stride < n ? (double)av[u] * (double)wv[u] : 0.0;
  }
  const double tot = pm_block_sum(s, sm);
  if (threadIdx.x == 0) {
    // (a device-scope store, then the ticket: a release FENCE here is a write-back of this XCD's whole L2 -- the reward
    //  launch's stash is still in it)
    __hip_atomic_store(&g_red_part[0][blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ticket = __hip_atomic_fetch_add(&g_red_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (ticket == gridDim.x - 1) {          // last block: every partial is published
    // fixed-shape tree over the published partials (deterministic), not a serial chain of
    // L2 round trips
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double t = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
      t += __hip_atomic_load(&g_red_part[0][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tt = pm_block_sum(t, sm);
    if (threadIdx.x == 0) {
      out[0] = (float)tt;
      __hip_atomic_store(&g_red_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// clip_grad_norm_ + Adam in two launches: (1) partial sums of g^2 over blocks of 256 elements (pmbrl_dw.h, pm_sq4_wave:
// the same partial sums the fused iteration's gradient reduction forms on the way), (2) every block adds the partials in
// the same order (identical norm everywhere) and updates its slice.
// Guarded form (pmbrl_clip_adam_guarded): block 0 also decides whether the step is taken (the rollout's status word says
// every horizon step completed) and, if so, advances the device-side step counter; pm_clip_adam_kernel (next launch)
// reads both.
__global__ __launch_bounds__(64) void pm_gradnorm_kernel(const float* __restrict__ g, long long n,
                                                         const int* __restrict__ status, int expect,
                                                         long long* __restrict__ step, int guarded) {
  if (guarded && blockIdx.x == 0 && threadIdx.x == 0) pm_adam_decide(status, expect, step);
  // (one block of 256 elements per workgroup up to PM_NORM_MAXB of them; a longer vector is walked in strides, the
  //  blocks' sums added in block order)
  double tot = 0.0;
  for (long long eb = blockIdx.x; eb * 256 < n; eb += gridDim.x) {
    const long long i0 = eb * 256 + 4 * (long long)threadIdx.x;
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i0 + 3 < n) t = *reinterpret_cast<const f32x4*>(g + i0);
    else
      for (int r = 0; r < 4; ++r) t[r] = i0 + r < n ? g[i0 + r] : 0.f;
    tot += pm_sq4_wave(t, i0, n);
  }
  if (threadIdx.x == 0) g_norm_part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void pm_clip_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           long long n, int n_part, float lr, float b1,
                                                           float b2, float omb1, float omb2, float eps,
                                                           float bc1, float bc2_sqrt, float max_norm,
                                                           float* __restrict__ norm_out,
                                                           const long long* __restrict__ step, int guarded,
                                                           double ln_b1, double ln_b2, double lr_d,
                                                           float* __restrict__ loss_out, int n_loss_part) {
  // The first four elements of this thread and the norm's partial sums are requested before anything is decided: as
  // "go? -> step counter -> partial sums (a loop of dependent loads) -> elements (a loop of four)" this launch was ten
  // memory round trips in a row around a few hundred instructions.
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float gq[4], mq[4], vq[4], pq[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    // (beyond n: the last element's address, never used -- loads under a condition are waited for one by one)
    const long long j = min(i0 + u * stride, n - 1);
    gq[u] = g[j];
    mq[u] = m[j];
    vq[u] = v[j];
    pq[u] = p[j];
  }
  const int lane = threadIdx.x & 63;
  double part[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const double x = g_norm_part[min(lane + 64 * u, PM_NORM_MAXB - 1)];
    part[u] = lane + 64 * u < n_part ? x : 0.0;
  }
  // the fused iteration's loss: the sums pm_dw_reduce's workgroups left, added in workgroup order by one wave (taken or
  // not, the step's loss is reported)
  if (loss_out && blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {
    double ls = 0.0;
    for (int b = lane; b < n_loss_part; b += 64) ls += g_loss_part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o);
    if (lane == 0) loss_out[0] = (float)ls;
  }
  if (guarded && !g_adam_go) return;   // the rollout failed: leave parameters and moments alone
  double step_size_d = lr_d / (double)bc1;
  if (step) {
    // guarded form: bias corrections 1 - beta^t of the device-side step counter as -expm1(t ln beta), the exponent formed
    // and evaluated in double (torch forms them in Python floats; in fp32 the sixth digit of the step size differs and a
    // five-iteration comparison of the moments notices), ln beta from the host -- two double-precision pow calls were a
    // thousand instructions at the head of every thread, expm1 is a third of that
    const double st = (double)step[0];
    bc1 = (float)-expm1(st * ln_b1);
    bc2_sqrt = (float)sqrt(-expm1(st * ln_b2));
    step_size_d = lr_d / -expm1(st * ln_b1);      // (torch: lr / bias_correction1 in Python floats, rounded once)
  }
  // the partial sums of squares, block b on lane b mod 64 in block order, and a fixed butterfly: the same bits in every wave
  // of every block
  double t = 0.0;
  {
#pragma unroll
    for (int u = 0; u < 4; ++u) t += part[u];                         // (absent: + 0)
    for (int b = lane + 256; b < n_part; b += 64) t += g_norm_part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  }
  const float norm = (float)sqrt(t);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  const float step_size = (float)step_size_d;
  for (long long i = i0;;) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * stride;
      if (j < n) {
        const float gi = gq[u] * coef;
        const float mi = mq[u] * b1 + omb1 * gi;
        const float vi = vq[u] * b2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        g[j] = gi;
        m[j] = mi;
        v[j] = vi;
        p[j] = pq[u] - step_size * (mi / denom);
      }
    }
    i += 4 * stride;
    if (i >= n) break;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = min(i + u * stride, n - 1);
      gq[u] = g[j];
      mq[u] = m[j];
      vq[u] = v[j];
      pq[u] = p[j];
    }
  }
}

// all weight / bias repacking of one forward call in ONE launch (blockIdx.y = job)
struct PackJob {
  const float* src;
  float* dst;
  int O, K, transpose, mult, is_bias;
  int planes;   // 0: fp32 fragments of 16-wide k-blocks; 2 / 3: bf16 pieces of K32 blocks (pmbrl_split.h), mult in K32 blocks
  int f16;      // pieces are fp16 instead of bf16
};
struct PackArgs {
  int n;
  int* wflag;       // fp16 pieces: set to `gen` when a weight is beyond +-65504 or not finite (nullptr: no check)
  int gen;
  int* status;      // reset to "no failure" by the same launch (nullptr: leave alone)
  PackJob job[6 * PM_MAXL];
};
__global__ void pm_pack_all(const PackArgs P) {
  if (P.status && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *P.status = 0x7fffffff;
  const PackJob j = P.job[blockIdx.y];
  if (j.is_bias) {
    const int O16 = (j.O + 15) / 16 * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < O16; i += gridDim.x * blockDim.x)
      j.dst[i] = i < j.O ? j.src[i] : 0.f;
    return;
  }
  const int n_out = j.transpose ? j.K : j.O, n_in = j.transpose ? j.O : j.K;
  const int n_ot = (n_out + 15) / 16;
  if (j.planes) {
    // dst[((ot*n_kb + kb)*planes + p)*64 + lane][8 bf16] = piece p of W[ot*16 + (lane&15)][kb*32 + 8*(lane>>4) + e]
    const int n_kb = ((n_in + 31) / 32 + j.mult - 1) / j.mult * j.mult;
    const size_t total = (size_t)n_ot * n_kb * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int lane = i & 63;
      const size_t tb = i >> 6;
      const int kb = tb % n_kb, ot = tb / n_kb;
      const int o = ot * 16 + (lane & 15);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kb * 32 + 8 * (lane >> 4) + e;
        v[e] = (o < n_out && k < n_in) ? (j.transpose ? j.src[(size_t)k * j.K + o] : j.src[(size_t)o * j.K + k]) : 0.f;
      }
      for (int p = 0; p < j.planes; ++p) {
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (j.f16) {
            if (p == 0 && P.wflag && (!(fabsf(v[2 * e]) <= 65504.f) || !(fabsf(v[2 * e + 1]) <= 65504.f)))
              atomicMax(P.wflag, P.gen);
            // (the low piece is stored scaled by 2^11: pmbrl_split.h, PM_F16_LO_SCALE)
            const float sc = p ? PM_F16_LO_SCALE : 1.f;
            w[e] = pm_pk_f16(v[2 * e] * sc, v[2 * e + 1] * sc);
            const pm_f32x2 f = pm_unpk_f16(w[e]);
            v[2 * e] -= f[0];
            v[2 * e + 1] -= f[1];
          } else {
            w[e] = pm_pk_bf16(v[2 * e], v[2 * e + 1]);
            v[2 * e] -= pm_bf_lo(w[e]);
            v[2 * e + 1] -= pm_bf_hi(w[e]);
          }
        }
        *reinterpret_cast<uint4*>(j.dst + ((tb * j.planes + p) * 64 + lane) * 4) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return;
  }
  const int n_kb = ((n_in + 15) / 16 + j.mult - 1) / j.mult * j.mult;
  const size_t total = (size_t)n_ot * n_kb * 256;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int jj = i & 3, lane = (i >> 2) & 63;
    const size_t tb = i >> 8;
    const int kb = tb % n_kb, ot = tb / n_kb;
    const int o = ot * 16 + (lane & 15);
    const int k = kb * 16 + 4 * (lane >> 4) + jj;
    float v = 0.f;
    if (o < n_out && k < n_in)
      v = j.transpose ? j.src[(size_t)k * j.K + o] : j.src[(size_t)o * j.K + k];
    j.dst[i] = v;
  }
}


// external moment matching (groups larger than a workgroup's rows): one workgroup of PM_MM_NW
// waves per group, rows in HBM; every wave takes a slice of the rows (pmbrl_mm.h, multi-wave
// forms).  Widths without a compile-time instantiation run the general code on wave 0.
#define PM_MM_NW 16
// (widths beyond the compile-time instances -- D > 6 -- run the one-wave routine: one wave's scratch; sixteen
//  of them would be 446 KB at D = 32)
__host__ __device__ inline int pm_mm_kernel_scratch_waves(int D) { return D <= 6 ? PM_MM_NW : 1; }
__host__ __device__ inline size_t pm_mm_kernel_doubles(int D) {
  return (size_t)pm_mm_kernel_scratch_waves(D) * pm_mm_scratch_doubles(D) + (size_t)PM_MM_NW * 256;
}
#define PM_MM_MW_SWITCH(D, CALL, ELSE) \
  switch (D) {                         \
    case 1: { CALL(1); break; }        \
    case 2: { CALL(2); break; }        \
    case 3: { CALL(3); break; }        \
    case 4: { CALL(4); break; }        \
    case 5: { CALL(5); break; }        \
    case 6: { CALL(6); break; }        \
    default: { ELSE; }                 \
  }
// the noise table of split groups in a launch of its own (pmbrl_mm.h: pm_ztab_block; the register-resident family's pack
// launch forms it with extra workgroups instead)
__global__ __launch_bounds__(256) void pm_mm_ztable_kernel(const ZtabArgs Z) { pm_ztab_block(Z, (int)blockIdx.x, (int)threadIdx.x); }
static ZtabArgs pm_ztab_args(const pmbrl_plan* p, const RolloutArgs& A) {
  ZtabArgs Z;
  Z.zmm = A.zmm; Z.tab = const_cast<double*>(A.mm_ztab);
  Z.pack = (p->M <= 64 && p->cfg.D <= 8) ? 1 : 0;
  Z.H = p->cfg.H; Z.G = p->G; Z.M = p->M; Z.D = p->cfg.D; Z.Bg = A.Bg;
  Z.per_step = (A.flags & PMBRL_FLAG_ZMM_PER_STEP) ? 1 : 0;
  Z.row_off = A.row_off;
  Z.n_blocks = pm_ztab_blocks(Z.H, Z.G, Z.pack);
  return Z;
}

__global__ __launch_bounds__(PM_MM_NW * 64) void pm_mm_fwd_kernel(RolloutArgs A, int t) {
  extern __shared__ __attribute__((aligned(16))) double mmscr[];
  const int gi = blockIdx.x, lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* part = mmscr + (size_t)pm_mm_kernel_scratch_waves(A.D) * pm_mm_scratch_doubles(A.D);
  const int r0 = gi * A.M;
  const bool ins = (A.flags & PMBRL_FLAG_INFER_NS) != 0;
  const int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags);
  const float* zmm = pm_zbase(A.zmm, A.D, t, A.Bg, A.flags);
  const float* zrr = pm_zbase(A.zrr, 1, t, A.Bg, A.flags);
  if (A.flags & PMBRL_FLAG_MM_STATES) {
    const float* s = A.xt + ((size_t)t * A.B + r0) * A.D;
    float* out = A.states + ((size_t)(t + 1) * A.B + r0) * A.D;
    bool ok = true;
#define PM_CALL(DD) ok = pm_mm_fwd_mw<DD>(s, A.D, A.M, zmm, A.D, zrow0, A.Bg, out, A.D, mmscr, part, PM_MM_NW, wid, lane)
    if (ins) {   // infer_noise_variables: the general single-wave routine
      if (wid == 0) ok = pm_mm_fwd(s, A.D, A.M, A.D, zmm, A.D, zrow0, A.Bg, true, out, A.D, mmscr, lane);
    } else {
      PM_MM_MW_SWITCH(A.D, PM_CALL,
                      if (wid == 0) ok = pm_mm_fwd(s, A.D, A.M, A.D, zmm, A.D, zrow0, A.Bg, false, out, A.D, mmscr, lane))
    }
#undef PM_CALL
    if (!ok && threadIdx.x == 0) atomicMin(A.status, t);
  }
  if (A.flags & PMBRL_FLAG_MM_REWARDS) {
    __syncthreads();
    bool ok = true;
    if (ins) {
      if (wid == 0) ok = pm_mm_fwd(A.rt + (size_t)t * A.B + r0, 1, A.M, 1, zrr, 1, zrow0, A.Bg, true,
                                   A.rewards + (size_t)t * A.B + r0, 1, mmscr, lane);
    } else {
      ok = pm_mm_fwd_mw<1>(A.rt + (size_t)t * A.B + r0, 1, A.M, zrr, 1, zrow0, A.Bg,
                           A.rewards + (size_t)t * A.B + r0, 1, mmscr, part, PM_MM_NW, wid, lane);
    }
    if (!ok && threadIdx.x == 0) atomicMin(A.status, t);
  }
}
// adjoint: (gx_carry = dL/dx_{t+1}, grad_rewards[t]) -> (gx_carry = dL/dx~, gr_tilde[t] = dL/dr~)
__global__ __launch_bounds__(PM_MM_NW * 64) void pm_mm_bwd_kernel(RolloutArgs A, int t, float* gr_tilde) {
  extern __shared__ __attribute__((aligned(16))) double mmscr[];
  const int gi = blockIdx.x, lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* part = mmscr + (size_t)pm_mm_kernel_scratch_waves(A.D) * pm_mm_scratch_doubles(A.D);
  const int r0 = gi * A.M;
  if (A.nvalid && t >= *A.nvalid) return;   // a step the forward sweep did not complete
  const bool ins = (A.flags & PMBRL_FLAG_INFER_NS) != 0;
  const int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags);
  const float* zmm = pm_zbase(A.zmm, A.D, t, A.Bg, A.flags);
  const float* zrr = pm_zbase(A.zrr, 1, t, A.Bg, A.flags);
  if (A.flags & PMBRL_FLAG_MM_STATES) {
    float* g = A.gx_carry + (size_t)r0 * A.D;
    const float* s = A.xt + ((size_t)t * A.B + r0) * A.D;
    // in place (g -> g): every read of g happens before the first write (pmbrl_mm.h)
#define PM_CALL(DD) pm_mm_bwd_mw<DD>(s, A.D, A.M, zmm, A.D, zrow0, A.Bg, g, A.D, g, A.D, mmscr, part, PM_MM_NW, wid, lane)
    if (ins) {
      if (wid == 0) pm_mm_bwd(s, A.D, A.M, A.D, zmm, A.D, zrow0, A.Bg, true, g, A.D, g, A.D, mmscr, lane);
    } else {
      PM_MM_MW_SWITCH(A.D, PM_CALL,
                      if (wid == 0) pm_mm_bwd(s, A.D, A.M, A.D, zmm, A.D, zrow0, A.Bg, false, g, A.D, g, A.D, mmscr, lane))
    }
#undef PM_CALL
  }
  const float* gsrc = A.grad_rewards + (size_t)t * A.B + r0;
  float* gdst = gr_tilde + (size_t)t * A.B + r0;
  if (A.flags & PMBRL_FLAG_MM_REWARDS) {
    __syncthreads();
    if (ins) {
      if (wid == 0) pm_mm_bwd(A.rt + (size_t)t * A.B + r0, 1, A.M, 1, zrr, 1, zrow0, A.Bg, true, gsrc, 1, gdst, 1, mmscr, lane);
    } else {
      pm_mm_bwd_mw<1>(A.rt + (size_t)t * A.B + r0, 1, A.M, zrr, 1, zrow0, A.Bg, gsrc, 1, gdst, 1, mmscr, part,
                      PM_MM_NW, wid, lane);
    }
  } else {
    for (int i = threadIdx.x; i < A.M; i += PM_MM_NW * 64) gdst[i] = gsrc[i];
  }
}

// wide states (6 < D <= 32), rows of a group staged in LDS (pmbrl_mm_wide.h); the forward leaves the factor block of
// (step, group) in fac_all for the adjoint
__global__ __launch_bounds__(PM_MMW_NT) void pm_mmw_fwd_kernel(RolloutArgs A, int t, double* fac_all) {
  extern __shared__ __attribute__((aligned(16))) double mmscr[];
  const int gi = blockIdx.x, r0 = gi * A.M;
  const int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags);
  const float* zmm = pm_zbase(A.zmm, A.D, t, A.Bg, A.flags);
  const float* s = A.xt + ((size_t)t * A.B + r0) * A.D;
  float* out = A.states + ((size_t)(t + 1) * A.B + r0) * A.D;
  double* fac = fac_all + ((size_t)t * gridDim.x + gi) * pm_mmw_fac_doubles(A.D);
  const bool ok = pm_mmw_fwd(s, A.M, A.D, zmm, zrow0, A.Bg, out, fac, mmscr,
                             (A.prof && gi == 0) ? A.prof + (size_t)t * 32 : nullptr);
  if (!ok && threadIdx.x == 0) atomicMin(A.status, t);
}
__global__ __launch_bounds__(PM_MMW_NT) void pm_mmw_bwd_kernel(RolloutArgs A, int t, const double* fac_all) {
  extern __shared__ __attribute__((aligned(16))) double mmscr[];
  const int gi = blockIdx.x, r0 = gi * A.M;
  if (A.nvalid && t >= *A.nvalid) return;   // a step the forward sweep did not complete
  const int zrow0 = pm_zrow0(t, A.row_off + r0, A.flags);
  const float* zmm = pm_zbase(A.zmm, A.D, t, A.Bg, A.flags);
  const float* s = A.xt + ((size_t)t * A.B + r0) * A.D;
  float* g = A.gx_carry + (size_t)r0 * A.D;
  const double* fac = fac_all + ((size_t)t * gridDim.x + gi) * pm_mmw_fac_doubles(A.D);
  pm_mmw_bwd(s, A.M, A.D, zmm, zrow0, A.Bg, g, g, fac, mmscr);
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int net_plan(const pmbrl_mlp& m, NetPlan& n, int in_expect, int out_expect,
                    const char* name) {
  if (m.n_layers < 1 || m.n_layers > PM_MAXL)
    return fail(-2, std::string(name) + ": n_layers out of range");
  n.nl = m.n_layers;
  size_t off = 0;
  for (int i = 0; i <= n.nl; ++i) {
    if (m.dims[i] < 1 || m.dims[i] > 4096) return fail(-2, std::string(name) + ": bad layer width");
    n.dim[i] = m.dims[i];
    n.nt[i] = (m.dims[i] + 15) / 16;
  }
  if (n.dim[0] != in_expect || n.dim[n.nl] != out_expect)
    return fail(-2, std::string(name) + ": input/output width does not match D/U");
  for (int l = 0; l < n.nl; ++l) {
    n.keep[l] = (l < n.nl - 1) ? m.keep[l] : 1.f;
    if (l < n.nl - 1 && !(n.keep[l] > 0.f)) return fail(-2, std::string(name) + ": keep must be > 0");
    n.w_off[l] = off;
    off += (size_t)n.dim[l + 1] * n.dim[l];
    n.b_off[l] = off;
    off += n.dim[l + 1];
  }
  n.n_params = off;
  return 0;
}

template <int RT>
static int set_attr(size_t lds) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd<RT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd<RT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return 0;
}

// dW wave blocks: balanced splits of every layer's output tile grid into <= 4 x 7 tile
// blocks, dealt to the four SIMDs (waves w and w+4 share SIMD w) by decreasing size so that
// every SIMD issues the same number of MFMAs.  Returns the number of blocks; `blocks` comes back
// sorted by wave, wave w owning [wave_first[w], wave_first[w+1]).
static int build_dw_blocks(int nl, const int* nt, std::vector<DwBlock>& blocks, int* wave_first, int tn_max = PM_DW_TN,
                           const bool* skip = nullptr) {
  blocks.clear();
  auto split = [](int n, int maxsz, std::vector<std::pair<int, int>>& out) {
    const int parts = (n + maxsz - 1) / maxsz;
    int lo = 0;
    for (int i = 0; i < parts; ++i) {
      const int sz = n / parts + (i < n % parts ? 1 : 0);
      out.push_back({lo, sz});
      lo += sz;
    }
  };
  for (int l = 0; l < nl; ++l) {
    if (skip && skip[l]) continue;      // a wide layer: pm_dw_wide_kernel
    std::vector<std::pair<int, int>> os, is;
    split(nt[l + 1], PM_DW_TM, os);
    split(nt[l], tn_max, is);
    for (auto& o : os)
      for (auto& i : is) {
        DwBlock b;
        b.layer = (int16_t)l;
        b.ot0 = (int16_t)o.first;
        b.n_ot = (int16_t)o.second;
        b.it0 = (int16_t)i.first;
        b.n_it = (int16_t)i.second;
        b.pad = 0;
        blocks.push_back(b);
      }
  }
  const int n_blocks = (int)blocks.size();
  // cost of a block = MFMAs per chunk in the shape class it runs in (see pm_dw_kernel)
  auto cost = [tn_max](const DwBlock& b) {
    const int ni = b.n_ot <= 1 ? 1 : (b.n_ot <= 3 ? 3 : 4);
    const int nj = b.n_it <= 1 ? 1 : (b.n_it <= 4 ? 4 : (b.n_it <= 6 ? 6 : 7));
    return ni * (tn_max <= 4 && nj > 1 ? 4 : nj);
  };
  std::stable_sort(blocks.begin(), blocks.end(),
                   [&](const DwBlock& a, const DwBlock& b) { return cost(a) > cost(b); });
  std::vector<DwBlock> per_wave[PM_DW_NW];
  long wave_load[PM_DW_NW] = {0};
  // waves w and w+4 of a workgroup share SIMD w (measured: pairing (2k, 2k+1) instead is 5% slower)
  for (const DwBlock& b : blocks) {
    int best_simd = 0;
    for (int sm = 1; sm < 4; ++sm)
      if (wave_load[sm] + wave_load[sm + 4] < wave_load[best_simd] + wave_load[best_simd + 4]) best_simd = sm;
    const int w = wave_load[best_simd] <= wave_load[best_simd + 4] ? best_simd : best_simd + 4;
    per_wave[w].push_back(b);
    wave_load[w] += cost(b);
  }
  blocks.clear();
  // the small blocks are latency-bound (a handful of MFMAs per chunk): the second wave of a SIMD runs
  // its FIRST, while the SIMD's other wave keeps the MFMA pipe busy with its big block
  for (int w = 0; w < PM_DW_NW; ++w) {
    wave_first[w] = (int)blocks.size();
    if (w >= 4) std::reverse(per_wave[w].begin(), per_wave[w].end());
    for (const DwBlock& b : per_wave[w]) blocks.push_back(b);
  }
  wave_first[PM_DW_NW] = (int)blocks.size();
  return n_blocks;
}

extern "C" int pmbrl_plan_create(const pmbrl_config* cfg, int device, pmbrl_plan** out) {
  if (!cfg || !out) return fail(-1, "null argument");
  const pmbrl_config& c = *cfg;
  if (c.B < 1 || c.D < 1 || c.U < 1 || c.H < 1) return fail(-2, "B, D, U, H must be >= 1");
  if (c.D > 32 || c.U > 16) return fail(-2, "supported widths: D <= 32, U <= 16");
  if (c.reward.k < 1 || c.reward.k > PMBRL_MAX_TIP) return fail(-2, "reward.k out of range");
  pmbrl_plan* p = new pmbrl_plan();
  memset(p, 0, sizeof(*p));
  p->cfg = c;
  if (p->cfg.B_global <= 0) p->cfg.B_global = c.B;
  p->device = device;
  p->replay = 1;
  if (const char* e = getenv("PMBRL_REPLAY")) p->replay = std::max(0, std::min(2, atoi(e)));
  if (c.n_pol_angle < 0 || c.n_pol_angle > PMBRL_MAX_ANGLE || c.n_dyn_angle < 0 || c.n_dyn_angle > PMBRL_MAX_ANGLE) {
    delete p;
    return fail(-2, "n_pol_angle / n_dyn_angle out of range");
  }
  const bool angles = c.n_pol_angle > 0 || c.n_dyn_angle > 0;
  int rc = net_plan(c.pol, p->pol, c.D + c.n_pol_angle, 2 * c.U, "policy");
  const bool gmm = c.dyn_components > 1;
  if (c.dyn_components < 0 || c.dyn_components > PMBRL_MAX_COMP) {
    delete p;
    return fail(-2, "dyn_components out of range (<= PMBRL_MAX_COMP)");
  }
  // (mixture head: n D means, n D log-stds, n logits, one log-temperature -- examples/deep_pilco_mm.py:117-121)
  if (rc == 0)
    rc = net_plan(c.dyn, p->dyn, c.D + c.U + c.n_dyn_angle, gmm ? (2 * c.D + 1) * c.dyn_components + 1 : 2 * c.D,
                  "dynamics");
  if (rc) { delete p; return rc; }

  // LDS leading dimension: widest activation (any layer of either net), +8 so that
  // LD % 16 == 8 (conflict-free ds_read_b128 for the MFMA B operand)
  int maxnt = 1;
  for (int i = 0; i <= p->pol.nl; ++i) maxnt = std::max(maxnt, p->pol.nt[i]);
  for (int i = 0; i <= p->dyn.nl; ++i) maxnt = std::max(maxnt, p->dyn.nt[i]);
  p->LD = maxnt * 16 + 8;

  const bool mm = (c.flags & (PMBRL_FLAG_MM_STATES | PMBRL_FLAG_MM_REWARDS)) != 0;
  const size_t lds_cap = 160 * 1024;
  // infer_noise_variables (utils/rollout.py:6-17, not a default anywhere) lives in the general
  // single-wave moment-matching routines only: general kernel family, mm_mode 1 or 2
  p->fast = !(c.flags & (PMBRL_FLAG_FORCE_GENERIC | PMBRL_FLAG_INFER_NS | PMBRL_FLAG_POL_MASKS_PER_STEP |
                         PMBRL_FLAG_DYN_MASKS_PER_STEP)) && !angles && !gmm &&
            pm_fast_net_ok(p->pol.dim, p->pol.nt, p->pol.nl) &&
            pm_fast_net_ok(p->dyn.dim, p->dyn.nt, p->dyn.nl);
  // general family on split operands (pmbrl_gsplit.h): two fp16 pieces forward / two bf16 pieces in the adjoint;
  // an activation buffer row is then LD 16-bit elements per piece plane, LD = 16 (mod 32)
  const bool gsplit = !p->fast && c.precision == PMBRL_PREC_SPLIT_F16 && !getenv("PMBRL_FORCE_F32");
  const int LD_generic = gsplit ? (maxnt * 16 + 31) / 32 * 32 + 16 : p->LD;
  // stage sizes (k-blocks) of the weight stream: every streamed layer is padded to a whole number
  // of stage PAIRS (CA + CB k-blocks); pick, among the instantiated pairs, the one that pads least
  struct StagePair { int ca, cb; };
  // split-bf16 precision: offered by the fast family for workgroups of up to 32 rows
  const bool want_split = (c.precision == PMBRL_PREC_SPLIT || c.precision == PMBRL_PREC_SPLIT_F16) && p->fast &&
                          !getenv("PMBRL_FORCE_F32");
  // (64-row workgroups: two fp16 piece planes fit the LDS next to everything else, three bf16 planes do not)
  // 64-row workgroups on split operands: two fp16 piece planes fit the LDS (three bf16 planes do not), and
  // only the shape-specialised instance keeps its inline-asm loads clear of register-allocator copies
  // (tools/check_inflight.py; the generic 64-row split instances fail that lint and are not compiled):
  // the double cart-pole shape with in-kernel moment matching (PM_SPLIT_SHAPED_RT4), fp32 otherwise
  bool rt4_split = c.precision == PMBRL_PREC_SPLIT_F16 && mm && c.D == 6 && c.U == 1 && p->pol.nl == 3 &&
                   p->dyn.nl == 3 && p->pol.nt[1] == 13 && p->pol.nt[2] == 13 && p->dyn.nt[1] == 13 &&
                   p->dyn.nt[2] == 13 && !(c.flags & PMBRL_FLAG_NO_SHAPED);
  auto prec_for = [&](int RT) {
    if (!p->fast) return gsplit ? (int)PMBRL_PREC_SPLIT_F16 : 0;
    return (want_split && (RT <= 2 || rt4_split)) ? (int)c.precision : 0;
  };
  // (kdiv: 16-wide k-blocks per stage unit -- 1 on the fp32 path, 2 = one K32 block on the split path)
  auto stream_work = [&](int m, int kdiv) {
    long w = 0;
    const NetPlan* nets[2] = {&p->pol, &p->dyn};
    for (const NetPlan* n : nets)
      for (int l = 1; l <= n->nl - 2; ++l) {
        const int ki = (n->nt[l] + kdiv - 1) / kdiv, ko = (n->nt[l + 1] + kdiv - 1) / kdiv;
        w += (long)n->nt[l + 1] * ((ki + m - 1) / m * m) + (long)n->nt[l] * ((ko + m - 1) / m * m);
      }
    return w;
  };
  auto stages_for = [&](int RT) {
    static const StagePair cand1[] = {{8, 8}, {7, 6}, {4, 4}, {2, 2}}, cand2[] = {{4, 4}, {4, 3}, {2, 2}}, cand4[] = {{2, 2}, {1, 1}};
    static const StagePair cands[] = {{4, 3}, {1, 1}};
    const bool sp = prec_for(RT) != 0;
    const StagePair* cand = sp ? cands : RT == 1 ? cand1 : (RT == 2 ? cand2 : cand4);
    const int nc = sp ? 2 : RT == 1 ? 4 : (RT == 2 ? 3 : 2);
    const int kdiv = sp ? 2 : 1;
    StagePair best = cand[0];
    for (int i = 1; i < nc; ++i)
      if (stream_work(cand[i].ca + cand[i].cb, kdiv) < stream_work(best.ca + best.cb, kdiv)) best = cand[i];
    return best;
  };
  // split path: elements per row of a bf16 piece plane = padded K + 16 (conflict-free ds_read_b128)
  auto ldb_for = [&](int RT) {
    if (!prec_for(RT)) return 0;
    if (!p->fast) return LD_generic;
    const StagePair sp = stages_for(RT);
    const int m = sp.ca + sp.cb;
    return ((maxnt + 1) / 2 + m - 1) / m * m * 32 + 16;
  };
  auto ld_for = [&](int RT) {
    if (!p->fast) return LD_generic;
    if (prec_for(RT))   // floats per row of a buffer holding the piece planes (3 bf16 / 2 fp16 forward, 2 bf16 adjoint)
      return ((prec_for(RT) == PMBRL_PREC_SPLIT_F16 ? 2 : 3) * ldb_for(RT) / 2 + 3) / 4 * 4;
    const StagePair sp = stages_for(RT);
    const int m = sp.ca + sp.cb;
    return (maxnt + m - 1) / m * m * 16 + 8;   // room for the zero K padding
  };
  auto lds_need = [&](int RT, int mmd, int parts = 1) {
    p->LD = ld_for(RT);
    if (p->fast) {
      // in-kernel moment matching: one wave per whole group of the workgroup (a group split over `parts`
      // workgroups: wave 0, and the whole group's rows in LDS)
      const int mw = parts > 1 ? 1 : (mmd && p->M <= 16 * RT) ? std::min(PF_NW, std::max(1, 16 * RT / p->M)) : PF_NW;
      return pm_fast_lds_floats(16 * RT, p->LD, c.D, c.U, RT, p->pol.nt, p->pol.nl, p->dyn.nt,
                                p->dyn.nl, mmd, prec_for(RT), mw, parts > 1 ? (parts > 8 ? 16 : p->M) : 0, parts > 1 ? c.H : 0) * sizeof(float);
    }
    return pm_lds_floats(16 * RT, p->LD, c.D, c.U, RT, mmd, p->inplace != 0) * sizeof(float);
  };
  p->G = 1;
  p->M = c.B;
  p->mm_mode = 0;
  p->mm_parts = 1;
  if (mm) {
    p->G = c.mm_groups > 0 ? c.mm_groups : 1;
    if (c.B % p->G) { delete p; return fail(-2, "B must be divisible by mm_groups"); }
    p->M = c.B / p->G;
    if (p->M < 2) { delete p; return fail(-2, "moment matching needs >= 2 rows per group"); }
    if (rt4_split && p->fast && lds_need(4, c.D) > lds_cap) rt4_split = false;   // 64-row workgroups: fp32 then
    // groups spread over ranks: the statistics cross devices between the two halves of the moment matching, so it
    // stays outside the sweep kernels (mm_mode 2: one sweep launch per step)
    p->span = c.mm_span_rows > 0;
    if (p->span) {
      if (c.mm_span_ranks < 1 || c.mm_span_rank < 0 || c.mm_span_rank >= c.mm_span_ranks) {
        delete p; return fail(-2, "mm_span_ranks / mm_span_rank out of range");
      }
      if (c.mm_span_offset < 0 || c.mm_span_offset + p->M > c.mm_span_rows) {
        delete p; return fail(-2, "mm_span_offset + rows per group on this rank exceeds mm_span_rows");
      }
      if ((long long)p->G * c.mm_span_rows > p->cfg.B_global) {
        delete p; return fail(-2, "mm_groups x mm_span_rows exceeds B_global");
      }
    }
    // in-kernel if a whole number of groups fits a workgroup's row tiles and LDS
    p->mm_mode = 2;
    for (int RT : {1, 2, 4}) {
      if (p->span) break;
      const int R = 16 * RT;
      if (p->M <= R && lds_need(RT, c.D) <= lds_cap) {
        p->mm_mode = 1;
        p->RT = RT;
        p->rows_per_wg = (R / p->M) * p->M;
        break;
      }
    }
    // A group of 33..64 rows in ONE 64-row workgroup is throughput-bound on its CU (52 tile epilogues and
    // 52 x 21 MFMAs per layer: 67 k cycles per step at the double cart-pole shape) while most of the chip idles.
    // Split over two workgroups of <= 32 rows a step is 26 k cycles of GEMMs; the halves exchange their
    // rows through HBM and meet at a group-local flag barrier (pm_group_sync), and each factors the whole
    // group redundantly (the d x d fp64 chain is serial anyway).  Needs every workgroup resident (checked
    // here like the device-wide barrier form).  PMBRL_MM_PARTS=1 keeps one workgroup per group.
    p->mm_parts = 1;
    {
      const char* e = getenv("PMBRL_MM_PARTS");
      int cus = 0;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      const int max_wg = std::min(cus, 1024);
      // By default where a group needs more than 16 rows: 64-row workgroups (M = 33..64) are throughput-bound on
      // their CU, and a 32-row workgroup (M = 17..32) still takes 30 k cycles per step against 24 k for two
      // 16-row ones with the exchange (measured on the cart-pole shapes, also with unshaped networks) -- and
      // where a group fits no workgroup at all (more than 64 rows, or LDS): the device-wide-barrier form would
      // take over otherwise (D = 6, 2 x 256 hidden units, 50-row groups: 8.9 ms against 4.0 ms per iteration).
      // Groups of up to 128 rows (beyond 64 the one-wave routines walk the rows in strides).  The fewest parts
      // that bring a part down to 16 rows while every workgroup stays resident, else to 32 rows.
      // PMBRL_MM_PARTS=n: n parts wherever that can be done (tests); 1: whole groups.
      const bool big = !p->span && ((p->mm_mode == 1 && p->RT >= 2) || p->mm_mode == 2 || (e && p->mm_mode == 1));
      auto fits = [&](int parts, int max_rows, bool resident = true) {
        const int rpw = (p->M + parts - 1) / parts;      // the last part takes what is left of the group
        return rpw <= max_rows && (parts - 1) * rpw < p->M && (!resident || p->G * parts <= max_wg);
      };
      int want = 0;
      // Round 5: 16-row parts even where they are more workgroups than CUs (the double cart-pole shape: 100 groups of 50
      // rows = 400 parts of <= 13 rows) IF the register-resident family takes the shape (pmbrl_reg_mm.h) -- its 16-row step
      // is a third of the latency-optimised family's 32-row step, so two launches of 200 workgroups (batches of whole groups:
      // what the statistics exchange needs resident together is one GROUP's workgroups) beat one launch of 200 32-row ones
      bool batched = false;
      if (e) want = atoi(e);
      else {
        for (int parts = 2; parts <= 8 && !want; ++parts) if (fits(parts, 16)) want = parts;
        if (!want && max_wg >= 64 && pm_reg_mm_shape_ok(p, prec_for(1)) && !getenv("PMBRL_MM_NO_BATCH"))
          for (int parts = 2; parts <= 8 && !want; ++parts)
            if (fits(parts, 16, false) && p->G * parts <= 1024) { want = parts; batched = true; }
        for (int parts = 2; parts <= 8 && !want; ++parts) if (fits(parts, 32)) want = parts;
      }
      p->mm_gpb = 0;
      if (p->fast && want >= 2 && want <= 8 && big && p->M <= 128 && (batched ? fits(want, 16, false) : fits(want, 32)) &&
          (c.flags & PMBRL_FLAG_MM_STATES)) {
        const int rpw = (p->M + want - 1) / want;
        const int rt = rpw <= 16 ? 1 : 2;
        if (lds_need(rt, c.D, want) <= lds_cap) {
          p->mm_mode = 1;
          p->mm_parts = want;
          p->RT = rt;
          p->rows_per_wg = rpw;
          if (batched) {      // balanced batches of whole groups, each batch's workgroups resident together
            const int nb = (p->G * want + max_wg - 1) / max_wg;
            p->mm_gpb = (p->G + nb - 1) / nb;
            // (the register-resident family deals hardware workgroups in blocks of 8 groups -- a group's parts on one XCD --
            //  and launches whole blocks: a batch rounded UP to blocks must still be resident together, or the parts of
            //  the block that straddles the chip spin until other workgroups finish a whole sweep.  G = 97 in 5 parts on
            //  256 CUs: 49 groups -> 280 workgroups; 48 -> 240)
            if (!(getenv("PMBRL_XCH_XCD") && atoi(getenv("PMBRL_XCH_XCD")) == 0)) {
              const int cap = max_wg / want / 8 * 8;
              if (cap >= 8) p->mm_gpb = std::min((p->mm_gpb + 7) / 8 * 8, cap);
            }
          }
        }
      }
      // A group beyond 8 parts of 32 rows -- ONE group over the whole batch, the reference's default (mm_groups=None,
      // examples/deep_pilco_mm.py:31): 16-row parts whose sums travel over two levels (pm_xch_get_tree), every workgroup
      // resident.  Only where the instance that exchanges sums runs (compile-time state width, the cart-pole shape:
      // PM_SPLIT_SHAPED_CASES) -- the rows + flags form of the generic instances keeps the whole group's rows in LDS.
      // 2 500 rows: 157 parts, 13 collectors; against the device-wide-barrier form (every workgroup re-reads and
      // re-reduces the whole group's rows every step) 3.0 -> see profiles/r04*_single_group.txt.  PMBRL_MM_TREE=0: off.
      const bool shaped_mm1 = p->fast && prec_for(1) == PMBRL_PREC_SPLIT_F16 && c.D == 4 && c.U == 1 && p->pol.nl == 3 &&
                              p->dyn.nl == 3 && p->pol.nt[1] == 13 && p->pol.nt[2] == 13 && p->dyn.nt[1] == 13 && p->dyn.nt[2] == 13 &&
                              !(c.flags & PMBRL_FLAG_NO_SHAPED) && !(getenv("PMBRL_LDS_TILES") && atoi(getenv("PMBRL_LDS_TILES")) == 0) &&
                              !(getenv("PMBRL_MM_XCH") && atoi(getenv("PMBRL_MM_XCH")) == 0);
      if (p->mm_mode == 2 && !p->span && !e && shaped_mm1 && (c.flags & PMBRL_FLAG_MM_STATES) &&
          !(getenv("PMBRL_MM_TREE") && atoi(getenv("PMBRL_MM_TREE")) == 0)) {
        const int parts = (p->M + 15) / 16;
        if (parts > 8 && (long long)p->G * parts <= max_wg && ld_for(1) == 240) {
          const size_t tiles = (size_t)PF_NW * pm_lds_tile_floats(14, 2, 16 * ((16 * 13 - 32 * 6) / 8)) * sizeof(float);
          if (lds_need(1, c.D, parts) + tiles + 16 <= lds_cap) {
            p->mm_mode = 1;
            p->mm_parts = parts;
            p->RT = 1;
            p->rows_per_wg = 16;
            int fan = 2;
            while (fan * fan < parts) ++fan;
            // ONE group (mm_groups=None): two collectors per XCD -- 2 fan parts of the group on every XCD, so that the
            // members' hop to their collector stays inside that XCD's L2 (the register-resident family deals its workgroups
            // that way: pmbrl_reg.h, pr_wg; the latency-optimised family walks the same slots in dispatch order)
            if (p->G == 1 && parts <= 256) fan = (parts + 15) / 16;
            p->mm_fan = fan;
          }
        }
      }
    }
  }
  if (p->mm_mode != 1) {
    rt4_split = false;
    int RT = 1;
    if (c.rows_per_wg_hint >= 64) RT = 4;
    else if (c.rows_per_wg_hint >= 32) RT = 2;
    else if (c.rows_per_wg_hint == 0) {
      // fill the chip first: >= 2 waves of workgroups over 256 CUs before growing tiles
      if ((c.B + 15) / 16 > 1024) RT = 2;
      if ((c.B + 31) / 32 > 1024) RT = 4;
      // general family (wide networks, throughput-bound): 32-row workgroups halve the weight traffic
      // per row as soon as they still give every CU two workgroups' worth of rows
      if (!p->fast && RT == 1 && (c.B + 31) / 32 >= 512) RT = 2;
    }
    while (RT > 1 && lds_need(RT, 0) > lds_cap) RT /= 2;
    // General family on split operands: 64-row workgroups with IN-PLACE layers (one activation buffer, pmbrl_gsplit.h:
    // gemm_layer_inplace_s) where two buffers of 64 rows do not fit the LDS.  Built to stream a 512 x 512 layer's weights
    // once per 64 rows instead of once per 32; measured at the C5 shape (profiles/r03b_inplace_*): a hidden layer's K loop
    // does become MFMA-bound (25 k cycles per 64 rows), but with every wave's epilogues behind one barrier they no
    // longer overlap another wave's MFMAs (+36 k), and the stash stores of a policy layer, issued back to back, cost
    // 26 k cycles more -- 46.7 ms per iteration against 45.3 ms for the two-buffer 32-row form.  So it is taken only
    // on request (rows_per_wg_hint >= 64): widths <= 512, every narrow width (2U, 2D, D + U) <= 64, no mixture head,
    // no angle features.
    if (gsplit && !angles && !gmm && maxnt <= 32 && 2 * c.D <= PM_IP_NOFF && 2 * c.U <= PM_IP_NOFF &&
        c.D + c.U <= PM_IP_NOFF && RT < 4) {
      // every hidden layer 512 wide: the in-place layers of pmbrl_wide.h (compile-time shape, lean epilogue) -- taken by
      // default as soon as 64-row workgroups still give every CU one (PMBRL_WIDE=0: the generic in-place form on
      // request only, as before)
      bool wide = true;
      for (int i = 1; i < p->pol.nl; ++i) wide = wide && p->pol.nt[i] == 32;
      for (int i = 1; i < p->dyn.nl; ++i) wide = wide && p->dyn.nt[i] == 32;
      if (const char* e = getenv("PMBRL_WIDE")) wide = wide && atoi(e) != 0;
      int cus = 0;
      if (wide && c.rows_per_wg_hint == 0)      // (of the device the plan is for, not of the current one)
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
      const bool want = c.rows_per_wg_hint >= 64 || (wide && c.rows_per_wg_hint == 0 && cus > 0 && (c.B + 63) / 64 >= cus);
      if (want) {
        p->inplace = wide ? 2 : 1;
        if (lds_need(4, 0) <= lds_cap) RT = 4;
        else p->inplace = 0;
      }
    }
    p->RT = RT;
    p->rows_per_wg = 16 * RT;
  } else if (c.rows_per_wg_hint > 0 && p->mm_parts <= 1) {
    // allow the caller to force fewer groups per workgroup
    const int want = std::max(p->M, (c.rows_per_wg_hint / p->M) * p->M);
    if (want <= 16 * p->RT) p->rows_per_wg = want;
  }
  // groups that span workgroups: the fast family does the state moment matching in the prologue of
  // its per-step launches (mm_mode 3) when it has a compile-time-width instance and the partial
  // Gram tiles fit the activation buffers; otherwise separate kernels per step (mm_mode 2)
  if (p->mm_mode == 2 && !p->span && p->fast && (c.flags & PMBRL_FLAG_MM_STATES) && c.D >= 4 && c.D <= 6 &&
      !getenv("PMBRL_MM_MODE2") && lds_need(p->RT, c.D) <= lds_cap &&
      (size_t)2 * 16 * p->RT * ld_for(p->RT) * sizeof(float) >= (size_t)8 * 256 * sizeof(double))
    p->mm_mode = 3;
  p->lds_bytes = lds_need(p->RT, (p->mm_mode == 1 || p->mm_mode == 3) ? c.D : 0, p->mm_parts);   // also fixes p->LD for the chosen RT
  { const StagePair sp = stages_for(p->RT); p->CA = sp.ca; p->CB = sp.cb; }
  p->prec = prec_for(p->RT);
  p->LDB = ldb_for(p->RT);
  if (p->lds_bytes > lds_cap) { delete p; return fail(-3, "network too wide for the fused kernel's LDS budget"); }
  p->nwg = p->mm_parts > 1 ? p->G * p->mm_parts : (c.B + p->rows_per_wg - 1) / p->rows_per_wg;
  // groups spanning workgroups, every workgroup resident at once (one per CU always fits): the
  // per-step launches become one launch whose workgroups meet at a device-wide barrier per step
  p->mm_grid = 0;
  if (p->mm_mode == 3 && p->RT == 1 && !getenv("PMBRL_MM_PERSTEP")) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && p->nwg <= cus &&
        p->nwg <= 1024)
      p->mm_grid = 1;
  }
  // More workgroups than CUs under ONE launch cannot meet at a barrier: the sweeps would run one launch per step with
  // the whole group's moment matching redone in the prologue of EVERY workgroup (one group of 5000 rows: 9.7 ms per
  // forward + adjoint, 10 000 rows: 23.4 ms).  The form built for groups spread over ranks does better there --
  // statistics over 16 workgroups, one factorisation, rows by all threads (pmbrl_mmx.h): 4.3 / 6.4 ms -- so such a
  // plan is the span form with ONE rank, whose exchange is the identity (profiles/r02n_span_cost.txt).
  // PMBRL_MM_PERSTEP keeps the per-step prologue form (tests).
  int cus_now = 0;
  (void)hipDeviceGetAttribute(&cus_now, hipDeviceAttributeMultiprocessorCount, p->device);
  if (p->mm_mode == 3 && !p->mm_grid && !p->span && p->M >= 2048 && cus_now > 0 && p->nwg > cus_now &&
      !(c.flags & PMBRL_FLAG_INFER_NS) && !getenv("PMBRL_MM_PERSTEP") && !getenv("PMBRL_MM_NO_SPAN1")) {
    p->span = 1;
    p->cfg.mm_span_rows = p->M;
    p->cfg.mm_span_offset = p->cfg.row_offset;   // (the groups are local: only the noise row needs the shard's offset)
    p->cfg.mm_span_ranks = 1;
    p->cfg.mm_span_rank = 0;
    p->mm_mode = 2;
    p->lds_bytes = lds_need(p->RT, 0, 1);
  }

  // LDS-resident tiles of the sweeps' second streamed layer (pmbrl_fast.h, lds_tile_s): the shape-specialised 16-row
  // split-precision instances (two streamed layers of 13 or 14 tiles, stage pair = one tile) keep 8 tiles of it in the
  // LDS the workgroup has to itself anyway; the launch decides per kernel variant (pmbrl_fast_split.hip).
  // PMBRL_LDS_TILES=0: off.
  p->wlds_off = 0;
  p->lds_last_lanes = 0;
  if (p->fast && p->RT == 1 && p->prec == PMBRL_PREC_SPLIT_F16 && p->CA + p->CB == 7 && p->pol.nl == 3 && p->dyn.nl == 3 &&
      p->pol.nt[1] == p->dyn.nt[1] && p->pol.nt[2] == p->pol.nt[1] && p->dyn.nt[2] == p->pol.nt[1] &&
      (p->pol.nt[1] == 13 || p->pol.nt[1] == 14) && !(c.flags & PMBRL_FLAG_NO_SHAPED) &&
      !(getenv("PMBRL_LDS_TILES") && atoi(getenv("PMBRL_LDS_TILES")) == 0)) {
    const int last_lanes = 16 * ((16 * p->pol.nt[1] - 32 * 6) / 8);
    size_t base = (p->lds_bytes + 15) / 16 * 16;
    if (p->mm_parts > 1) {
      // split groups: the instances that hold the tiles exchange sums and never touch the row blocks of the rows +
      // flags form, the generic instances that use those never copy tiles: the two share the end of the LDS (a
      // 100-row group in seven parts would not fit both)
      const int mw = 1;
      base = pm_fast_rows_area_off(16 * p->RT, p->LD, c.D, c.U, p->RT, p->pol.nt, p->pol.nl, p->dyn.nt, p->dyn.nl, c.D,
                                   prec_for(p->RT), mw, p->mm_parts > 8 ? 16 : p->M, c.H) * sizeof(float);
      base = (base + 15) / 16 * 16;
    }
    const size_t tiles = (size_t)PF_NW * pm_lds_tile_floats(14, 2, last_lanes) * sizeof(float);
    if (base + tiles <= lds_cap) {
      p->wlds_off = (int)(base / sizeof(float));
      p->lds_last_lanes = last_lanes;
      p->lds_bytes = std::max(p->lds_bytes, base + tiles);
    }
  }

  // wide states between per-step launches: the LDS-staged multi-wave kernels (PMBRL_MM_NO_WIDE: the one-wave routines)
  p->mm_wide = p->mm_mode == 2 && !p->span && (c.flags & PMBRL_FLAG_MM_STATES) && pm_mmw_ok(p->M, c.D, c.flags) &&
               !getenv("PMBRL_MM_NO_WIDE");
  HIPCHK(hipSetDevice(device));
  // reward constants
  {
    RewardDev r;
    memset(&r, 0, sizeof(r));
    const pmbrl_reward& s = c.reward;
    r.kind = s.kind;
    r.expand = s.expand;
    r.k = s.k;
    r.n_angle = s.expand ? s.n_angle : 0;
    if (r.n_angle > PMBRL_MAX_ANGLE) { delete p; return fail(-2, "too many angle dims"); }
    int no = 0;
    for (int i = 0; i < c.D; ++i) {
      bool isang = false;
      for (int j = 0; j < r.n_angle; ++j) isang |= (s.angle_dims[j] == i);
      if (!isang) r.other_dims[no++] = i;
    }
    for (int j = 0; j < r.n_angle; ++j) {
      if (s.angle_dims[j] < 0 || s.angle_dims[j] >= c.D) { delete p; return fail(-2, "angle dim out of range"); }
      r.angle_dims[j] = s.angle_dims[j];
    }
    r.n_other = s.expand ? no : c.D;
    r.De = s.expand ? no + 2 * r.n_angle : c.D;
    if (r.De > PMBRL_MAX_DIM) { delete p; return fail(-2, "expanded state too wide"); }
    for (int i = 0; i < s.k; ++i) {
      for (int j = 0; j < r.De; ++j) r.C[i * r.De + j] = s.C[i * r.De + j] / s.norm;
      r.tt[i] = s.tip_target[i] / s.norm;
    }
    r.w = s.w;
    for (int i = 0; i < s.k; ++i)
      for (int j = 0; j < s.k; ++j) {
        r.Q[i * s.k + j] = s.Q[i * s.k + j];
        r.QQ[i * s.k + j] = s.Q[i * s.k + j] + s.Q[j * s.k + i];
      }
    for (int i = 0; i < c.U; ++i)
      for (int j = 0; j < c.U; ++j) {
        r.R[i * c.U + j] = s.R[i * c.U + j];
        r.RR[i * c.U + j] = s.R[i * c.U + j] + s.R[j * c.U + i];
      }
    for (int d = 0; d < PMBRL_MAX_DIM; ++d) r.d_copy[d] = r.d_sin[d] = r.d_cos[d] = -1;
    if (s.expand) {
      for (int i = 0; i < no; ++i) { r.phi_src[i] = r.other_dims[i]; r.phi_mode[i] = 0; r.d_copy[r.other_dims[i]] = i; }
      for (int j = 0; j < r.n_angle; ++j) {
        r.phi_src[no + j] = r.angle_dims[j]; r.phi_mode[no + j] = 1; r.d_sin[r.angle_dims[j]] = no + j;
        r.phi_src[no + r.n_angle + j] = r.angle_dims[j]; r.phi_mode[no + r.n_angle + j] = 2;
        r.d_cos[r.angle_dims[j]] = no + r.n_angle + j;
      }
    } else {
      for (int i = 0; i < c.D; ++i) { r.phi_src[i] = i; r.phi_mode[i] = 0; r.d_copy[i] = i; }
    }
    if ((size_t)256 * (c.D | 1) * sizeof(float) > 64 * 1024)      // (rows of a block staged in LDS)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(pm_reward_kernel_for(c.U, r.k)),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 256 * (c.D | 1) * (int)sizeof(float)));
    p->rew_k = r.k;
    HIPCHK(hipMalloc(&p->rew_d, sizeof(RewardDev)));
    HIPCHK(hipMemcpy(p->rew_d, &r, sizeof(RewardDev), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&p->wflag_d, sizeof(int)));
    HIPCHK(hipMemset(p->wflag_d, 0, sizeof(int)));
    p->wgen = 0;
  }
  // angle_dims feature maps (utils/angles.py:29-42: others in order, then sin, then cos)
  if (angles) {
    AngleDev a;
    memset(&a, 0, sizeof(a));
    auto build = [&](FeatMap& m, int width, int n_ang, const int32_t* dims, int max_dim) -> bool {
      for (int d = 0; d < PMBRL_MAX_DIM; ++d) m.f_copy[d] = m.f_sin[d] = m.f_cos[d] = -1;
      for (int j = 0; j < n_ang; ++j) {
        if (dims[j] < 0 || dims[j] >= max_dim) return false;
        for (int i = 0; i < j; ++i)
          if (dims[i] == dims[j]) return false;
      }
      int no = 0;
      for (int i = 0; i < width; ++i) {
        bool isang = false;
        for (int j = 0; j < n_ang; ++j) isang |= (dims[j] == i);
        if (!isang) { m.src[no] = i; m.mode[no] = 0; m.f_copy[i] = no; ++no; }
      }
      for (int j = 0; j < n_ang; ++j) {
        m.src[no + j] = dims[j]; m.mode[no + j] = 1; m.f_sin[dims[j]] = no + j;
        m.src[no + n_ang + j] = dims[j]; m.mode[no + n_ang + j] = 2; m.f_cos[dims[j]] = no + n_ang + j;
      }
      m.n_feat = no + 2 * n_ang;
      return m.n_feat <= PMBRL_MAX_DIM;
    };
    if (!build(a.pol, c.D, c.n_pol_angle, c.pol_angle_dims, c.D) ||
        !build(a.dyn, c.D + c.U, c.n_dyn_angle, c.dyn_angle_dims, c.D)) {
      pmbrl_plan_destroy(p);
      return fail(-2, "angle dims must be distinct state dims (0 <= d < D)");
    }
    HIPCHK(hipMalloc(&p->ang_d, sizeof(AngleDev)));
    HIPCHK(hipMemcpy(p->ang_d, &a, sizeof(AngleDev), hipMemcpyHostToDevice));
  }
  {
    std::vector<DwBlock> blocks;
    // split-operand form of the dW GEMM (pmbrl_dw.h, pm_dw_kernel_s): where the fp32 form is bound by the matrix
    // core -- 32- / 64-row workgroups and the general family's wide networks; the 16-row sweeps' dW is HBM-bound
    p->dw_split = p->prec != 0 && (p->RT >= 2 || !p->fast) && !getenv("PMBRL_DW_F32");
    if (getenv("PMBRL_DW_SPLIT") && p->prec != 0) p->dw_split = 1;
    // wide layers of the split form go to the LDS-staged tile kernel (pmbrl_dw.h, pm_dw_wide_kernel)
    bool wide[PM_MAXL] = {false};
    std::vector<DwUnit> units;
    // (layers of at least 128 x 128 on the split-precision plans: also the 200 x 200 layer of the cart-pole shapes,
    //  whose block-kernel GEMM re-read its stash 1.5 x -- C2 0.116 -> 0.102 ms, C4 0.50 -> 0.38 ms; the exact-fp32
    //  plans keep the fp32 block kernel for every layer)
    const int wide_min = getenv("PMBRL_DW_WIDE_MIN") ? atoi(getenv("PMBRL_DW_WIDE_MIN")) : 128;
    p->dw_layer13 = -1;
    if (p->prec != 0 && !getenv("PMBRL_DW_NO_WIDE"))
      for (int l = 0; l < p->pol.nl; ++l)
        if (p->pol.nt[l + 1] * 16 >= wide_min && p->pol.nt[l] * 16 >= wide_min) {
          if (p->dw_layer13 < 0 && p->pol.nt[l + 1] <= PM_DWL_NT && p->pol.nt[l] <= PM_DWL_NT && !getenv("PMBRL_DW_NO_LAYER13")) {
            // a whole layer of up to 13 x 13 tiles per workgroup
            wide[l] = true;
            p->dw_layer13 = l;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_dw_layer_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PM_DWL_LDS_BYTES));
            continue;
          }
          wide[l] = true;
          for (int m0 = 0; m0 < p->pol.nt[l + 1] * 16; m0 += PM_DWW_TM)
            for (int n0 = 0; n0 < p->pol.nt[l] * 16; n0 += PM_DWW_TN) units.push_back(DwUnit{(int16_t)l, (int16_t)m0, (int16_t)n0, 0});
        }
    // Round 6: the wide sweeps (64-row workgroups, in-place 512-wide layers: pmbrl_wide.h) write the stashes of the 512 x 512
    // layers PRE-SPLIT, in the layout pm_dw_wide_pre_kernel streams into LDS without touching a register (pmbrl_dw.h).
    // Every wide layer 512 x 512 exactly (32 tiles: the stash block's tile count); PMBRL_DW_PRE=0: the fp32 stash.
    p->dw_pre_mask = 0;
    if (!units.empty() && p->inplace == 2 && p->RT == 4 && !(getenv("PMBRL_DW_PRE") && atoi(getenv("PMBRL_DW_PRE")) == 0)) {
      bool all = true;
      for (const DwUnit& u : units) all = all && p->pol.nt[u.layer] == 32 && p->pol.nt[u.layer + 1] == 32 &&
                                          p->pol.dim[u.layer] == 512 && p->pol.dim[u.layer + 1] == 512;
      // ... and every OTHER layer narrow on one side, 512 wide on the other (first layer: <= 32 inputs; head: <= 32 outputs):
      // pm_dw_narrow_pre_kernel -- so that EVERY 512-wide stash is pre-split and the sweeps carry one form of stash store
      for (int l = 0; l < p->pol.nl && all; ++l) {
        bool is = false;
        for (const DwUnit& u : units) is = is || u.layer == l;
        p->dw_narrow[l] = 0;
        if (is) continue;
        if (p->pol.nt[l + 1] == 32 && p->pol.dim[l + 1] == 512 && p->pol.nt[l] <= 2) p->dw_narrow[l] = 1;
        else if (p->pol.nt[l] == 32 && p->pol.dim[l] == 512 && p->pol.nt[l + 1] <= 2) p->dw_narrow[l] = 2;
        else all = false;
      }
      if (all) {
        std::vector<DwUnit> pre;
        for (int l = 0; l < p->pol.nl; ++l) {
          if (p->dw_narrow[l]) { wide[l] = true; continue; }      // (not the block kernels': skipped by build_dw_blocks)
          p->dw_pre_mask |= 1u << l;
          for (int m0 = 0; m0 < 512; m0 += PM_DWP_TM)
            for (int n0 = 0; n0 < 512; n0 += PM_DWP_TN) pre.push_back(DwUnit{(int16_t)l, (int16_t)m0, (int16_t)n0, 0});
        }
        units = pre;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_dw_narrow_pre_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, PM_DWP_LDS_BYTES));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_dw_narrow_pre_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, PM_DWP_LDS_BYTES));
      } else {
        for (int l = 0; l < PM_MAXL; ++l) p->dw_narrow[l] = 0;
      }
    }
    p->n_dw_units = (int)units.size();
    if (p->n_dw_units) {
      HIPCHK(hipMalloc(&p->dw_units_d, units.size() * sizeof(DwUnit)));
      HIPCHK(hipMemcpy(p->dw_units_d, units.data(), units.size() * sizeof(DwUnit), hipMemcpyHostToDevice));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_dw_wide_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, PM_DWW_LDS_BYTES));
      if (p->dw_pre_mask)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_dw_wide_pre_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, PM_DWP_LDS_BYTES));
    }
    p->n_dw_blocks = build_dw_blocks(p->pol.nl, p->pol.nt, blocks, p->dw_wave_first, p->dw_split ? PM_DW_TN_S : PM_DW_TN, wide);
    p->dw_n_chunks = c.H * p->nwg * p->RT;
    int nsplit = std::min(256, p->dw_n_chunks);   // one 8-wave workgroup per CU
    p->dw_chunks_per_split = (p->dw_n_chunks + nsplit - 1) / nsplit;
    if (p->dw_split) p->dw_chunks_per_split = (p->dw_chunks_per_split + 1) & ~1;   // whole chunk pairs
    p->dw_nsplit = (p->dw_n_chunks + p->dw_chunks_per_split - 1) / p->dw_chunks_per_split;
    HIPCHK(hipMalloc(&p->dw_blocks_d, std::max<size_t>(1, blocks.size()) * sizeof(DwBlock)));
    if (!blocks.empty())
      HIPCHK(hipMemcpy(p->dw_blocks_d, blocks.data(), blocks.size() * sizeof(DwBlock),
                       hipMemcpyHostToDevice));
  }
  {
    // dW GEMM behind the adjoint sweep.  The sweep is latency-bound and holds nwg CUs completely (its
    // workgroups take a CU's whole register file); with fewer workgroups than CUs the rest of the chip idles
    // for its duration, and the dW GEMM -- 12-17 % of an iteration -- used to start when the sweep had
    // finished.  It needs only the deltas of the steps already swept: the sweep becomes pipe_K launches
    // over descending step ranges (the state gradient is carried through HBM, as in the per-step launch
    // modes) and the GEMM of range k runs next to the sweep of range k+1, on a second stream, in a grid that
    // fits the idle CUs (workgroups go round-robin to the 8 XCDs, so per XCD: CUs/8 - ceil(nwg/8) are free)
    // with workgroups that cannot share a CU with the sweep's.  Only the last, short range is exposed.
    // Correctness never depends on the overlap: every dependency is a stream event.
    p->pipe_K = 1;
    const bool no_pipe_pre = p->dw_pre_mask != 0;      // (the pre-split GEMMs run once, behind the whole sweep)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
    const char* e = getenv("PMBRL_DW_PIPE");
    const int n_free = cus >= 64 && cus % 8 == 0 ? 8 * (cus / 8 - (p->nwg + 7) / 8) : 0;
    const bool single = (p->mm_mode == 0 || p->mm_mode == 1) && p->mm_parts <= 1;   // (no second kernel next to flag barriers)
    // Worth it where a range of the sweep is long against a launch boundary (tail of the last step, launch
    // gap, the next launch's prologue: 15 us at 16 rows per workgroup, 40 us at 64) -- measured: the double
    // cart-pole shape (64-row workgroups, 30 us per step) gains 7 % of an iteration, the cart-pole shapes
    // (6 / 13 us per step) lose 1-5 %.  Proxy for the step time: rows per workgroup x weights of both nets.
    const bool explicit_ranges = e && e[0] >= '0' && e[0] <= '9';
    const bool long_steps = (long long)p->RT * (long long)(p->pol.n_params + p->dyn.n_params) >= 256 * 1024 && c.H >= 16;
    if (single && !no_pipe_pre && n_free >= cus / 8 && c.H >= 2 && !(e && !strcmp(e, "off")) && (long_steps || explicit_ranges)) {
      std::vector<int> cnt;
      if (explicit_ranges) {     // PMBRL_DW_PIPE=n0,n1,...: steps per range, highest steps first (tests, tuning)
        int sum = 0;
        for (const char* q = e; *q;) {
          const int v = atoi(q);
          if (v > 0) { cnt.push_back(v); sum += v; }
          while (*q && *q != ',') ++q;
          if (*q == ',') ++q;
        }
        if (sum != c.H || (int)cnt.size() > PM_PIPE_MAX) cnt.clear();
      }
      if (cnt.empty() && long_steps) {
        // 40 / 30 / 20 / 10 % of the horizon: the exposed tail is the GEMM of the last tenth
        const int a3 = std::max(2, c.H / 10), a2 = a3 + std::max(2, c.H / 5), a1 = a2 + std::max(2, (3 * c.H) / 10);
        cnt = {c.H - a1, a1 - a2, a2 - a3, a3};
      }
      // lower step bound of every range (the last one ends at step 0; ranges emptied by the adjustment are dropped)
      int K = 0, lo = c.H;
      for (size_t k = 0; k < cnt.size() && lo > 0; ++k) {
        lo = k + 1 == cnt.size() ? 0 : std::max(0, lo - cnt[k]);
        // split-operand GEMM: a K = 32 chunk pair must not straddle two launches
        if (p->dw_split && ((p->nwg * p->RT) & 1) && (lo & 1)) lo -= 1;
        if (K > 0 && lo >= p->pipe_lo[K - 1]) continue;
        p->pipe_lo[K++] = lo;
      }
      if (K >= 2 && p->pipe_lo[K - 1] == 0) {
        p->pipe_K = K;
        p->pipe_grid = n_free;                     // workgroups (= partial rows) of a GEMM launch next to the sweep
        p->pipe_rows = std::max(n_free, cus);      // ... of the last launch, which has the chip to itself
        HIPCHK(hipStreamCreateWithFlags(&p->pipe_stream, hipStreamNonBlocking));
        for (int k = 0; k < p->pipe_K; ++k) HIPCHK(hipEventCreateWithFlags(&p->pipe_ev[k], hipEventDisableTiming));
      }
    }
  }
  p->reg = pm_reg_plan_ok(p) ? 1 : 0;
  p->reg_mm = pm_reg_mm_width(p);
  p->reg_bwd = !(getenv("PMBRL_REG_BWD") && atoi(getenv("PMBRL_REG_BWD")) == 0);      // (read once: a recorded call bakes the choice in)
  // workspace carve-up
  {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      const size_t o = off;
      off = align_up(off + bytes, 256);
      return o;
    };
    NetPlan* nets[2] = {&p->pol, &p->dyn};
    for (NetPlan* n : nets) {
      for (int l = 0; l < n->nl; ++l) {
        const size_t fr = (size_t)(n->nt[l + 1] + 15) * (n->nt[l] + 15) * 256 * sizeof(float);
        n->wf[l] = take(fr);
        n->wb[l] = take(fr);
        n->bias[l] = take((size_t)n->nt[l + 1] * 16 * sizeof(float));
        if (l < n->nl - 1)
          // u16 (generic) or 4 nibble-bytes (fast); 256 bytes of slack behind it: where the register-resident family's
          // rows past the batch put their (zero) bytes -- lane (< 64) + 4 x tile (< 64) past the end (pmbrl_reg.h)
          n->abits[l] = take((size_t)c.H * c.B * n->nt[l + 1] * 4 + 256);
      }
    }
    const size_t Rw = 16 * p->RT;
    for (int l = 0; l < p->pol.nl; ++l) {
      p->off_actT[l] = take((size_t)c.H * p->nwg * p->pol.nt[l] * 16 * Rw * sizeof(float));
      p->off_gT[l] = take((size_t)c.H * p->nwg * p->pol.nt[l + 1] * 16 * Rw * sizeof(float));
    }
    p->off_Tp = take((size_t)c.H * c.B * c.U * sizeof(float));
    p->off_Td = take((size_t)c.H * c.B * c.D * sizeof(float));
    p->off_gmm_c = take(gmm ? (size_t)c.H * c.B * (c.dyn_components + 1) * c.D * sizeof(float) : 0);
    p->off_gmm_k = take(gmm ? (size_t)c.H * c.B * sizeof(int) : 0);
    p->off_xt = take((size_t)c.H * c.B * c.D * sizeof(float));
    p->off_rt = take((size_t)c.H * c.B * sizeof(float));
    p->off_mmfac = take(p->mm_mode == 1 ? (size_t)c.H * (c.B / p->M) * pm_mm_fac_doubles(c.D) * sizeof(double)
                        : p->mm_wide    ? (size_t)c.H * p->G * pm_mmw_fac_doubles(c.D) * sizeof(double)
                                        : 0);
    p->off_Jx = take((size_t)c.H * c.B * c.D * sizeof(float));
    p->off_Ja = take((size_t)c.H * c.B * c.U * sizeof(float));
    p->off_gxc = take((size_t)c.B * c.D * sizeof(float));
    p->off_gxc2 = take((size_t)c.B * c.D * sizeof(float));
    p->off_gsync = take(2 * 1024 * sizeof(unsigned));   // one flag per workgroup for the device-wide barriers (forward, backward)
    // groups split over workgroups: the granules of their statistics exchange (pm_xch_sum; PMBRL_MM_XCH=0: rows + flags)
    p->xch_bytes = 0;
    if (p->mm_parts > 1 && !(getenv("PMBRL_MM_XCH") && atoi(getenv("PMBRL_MM_XCH")) == 0)) {
      // (more than 8 parts: a second set of slots, the collectors')
      p->xch_bytes = (size_t)(p->mm_parts > 8 ? 2 : 1) * p->nwg * PM_XCH_WG_WORDS(2) * sizeof(unsigned long long);
      p->off_xch = take(p->xch_bytes);
    }
    p->off_ztab = take((p->mm_fan || p->reg_mm) ? (size_t)c.H * p->G * 2 * c.D * sizeof(double) : 0);
    p->off_reg_linv = take(p->reg_mm ? (size_t)c.H * p->G * c.D * c.D * sizeof(double) : 0);
    p->off_grt = take((size_t)c.H * c.B * sizeof(float));
    {
      // statistics exchange of groups spread over ranks: forward slots of every rank (states of one step, or the
      // rewards of all steps), the factors the adjoint reuses
      const size_t n_s = (size_t)p->G * pm_mmx_slot_doubles(c.D), n_r = (size_t)c.H * p->G * pm_mmx_slot_doubles(1);
      p->off_mmx_buf = take(p->span ? (size_t)p->cfg.mm_span_ranks * std::max(n_s * PM_MMX_NB, n_r) * sizeof(double) : 0);
      p->off_mmx_fac = take(p->span ? (size_t)c.H * p->G * pm_mm_fac_doubles(c.D) * sizeof(double) : 0);
      p->off_mmx_rfac = take(p->span ? (size_t)c.H * p->G * pm_mm_fac_doubles(1) * sizeof(double) : 0);
    }
    p->off_reg_pack = take(p->reg ? pm_reg_pack_bytes() : 0);
    for (int n = 0; n < 2; ++n)
      for (int l = 0; l < 2; ++l) p->off_reg_ab[n][l] = take(p->reg ? (size_t)c.H * p->nwg * 1024 : 0);
    p->off_part = take((size_t)std::max(p->dw_nsplit, p->pipe_K > 1 ? p->pipe_rows : 0) *
                       ((p->pol.n_params + 3) / 4 * 4) * sizeof(float));
    p->ws_bytes = off;
    if (p->ws_bytes >= ((size_t)1 << 32)) p->reg = p->reg_mm = 0;      // (the family addresses its stashes with 32-bit workspace offsets)
  }
  int rc2 = 0;
  if (!p->fast && p->prec) {
    rc2 = pm_general_split_set_attr(p);
  } else if (!p->fast) {
    switch (p->RT) {
      case 1: rc2 = set_attr<1>(p->lds_bytes); break;
      case 2: rc2 = set_attr<2>(p->lds_bytes); break;
      default: rc2 = set_attr<4>(p->lds_bytes); break;
    }
  } else if (p->prec) {
    rc2 = p->prec == PMBRL_PREC_SPLIT_F16 ? pm_fast_split2_set_attr(p) : pm_fast_split1_set_attr(p);
  } else {
    rc2 = pm_fast_f32_set_attr(p);
  }
  if (p->mm_wide) {
    const int smem = (int)pm_mmw_lds_bytes(p->M, c.D, true);
    if (smem > 64 * 1024 &&
        (hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mmw_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             smem) != hipSuccess ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mmw_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             smem) != hipSuccess)) {
      (void)hipGetLastError();
      pmbrl_plan_destroy(p);
      return fail(-3, "hipFuncSetAttribute(wide moment-matching kernels) failed");
    }
  }
  if (p->mm_mode == 2 || p->mm_mode == 3) {
    const int smem = (int)(pm_mm_kernel_doubles(c.D) * sizeof(double));
    if (smem > (int)lds_cap) { pmbrl_plan_destroy(p); return fail(-3, "moment-matching scratch exceeds the LDS"); }
    if (smem > 64 * 1024) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mm_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mm_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        (void)hipGetLastError();
        pmbrl_plan_destroy(p);
        return fail(-3, "hipFuncSetAttribute(moment-matching kernels) failed");
      }
    }
  }
  if (!rc2 && p->reg) rc2 = pm_reg_set_attr(p);
  if (rc2) { pmbrl_plan_destroy(p); return rc2; }
  if (p->mm_grid) {
    // the barrier-form sweep needs every workgroup resident at once: ask the runtime how many of THIS
    // kernel (registers, LDS) fit a CU instead of assuming one
    const int per_cu = p->prec == PMBRL_PREC_SPLIT_F16 ? pm_fast_split2_mmg_blocks_per_cu(p)
                       : p->prec                        ? pm_fast_split1_mmg_blocks_per_cu(p)
                                                        : pm_fast_f32_mmg_blocks_per_cu(p);
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
    if (per_cu < 1 || p->nwg > cus) p->mm_grid = 0;     // (one per CU is all that is relied upon)
  }
  *out = p;
  return 0;
}

// ---------------------------------------------------------------------------
// replay of repeated calls (include/pmbrl.h: pmbrl_plan_set_replay)
// ---------------------------------------------------------------------------
static bool per_step_launches(const pmbrl_plan* p) {
  return (p->mm_mode == 2 || (p->mm_mode == 3 && !p->mm_grid)) && p->cfg.H >= 4;
}
static bool replay_wanted(const pmbrl_plan* p) {
  if (p->replay == 0 || p->span || p->coll) return false;
  return p->replay >= 2 || per_step_launches(p);
}
static void replay_drop(pmbrl_plan* p, int slot) {
  if (p->rp[slot].exec) (void)hipGraphExecDestroy(p->rp[slot].exec);
  p->rp[slot].exec = nullptr;
  p->rp[slot].seen = 0;
}
// what makes two calls "the same": every argument byte, kept (a hash alone would replay a stale recording on a collision)
struct ReplayKey {
  unsigned long long h = 1469598103934665603ull;
  std::vector<unsigned char> bytes;
  void add(const void* q, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(q);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    bytes.insert(bytes.end(), b, b + n);
  }
  template <class T> void val(const T& v) { add(&v, sizeof(T)); }
};
// is this stream recording a graph?  (the legacy default stream cannot be asked, and cannot be recording)
static bool pm_stream_recording(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (!s) return false;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}
// body(): queues the call on s.  Returns its code; *replayed = the call went out as a graph launch.
template <class Body>
static int replay_call(pmbrl_plan* p, int slot, const ReplayKey& K, hipStream_t s, Body body) {
  const unsigned long long key = K.h;
  pmbrl_plan::ReplaySlot& R = p->rp[slot];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  // (the legacy default stream cannot be asked, and cannot be recording)
  const bool usable = replay_wanted(p) && !p->timing && !p->prof_fwd && !p->prof_bwd && !R.dead &&
                      (!s || (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone));
  if (!usable) {
    if (R.exec || R.seen) replay_drop(p, slot);
    return body(s);
  }
  const bool same = R.key == key && R.bytes == K.bytes;
  if (R.exec && same) {
    HIPCHK(hipGraphLaunch(R.exec, s));
    ++R.launches;
    if (slot == 0) { p->old_pack_stale = R.aux & 1; p->abits_packed = (R.aux >> 1) & 1; }   // (the host-side notes the recorded call left: which families' weights it packed, the form of its activity bits)
    return 0;
  }
  if (!same || R.seen == 0) {      // first sight of these arguments: as usual, remember them
    replay_drop(p, slot);
    R.key = key;
    R.bytes = K.bytes;
    R.seen = 1;
    return body(s);
  }
  // second identical call: record it on a stream of the plan's own (the caller's may be the legacy default stream, which
  // cannot record; nothing runs while recording), then launch what was recorded on the caller's
  hipGraph_t g = nullptr;
  if (!p->cap_stream && hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess) p->cap_stream = nullptr;
  if (!p->cap_stream || hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    R.dead = 1;
    return body(s);
  }
  const int rc = body(p->cap_stream);
  const hipError_t e = hipStreamEndCapture(p->cap_stream, &g);
  if (rc != 0 || e != hipSuccess || !g) {
    // (a call inside invalidated the capture, or the body failed: nothing was queued -- run it eagerly, never try again)
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    R.dead = 1;
    replay_drop(p, slot);
    return rc != 0 ? rc : body(s);
  }
  hipGraphExec_t ex = nullptr;
  const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ei != hipSuccess || !ex) {
    (void)hipGetLastError();
    R.dead = 1;
    replay_drop(p, slot);
    return body(s);
  }
  R.exec = ex;
  R.aux = (p->old_pack_stale & 1) | ((p->abits_packed & 1) << 1);
  HIPCHK(hipGraphLaunch(R.exec, s));
  ++R.launches;
  return 0;
}

extern "C" int pmbrl_plan_set_replay(pmbrl_plan* p, int on) {
  if (!p || on < 0 || on > 2) return fail(-1, "bad argument");
  p->replay = on;
  for (int k = 0; k < 2; ++k) { replay_drop(p, k); p->rp[k].dead = 0; }
  return 0;
}

extern "C" int pmbrl_plan_replay_count(const pmbrl_plan* p, int64_t* n_out) {
  if (!p || !n_out) return fail(-1, "null argument");
  for (int k = 0; k < 2; ++k) n_out[k] = p->rp[k].launches;
  return 0;
}

extern "C" int pmbrl_plan_set_prof(pmbrl_plan* p, long long* fwd_d, long long* bwd_d) {
  if (!p) return fail(-1, "null argument");
  p->prof_fwd = fwd_d;
  p->prof_bwd = bwd_d;
  return 0;
}

extern "C" int pmbrl_plan_set_timing(pmbrl_plan* p, int on) {
  if (!p) return fail(-1, "null argument");
  if (on && !p->ev[0][0]) {
    for (int i = 0; i < PMBRL_TIMER_COUNT; ++i)
      for (int j = 0; j < 2; ++j) HIPCHK(hipEventCreate(&p->ev[i][j]));
  }
  p->timing = on ? 1 : 0;
  for (int i = 0; i < PMBRL_TIMER_COUNT; ++i) p->ev_set[i] = false;
  return 0;
}

extern "C" int pmbrl_plan_read_timing(pmbrl_plan* p, float* ms) {
  if (!p || !ms) return fail(-1, "null argument");
  for (int i = 0; i < PMBRL_TIMER_COUNT; ++i) {
    ms[i] = -1.f;
    if (p->timing && p->ev_set[i]) {
      HIPCHK(hipEventSynchronize(p->ev[i][1]));
      HIPCHK(hipEventElapsedTime(&ms[i], p->ev[i][0], p->ev[i][1]));
    }
  }
  return 0;
}

extern "C" void pmbrl_plan_destroy(pmbrl_plan* p) {
  if (!p) return;
  if (p->ev[0][0])
    for (int i = 0; i < PMBRL_TIMER_COUNT; ++i)
      for (int j = 0; j < 2; ++j) (void)hipEventDestroy(p->ev[i][j]);
  for (int k = 0; k < 2; ++k) replay_drop(p, k);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  if (p->rew_d) (void)hipFree(p->rew_d);
  if (p->wflag_d) (void)hipFree(p->wflag_d);
  if (p->ang_d) (void)hipFree(p->ang_d);
  if (p->dw_blocks_d) (void)hipFree(p->dw_blocks_d);
  if (p->dw_units_d) (void)hipFree(p->dw_units_d);
  if (p->pipe_stream) {
    (void)hipStreamDestroy(p->pipe_stream);
    for (int k = 0; k < p->pipe_K; ++k) (void)hipEventDestroy(p->pipe_ev[k]);
  }
  delete p;
}

extern "C" size_t pmbrl_plan_workspace_bytes(const pmbrl_plan* p) { return p ? p->ws_bytes : 0; }

extern "C" int pmbrl_plan_info(const pmbrl_plan* p, int32_t* info) {
  if (!p || !info) return fail(-1, "null argument");
  memset(info, 0, sizeof(int32_t) * PMBRL_INFO_COUNT);
  info[PMBRL_INFO_ROWS_PER_WG] = p->rows_per_wg;
  info[PMBRL_INFO_N_WG] = p->nwg;
  info[PMBRL_INFO_ROW_TILES] = p->RT;
  info[PMBRL_INFO_LDS_BYTES] = (int32_t)p->lds_bytes;
  info[PMBRL_INFO_N_POL_PARAMS] = (int32_t)p->pol.n_params;
  info[PMBRL_INFO_N_DYN_PARAMS] = (int32_t)p->dyn.n_params;
  info[PMBRL_INFO_DW_SPLITS] = p->dw_nsplit;
  info[7] = p->mm_mode;
  info[8] = p->LD;
  info[9] = p->n_dw_blocks;
  info[10] = p->fast;
  info[11] = p->CA * 16 + p->CB;
  info[12] = p->mm_grid;
  info[PMBRL_INFO_PRECISION] = p->prec;
  info[PMBRL_INFO_DW_PIPE] = p->pipe_K;
  info[PMBRL_INFO_MM_PARTS] = p->mm_parts;
  info[PMBRL_INFO_REG] = p->reg;
  info[PMBRL_INFO_REPLAY] = replay_wanted(p) ? 1 : 0;
  info[PMBRL_INFO_INPLACE] = p->inplace;
  info[PMBRL_INFO_REG_FWD_CALLS] = p->reg_calls[0];
  info[PMBRL_INFO_REG_BWD_CALLS] = p->reg_calls[1];
  return 0;
}

extern "C" int pmbrl_pack_mask(void* stream, const float* mask_d, int32_t B, int32_t h,
                               int32_t src_ld, uint16_t* bits_d) {
  if (!mask_d || !bits_d || B < 1 || h < 1 || src_ld < h) return fail(-1, "bad argument");
  const int nt = (h + 15) / 16;
  const int n = B * nt;
  hipLaunchKernelGGL(pm_pack_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     mask_d, B, h, src_ld, bits_d);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_draw_masks(void* stream, int32_t kind, uint64_t seed, uint64_t offset, const float* param_d,
                                int32_t param_len, float temp, int32_t rows, int32_t h, const float* u_d, const float* v_d,
                                uint16_t* bits_d, int32_t aux_row0, int32_t aux_rows, float* u_out_d, float* hard_out_d,
                                float* probs_out_d) {
  if (!param_d || !bits_d || rows < 1 || h < 1 || (kind != 0 && kind != 1) || (param_len != 1 && param_len != h) ||
      aux_row0 < 0 || aux_rows < 0 || aux_row0 + aux_rows > rows)
    return fail(-1, "bad argument");
  if (kind == 1 && !(temp > 0.f)) return fail(-1, "concrete dropout needs a positive temperature");
  DrawArgs A;
  A.kind = kind; A.seed = seed; A.offset = offset; A.param = param_d; A.param_len = param_len; A.temp = temp;
  A.rows = rows; A.h = h; A.u_in = u_d; A.v_in = v_d; A.bits = bits_d;
  A.aux_row0 = aux_row0; A.aux_rows = aux_rows; A.u_out = u_out_d; A.hard_out = hard_out_d; A.probs_out = probs_out_d;
  const long long n = (long long)rows * ((h + 15) / 16);
  hipLaunchKernelGGL(pm_draw_masks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// argument assembly
// ---------------------------------------------------------------------------
static void fill_net(const NetPlan& n, char* ws, const uint16_t* const* masks, NetDev& d) {
  d.nl = n.nl;
  for (int i = 0; i <= n.nl; ++i) {
    d.dim[i] = n.dim[i];
    d.nt[i] = n.nt[i];
  }
  for (int l = 0; l < n.nl; ++l) {
    d.keep[l] = n.keep[l];
    d.inv_keep[l] = 1.f / n.keep[l];
    d.wf[l] = reinterpret_cast<const float*>(ws + n.wf[l]);
    d.wb[l] = reinterpret_cast<const float*>(ws + n.wb[l]);
    d.bias[l] = reinterpret_cast<const float*>(ws + n.bias[l]);
    d.mask[l] = (l < n.nl - 1) ? masks[l] : nullptr;
    d.abits[l] = (l < n.nl - 1) ? reinterpret_cast<uint16_t*>(ws + n.abits[l]) : nullptr;
  }
}

static int fill_args(const pmbrl_plan* p, void* workspace, const pmbrl_inputs* in, RolloutArgs& A) {
  const pmbrl_config& c = p->cfg;
  char* ws = static_cast<char*>(workspace);
  memset(&A, 0, sizeof(A));
  A.B = c.B; A.D = c.D; A.U = c.U; A.H = c.H;
  A.Bg = c.B_global; A.row_off = c.row_offset; A.flags = c.flags;
  // split groups without the statistics exchange (PMBRL_MM_XCH=0): only the generic instances carry the rows + flags form
  if (p->mm_parts > 1 && !p->xch_bytes) A.flags |= PMBRL_FLAG_NO_SHAPED;
  A.G = p->G; A.M = p->M; A.mm_mode = p->mm_mode;
  A.t0 = 0; A.t1 = c.H;
  A.rows_per_wg = p->rows_per_wg; A.nwg = p->nwg; A.Rw = 16 * p->RT; A.LD = p->LD; A.LDB = p->LDB;
  A.stash_pre = p->dw_pre_mask;
  A.mls_pol = c.max_log_std_pol; A.mls_dyn = c.max_log_std_dyn;
  for (int l = 0; l < p->pol.nl - 1; ++l)
    if (!in->pol_mask_bits_d[l]) return fail(-1, "missing policy mask bits");
  for (int l = 0; l < p->dyn.nl - 1; ++l)
    if (!in->dyn_mask_bits_d[l]) return fail(-1, "missing dynamics mask bits");
  fill_net(p->pol, ws, in->pol_mask_bits_d, A.pol);
  fill_net(p->dyn, ws, in->dyn_mask_bits_d, A.dyn);
  A.rew = p->rew_d;
  A.ang = p->ang_d;
  A.x0 = in->x0_d; A.mx = in->mx_d; A.iSx = in->iSx_d; A.my = in->my_d; A.Sy = in->Sy_d;
  A.pscale = in->pol_scale_d; A.pbias = in->pol_bias_d;
  A.zpol = in->z_pol_d; A.zdyn = in->z_dyn_d; A.zmm = in->z_mm_d; A.zrr = in->z_rr_d;
  A.zpol_ss = in->z_pol_step_stride; A.zdyn_ss = in->z_dyn_step_stride;
  if (!A.x0 || !A.mx || !A.iSx || !A.my || !A.Sy || !A.pscale || !A.pbias || !A.zpol || !A.zdyn)
    return fail(-1, "null input pointer");
  if ((c.flags & PMBRL_FLAG_MM_STATES) && !A.zmm) return fail(-1, "mm_states needs z_mm");
  if ((c.flags & PMBRL_FLAG_MM_REWARDS) && !A.zrr) return fail(-1, "mm_rewards needs z_rr");
  for (int l = 0; l < p->pol.nl; ++l) {
    A.actT[l] = reinterpret_cast<float*>(ws + p->off_actT[l]);
    A.gT[l] = reinterpret_cast<float*>(ws + p->off_gT[l]);
  }
  A.Tp = reinterpret_cast<float*>(ws + p->off_Tp);
  A.Td = reinterpret_cast<float*>(ws + p->off_Td);
  A.gmm_n = p->cfg.dyn_components > 1 ? p->cfg.dyn_components : 0;
  A.zpi = A.ucat = A.zdyn_grad = nullptr;
  A.gmm_k = nullptr;
  A.gmm_c = nullptr;
  if (A.gmm_n) {
    if (!in->z_pi_d || !in->u_cat_d) return fail(-1, "mixture head: z_pi_d and u_cat_d are required");
    if (in->z_dyn_step_stride == 0) return fail(-1, "mixture head: z_dyn_d must be a per-step draw [H, B, D]");
    A.zpi = in->z_pi_d;
    A.ucat = in->u_cat_d;
    // the reference's autograd differentiates the noise term with the LAST step's noise at every step
    if (!(p->cfg.flags & PMBRL_FLAG_GMM_EXACT_NOISE_GRAD))
      A.zdyn_grad = in->z_dyn_d + (size_t)(p->cfg.H - 1) * in->z_dyn_step_stride;
    A.gmm_c = reinterpret_cast<float*>(ws + p->off_gmm_c);
    A.gmm_k = reinterpret_cast<int*>(ws + p->off_gmm_k);
  }
  A.xt = reinterpret_cast<float*>(ws + p->off_xt);
  A.rt = reinterpret_cast<float*>(ws + p->off_rt);
  A.mmfac = reinterpret_cast<double*>(ws + p->off_mmfac);
  A.mmfac_groups = p->mm_mode == 1 ? c.B / p->M : 0;
  A.Jx = reinterpret_cast<float*>(ws + p->off_Jx);
  A.Ja = reinterpret_cast<float*>(ws + p->off_Ja);
  A.gx_carry = reinterpret_cast<float*>(ws + p->off_gxc);
  A.mm_grid = p->mm_grid;
  A.mm_parts = p->mm_parts;
  A.mm_fan = p->mm_fan;
  A.mm_ztab = (p->mm_fan || p->reg_mm) ? reinterpret_cast<const double*>(ws + p->off_ztab) : nullptr;
  A.gsync = reinterpret_cast<unsigned*>(ws + p->off_gsync);
  A.xch = p->xch_bytes ? reinterpret_cast<unsigned long long*>(ws + p->off_xch) : nullptr;
  A.gx_carry_out = nullptr;
  if (p->fast) {
    // weight streams (hidden->hidden layers) and LDS offsets: same walk as pm_fast_carve
    const NetDev& P = A.pol;
    const NetDev& F = A.dyn;
    const int pm = p->CA + p->CB;
    auto padk = [pm](int nkb) { return (nkb + pm - 1) / pm * pm; };
    // streamed layer: weights, output tiles, padded / real k-blocks; a layer with 4m+1 output
    // tiles hands its last tile to the K-split path (pm_fast_ksplit)
    auto add_stream = [&](StreamDesc& sd, const float* wf, int n_ot, int n_kb_real, int& tw_off) {
      const int i = sd.n++;
      sd.wf[i] = wf;
      // split precision: the stream counts 1 KiB weight loads = pieces of K32 blocks (3 per block in the
      // forward sweep's table, 2 in the adjoint's)
      const int np = (&sd == &A.sd_fwd && p->prec != PMBRL_PREC_SPLIT_F16) ? 3 : 2;
      sd.n_kb[i] = p->prec ? padk((n_kb_real + 1) / 2) * np : padk(n_kb_real);
      sd.n_kb_real[i] = n_kb_real;
      sd.ks[i] = pm_fast_ksplit(n_ot, p->RT, p->prec) ? 1 : 0;
      sd.n_ot[i] = n_ot - sd.ks[i];
      sd.tw_off[i] = tw_off;
      if (sd.ks[i]) tw_off += n_kb_real * 256;
    };
    StreamDesc& f = A.sd_fwd;
    f.n = 0;
    int twf = 0, twb = 0;
    for (int l = 1; l < P.nl - 1; ++l) add_stream(f, P.wf[l], P.nt[l + 1], P.nt[l], twf);
    for (int l = 1; l < F.nl - 1; ++l) add_stream(f, F.wf[l], F.nt[l + 1], F.nt[l], twf);
    StreamDesc& b = A.sd_bwd;
    b.n = 0;
    for (int l = F.nl - 2; l >= 1; --l) add_stream(b, F.wb[l], F.nt[l], F.nt[l + 1], twb);
    for (int l = P.nl - 2; l >= 1; --l) add_stream(b, P.wb[l], P.nt[l], P.nt[l + 1], twb);
    const size_t R = 16 * (size_t)p->RT;
    size_t o = 2 * R * p->LD + 2 * R * c.D + R * c.U + R * 16 + 2 * R +
               (pm_fast_hp_alias((int)R, p->LD, p->RT) ? 0 : (size_t)PF_NW * p->RT * 256);   // up to and incl. the hp region
    for (int l = 0; l < P.nl; ++l) { A.fo.pbias[l] = (int)o; o += (size_t)P.nt[l + 1] * 16; }
    for (int l = 0; l < F.nl; ++l) { A.fo.dbias[l] = (int)o; o += (size_t)F.nt[l + 1] * 16; }
    for (int l = 0; l < P.nl - 1; ++l) { A.fo.pmask[l] = (int)o; o += (R * P.nt[l + 1] + 1) / 2; }
    for (int l = 0; l < F.nl - 1; ++l) { A.fo.dmask[l] = (int)o; o += (R * F.nt[l + 1] + 1) / 2; }
  }
  if (in->pol_params_d && in->dyn_params_d) {
    A.pol_head_w = in->pol_params_d + p->pol.w_off[p->pol.nl - 1];
    A.dyn_head_w = in->dyn_params_d + p->dyn.w_off[p->dyn.nl - 1];
    A.pol_first_w = in->pol_params_d + p->pol.w_off[0];
    A.dyn_first_w = in->dyn_params_d + p->dyn.w_off[0];
  }
  return 0;
}

static void pack_jobs(const NetPlan& n, char* ws, const float* params, int pair_kb, PackArgs& P, int prec = 0,
                      bool general = false) {
  for (int l = 0; l < n.nl; ++l) {
    const int O = n.dim[l + 1], K = n.dim[l];
    if (general && prec) {
      // general family on split operands: every layer as two piece planes of whole K32 blocks
      P.job[P.n++] = PackJob{params + n.w_off[l], reinterpret_cast<float*>(ws + n.wf[l]), O, K, 0, 1, 0, 2, 1};
      P.job[P.n++] = PackJob{params + n.w_off[l], reinterpret_cast<float*>(ws + n.wb[l]), O, K, 1, 1, 0, 2, 0};
      P.job[P.n++] = PackJob{params + n.b_off[l], reinterpret_cast<float*>(ws + n.bias[l]), O, K, 0, 1, 1, 0, 0};
      continue;
    }
    // hidden->hidden layers feed the streamed GEMMs of the fast kernels: k-blocks padded to CA+CB
    const int mult = (pair_kb >= 1 && l >= 1 && l <= n.nl - 2) ? pair_kb : 1;
    // split precision: bf16 pieces for every product with K = a hidden width -- forward: hidden->hidden layers
    // (streamed, padded to stage pairs) and the head (3 pieces); adjoint: their transposes and the tail (2 pieces)
    const bool hid_in = l >= 1, hid_out = l <= n.nl - 2;   // K is a hidden width forward / in the transposed product
    const int pf = (prec && hid_in) ? (prec == PMBRL_PREC_SPLIT_F16 ? 2 : 3) : 0, pb = (prec && hid_out) ? 2 : 0;
    const int mf = pf ? ((l <= n.nl - 2) ? pair_kb : 1) : mult, mb = pb ? ((l >= 1) ? pair_kb : 1) : mult;
    P.job[P.n++] = PackJob{params + n.w_off[l], reinterpret_cast<float*>(ws + n.wf[l]), O, K, 0, mf, 0, pf,
                           prec == PMBRL_PREC_SPLIT_F16 ? 1 : 0};
    P.job[P.n++] = PackJob{params + n.w_off[l], reinterpret_cast<float*>(ws + n.wb[l]), O, K, 1, mb, 0, pb, 0};
    P.job[P.n++] = PackJob{params + n.b_off[l], reinterpret_cast<float*>(ws + n.bias[l]), O, K, 0, 1, 1, 0, 0};
  }
}

template <int RT>
static void launch_fwd(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s) {
  hipLaunchKernelGGL(pm_rollout_fwd<RT>, dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
}
template <int RT>
static void launch_bwd(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s) {
  hipLaunchKernelGGL(pm_rollout_bwd<RT>, dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
}
static void launch_fast_rt(const pmbrl_plan* p, const RolloutArgs& A0, hipStream_t s, bool fwd) {
  // (more groups than fit the chip at once: batches of whole groups, one launch each -- pmbrl_host.h, mm_gpb)
  const int per = (p->mm_gpb > 0 && A0.mm_mode == 1) ? p->mm_gpb * p->mm_parts : p->nwg;
  for (int w0 = 0; w0 < p->nwg; w0 += per) {
    RolloutArgs A = A0;
    A.wg0 = w0;
    A.launch_wg = std::min(per, p->nwg - w0);
    if (p->prec == PMBRL_PREC_SPLIT_F16) pm_fast_split2_launch(p, A, s, fwd);
    else if (p->prec) pm_fast_split1_launch(p, A, s, fwd);
    else pm_fast_f32_launch(p, A, s, fwd);
  }
}
static void launch_fwd_rt(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s) {
  if (p->fast) return launch_fast_rt(p, A, s, true);
  if (p->prec) return pm_general_split_launch(p, A, s, true);
  switch (p->RT) {
    case 1: launch_fwd<1>(p, A, s); break;
    case 2: launch_fwd<2>(p, A, s); break;
    default: launch_fwd<4>(p, A, s); break;
  }
}
static void launch_bwd_rt(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s) {
  if (p->fast) return launch_fast_rt(p, A, s, false);
  if (p->prec) return pm_general_split_launch(p, A, s, false);
  switch (p->RT) {
    case 1: launch_bwd<1>(p, A, s); break;
    case 2: launch_bwd<2>(p, A, s); break;
    default: launch_bwd<4>(p, A, s); break;
  }
}

// groups spread over ranks (pmbrl_mmx.h): the in-place fp64 sum of the statistics buffer over the ranks
static int mmx_exchange(pmbrl_plan* p, hipStream_t s, double* buf, size_t n) {
  if (p->cfg.mm_span_ranks == 1 && !p->coll) return 0;   // one rank: the sum over the ranks is the identity
  if (!p->coll) return fail(-3, "moment-matching groups spread over ranks: attach a collective first (pmbrl_plan_set_comm / pmbrl_plan_set_collective)");
  if (int rc = p->coll(p->coll_ctx, (void*)s, buf, (int64_t)n)) return fail(-4, "statistics all-reduce failed (" + std::to_string(rc) + ")");
  return 0;
}
// dynamic LDS of pm_mmx_apply_kernel: room for the slots of every (rank, workgroup) if that fits (pmbrl_mmx.h)
static size_t mmx_apply_lds(int d, const MmxArgs& X, size_t other) {
  const size_t want = pm_mmx_apply_lds_doubles(d, X.nranks * X.nb) * sizeof(double);
  return want <= PM_MMX_LDS_MAX ? std::max(want, other) : other;
}
static MmxArgs mmx_args(const pmbrl_plan* p, char* ws, bool rewards) {
  MmxArgs X;
  X.nranks = p->cfg.mm_span_ranks;
  X.rank = p->cfg.mm_span_rank;
  X.span_rows = p->cfg.mm_span_rows;
  X.span_off = p->cfg.mm_span_offset;
  X.buf = reinterpret_cast<double*>(ws + p->off_mmx_buf);
  X.fac = reinterpret_cast<double*>(ws + (rewards ? p->off_mmx_rfac : p->off_mmx_fac));
  X.nb = rewards ? 1 : pm_mmx_blocks(p->M);   // (rewards: H x G workgroups already)
  return X;
}

// the sweep families' fragment-packed weights of this call's parameters (pm_pack_all); status_d: the forward
// sweep's status word, reset by the launch (nullptr: leave it)
static void launch_pack_all(pmbrl_plan* p, char* ws, const pmbrl_inputs* in, hipStream_t s, int32_t* status_d) {
  PackArgs PK;
  PK.n = 0;
  pack_jobs(p->pol, ws, in->pol_params_d, p->fast ? p->CA + p->CB : 0, PK, p->prec, !p->fast);
  pack_jobs(p->dyn, ws, in->dyn_params_d, p->fast ? p->CA + p->CB : 0, PK, p->prec, !p->fast);
  PK.status = status_d;
  PK.wflag = p->prec == PMBRL_PREC_SPLIT_F16 ? p->wflag_d : nullptr;
  PK.gen = p->wgen;
  hipLaunchKernelGGL(pm_pack_all, dim3(32, PK.n), dim3(256), 0, s, PK);
}

static int rollout_fwd_impl(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                            float* states_d, float* actions_d, float* rewards_d, int32_t* status_d);

extern "C" int pmbrl_rollout_fwd(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                                 float* states_d, float* actions_d, float* rewards_d,
                                 int32_t* status_d) {
  if (!p || !workspace || !in || !states_d || !actions_d || !rewards_d || !status_d)
    return fail(-1, "null argument");
  if (!in->pol_params_d || !in->dyn_params_d) return fail(-1, "null parameter pointer");
  HIPCHK(hipSetDevice(p->device));
  ReplayKey K;
  K.val(stream); K.val(workspace); K.add(in, sizeof(*in));
  K.val(states_d); K.val(actions_d); K.val(rewards_d); K.val(status_d); K.val(p->loss_w); K.val(p->loss_out);
  return replay_call(p, 0, K, (hipStream_t)stream, [&](hipStream_t on) {
    return rollout_fwd_impl(p, on, workspace, in, states_d, actions_d, rewards_d, status_d);
  });
}

static int rollout_fwd_impl(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                            float* states_d, float* actions_d, float* rewards_d, int32_t* status_d) {
  hipStream_t s = (hipStream_t)stream;
  RolloutArgs A;
  int rc = fill_args(p, workspace, in, A);
  if (rc) return rc;
  A.states = states_d; A.actions = actions_d; A.rewards = rewards_d; A.status = status_d;
  A.prof = p->prof_fwd;
  char* ws = static_cast<char*>(workspace);
  // the register-resident family serves this call (pmbrl_reg.h): only ITS weights are packed now -- the
  // latency-optimised family's are packed by the adjoint call if it turns out to need them (optional outputs)
  const bool reg_fwd = pm_reg_can_run(p, A, true);
  bool ztab_done = false;
  {
    ScopedTimer tm(p, PMBRL_TIMER_PACK, s);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    // (the flag only ever grows: start over before the counter wraps -- and in every recorded call, whose replays all
    //  carry the generation number of the recording)
    if (p->wgen >= 0x7ffffff0 || cap == hipStreamCaptureStatusActive) {
      HIPCHK(hipMemsetAsync(p->wflag_d, 0, sizeof(int), s));
      if (p->wgen >= 0x7ffffff0) p->wgen = 0;
    }
    ++p->wgen;
    if (!reg_fwd) {
      launch_pack_all(p, ws, in, s, status_d);
      p->old_pack_stale = 0;
    } else {
      p->old_pack_stale = 1;
    }
    p->abits_packed = reg_fwd ? 1 : 0;
    // (the noise table of split moment-matching groups rides in the same launch: extra workgroups, PMBRL_ZTAB_MERGE=0 for
    //  the launch of its own)
    ZtabArgs Zt;
    Zt.n_blocks = 0;
    if (p->reg && reg_fwd && (p->mm_fan || p->reg_mm) && p->mm_mode == 1 &&
        !(getenv("PMBRL_ZTAB_MERGE") && atoi(getenv("PMBRL_ZTAB_MERGE")) == 0)) {
      Zt = pm_ztab_args(p, A);
      ztab_done = true;
    }
    if (p->reg) pm_reg_pack_launch(p, ws, in->pol_params_d, in->dyn_params_d, p->wflag_d, p->wgen, s, reg_fwd ? status_d : nullptr, &Zt);
  }
  A.wflag = p->prec == PMBRL_PREC_SPLIT_F16 ? p->wflag_d : nullptr;
  A.wgen = p->wgen;
  const bool mm_r = (p->cfg.flags & PMBRL_FLAG_MM_REWARDS) != 0;
  {
  ScopedTimer tm(p, PMBRL_TIMER_FWD, s);
  if (p->mm_mode == 3 && p->mm_grid) {
    // one launch; the workgroups meet after every step and moment-match the sampled states (x_H included)
    HIPCHK(hipMemsetAsync(A.gsync, 0, 1024 * sizeof(unsigned), s));
    A.t0 = 0; A.t1 = p->cfg.H;
    launch_fwd_rt(p, A, s);
  } else if (p->mm_mode == 3) {
    // one launch per step; step t's launch first moment-matches the states sampled by step t-1
    for (int t = 0; t < p->cfg.H; ++t) {
      A.t0 = t; A.t1 = t + 1;
      launch_fwd_rt(p, A, s);
    }
    RolloutArgs Am = A;
    Am.flags &= ~PMBRL_FLAG_MM_REWARDS;
    hipLaunchKernelGGL(pm_mm_fwd_kernel, dim3(p->G), dim3(PM_MM_NW * 64),
                       pm_mm_kernel_doubles(p->cfg.D) * sizeof(double), s, Am, p->cfg.H - 1);   // x_H
  } else if (p->mm_mode != 2) {
    RolloutArgs As = A;
    if (!p->fast) { As.ext_reward = 1; As.flags &= ~PMBRL_FLAG_MM_REWARDS; }   // general family: rewards after the sweep
    if ((p->mm_fan || p->reg_mm) && !ztab_done) {
      const ZtabArgs Z = pm_ztab_args(p, As);
      hipLaunchKernelGGL(pm_mm_ztable_kernel, dim3(Z.n_blocks), dim3(256), 0, s, Z);
    }
    // (the granules' tags: zeroed per launch for the latency-optimised family, whose tags count the steps from 1; the
    //  register-resident family's carry a launch generation instead -- once zeroed, never again: two 5 us fill kernels less;
    //  nor does that family touch the group-local barrier flags: two more)
    const bool reg_now = pm_reg_can_run(p, As, true);
    if (p->mm_parts > 1 && !reg_now) HIPCHK(hipMemsetAsync(As.gsync, 0, 1024 * sizeof(unsigned), s));   // group-local barriers
    // (a RECORDED launch is replayed with the generation it was recorded with: two replays of a forward call with no
    //  adjoint call between them would meet their own tags of the replay before -- stale sums accepted.  A recording
    //  therefore carries the fill; what it leaves behind is its generation's tags, which no later launch takes for its own)
    const bool xch_rec = pm_stream_recording(s);
    if (p->mm_parts > 1 && As.xch && (!reg_now || !p->xch_zeroed || xch_rec)) {
      HIPCHK(hipMemsetAsync(As.xch, 0, p->xch_bytes, s));
      p->xch_zeroed = (reg_now && !xch_rec) ? 1 : 0;
    }
    if (reg_now) pm_reg_launch(p, ws, As, in->pol_params_d, in->dyn_params_d, s, true);
    else launch_fwd_rt(p, As, s);
  } else {
    const size_t smem = pm_mm_kernel_doubles(p->cfg.D) * sizeof(double);
    RolloutArgs Am = A, As = A;
    Am.flags &= ~PMBRL_FLAG_MM_REWARDS;   // both families: rewards are handled after the sweep
    if (!p->fast) { As.ext_reward = 1; As.flags &= ~PMBRL_FLAG_MM_REWARDS; }
    for (int t = 0; t < p->cfg.H; ++t) {
      As.t0 = t; As.t1 = t + 1;
      launch_fwd_rt(p, As, s);
      if (p->mm_wide) {
        hipLaunchKernelGGL(pm_mmw_fwd_kernel, dim3(p->G), dim3(PM_MMW_NT), pm_mmw_lds_bytes(p->M, p->cfg.D, false), s, Am, t,
                           reinterpret_cast<double*>(ws + p->off_mmfac));
      } else if (!p->span) {
        hipLaunchKernelGGL(pm_mm_fwd_kernel, dim3(p->G), dim3(PM_MM_NW * 64), smem, s, Am, t);
      } else if (p->cfg.flags & PMBRL_FLAG_MM_STATES) {
        // groups spread over ranks: own statistics -> sum over the ranks -> factor + own rows
        MmxArgs X = mmx_args(p, ws, false);
        X.fac += (size_t)t * p->G * pm_mm_fac_doubles(p->cfg.D);
        const int nw = pm_mmx_waves(p->M, p->cfg.D), nws = pm_mmx_waves((p->M + X.nb - 1) / X.nb, p->cfg.D);
        const size_t scr = pm_mmx_lds_doubles(p->cfg.D, nw) * sizeof(double);
        hipLaunchKernelGGL(pm_mmx_stats_kernel<0>, dim3(p->G * X.nb, X.nranks), dim3(64 * nws), scr, s, Am, X, t);
        if (int rc = mmx_exchange(p, s, X.buf, (size_t)X.nranks * X.nb * p->G * pm_mmx_slot_doubles(p->cfg.D))) return rc;
        hipLaunchKernelGGL(pm_mmx_apply_kernel<0>, dim3(p->G), dim3(64 * nw), mmx_apply_lds(p->cfg.D, X, scr), s, Am, X, t);
      }
    }
  }
  }
  {
    // rewards (+ Jacobians) of all row-steps in one parallel pass, then their moment matching
    ScopedTimer tm(p, PMBRL_TIMER_REWARD, s);
    A.t0 = 0; A.t1 = p->cfg.H;
    const long long n = (long long)p->cfg.H * p->cfg.B;
    hipLaunchKernelGGL(pm_reward_kernel_for(p->cfg.U, p->rew_k), dim3((unsigned)((n + 255) / 256)), dim3(256),
                       (size_t)256 * (p->cfg.D | 1) * sizeof(float), s, A);
    if (mm_r && p->span) {
      // the rewards of all steps: one exchange for the whole horizon
      const MmxArgs X = mmx_args(p, ws, true);
      const int nw = std::min(4, pm_mmx_waves(p->M, 1));   // (H x G workgroups: fewer waves each)
      const size_t scr = pm_mmx_lds_doubles(1, nw) * sizeof(double);
      const int n_items = p->cfg.H * p->G;
      hipLaunchKernelGGL(pm_mmx_stats_kernel<1>, dim3(n_items, X.nranks), dim3(64 * nw), scr, s, A, X, 0);
      if (int rc = mmx_exchange(p, s, X.buf, (size_t)X.nranks * n_items * pm_mmx_slot_doubles(1))) return rc;
      hipLaunchKernelGGL(pm_mmx_apply_kernel<1>, dim3(n_items), dim3(64 * nw), mmx_apply_lds(1, X, scr), s, A, X, 0);
    } else if (mm_r)
      hipLaunchKernelGGL(pm_mm_rewards_fwd_kernel, dim3(p->cfg.H * p->G), dim3(64), pm_mmr_lds_bytes(p->M, 2), s, A);
    if (p->loss_w) {
      // (folding this sum into the reward launch -- per-block partials, last block finishes -- was measured: the
      //  reward launch grew by what the separate reduction costs, 391 arrivals on one counter; not kept)
      const long long nn = (long long)p->cfg.H * p->cfg.B;
      const int nb = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (nn + 2047) / 2048));
      hipLaunchKernelGGL(pm_weighted_sum_kernel, dim3(nb), dim3(256), 0, s, (const float*)rewards_d, p->loss_w, nn,
                         p->loss_out, (const int*)status_d, (long long)p->cfg.B);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_plan_set_loss(pmbrl_plan* p, const float* loss_weights_d, float* loss_out_d) {
  if (!p || (loss_weights_d && !loss_out_d)) return fail(-1, "bad argument");
  p->loss_w = loss_weights_d;
  p->loss_out = loss_weights_d ? loss_out_d : nullptr;
  return 0;
}

static int rollout_bwd(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                       const float* states_d, const float* actions_d,
                       const float* rewards_d, const float* grad_rewards_d,
                       const float* grad_states_d, const float* grad_actions_d,
                       float* grad_pol_flat_d, float* grad_x0_d,
                       float* action_grad_norms_d, int32_t* status_d, const pmbrl_adam* opt) {
  if (!p || !workspace || !in || !states_d || !actions_d || !rewards_d || !grad_rewards_d ||
      !grad_pol_flat_d)
    return fail(-1, "null argument");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(p->device));
  RolloutArgs A;
  int rc = fill_args(p, workspace, in, A);
  if (rc) return rc;
  char* ws = static_cast<char*>(workspace);
  A.states = const_cast<float*>(states_d);
  A.actions = const_cast<float*>(actions_d);
  A.rewards = const_cast<float*>(rewards_d);
  A.grad_rewards = grad_rewards_d;
  A.grad_states = grad_states_d;
  A.grad_actions = grad_actions_d;
  A.grad_x0 = grad_x0_d;
  A.agn = action_grad_norms_d;
  A.prof = p->prof_bwd;
  // status_d[0]: valid steps of the forward sweep (the adjoint covers only those); status_d[1]: set by
  // the sweep if its device-wide barrier timed out
  A.nvalid = status_d;
  A.status = status_d ? status_d + 1 : nullptr;
  // (cleared by the FORWARD call's reward launch -- pm_reward_all_kernel -- since round 5: as a memset of its own here it
  //  was a 5 us fill kernel in every iteration, 1 % of the cart-pole shape's)
  // this call goes to the latency-optimised family after a forward call that packed the register-resident family's
  // weights only: pack the others now (same parameters: the caller hands the same inputs to both calls)
  const bool reg_bwd = pm_reg_can_run(p, A, false);
  if (p->reg && p->old_pack_stale && !reg_bwd) {
    if (!in->pol_params_d || !in->dyn_params_d) return fail(-1, "null parameter pointer");
    launch_pack_all(p, ws, in, s, nullptr);
  }
  // ... and its activity bits are in that family's form.  (Neither note is cleared here: a recorded adjoint call --
  // pmbrl_plan_set_replay -- must contain both conversions whenever the forward call before it was of that kind.)
  if (p->reg && p->abits_packed && !reg_bwd) pm_reg_unpack_abits(p, ws, s);
  float* grt = reinterpret_cast<float*>(ws + p->off_grt);
  const bool mm_r = (p->cfg.flags & PMBRL_FLAG_MM_REWARDS) != 0;
  if (!p->fast) { A.ext_reward = 1; }
  if (mm_r && p->span) {
    const MmxArgs X = mmx_args(p, ws, true);
    const int nw = std::min(4, pm_mmx_waves(p->M, 1));
    const size_t scr = pm_mmx_lds_doubles(1, nw) * sizeof(double);
    const int n_items = p->cfg.H * p->G;
    hipLaunchKernelGGL(pm_mmx_bwd_sums_kernel<1>, dim3(n_items), dim3(64 * nw), scr, s, A, X, 0);
    if (int rc = mmx_exchange(p, s, X.buf, (size_t)n_items * pm_mmx_bwd_doubles(1))) return rc;
    hipLaunchKernelGGL(pm_mmx_bwd_apply_kernel<1>, dim3(n_items), dim3(64 * nw), scr, s, A, X, 0, grt);
    A.grad_rewards = grt;
  } else if (mm_r) {
    // adjoint of the reward moment matching for all (t, group) up front
    hipLaunchKernelGGL(pm_mm_rewards_bwd_kernel, dim3(p->cfg.H * p->G), dim3(64), pm_mmr_lds_bytes(p->M, 3), s, A, grt);
    A.grad_rewards = grt;
  }
  if (!p->fast) A.flags &= ~PMBRL_FLAG_MM_REWARDS;   // (general sweeps: the reward side is done, see ext_reward)
  // dW GEMM over the stashes + deterministic reduction
  DwArgs W;
  memset(&W, 0, sizeof(W));
  W.nl = p->pol.nl;
  W.nsplit = p->dw_nsplit;
  W.n_chunks = p->dw_n_chunks;
  W.chunks_per_split = p->dw_chunks_per_split;
  W.RT = p->RT;
  W.Rw = 16 * p->RT;
  W.n_params = (int)p->pol.n_params;
  W.part_stride = (W.n_params + 3) / 4 * 4;
  for (int i = 0; i <= p->pol.nl; ++i) {
    W.dim[i] = p->pol.dim[i];
    W.nt[i] = p->pol.nt[i];
  }
  for (int l = 0; l < p->pol.nl; ++l) {
    W.w_off[l] = (int)p->pol.w_off[l];
    W.b_off[l] = (int)p->pol.b_off[l];
    W.actT[l] = A.actT[l];
    W.gT[l] = A.gT[l];
  }
  W.blocks = p->dw_blocks_d;
  for (int w = 0; w <= PM_DW_NW; ++w) W.wave_first[w] = p->dw_wave_first[w];
  W.part = reinterpret_cast<float*>(ws + p->off_part);
  W.nvalid = status_d;
  W.chunks_per_step = p->nwg * p->RT;
  W.split_prec = p->dw_split;
  const bool pipe = p->pipe_K > 1 && !p->prof_bwd;
  auto launch_dw = [&](int k, hipStream_t st) {
    DwArgs Wk = W;
    const int cps = W.chunks_per_step;
    Wk.pipe = 1;
    Wk.pipe_k = k;
    const bool last = k + 1 == p->pipe_K;
    Wk.nsplit = last ? p->pipe_rows : p->pipe_grid;
    Wk.rows_before = p->pipe_grid;
    Wk.c_begin = p->pipe_lo[k] * cps;
    Wk.c_end = (k ? p->pipe_lo[k - 1] : p->cfg.H) * cps;
    Wk.chunks_per_split = (Wk.c_end - Wk.c_begin + Wk.nsplit - 1) / Wk.nsplit;
    if (p->dw_split) Wk.chunks_per_split = (Wk.chunks_per_split + 1) & ~1;
    if (p->dw_split) hipLaunchKernelGGL(pm_dw_kernel_s, dim3(Wk.nsplit), dim3(PM_DW_NT), 0, st, Wk);
    else hipLaunchKernelGGL(pm_dw_kernel, dim3(Wk.nsplit), dim3(PM_DW_NT), 0, st, Wk);
    if (p->n_dw_units && p->dw_pre_mask)
      hipLaunchKernelGGL(pm_dw_wide_pre_kernel, dim3((Wk.nsplit + 7) / 8 * 8 * p->n_dw_units), dim3(PM_DWP_NT), PM_DWP_LDS_BYTES, st, Wk,
                         p->dw_units_d, p->n_dw_units);
    else if (p->n_dw_units)
      hipLaunchKernelGGL(pm_dw_wide_kernel, dim3((Wk.nsplit + 7) / 8 * 8 * p->n_dw_units), dim3(PM_DW_NT), PM_DWW_LDS_BYTES, st, Wk,
                         p->dw_units_d, p->n_dw_units);
    if (p->dw_layer13 >= 0)
      hipLaunchKernelGGL(pm_dw_layer_kernel, dim3(Wk.nsplit), dim3(PM_DW_NT), PM_DWL_LDS_BYTES, st, Wk, p->dw_layer13);
  };
  if (p->mm_mode == 3) {
    ScopedTimer tm(p, PMBRL_TIMER_BWD, s);
    if (grad_states_d) return fail(-3, "grad_states with moment-matching groups spanning workgroups: not offered");
    float* cbuf[2] = {A.gx_carry, reinterpret_cast<float*>(ws + p->off_gxc2)};
    HIPCHK(hipMemsetAsync(cbuf[0], 0, (size_t)p->cfg.B * p->cfg.D * sizeof(float), s));
    A.gx_from_carry = 1;
    int k = 0;
    if (p->mm_grid) {
      A.gsync += 1024;
      HIPCHK(hipMemsetAsync(A.gsync, 0, 1024 * sizeof(unsigned), s));
      A.gx_carry_out = cbuf[1];
      A.t0 = 0; A.t1 = p->cfg.H;
      launch_bwd_rt(p, A, s);
    } else
    for (int t = p->cfg.H - 1; t >= 0; --t, k ^= 1) {
      A.gx_carry = cbuf[k];           // dL/dx_{t+1}, all rows (read by every workgroup of the group)
      A.gx_carry_out = cbuf[k ^ 1];   // dL/dx_t of this workgroup's rows
      A.t0 = t; A.t1 = t + 1;
      launch_bwd_rt(p, A, s);
    }
    A.gx_carry = cbuf[0];
    A.gx_carry_out = nullptr;
  } else if (p->mm_mode != 2 && pipe) {
    // range k of the sweep on s; the dW GEMM of range k on the second stream behind it (the last one below, on s)
    ScopedTimer tm(p, PMBRL_TIMER_BWD, s);
    for (int k = 0; k < p->pipe_K; ++k) {
      A.t0 = p->pipe_lo[k];
      A.t1 = k ? p->pipe_lo[k - 1] : p->cfg.H;
      A.gx_from_carry = k > 0;
      if (reg_bwd) pm_reg_launch(p, ws, A, in->pol_params_d, in->dyn_params_d, s, false);
      else launch_bwd_rt(p, A, s);
      if (k + 1 < p->pipe_K) {
        HIPCHK(hipEventRecord(p->pipe_ev[k], s));
        HIPCHK(hipStreamWaitEvent(p->pipe_stream, p->pipe_ev[k], 0));
        launch_dw(k, p->pipe_stream);
      }
    }
    A.t0 = 0; A.t1 = p->cfg.H;
  } else if (p->mm_mode != 2) {
    ScopedTimer tm(p, PMBRL_TIMER_BWD, s);
    A.gx_from_carry = 0;
    if (p->mm_parts > 1) {
      // groups split over workgroups: their own flags, and two buffers for the rows of dL/dx they exchange
      A.gsync += 1024;
      if (!reg_bwd) HIPCHK(hipMemsetAsync(A.gsync, 0, 1024 * sizeof(unsigned), s));
      const bool xch_rec = pm_stream_recording(s);      // (see the forward call)
      if (A.xch && (!reg_bwd || !p->xch_zeroed || xch_rec)) {
        HIPCHK(hipMemsetAsync(A.xch, 0, p->xch_bytes, s));
        p->xch_zeroed = (reg_bwd && !xch_rec) ? 1 : 0;
      }
      A.gx_carry_out = reinterpret_cast<float*>(ws + p->off_gxc2);
    }
    if (reg_bwd) pm_reg_launch(p, ws, A, in->pol_params_d, in->dyn_params_d, s, false);
    else launch_bwd_rt(p, A, s);
    A.gx_carry_out = nullptr;
  } else {
    ScopedTimer tm(p, PMBRL_TIMER_BWD, s);
    if (grad_states_d) return fail(-3, "grad_states with external moment matching: not offered");
    const size_t smem = pm_mm_kernel_doubles(p->cfg.D) * sizeof(double);
    HIPCHK(hipMemsetAsync(A.gx_carry, 0, (size_t)p->cfg.B * p->cfg.D * sizeof(float), s));
    RolloutArgs Am = A;
    Am.flags &= ~PMBRL_FLAG_MM_REWARDS;   // rewards already handled above
    A.gx_from_carry = 1;
    for (int t = p->cfg.H - 1; t >= 0; --t) {
      if (p->mm_wide) {
        hipLaunchKernelGGL(pm_mmw_bwd_kernel, dim3(p->G), dim3(PM_MMW_NT), pm_mmw_lds_bytes(p->M, p->cfg.D, true), s, Am, t,
                           reinterpret_cast<const double*>(ws + p->off_mmfac));
      } else if (!p->span) {
        hipLaunchKernelGGL(pm_mm_bwd_kernel, dim3(p->G), dim3(PM_MM_NW * 64), smem, s, Am, t, grt);
      } else if (p->cfg.flags & PMBRL_FLAG_MM_STATES) {
        MmxArgs X = mmx_args(p, ws, false);
        X.fac += (size_t)t * p->G * pm_mm_fac_doubles(p->cfg.D);
        const int nw = pm_mmx_waves(p->M, p->cfg.D), nws = pm_mmx_waves((p->M + X.nb - 1) / X.nb, p->cfg.D);
        const size_t scr = pm_mmx_lds_doubles(p->cfg.D, nw) * sizeof(double);
        hipLaunchKernelGGL(pm_mmx_bwd_sums_kernel<0>, dim3(p->G * X.nb), dim3(64 * nws), scr, s, Am, X, t);
        if (int rc = mmx_exchange(p, s, X.buf, (size_t)X.nb * p->G * pm_mmx_bwd_doubles(p->cfg.D))) return rc;
        hipLaunchKernelGGL(pm_mmx_bwd_apply_kernel<0>, dim3(p->G), dim3(64 * nw), scr, s, Am, X, t, (float*)nullptr);
      }
      A.t0 = t; A.t1 = t + 1;
      launch_bwd_rt(p, A, s);
    }
  }
  const bool piped = pipe && p->mm_mode != 2 && p->mm_mode != 3;
  {
    ScopedTimer tm(p, PMBRL_TIMER_DW, s);
    if (piped) {
      // the rows of the last range are added to what the second stream left: join it first
      HIPCHK(hipEventRecord(p->pipe_ev[p->pipe_K - 1], p->pipe_stream));
      HIPCHK(hipStreamWaitEvent(s, p->pipe_ev[p->pipe_K - 1], 0));
      launch_dw(p->pipe_K - 1, s);
    }
    else {
      if (p->n_dw_blocks == 0) {}      // (every layer has a GEMM kernel of its own: below)
      else if (p->dw_split) hipLaunchKernelGGL(pm_dw_kernel_s, dim3(p->dw_nsplit), dim3(PM_DW_NT), 0, s, W);
      else hipLaunchKernelGGL(pm_dw_kernel, dim3(p->dw_nsplit), dim3(PM_DW_NT), 0, s, W);
      if (p->dw_pre_mask)
        for (int l = 0; l < p->pol.nl; ++l) {
          if (!p->dw_narrow[l]) continue;
          const int g_wide = p->dw_narrow[l] == 1 ? 1 : 0, nt_n = g_wide ? p->pol.nt[l] : p->pol.nt[l + 1];
          if (nt_n <= 1) hipLaunchKernelGGL(pm_dw_narrow_pre_kernel<1>, dim3(p->dw_nsplit), dim3(PM_DWP_NT), PM_DWP_LDS_BYTES, s, W, l, g_wide);
          else hipLaunchKernelGGL(pm_dw_narrow_pre_kernel<2>, dim3(p->dw_nsplit), dim3(PM_DWP_NT), PM_DWP_LDS_BYTES, s, W, l, g_wide);
        }
      if (p->n_dw_units && p->dw_pre_mask)
        hipLaunchKernelGGL(pm_dw_wide_pre_kernel, dim3((p->dw_nsplit + 7) / 8 * 8 * p->n_dw_units), dim3(PM_DWP_NT), PM_DWP_LDS_BYTES, s, W,
                           p->dw_units_d, p->n_dw_units);
      else if (p->n_dw_units)
        hipLaunchKernelGGL(pm_dw_wide_kernel, dim3((p->dw_nsplit + 7) / 8 * 8 * p->n_dw_units), dim3(PM_DW_NT), PM_DWW_LDS_BYTES, s, W,
                           p->dw_units_d, p->n_dw_units);
      if (p->dw_layer13 >= 0)
        hipLaunchKernelGGL(pm_dw_layer_kernel, dim3(p->dw_nsplit), dim3(PM_DW_NT), PM_DWL_LDS_BYTES, s, W, p->dw_layer13);
    }
  }
  const int n = (int)p->pol.n_params;
  // the fused iteration (one process, optimiser attached): the reduction forms the gradient norm's partial sums and the
  // go / no-go decision on the way -- pm_gradnorm_kernel's launch is not needed (PMBRL_FUSE_NORM=0: the separate launch)
  const bool norm_fused = opt && !piped && (n + 255) / 256 <= PM_NORM_MAXB &&
                          !(getenv("PMBRL_FUSE_NORM") && atoi(getenv("PMBRL_FUSE_NORM")) == 0);
  // ... and, asked for (pmbrl_adam::loss_out_d), the loss' partial sums: the optimiser launch adds them -- no loss launch
  // behind the forward call (PMBRL_FUSE_LOSS=0: pm_weighted_sum_kernel, here)
  const bool loss_fused = norm_fused && opt->loss_out_d && !(getenv("PMBRL_FUSE_LOSS") && atoi(getenv("PMBRL_FUSE_LOSS")) == 0);
  {
    ScopedTimer tm(p, PMBRL_TIMER_DW_REDUCE, s);
    if (piped)   // every partial row was written (pm_dw_range)
      hipLaunchKernelGGL(pm_dw_reduce, dim3((n + 255) / 256), dim3(512), 0, s, W.part, p->pipe_rows, n,
                         W.part_stride, grad_pol_flat_d, (const int*)nullptr, 0, 1);
    else
      hipLaunchKernelGGL(pm_dw_reduce, dim3((n + 255) / 256), dim3(512), 0, s, W.part, p->dw_nsplit, n,
                         W.part_stride, grad_pol_flat_d, (const int*)status_d, W.chunks_per_step, W.chunks_per_split,
                         norm_fused ? 1 : 0, (const int*)status_d, opt ? (opt->expect_steps > 0 ? opt->expect_steps : p->cfg.H) : 0,
                         opt ? reinterpret_cast<long long*>(opt->step_d) : (long long*)nullptr,
                         loss_fused ? rewards_d : (const float*)nullptr, grad_rewards_d, (long long)p->cfg.H * p->cfg.B,
                         (long long)p->cfg.B);
  }
  // the loss asked for with the optimiser step, where the reduction does not form it on the way: a launch of its own
  if (opt && opt->loss_out_d && !loss_fused) {
    const long long nn = (long long)p->cfg.H * p->cfg.B;
    const int nbl = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (nn + 2047) / 2048));
    hipLaunchKernelGGL(pm_weighted_sum_kernel, dim3(nbl), dim3(256), 0, s, rewards_d, grad_rewards_d, nn, opt->loss_out_d,
                       (const int*)status_d, (long long)p->cfg.B);
  }
  HIPCHK(hipGetLastError());
  // the optimiser step behind it, decided on the device (a form with the reduction, the norm and the update in ONE launch
  // around a device-wide barrier was measured at the C2 shape: 25.6 us against 13.7 + 4.7 + 7.1 us for the three launches
  // -- the barrier's L2 round trips cost what two launches do; not kept)
  if (opt)
    return clip_adam_guarded_impl(stream, opt->params_d, grad_pol_flat_d, opt->exp_avg_d, opt->exp_avg_sq_d, n, opt->step_d,
                                  opt->lr, opt->beta1, opt->beta2, opt->eps, opt->max_norm, opt->norm_out_d, status_d,
                                  opt->expect_steps > 0 ? opt->expect_steps : p->cfg.H, norm_fused,
                                  loss_fused ? opt->loss_out_d : nullptr, loss_fused ? (n + 255) / 256 : 0);
  return 0;
}

static int rollout_bwd_replay(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                              const float* states_d, const float* actions_d,
                              const float* rewards_d, const float* grad_rewards_d,
                              const float* grad_states_d, const float* grad_actions_d,
                              float* grad_pol_flat_d, float* grad_x0_d,
                              float* action_grad_norms_d, int32_t* status_d, const pmbrl_adam* opt) {
  if (!p || !in) return fail(-1, "null argument");
  HIPCHK(hipSetDevice(p->device));
  ReplayKey K;
  K.val(stream); K.val(workspace); K.add(in, sizeof(*in));
  K.val(states_d); K.val(actions_d); K.val(rewards_d); K.val(grad_rewards_d); K.val(grad_states_d); K.val(grad_actions_d);
  K.val(grad_pol_flat_d); K.val(grad_x0_d); K.val(action_grad_norms_d); K.val(status_d);
  const int has_opt = opt ? 1 : 0;
  K.val(has_opt);
  if (opt) {      // field by field: the struct's padding bytes are not arguments (a caller's stack copy need not repeat them)
    K.val(opt->params_d); K.val(opt->exp_avg_d); K.val(opt->exp_avg_sq_d); K.val(opt->step_d);
    K.val(opt->lr); K.val(opt->beta1); K.val(opt->beta2); K.val(opt->eps); K.val(opt->max_norm);
    K.val(opt->norm_out_d); K.val(opt->expect_steps);
  }
  // what the forward call left behind decides which conversions the adjoint call queues (weights repacked for the other
  // family, activity words unpacked): a recording is only good for the state it was recorded in
  K.val(p->old_pack_stale); K.val(p->abits_packed);
  return replay_call(p, 1, K, (hipStream_t)stream, [&](hipStream_t on) {
    return rollout_bwd(p, on, workspace, in, states_d, actions_d, rewards_d, grad_rewards_d, grad_states_d,
                       grad_actions_d, grad_pol_flat_d, grad_x0_d, action_grad_norms_d, status_d, opt);
  });
}

extern "C" int pmbrl_rollout_bwd(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                                 const float* states_d, const float* actions_d,
                                 const float* rewards_d, const float* grad_rewards_d,
                                 const float* grad_states_d, const float* grad_actions_d,
                                 float* grad_pol_flat_d, float* grad_x0_d,
                                 float* action_grad_norms_d, int32_t* status_d) {
  return rollout_bwd_replay(p, stream, workspace, in, states_d, actions_d, rewards_d, grad_rewards_d, grad_states_d,
                            grad_actions_d, grad_pol_flat_d, grad_x0_d, action_grad_norms_d, status_d, nullptr);
}

extern "C" int pmbrl_rollout_bwd_adam(pmbrl_plan* p, void* stream, void* workspace, const pmbrl_inputs* in,
                                      const float* states_d, const float* actions_d,
                                      const float* rewards_d, const float* grad_rewards_d,
                                      const float* grad_states_d, const float* grad_actions_d,
                                      float* grad_pol_flat_d, float* grad_x0_d,
                                      float* action_grad_norms_d, int32_t* status_d, const pmbrl_adam* opt) {
  if (!opt || !status_d || !opt->params_d || !opt->exp_avg_d || !opt->exp_avg_sq_d || !opt->step_d)
    return fail(-1, "null argument");
  return rollout_bwd_replay(p, stream, workspace, in, states_d, actions_d, rewards_d, grad_rewards_d, grad_states_d,
                            grad_actions_d, grad_pol_flat_d, grad_x0_d, action_grad_norms_d, status_d, opt);
}

// ---------------------------------------------------------------------------
// stand-alone network evaluation
// ---------------------------------------------------------------------------
struct MlpLayout {
  int nl, LD;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  size_t wf[PM_MAXL], wb[PM_MAXL], bias[PM_MAXL], w_off[PM_MAXL], b_off[PM_MAXL], bytes, lds, lds_bwd;
};
static int mlp_layout(const pmbrl_mlp_call* c, MlpLayout& L) {
  if (!c) return fail(-1, "null argument");
  const pmbrl_mlp& m = c->net;
  if (m.n_layers < 2 || m.n_layers > PM_MAXL) return fail(-2, "mlp: 2..8 layers");
  if (c->B < 1) return fail(-2, "mlp: B must be >= 1");
  if (m.dims[m.n_layers] % 2) return fail(-2, "mlp: the output layer must be 2 x n_out wide");
  L.nl = m.n_layers;
  int maxnt = 1;
  size_t off = 0, po = 0;
  for (int i = 0; i <= L.nl; ++i) {
    if (m.dims[i] < 1 || m.dims[i] > 16 * 64) return fail(-2, "mlp: layer width out of range");
    L.dim[i] = m.dims[i];
    L.nt[i] = (m.dims[i] + 15) / 16;
    maxnt = std::max(maxnt, L.nt[i]);
  }
  for (int l = 0; l < L.nl; ++l) {
    L.w_off[l] = po; po += (size_t)L.dim[l + 1] * L.dim[l];
    L.b_off[l] = po; po += (size_t)L.dim[l + 1];
    L.wf[l] = off; off += (size_t)L.nt[l + 1] * L.nt[l] * 256 * sizeof(float);
    L.wb[l] = off; off += (size_t)L.nt[l + 1] * L.nt[l] * 256 * sizeof(float);
    L.bias[l] = off; off += (size_t)L.nt[l + 1] * 16 * sizeof(float);
    if (l < L.nl - 1 && !(m.keep[l] > 0.f)) return fail(-2, "mlp: keep must be > 0");
  }
  L.bytes = off;
  L.LD = maxnt * 16 + 8;
  L.lds = (2 * (size_t)16 * L.LD + (size_t)PM_NW * PM_KS_NT * 256) * sizeof(float);
  L.lds_bwd = pm_mlp_bwd_lds_floats(L.nl, L.LD) * sizeof(float);
  if (L.lds > 160 * 1024) return fail(-3, "mlp: network too wide for LDS");
  return 0;
}

extern "C" size_t pmbrl_mlp_workspace_bytes(const pmbrl_mlp_call* c) {
  MlpLayout L;
  return mlp_layout(c, L) ? 0 : L.bytes;
}

extern "C" int pmbrl_mlp_forward(void* stream, const pmbrl_mlp_call* c, void* workspace_d,
                                 const float* x_d, const float* params_flat_d,
                                 const uint16_t* const* mask_bits_d, const float* z_d,
                                 const float* in_shift_d, const float* in_iscale_d,
                                 const float* out_scale_d, const float* out_shift_d,
                                 const float* sq_scale_d, const float* sq_bias_d,
                                 float* sample_d, float* mean_d, float* log_std_d) {
  MlpLayout L;
  int rc = mlp_layout(c, L);
  if (rc) return rc;
  if (!workspace_d || !x_d || !params_flat_d) return fail(-1, "null argument");
  if ((in_shift_d == nullptr) != (in_iscale_d == nullptr) || (out_scale_d == nullptr) != (out_shift_d == nullptr) ||
      (sq_scale_d == nullptr) != (sq_bias_d == nullptr))
    return fail(-1, "mlp: shift/scale pointers come in pairs");
  hipStream_t s = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace_d);
  PackArgs PK;
  PK.wflag = nullptr;
  PK.gen = 0;
  PK.n = 0;
  PK.status = nullptr;
  MlpArgs A;
  memset(&A, 0, sizeof(A));
  A.B = c->B; A.nl = L.nl; A.LD = L.LD; A.n_out = L.dim[L.nl] / 2;
  for (int i = 0; i <= L.nl; ++i) { A.dim[i] = L.dim[i]; A.nt[i] = L.nt[i]; }
  for (int l = 0; l < L.nl; ++l) {
    float* wf = reinterpret_cast<float*>(ws + L.wf[l]);
    float* bs = reinterpret_cast<float*>(ws + L.bias[l]);
    PK.job[PK.n++] = PackJob{params_flat_d + L.w_off[l], wf, L.dim[l + 1], L.dim[l], 0, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + L.b_off[l], bs, L.dim[l + 1], L.dim[l], 0, 1, 1, 0, 0};
    A.wf[l] = wf;
    A.bias[l] = bs;
    A.mask[l] = (l < L.nl - 1 && mask_bits_d) ? mask_bits_d[l] : nullptr;
    A.keep[l] = l < L.nl - 1 ? c->net.keep[l] : 1.f;
  }
  A.x = x_d; A.z = z_d; A.in_shift = in_shift_d; A.in_iscale = in_iscale_d;
  A.out_scale = out_scale_d; A.out_shift = out_shift_d; A.sq_scale = sq_scale_d; A.sq_bias = sq_bias_d;
  A.mls = c->max_log_std;
  A.sample = sample_d; A.mean = mean_d; A.log_std = log_std_d;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mlp_fwd_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
  hipLaunchKernelGGL(pm_pack_all, dim3(32, PK.n), dim3(256), 0, s, PK);
  hipLaunchKernelGGL(pm_mlp_fwd_kernel, dim3((c->B + 15) / 16), dim3(PM_NT), L.lds, s, A);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_mlp_grad_input(void* stream, const pmbrl_mlp_call* c, void* workspace_d,
                                    const float* x_d, const float* params_flat_d,
                                    const uint16_t* const* mask_bits_d, const float* z_d,
                                    const float* in_shift_d, const float* in_iscale_d,
                                    const float* out_scale_d, const float* out_shift_d,
                                    const float* sq_scale_d, const float* sq_bias_d,
                                    const float* g_sample_d, const float* g_mean_d,
                                    const float* g_log_std_d, float* grad_x_d) {
  MlpLayout L;
  int rc = mlp_layout(c, L);
  if (rc) return rc;
  if (!workspace_d || !x_d || !params_flat_d || !grad_x_d) return fail(-1, "null argument");
  if (L.lds_bwd > 160 * 1024) return fail(-3, "mlp: network too wide / deep for the input-gradient kernel's LDS budget");
  if ((in_shift_d == nullptr) != (in_iscale_d == nullptr) || (out_scale_d == nullptr) != (out_shift_d == nullptr) ||
      (sq_scale_d == nullptr) != (sq_bias_d == nullptr))
    return fail(-1, "mlp: shift/scale pointers come in pairs");
  hipStream_t s = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace_d);
  PackArgs PK;
  PK.wflag = nullptr;
  PK.gen = 0;
  PK.n = 0;
  PK.status = nullptr;
  MlpBwdArgs Bw;
  memset(&Bw, 0, sizeof(Bw));
  MlpArgs& A = Bw.f;
  A.B = c->B; A.nl = L.nl; A.LD = L.LD; A.n_out = L.dim[L.nl] / 2;
  for (int i = 0; i <= L.nl; ++i) { A.dim[i] = L.dim[i]; A.nt[i] = L.nt[i]; }
  for (int l = 0; l < L.nl; ++l) {
    float* wf = reinterpret_cast<float*>(ws + L.wf[l]);
    float* wb = reinterpret_cast<float*>(ws + L.wb[l]);
    float* bs = reinterpret_cast<float*>(ws + L.bias[l]);
    PK.job[PK.n++] = PackJob{params_flat_d + L.w_off[l], wf, L.dim[l + 1], L.dim[l], 0, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + L.w_off[l], wb, L.dim[l + 1], L.dim[l], 1, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + L.b_off[l], bs, L.dim[l + 1], L.dim[l], 0, 1, 1, 0, 0};
    A.wf[l] = wf; Bw.wb[l] = wb; A.bias[l] = bs;
    A.mask[l] = (l < L.nl - 1 && mask_bits_d) ? mask_bits_d[l] : nullptr;
    A.keep[l] = l < L.nl - 1 ? c->net.keep[l] : 1.f;
  }
  A.x = x_d; A.z = z_d; A.in_shift = in_shift_d; A.in_iscale = in_iscale_d;
  A.out_scale = out_scale_d; A.out_shift = out_shift_d; A.sq_scale = sq_scale_d; A.sq_bias = sq_bias_d;
  A.mls = c->max_log_std;
  Bw.g_sample = g_sample_d; Bw.g_mean = g_mean_d; Bw.g_log_std = g_log_std_d; Bw.grad_x = grad_x_d;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_mlp_bwdx_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bwd));
  hipLaunchKernelGGL(pm_pack_all, dim3(32, PK.n), dim3(256), 0, s, PK);
  hipLaunchKernelGGL(pm_mlp_bwdx_kernel, dim3((c->B + 15) / 16), dim3(PM_NT), L.lds_bwd, s, Bw);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// BNN training step
// ---------------------------------------------------------------------------
struct pmbrl_bnn_plan {
  pmbrl_bnn_config cfg;
  int device, nl, LD, nwg, sum_h, n_params;
  int dim[PM_MAXL + 1], nt[PM_MAXL + 1];
  int w_off[PM_MAXL], b_off[PM_MAXL], lp_poff[PM_MAXL], lp_off[PM_MAXL], has_drop[PM_MAXL];
  size_t off_wf[PM_MAXL], off_wb[PM_MAXL], off_bias[PM_MAXL], off_actT[PM_MAXL], off_gT[PM_MAXL];
  size_t off_part_lp, off_part_loss, off_part, off_reg_part, ws_bytes, lds;
  int part_stride, n_dw_blocks, dw_wave_first[PM_DW_NW + 1];
  DwBlock* dw_blocks_d;
};

extern "C" int pmbrl_bnn_plan_create(const pmbrl_bnn_config* cfg, int device, pmbrl_bnn_plan** out) {
  if (!cfg || !out) return fail(-1, "null argument");
  const pmbrl_mlp& m = cfg->net;
  if (m.n_layers < 2 || m.n_layers > PM_MAXL) return fail(-2, "bnn: 2..8 layers");
  if (cfg->M < 1 || cfg->N < 1) return fail(-2, "bnn: M, N must be >= 1");
  if (cfg->loss_kind == 2) {
    const int n = cfg->n_components;
    if (n < 2 || n > PMBRL_MAX_COMP) return fail(-2, "bnn: mixture loss needs 2..PMBRL_MAX_COMP components");
    if ((m.dims[m.n_layers] - 1) % n || ((m.dims[m.n_layers] - 1) / n - 1) % 2 || (m.dims[m.n_layers] - 1) / n < 3)
      return fail(-2, "bnn: mixture head must be (2 n_out + 1) n_components + 1 wide");
  } else if (m.dims[m.n_layers] % 2) return fail(-2, "bnn: the output layer must be 2 x n_out wide");
  pmbrl_bnn_plan* p = new pmbrl_bnn_plan();
  p->cfg = *cfg;
  p->device = device;
  p->nl = m.n_layers;
  int maxnt = 1, po = 0;
  p->sum_h = 0;
  for (int i = 0; i <= p->nl; ++i) {
    if (m.dims[i] < 1 || m.dims[i] > 16 * 64) { delete p; return fail(-2, "bnn: layer width out of range"); }
    p->dim[i] = m.dims[i];
    p->nt[i] = (m.dims[i] + 15) / 16;
    maxnt = std::max(maxnt, p->nt[i]);
  }
  for (int l = 0; l < p->nl; ++l) {
    p->w_off[l] = po; po += p->dim[l + 1] * p->dim[l];
    p->b_off[l] = po; po += p->dim[l + 1];
    p->has_drop[l] = (l < p->nl - 1 && cfg->temperature[l] > 0.f) ? 1 : 0;
    p->lp_poff[l] = po;
    p->lp_off[l] = p->sum_h;
    if (p->has_drop[l]) { po += p->dim[l + 1]; p->sum_h += p->dim[l + 1]; }
  }
  p->n_params = po;
  p->LD = maxnt * 16 + 8;
  p->nwg = (cfg->M + 15) / 16;
  p->lds = pm_bnn_lds_floats(p->nl, p->LD) * sizeof(float);
  if (p->lds > 160 * 1024) { delete p; return fail(-3, "bnn: network too wide / deep for the fused training step's LDS budget"); }
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  for (int l = 0; l < p->nl; ++l) {
    p->off_wf[l] = take((size_t)p->nt[l + 1] * p->nt[l] * 256 * sizeof(float));
    p->off_wb[l] = take((size_t)p->nt[l + 1] * p->nt[l] * 256 * sizeof(float));
    p->off_bias[l] = take((size_t)p->nt[l + 1] * 16 * sizeof(float));
    p->off_actT[l] = take((size_t)p->nwg * p->nt[l] * 16 * 16 * sizeof(float));
    p->off_gT[l] = take((size_t)p->nwg * p->nt[l + 1] * 16 * 16 * sizeof(float));
  }
  p->part_stride = (p->n_params + 3) / 4 * 4;
  p->off_part_lp = take((size_t)p->nwg * std::max(p->sum_h, 1) * sizeof(float));
  p->off_part_loss = take((size_t)p->nwg * sizeof(float));
  p->off_part = take((size_t)p->nwg * p->part_stride * sizeof(float));
  p->off_reg_part = take(1024 * sizeof(float));
  p->ws_bytes = off;
  HIPCHK(hipSetDevice(device));
  std::vector<DwBlock> blocks;
  p->n_dw_blocks = build_dw_blocks(p->nl, p->nt, blocks, p->dw_wave_first);
  HIPCHK(hipMalloc(&p->dw_blocks_d, blocks.size() * sizeof(DwBlock)));
  HIPCHK(hipMemcpy(p->dw_blocks_d, blocks.data(), blocks.size() * sizeof(DwBlock), hipMemcpyHostToDevice));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_bnn_fwd_bwd),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds));
  *out = p;
  return 0;
}

extern "C" void pmbrl_bnn_plan_destroy(pmbrl_bnn_plan* p) {
  if (!p) return;
  if (p->dw_blocks_d) (void)hipFree(p->dw_blocks_d);
  delete p;
}
extern "C" size_t pmbrl_bnn_plan_workspace_bytes(const pmbrl_bnn_plan* p) { return p ? p->ws_bytes : 0; }
extern "C" int64_t pmbrl_bnn_plan_n_params(const pmbrl_bnn_plan* p) { return p ? p->n_params : 0; }

extern "C" int pmbrl_bnn_loss_grad_ex(pmbrl_bnn_plan* p, void* stream, void* workspace_d, const float* Xn_d,
                                      const float* Yn_d, const int32_t* idx_d, const float* params_flat_d,
                                      const float* u_d, const float* bvar_d, float* grad_flat_d,
                                      float* loss_out_d, const float* row_weight_d, float* row_logprob_d,
                                      int32_t terms);
extern "C" int pmbrl_bnn_loss_grad(pmbrl_bnn_plan* p, void* stream, void* workspace_d, const float* Xn_d,
                                   const float* Yn_d, const int32_t* idx_d, const float* params_flat_d,
                                   const float* u_d, const float* bvar_d, float* grad_flat_d,
                                   float* loss_out_d) {
  return pmbrl_bnn_loss_grad_ex(p, stream, workspace_d, Xn_d, Yn_d, idx_d, params_flat_d, u_d, bvar_d,
                                grad_flat_d, loss_out_d, nullptr, nullptr, 3);
}

extern "C" int pmbrl_bnn_loss_grad_ex(pmbrl_bnn_plan* p, void* stream, void* workspace_d, const float* Xn_d,
                                      const float* Yn_d, const int32_t* idx_d, const float* params_flat_d,
                                      const float* u_d, const float* bvar_d, float* grad_flat_d,
                                      float* loss_out_d, const float* row_weight_d, float* row_logprob_d,
                                      int32_t terms) {
  if (!p || !workspace_d || !Xn_d || !Yn_d || !idx_d || !params_flat_d || !grad_flat_d || !loss_out_d)
    return fail(-1, "null argument");
  if (!(terms & 3)) return fail(-1, "bnn: terms must select the likelihood (1), the regulariser (2) or both");
  const bool lik = (terms & 1) != 0, regt = (terms & 2) != 0;
  if (p->sum_h > 0 && lik && (!u_d || !bvar_d)) return fail(-1, "bnn: dropout layers need u and bvar");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(p->device));
  char* ws = static_cast<char*>(workspace_d);
  PackArgs PK;
  PK.wflag = nullptr;
  PK.gen = 0;
  PK.n = 0;
  PK.status = nullptr;
  BnnArgs A;
  memset(&A, 0, sizeof(A));
  A.M = p->cfg.M; A.nl = p->nl; A.LD = p->LD; A.n_out = p->dim[p->nl] / 2; A.nwg = p->nwg;
  A.gmm_n = 0;
  if (p->cfg.loss_kind == 2) {
    A.gmm_n = p->cfg.n_components;
    A.n_out = ((p->dim[p->nl] - 1) / A.gmm_n - 1) / 2;
  }
  A.n_in = p->dim[0]; A.sum_h = p->sum_h;
  for (int i = 0; i <= p->nl; ++i) { A.dim[i] = p->dim[i]; A.nt[i] = p->nt[i]; }
  for (int l = 0; l < p->nl; ++l) {
    float* wf = reinterpret_cast<float*>(ws + p->off_wf[l]);
    float* wb = reinterpret_cast<float*>(ws + p->off_wb[l]);
    float* bs = reinterpret_cast<float*>(ws + p->off_bias[l]);
    PK.job[PK.n++] = PackJob{params_flat_d + p->w_off[l], wf, p->dim[l + 1], p->dim[l], 0, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + p->w_off[l], wb, p->dim[l + 1], p->dim[l], 1, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + p->b_off[l], bs, p->dim[l + 1], p->dim[l], 0, 1, 1, 0, 0};
    A.wf[l] = wf; A.wb[l] = wb; A.bias[l] = bs;
    A.actT[l] = reinterpret_cast<float*>(ws + p->off_actT[l]);
    A.gT[l] = reinterpret_cast<float*>(ws + p->off_gT[l]);
    A.lp_off[l] = p->lp_off[l];
    if (p->has_drop[l]) {
      A.logit_p[l] = params_flat_d + p->lp_poff[l];
      A.u[l] = u_d + (size_t)p->cfg.M * p->lp_off[l];
      A.bvar[l] = bvar_d + (size_t)p->cfg.M * p->lp_off[l];
      A.inv_temp[l] = 1.f / p->cfg.temperature[l];
    }
  }
  A.X = Xn_d; A.Y = Yn_d; A.idx = idx_d;
  A.mls = p->cfg.max_log_std;
  A.mse = p->cfg.loss_kind == 1;
  A.row_w = row_weight_d; A.row_lp = row_logprob_d;
  A.inv_M = 1.f / (float)p->cfg.M;
  A.part_lp = reinterpret_cast<float*>(ws + p->off_part_lp);
  A.part_loss = reinterpret_cast<float*>(ws + p->off_part_loss);
  if (!lik) {
    HIPCHK(hipMemsetAsync(grad_flat_d, 0, (size_t)p->n_params * sizeof(float), s));
  } else {
  hipLaunchKernelGGL(pm_pack_all, dim3(32, PK.n), dim3(256), 0, s, PK);
  hipLaunchKernelGGL(pm_bnn_fwd_bwd, dim3(p->nwg), dim3(PM_NT), p->lds, s, A);
  // dW / db of every layer: the policy-gradient GEMM over the same stash layout (one 16-row chunk per split)
  DwArgs W;
  memset(&W, 0, sizeof(W));
  W.nl = p->nl; W.nsplit = p->nwg; W.n_chunks = p->nwg; W.chunks_per_split = 1;
  W.RT = 1; W.Rw = 16; W.n_params = p->n_params; W.part_stride = p->part_stride;
  for (int i = 0; i <= p->nl; ++i) { W.dim[i] = p->dim[i]; W.nt[i] = p->nt[i]; }
  for (int l = 0; l < p->nl; ++l) {
    W.w_off[l] = p->w_off[l]; W.b_off[l] = p->b_off[l];
    W.actT[l] = A.actT[l]; W.gT[l] = A.gT[l];
  }
  W.blocks = p->dw_blocks_d;
  for (int w = 0; w <= PM_DW_NW; ++w) W.wave_first[w] = p->dw_wave_first[w];
  W.part = reinterpret_cast<float*>(ws + p->off_part);
  hipLaunchKernelGGL(pm_dw_kernel, dim3(p->nwg), dim3(PM_DW_NT), 0, s, W);
  hipLaunchKernelGGL(pm_dw_reduce, dim3((p->n_params + 255) / 256), dim3(512), 0, s, W.part, p->nwg, p->n_params,
                     p->part_stride, grad_flat_d, (const int*)nullptr, 0, 1);
  }
  BnnFinishArgs Fa;
  memset(&Fa, 0, sizeof(Fa));
  Fa.nl = p->nl; Fa.nwg = p->nwg; Fa.sum_h = p->sum_h; Fa.N = p->cfg.N;
  for (int i = 0; i <= p->nl; ++i) Fa.dim[i] = p->dim[i];
  for (int l = 0; l < p->nl; ++l) {
    Fa.w_off[l] = p->w_off[l]; Fa.b_off[l] = p->b_off[l]; Fa.lp_poff[l] = p->lp_poff[l]; Fa.lp_off[l] = p->lp_off[l];
    Fa.has_drop[l] = p->has_drop[l];
    Fa.reg_scale[l] = p->cfg.reg_scale[l]; Fa.drop_reg[l] = p->cfg.drop_reg[l];
  }
  Fa.reg_weight = regt ? p->cfg.reg_weight : 0.f; Fa.inv_M = A.inv_M;
  Fa.reg_only = lik ? 0 : 1;
  Fa.params = params_flat_d; Fa.grad = grad_flat_d; Fa.part_lp = A.part_lp; Fa.part_loss = A.part_loss;
  Fa.loss_out = loss_out_d; Fa.reg_part = reinterpret_cast<float*>(ws + p->off_reg_part);
  int nfb = 0;
  for (int l = 0; l < p->nl - 1; ++l)
    if (p->has_drop[l]) nfb += (p->dim[l + 1] + PM_BNN_CG - 1) / PM_BNN_CG;
  if (nfb < 1) nfb = 1;
  hipLaunchKernelGGL(pm_bnn_finish, dim3(nfb), dim3(256), 0, s, Fa);
  hipLaunchKernelGGL(pm_bnn_loss, dim3(1), dim3(64), 0, s, Fa, nfb);
  HIPCHK(hipGetLastError());
  return 0;
}

// Whole training iterations of utils/train_regressor.py:113-131 (likelihood + regulariser, Adam, no clipping) in TWO
// launches each: pm_bnn_fwd_bwd, pm_bnn_tail (pmbrl_bnn.h).  n_steps iterations are queued by ONE call: minibatch i takes
// rows idx_all_d[i * M ..]; the dropout noise comes from the in-kernel generator (seed, first_step + i) unless recorded
// draws are handed in (u_d / bvar_d: [n_steps][M * sum_h]).  The weights' fragments are packed once at the start of the
// call (the parameters may have changed outside) and kept current by the tail of every iteration.
extern "C" int pmbrl_bnn_train_steps(pmbrl_bnn_plan* p, void* stream, void* workspace_d, const float* Xn_d, const float* Yn_d,
                                     const int32_t* idx_all_d, int32_t n_steps, float* params_flat_d, float* exp_avg_d,
                                     float* exp_avg_sq_d, int64_t* step_d, double lr, double beta1, double beta2, double eps,
                                     uint64_t seed, uint64_t first_step, const float* u_d, const float* bvar_d,
                                     float* loss_out_d, float* loss_hist_d) {
  if (!p || !workspace_d || !Xn_d || !Yn_d || !idx_all_d || !params_flat_d || !exp_avg_d || !exp_avg_sq_d || !step_d ||
      !loss_out_d || n_steps < 1)
    return fail(-1, "bad argument");
  if (p->cfg.loss_kind != 0) return fail(-3, "pmbrl_bnn_train_steps: the diagonal-Gaussian likelihood only (use pmbrl_bnn_loss_grad_ex)");
  if ((u_d == nullptr) != (bvar_d == nullptr)) return fail(-1, "bnn: u and bvar come together");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(p->device));
  char* ws = static_cast<char*>(workspace_d);
  PackArgs PK;
  PK.wflag = nullptr; PK.gen = 0; PK.n = 0; PK.status = nullptr;
  BnnArgs A;
  memset(&A, 0, sizeof(A));
  A.M = p->cfg.M; A.nl = p->nl; A.LD = p->LD; A.n_out = p->dim[p->nl] / 2; A.nwg = p->nwg;
  A.n_in = p->dim[0]; A.sum_h = p->sum_h;
  BnnTailArgs T;
  memset(&T, 0, sizeof(T));
  T.nl = p->nl; T.n_chunks = p->nwg; T.sum_h = p->sum_h; T.N = p->cfg.N;
  int units = 0;
  for (int i = 0; i <= p->nl; ++i) { A.dim[i] = T.dim[i] = p->dim[i]; A.nt[i] = T.nt[i] = p->nt[i]; }
  for (int l = 0; l < p->nl; ++l) {
    float* wf = reinterpret_cast<float*>(ws + p->off_wf[l]);
    float* wb = reinterpret_cast<float*>(ws + p->off_wb[l]);
    float* bs = reinterpret_cast<float*>(ws + p->off_bias[l]);
    PK.job[PK.n++] = PackJob{params_flat_d + p->w_off[l], wf, p->dim[l + 1], p->dim[l], 0, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + p->w_off[l], wb, p->dim[l + 1], p->dim[l], 1, 1, 0, 0, 0};
    PK.job[PK.n++] = PackJob{params_flat_d + p->b_off[l], bs, p->dim[l + 1], p->dim[l], 0, 1, 1, 0, 0};
    A.wf[l] = wf; A.wb[l] = wb; A.bias[l] = bs;
    T.wf[l] = wf; T.wb[l] = wb; T.bias[l] = bs;
    A.actT[l] = reinterpret_cast<float*>(ws + p->off_actT[l]);
    A.gT[l] = reinterpret_cast<float*>(ws + p->off_gT[l]);
    T.actT[l] = A.actT[l]; T.gT[l] = A.gT[l];
    A.lp_off[l] = T.lp_off[l] = p->lp_off[l];
    T.w_off[l] = p->w_off[l]; T.b_off[l] = p->b_off[l]; T.lp_poff[l] = p->lp_poff[l];
    T.has_drop[l] = p->has_drop[l];
    T.reg_scale[l] = p->cfg.reg_scale[l]; T.drop_reg[l] = p->cfg.drop_reg[l];
    T.unit0[l] = units;
    units += p->nt[l];
    if (p->has_drop[l]) {
      A.logit_p[l] = params_flat_d + p->lp_poff[l];
      A.inv_temp[l] = 1.f / p->cfg.temperature[l];
    }
  }
  T.unit0[p->nl] = units;
  if (units > 512) return fail(-3, "pmbrl_bnn_train_steps: network too wide for the fused tail");
  A.X = Xn_d; A.Y = Yn_d;
  A.mls = p->cfg.max_log_std;
  A.inv_M = T.inv_M = 1.f / (float)p->cfg.M;
  A.part_lp = reinterpret_cast<float*>(ws + p->off_part_lp);
  A.part_loss = reinterpret_cast<float*>(ws + p->off_part_loss);
  T.part_lp = A.part_lp; T.part_loss = A.part_loss; T.n_part_loss = p->nwg;
  T.reg_weight = p->cfg.reg_weight;
  T.params = params_flat_d; T.m = exp_avg_d; T.v = exp_avg_sq_d;
  T.reg_part = reinterpret_cast<float*>(ws + p->off_reg_part);
  T.counter = reinterpret_cast<unsigned*>(ws + p->off_reg_part) + 1000;
  T.step = reinterpret_cast<long long*>(step_d);
  T.lr = (float)lr; T.b1 = (float)beta1; T.b2 = (float)beta2; T.eps = (float)eps;
  T.ln_b1 = log(beta1); T.ln_b2 = log(beta2);
  A.rng = u_d ? 0 : 1;
  A.prof = getenv("PMBRL_BNN_PROF") ? reinterpret_cast<long long*>(ws + p->off_reg_part) + 256 : nullptr;      // (debugging: stamps at floats 512.. of the scratch)
  A.rng_k0 = (unsigned)seed; A.rng_k1 = (unsigned)(seed >> 32);
  HIPCHK(hipMemsetAsync(T.counter, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(pm_pack_all, dim3(32, PK.n), dim3(256), 0, s, PK);
  const size_t noise = (size_t)p->cfg.M * p->sum_h;
  for (int i = 0; i < n_steps; ++i) {
    A.idx = idx_all_d + (size_t)i * p->cfg.M;
    A.rng_step = (unsigned)(first_step + (uint64_t)i);
    for (int l = 0; l < p->nl; ++l)
      if (p->has_drop[l] && u_d) {
        A.u[l] = u_d + (size_t)i * noise + (size_t)p->cfg.M * p->lp_off[l];
        A.bvar[l] = bvar_d + (size_t)i * noise + (size_t)p->cfg.M * p->lp_off[l];
      }
    T.loss_out = loss_hist_d ? loss_hist_d + (size_t)3 * i : loss_out_d;
    hipLaunchKernelGGL(pm_bnn_fwd_bwd, dim3(p->nwg), dim3(PM_NT), p->lds, s, A);
    hipLaunchKernelGGL(pm_bnn_tail, dim3(units), dim3(PM_BNT_NT), 0, s, T);
  }
  if (loss_hist_d)
    HIPCHK(hipMemcpyAsync(loss_out_d, loss_hist_d + (size_t)3 * (n_steps - 1), 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// gradient all-reduce: RCCL through dlopen (no link-time dependency)
// ---------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
struct RcclId { char internal[PMBRL_COMM_ID_BYTES]; };          // = ncclUniqueId (rccl.h:43)
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;      // ncclCommInitRank(comm*, nranks, id BY VALUE, rank)
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;                       // ncclCommCount(comm, int* count)
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy torch.distributed already loaded
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(-4, std::string("librccl not found: ") + dlerror());
  RcclApi a;
  a.lib = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy)
    return fail(-4, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
  g_rccl = a;
  return 0;
}
int rccl_fail(const char* what, int rc) {
  return fail(-200 - rc, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace
struct pmbrl_comm {
  void* comm;
  int rank, nranks, device;
};
extern "C" int pmbrl_comm_unique_id(void* id_out) {
  if (!id_out) return fail(-1, "null argument");
  if (int rc = rccl_load()) return rc;
  RcclId id;
  if (int rc = g_rccl.GetUniqueId(&id)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}
extern "C" int pmbrl_comm_init(const void* id_in, int32_t rank, int32_t nranks, int32_t device, pmbrl_comm** out) {
  if (!id_in || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "bad argument");
  if (int rc = rccl_load()) return rc;
  HIPCHK(hipSetDevice(device));
  RcclId id;
  memcpy(&id, id_in, sizeof(id));
  void* c = nullptr;
  if (int rc = g_rccl.CommInitRank(&c, nranks, id, rank)) return rccl_fail("ncclCommInitRank", rc);
  *out = new pmbrl_comm{c, rank, nranks, device};
  return 0;
}
extern "C" int pmbrl_allreduce_sum(pmbrl_comm* comm, void* stream, float* buf_d, int64_t n) {
  if (!comm || !buf_d || n < 0) return fail(-1, "bad argument");
  // ncclFloat32 = 7, ncclSum = 0 (rccl.h: ncclDataType_t, ncclRedOp_t)
  if (int rc = g_rccl.AllReduce(buf_d, buf_d, (size_t)n, 7, 0, comm->comm, (hipStream_t)stream))
    return rccl_fail("ncclAllReduce", rc);
  return 0;
}
extern "C" int pmbrl_comm_count(pmbrl_comm* comm, int32_t* nranks_out) {
  if (!comm || !nranks_out) return fail(-1, "null argument");
  if (!g_rccl.CommCount) return fail(-4, "librccl lacks ncclCommCount");
  int n = 0;
  if (int rc = g_rccl.CommCount(comm->comm, &n)) return rccl_fail("ncclCommCount", rc);
  *nranks_out = n;
  return 0;
}
extern "C" void pmbrl_comm_destroy(pmbrl_comm* comm) {
  if (!comm) return;
  if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
  delete comm;
}

static int pm_coll_rccl(void* ctx, void* stream, double* buf_d, int64_t n) {
  pmbrl_comm* comm = static_cast<pmbrl_comm*>(ctx);
  return g_rccl.AllReduce(buf_d, buf_d, (size_t)n, 8 /* ncclDouble */, 0 /* ncclSum */, comm->comm, (hipStream_t)stream);
}
extern "C" int pmbrl_plan_set_comm(pmbrl_plan* plan, pmbrl_comm* comm) {
  if (!plan || !comm) return fail(-1, "null argument");
  plan->coll = pm_coll_rccl;
  plan->coll_ctx = comm;
  return 0;
}
extern "C" int pmbrl_plan_set_collective(pmbrl_plan* plan, pmbrl_collective_fn fn, void* ctx) {
  if (!plan) return fail(-1, "null argument");
  plan->coll = fn;
  plan->coll_ctx = ctx;
  return 0;
}

extern "C" int pmbrl_weighted_sum(void* stream, const float* a_d, const float* w_d, int64_t n,
                                  float* out_d) {
  if (!a_d || !w_d || !out_d || n < 0) return fail(-1, "bad argument");
  const int nb = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (n + 2047) / 2048));
  hipLaunchKernelGGL(pm_weighted_sum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a_d, w_d,
                     (long long)n, out_d, (const int*)nullptr, 0ll);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_weighted_sum_steps(void* stream, const float* a_d, const float* w_d, int64_t n_per_step,
                                        int32_t n_steps, const int32_t* status_d, float* out_d) {
  if (!a_d || !w_d || !out_d || n_per_step < 0 || n_steps < 0) return fail(-1, "bad argument");
  const long long n = (long long)n_per_step * n_steps;
  const int nb = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (n + 2047) / 2048));
  hipLaunchKernelGGL(pm_weighted_sum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a_d, w_d, n, out_d,
                     (const int*)status_d, (long long)n_per_step);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_clip_adam(void* stream, float* params_d, float* grads_d, float* exp_avg_d,
                               float* exp_avg_sq_d, int64_t n, int64_t step, double lr, double beta1,
                               double beta2, double eps, double max_norm, float* norm_out_d) {
  if (!params_d || !grads_d || !exp_avg_d || !exp_avg_sq_d || n < 1 || step < 1)
    return fail(-1, "bad argument");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const int nb = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (n + 1023) / 1024));
  const int np = pm_norm_blocks(n);
  // without clipping and without a norm to report there is nothing for the norm kernel to do
  const bool need_norm = max_norm > 0.0 || norm_out_d != nullptr;
  if (need_norm)
    hipLaunchKernelGGL(pm_gradnorm_kernel, dim3(np), dim3(64), 0, (hipStream_t)stream, grads_d,
                       (long long)n, (const int*)nullptr, 0, (long long*)nullptr, 0);
  hipLaunchKernelGGL(pm_clip_adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params_d,
                     grads_d, exp_avg_d, exp_avg_sq_d, (long long)n, need_norm ? np : 0, (float)lr, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                     (float)bc1, (float)sqrt(bc2), (float)max_norm, norm_out_d, (const long long*)nullptr, 0, 0.0, 0.0, lr,
                     (float*)nullptr, 0);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int pmbrl_clip_adam_guarded(void* stream, float* params_d, float* grads_d, float* exp_avg_d,
                                       float* exp_avg_sq_d, int64_t n, int64_t* step_d, double lr,
                                       double beta1, double beta2, double eps, double max_norm,
                                       float* norm_out_d, const int32_t* status_d, int32_t expect) {
  if (!params_d || !grads_d || !exp_avg_d || !exp_avg_sq_d || !step_d || !status_d || n < 1)
    return fail(-1, "bad argument");
  return clip_adam_guarded_impl(stream, params_d, grads_d, exp_avg_d, exp_avg_sq_d, n, step_d, lr, beta1, beta2, eps, max_norm,
                                norm_out_d, status_d, expect, false);
}
// norm_done: the partial sums of squares and the go / no-go decision are already on the device (the fused iteration's
// gradient reduction formed them: pm_dw_reduce, norm_on)
static int clip_adam_guarded_impl(void* stream, float* params_d, float* grads_d, float* exp_avg_d, float* exp_avg_sq_d,
                                  int64_t n, int64_t* step_d, double lr, double beta1, double beta2, double eps,
                                  double max_norm, float* norm_out_d, const int32_t* status_d, int32_t expect, bool norm_done,
                                  float* loss_out_d, int n_loss_part) {
  const int nb = (int)std::max<long long>(1, std::min<long long>(PM_RED_MAXB, (n + 1023) / 1024));
  const int np = pm_norm_blocks(n);
  if (!norm_done)
    hipLaunchKernelGGL(pm_gradnorm_kernel, dim3(np), dim3(64), 0, (hipStream_t)stream, grads_d,
                       (long long)n, (const int*)status_d, (int)expect, reinterpret_cast<long long*>(step_d), 1);
  hipLaunchKernelGGL(pm_clip_adam_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params_d,
                     grads_d, exp_avg_d, exp_avg_sq_d, (long long)n, np, (float)lr, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, 1.f, 1.f,
                     (float)max_norm, norm_out_d, reinterpret_cast<const long long*>(step_d), 1, log(beta1), log(beta2), lr,
                     loss_out_d, n_loss_part);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// test hook: y = x W^T + b through gemm_tiles / gemm_narrow
// ---------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(PM_NT, 1) void pm_debug_linear_kernel(const float* x, const float* wf,
                                                                  const float* bias, int R, int K,
                                                                  int O, int LD, int narrow,
                                                                  float* y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int RR = 16 * RT;
  float* X = smem;
  float* Y = X + RR * LD;
  float* part = pm_part_alias_ok(RR, LD, RT) ? nullptr : Y + RR * LD;   // (same rule as the rollout kernels)
  const int n_kb = (K + 15) / 16, n_ot = (O + 15) / 16;
  for (int i = tid; i < RR * LD; i += PM_NT) {
    const int r = i / LD, k = i - r * LD;
    X[i] = (r < R && k < K) ? x[(size_t)r * K + k] : 0.f;
  }
  __syncthreads();
  if (narrow) {
    gemm_narrow<RT>(wf, n_ot, n_kb, bias, X, Y, LD, part, wid, lane, tid);
  } else {
    EpiPlain e{bias, Y, LD, lane};
    gemm_tiles<RT>(wf, n_ot, n_kb, X, LD, wid, lane, e);
    __syncthreads();
  }
  for (int i = tid; i < R * O; i += PM_NT) {
    const int r = i / O, o = i - r * O;
    y[i] = Y[r * LD + o];
  }
}

extern "C" int pmbrl_debug_linear(void* stream, const float* x_d, const float* W_d, const float* b_d,
                                  int32_t R, int32_t K, int32_t O, int32_t transpose_w, float* y_d,
                                  float* scratch_d) {
  // transpose_w: treat W as [K_out... ] i.e. compute y = x W (W is [K][O] row-major = stored [in][out])
  if (R < 1 || R > 64 || K < 1 || O < 1) return fail(-1, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int n_kb = (K + 15) / 16, n_ot = (O + 15) / 16;
  const int LD = std::max(n_kb, n_ot) * 16 + 8;
  float* wf = scratch_d;
  float* bias = scratch_d + (size_t)n_ot * n_kb * 256;
  const size_t tot = (size_t)n_ot * n_kb * 256;
  const int grid = (int)std::min<size_t>((tot + 255) / 256, 1024);
  if (!transpose_w)
    hipLaunchKernelGGL(pm_pack_frag, dim3(grid), dim3(256), 0, s, W_d, O, K, 0, 1, wf);
  else  // W_d is [K][O]: out feature index is the second one
    hipLaunchKernelGGL(pm_pack_frag, dim3(grid), dim3(256), 0, s, W_d, K, O, 1, 1, wf);
  hipLaunchKernelGGL(pm_pack_bias, dim3((n_ot * 16 + 255) / 256), dim3(256), 0, s, b_d, O, n_ot * 16,
                     bias);
  auto lds_for = [&](int RT) {
    return ((size_t)2 * 16 * RT * LD +
            (pm_part_alias_ok(16 * RT, LD, RT) ? 0 : (size_t)PM_NW * PM_KS_NT * RT * 256)) * sizeof(float);
  };
  const int narrow = n_ot <= PM_KS_NT ? 1 : 0;
  // rows in slabs of the largest tile height whose LDS fits
  int RT = R <= 16 ? 1 : (R <= 32 ? 2 : 4);
  while (RT > 1 && lds_for(RT) > (size_t)160 * 1024) RT /= 2;
  const size_t lds = lds_for(RT);
#define DBG_LAUNCH(RTV, X0, Y0, RR)                                                             \
  do {                                                                                          \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_debug_linear_kernel<RTV>),     \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
    hipLaunchKernelGGL(pm_debug_linear_kernel<RTV>, dim3(1), dim3(PM_NT), lds, s, X0, wf, bias, RR, \
                       K, O, LD, narrow, Y0);                                                   \
  } while (0)
  for (int r0 = 0; r0 < R; r0 += 16 * RT) {
    const int rr = std::min(R - r0, 16 * RT);
    const float* x0 = x_d + (size_t)r0 * K;
    float* y0 = y_d + (size_t)r0 * O;
    if (RT == 1) DBG_LAUNCH(1, x0, y0, rr);
    else if (RT == 2) DBG_LAUNCH(2, x0, y0, rr);
    else DBG_LAUNCH(4, x0, y0, rr);
  }
#undef DBG_LAUNCH
  HIPCHK(hipGetLastError());
  return 0;
}
