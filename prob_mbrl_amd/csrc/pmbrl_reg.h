// Register-resident sweep kernels for the cart-pole class of shapes (round 4): three-layer policy and dynamics
// networks with ONE hidden width of 177..208 units (13 output tiles), at most 8 network inputs (D + U <= 8), no moment
// matching inside the sweep, 16 rows per workgroup.  Same mathematics, same stashes and the same C ABI as the
// latency-optimised family (pmbrl_fast.h) -- what changes is where a step's time goes:
//
//   * FOUR waves per workgroup, one per SIMD, 512 registers each.  The hidden->hidden weights of BOTH networks
//     (2 x 200 x 200, two 16-bit pieces each: 336 registers per wave for the 12 tiles a wave quartet shares out evenly)
//     stay in registers for the whole launch -- 256 of them in the accumulator file, which the matrix core reads as
//     an A operand directly (inline-asm MFMAs: hipcc would copy them back through VGPRs).  Nothing is streamed from
//     L2 inside the horizon loop; the 13th tile, the first layers and the heads live in LDS (copied once per launch).
//   * a step has FOUR workgroup barriers (pmbrl_fast.h: nine): per network, one after the first layer's epilogue
//     (hidden activations -> LDS as MFMA B fragments, read back by every wave) and one after the head, which is
//     K-split over the waves STRAIGHT FROM REGISTERS -- the K permutation below makes the epilogue's output registers
//     the next product's B operand as they are, so the second hidden layer never goes through LDS at all.
//   * the elementwise phases (squash, sampling, normalisation) are lane-local and computed redundantly by all four
//     waves: input i of a network lives in lane group i / 2 of every wave, the head's output rows are permuted (at pack
//     time, for free) so that the mean / log-std of what becomes input i come out of the MFMA in that very lane group.
//     No LDS round trip, no barrier, no cross-lane traffic between a head and the next first layer.
//   * first layers (K <= 8 inputs) are ONE MFMA per tile: the three piece products hi.hi, hi.lo, lo.hi and the bias
//     sit in different k-slots of the same K = 32 block.  Hidden-layer biases ride in the free half of the last
//     K = 32 block (13 tiles = 6.5 blocks) against a constant 1 in the activation buffer: no bias add anywhere.
//
// K permutation of a hidden width (both operands of a product use it; a sum over k does not care): K32 block kb,
// lane group g = lane >> 4, slot j = 0..7  <->  feature 16 (2 kb + (j >> 2)) + 4 g + (j & 3), i.e. the four
// accumulator registers of output tile 2 kb (j < 4) and of tile 2 kb + 1 (j >= 4) in lane group g.
//
// Arithmetic: pmbrl_split.h's -- forward two fp16 pieces (low weight piece scaled by 2^11, own accumulator chain),
// adjoint two bf16 pieces.  Reference: utils/rollout.py:95-152, models/core.py:221-303, models/modules.py:46-160.
#pragma once
#include <cstddef>
#include <type_traits>
#include <utility>
#include "pmbrl_dev.h"
#include "pmbrl_split.h"
#include "pmbrl_reg_mm.h"

#define PR_NW 4                 // waves per workgroup: one per SIMD
#define PR_NTHR (PR_NW * 64)
#define PR_NT 13                // output tiles of a hidden layer (hidden width 177 .. 208)
#define PR_KB 7                 // K32 blocks of a hidden width
#define PR_SLOTS 3              // register-resident tiles per wave and hidden->hidden layer: tile 4 s + w
#define PR_XT 12                // the tile that lives in LDS (run by wave pr_xwave(net))
#define PR_FRAG 256             // floats of one fragment (64 lanes x 16 bytes)
#define PM_GLOBAL_ __attribute__((address_space(1)))
typedef unsigned pr_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// floats of the packed sections of ONE network and direction (pm_reg_pack_kernel writes them, the sweeps read them)
#define PR_RES_FLOATS (PR_NW * PR_SLOTS * PR_KB * 2 * PR_FRAG)    // [wave][slot][kb][piece][lane][4]
#define PR_XT_FLOATS (PR_KB * 2 * PR_FRAG)                        // [kb][piece][lane][4]
#define PR_L0P_FLOATS (PR_NT * 2 * PR_FRAG)                       // packed: [tile][block (forward: 1 used)][lane][4]
#define PR_L0_FLOATS (PR_NT * PR_FRAG)                            // forward, in LDS: [tile][lane][4]
#define PR_HEAD_FLOATS (PR_NW * 2 * 2 * PR_FRAG)                  // [wave][block][piece][lane][4]
#define PR_NET_FLOATS (PR_RES_FLOATS + PR_XT_FLOATS + PR_L0P_FLOATS + PR_HEAD_FLOATS)
#define PR_OFF_RES 0
#define PR_OFF_XT (PR_RES_FLOATS)
#define PR_OFF_L0 (PR_RES_FLOATS + PR_XT_FLOATS)
#define PR_OFF_HEAD (PR_RES_FLOATS + PR_XT_FLOATS + PR_L0P_FLOATS)
// the whole packed buffer: [direction (0 forward, 1 adjoint)][net (0 policy, 1 dynamics)][PR_NET_FLOATS]
#define PR_PACK_FLOATS (4 * PR_NET_FLOATS)

// the wave that runs the LDS-resident tile of a network (two different ones: they balance over a step)
__host__ __device__ constexpr int pr_xwave(int net) { return net; }

// LDS map (floats)
#define PR_LDS_ACT 0                                   // [kb][piece][lane][4]: hidden activations as B fragments
#define PR_LDS_PART (PR_LDS_ACT + PR_KB * 2 * PR_FRAG) // [wave][lane][4]: partial head / tail tiles
#define PR_LDS_L0(net) (PR_LDS_PART + PR_NW * PR_FRAG + (net) * (PR_L0_FLOATS + PR_XT_FLOATS + PR_HEAD_FLOATS))
#define PR_LDS_XT(net) (PR_LDS_L0(net) + PR_L0_FLOATS)
#define PR_LDS_HEAD(net) (PR_LDS_XT(net) + PR_XT_FLOATS)
// dropout multipliers {0, 1 / keep} of this lane's four values of a tile, in accumulator-register order: resident
// tiles [wave][net][layer][slot][lane][4], the LDS-resident tile [net][layer][lane][4] (forward: the mask; adjoint: the
// same table is rebuilt per step from the stashed activity bits)
#define PR_LDS_MF (PR_LDS_L0(2))
#define PR_LDS_MFX (PR_LDS_MF + PR_NW * 2 * 2 * PR_SLOTS * PR_FRAG)
#define PR_LDS_FLAG (PR_LDS_MFX + 2 * 2 * PR_FRAG)
#define PR_LDS_XB (PR_LDS_FLAG + 16)             // [16 rows][8]: the moment-matched rows, wave 0 -> every wave
#define PR_LDS_ZH (PR_LDS_XB + 128)              // [2][16 rows][<= 6] doubles: standardised noise rows, wave 1 -> wave 0
#define PR_LDS_REC (PR_LDS_ZH + 2 * PR_MM_ZH_DOUBLES(6))      // [3][64] doubles: the factor record's inputs, wave 0 -> wave 2
#define PR_LDS_FLOATS (PR_LDS_REC + 2 * PR_MM_REC_DOUBLES)
static_assert(PR_LDS_FLOATS * 4 <= 160 * 1024, "forward sweep: LDS");

struct RegNet {
  int w_off[3], b_off[3];       // offsets (floats) of W_l / b_l in the flat parameter vector
  int n_in, n_out;              // network inputs (D or D + U), head rows (2 U or 2 D)
  const uint16_t* mask[2];      // dropout bit rows [B][PR_NT] of the two hidden layers
  unsigned abits[2];            // stash [H][B][PR_NT][4] (byte offset in the workspace): mask & (pre-activation > 0), one
                                // nibble-byte per lane group
  float inv_keep[2];
};

struct RegArgs {
  int B, H, D, U, nwg, hid;
  float mls_pol, mls_dyn;
  RegNet pol, dyn;
  const float* packed;          // PR_PACK_FLOATS (this launch's weights, pm_reg_pack_kernel)
  const float *pol_params, *dyn_params;
  const float *x0, *mx, *iSx, *my, *Sy, *pscale, *pbias, *zpol, *zdyn;
  float *states, *actions;
  // everything the sweeps stash lives in ONE workspace (< 4 GB for the shapes of this family): a base and 32-bit byte
  // offsets instead of a dozen 64-bit pointers held in scalar registers across the horizon loop
  char* ws;
  unsigned Tp, Td;              // [H][B][U], [H][B][D]
  unsigned actT[3];             // policy layer inputs, feature-major blocks [H][nwg][nt * 16][16]
  unsigned gT[3];               // policy pre-activation gradients, same layout
  unsigned Jx, Ja;              // reward Jacobians [H][B][D], [H][B][U]
  int* status;
  const int* wflag;
  int wgen;
  // adjoint
  const float* grad_rewards;
  float* grad_x0;
  const int* nvalid;
  // adjoint sweep over the steps [t0, t1) only (the dW GEMM of a finished range runs behind the next one, pmbrl.hip):
  // dL/dx_{t1} comes from gx_in ([B][D]; nullptr: zero, the sweep starts at the horizon), dL/dx_{t0} goes to gx_out
  // (nullptr: not wanted) and, at t0 = 0, to grad_x0
  int t0, t1;
  const float* gx_in;
  float* gx_out;
  long long* prof;
  RegMM mm;                     // moment matching of states inside the sweep (pmbrl_reg_mm.h)
  int wg0;                      // first workgroup of this launch (groups launched in batches: workgroup = blockIdx.x + wg0)
};
// the moment-matching arguments where the launch put them, in the kernel-argument segment (NOT &A.mm: the address of a
// by-value kernel argument makes the compiler copy the whole struct to scratch)
// (the pointer KEEPS the constant address space: cast to a generic pointer -- as it was until round 6 -- the per-step copy of
//  the struct compiled to FLAT vector loads with a uniform address, waited for with vmcnt(0) lgkmcnt(0): a memory round trip
//  and a drain of the wave's stash stores at the head of every chain.  Scalar loads from the constant cache now.)
typedef __attribute__((address_space(4))) const RegMM* pr_mm_kptr;
__device__ __forceinline__ pr_mm_kptr pr_mm_args() {
  typedef __attribute__((address_space(4))) const char* kcp;
  return (pr_mm_kptr)((kcp)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RegArgs, mm));
}
// (the host pass type-checks kernel bodies too, and knows no copy out of the constant address space)
__device__ __forceinline__ RegMM pr_mm_load(pr_mm_kptr p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  (void)p;
  return RegMM{};
#endif
}
// Logical workgroup of this hardware workgroup (-1: none -- padding).  Workgroups go round-robin to the 8 XCDs, each with an
// L2 of its own: dealt in blocks of 8 groups -- hardware workgroup 8 j + i of a block = part j of the block's group i --
// the parts of a group land on ONE XCD and their statistics exchange meets in that L2 instead of crossing the fabric.
__device__ __forceinline__ int pr_wg(const RegArgs& A) {
  const int hw = (int)blockIdx.x + A.wg0;
  if (!A.mm.on || !A.mm.xcd) return hw;
  if (A.mm.xcd == 2) {
    // ONE group over the batch, two-level exchange: XCD x holds the parts [x * 2 fan, (x + 1) * 2 fan) -- two collectors and
    // their members: the first hop stays inside the XCD's L2, the second (<= 16 collectors' slots) crosses the fabric
    const int per = 2 * A.mm.fan, pw = (hw & 7) * per + (hw >> 3);
    return pw < A.mm.parts ? pw : -1;
  }
  const int P = A.mm.parts, blk = hw / (8 * P), r = hw - blk * (8 * P);
  const int gi = blk * 8 + (r & 7);
  return gi < A.mm.groups ? gi * P + (r >> 3) : -1;
}
// rows of workgroup wg: 16 consecutive rows, or -- moment matching -- part wg % parts of group wg / parts
__device__ __forceinline__ void pr_rows(const RegArgs& A, int wg, int& row0, int& nvalid) {
  if (A.mm.on) {
    const int gi = wg / A.mm.parts, me = wg - gi * A.mm.parts;
    row0 = gi * A.mm.M + me * A.mm.rpw;
    nvalid = min(A.mm.rpw, A.mm.M - me * A.mm.rpw);
  } else {
    row0 = wg * 16;
    nvalid = min(16, A.B - row0);
  }
}

template <int N, class F, int... I>
__device__ __forceinline__ void pr_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void pr_for(F&& f) {
  pr_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// feature of K slot (kb, g, j) of a hidden width
__host__ __device__ constexpr int pr_hid_feature(int kb, int g, int j) { return 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3); }

// head row that lands in accumulator register r of lane group g (row m = 4 g + r of the head tile): the pair
// (mean, log-std) of what becomes input i = 2 g + (r >> 1) of the NEXT first layer -- action i - D for the policy
// (inputs D .. D + U - 1 of the dynamics model), next-state dimension i for the dynamics model.  -1: nothing.
__host__ __device__ inline int pr_head_row(int net, int D, int U, int m) {
  const int i = 2 * (m >> 2) + ((m >> 1) & 1), kind = m & 1;
  if (net == 0) {
    const int j = i - D;
    return (j >= 0 && j < U) ? (kind ? U + j : j) : -1;
  }
  return i < D ? (kind ? D + i : i) : -1;
}

// ---------------------------------------------------------------------------
// weight packer: one thread per (fragment, lane) = 8 sixteen-bit values
// ---------------------------------------------------------------------------
struct RegPackArgs {
  const float* par[2];          // flat parameters of the policy / the dynamics model
  int w_off[2][3], b_off[2][3];
  int n_in[2], n_out[2];
  int D, U, hid;
  float* out;                   // PR_PACK_FLOATS
  int* wflag;
  int gen;
  int* status;                  // the forward sweep's status word: reset here (first launch of an iteration)
  int n_pack_blocks;            // workgroups of the pack proper; the ones behind them form the noise table (zt.n_blocks)
  ZtabArgs zt;
};

__device__ __forceinline__ unsigned short pr_f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ unsigned short pr_bf16_bits(float v) { return (unsigned short)(pm_pk_bf16(v, 0.f) & 0xffffu); }
// piece p of v: fp16 (low piece scaled by 2^11, pmbrl_split.h) or bf16
__device__ __forceinline__ unsigned short pr_piece(float v, int p, bool f16) {
  if (f16) {
    const _Float16 h = (_Float16)v;
    return p == 0 ? __builtin_bit_cast(unsigned short, h) : pr_f16_bits((v - (float)h) * PM_F16_LO_SCALE);
  }
  const unsigned a = pm_pk_bf16(v, 0.f) & 0xffffu;
  return p == 0 ? (unsigned short)a : pr_bf16_bits(v - pm_bf_lo(a));
}

// ZT: the launch has extra workgroups that form the noise table of split moment-matching groups (pmbrl_mm.h) -- an instance
// of its own, so that the shapes without moment matching launch the code they always did
template <bool ZT>
__global__ __launch_bounds__(256) void pm_reg_pack_kernel(const RegPackArgs P) {
  if constexpr (ZT) {
    if ((int)blockIdx.x >= P.n_pack_blocks) {
      pm_ztab_block(P.zt, (int)blockIdx.x - P.n_pack_blocks, (int)threadIdx.x);
      return;
    }
  }
  if (P.status && blockIdx.x == 0 && threadIdx.x == 0) *P.status = 0x7fffffff;
  const int n_frag_net = PR_NET_FLOATS / PR_FRAG;
  const int total = 4 * n_frag_net * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += P.n_pack_blocks * blockDim.x) {
    const int lane = idx & 63, fr_all = idx >> 6;
    const int dn = fr_all / n_frag_net, fr = fr_all - dn * n_frag_net;
    const int dir = dn >> 1, net = dn & 1;
    const bool f16 = dir == 0;
    const float* par = P.par[net];
    const float* W0 = par + P.w_off[net][0];
    const float* W1 = par + P.w_off[net][1];
    const float* W2 = par + P.w_off[net][2];
    const float* b0 = par + P.b_off[net][0];
    const float* b1 = par + P.b_off[net][1];
    const int hid = P.hid, n_in = P.n_in[net], n_out = P.n_out[net];
    const int m = lane & 15, g = lane >> 4;
    float v[8];
    int piece[8];       // which piece of v[e] is stored (0 / 1), per element
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = 0.f; piece[e] = 0; }
    const int n_res = PR_RES_FLOATS / PR_FRAG, n_xt = PR_XT_FLOATS / PR_FRAG, n_l0 = PR_L0P_FLOATS / PR_FRAG;
    if (fr < n_res + n_xt) {
      // hidden -> hidden layer: output tile ot, K32 block kb, piece p
      int ot, kb, p;
      if (fr < n_res) {
        const int w = fr / (PR_SLOTS * PR_KB * 2), r = fr - w * (PR_SLOTS * PR_KB * 2);
        const int s = r / (PR_KB * 2), r2 = r - s * (PR_KB * 2);
        kb = r2 >> 1; p = r2 & 1; ot = 4 * s + w;
      } else {
        const int r2 = fr - n_res;
        kb = r2 >> 1; p = r2 & 1; ot = PR_XT;
      }
      const int o = 16 * ot + m;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = pr_hid_feature(kb, g, e);
        piece[e] = p;
        if (o < hid && k < hid) v[e] = dir == 0 ? W1[(size_t)o * hid + k] : W1[(size_t)k * hid + o];
        // forward: the bias rides in slot 4 of the last block (tile 13 does not exist), lane group 0, against the
        // constant 1 the kernel keeps in the activation buffer
        if (dir == 0 && kb == PR_KB - 1 && e == 4 && g == 0 && o < hid) v[e] = b1[o];
      }
    } else if (fr < n_res + n_xt + n_l0) {
      const int r = fr - n_res - n_xt;
      const int ot = r >> 1, blk = r & 1;
      const int o = 16 * ot + m;
      if (dir == 0) {
        // first layer: slots 3 s + {0, 1, 2} = W.hi (x act.hi), W.hi (x act.lo), W.lo (x act.hi 2^-11) of input 2 g + s;
        // slots 6 / 7 of lane group 0: bias.hi (x 1), bias.lo (x 2^-11)
        if (blk == 0 && o < hid) {
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            const int i = 2 * g + e / 3;
            if (i < n_in) { v[e] = W0[(size_t)o * n_in + i]; piece[e] = (e % 3) == 2 ? 1 : 0; }
          }
          if (g == 0) { v[6] = b0[o]; piece[6] = 0; v[7] = b0[o]; piece[7] = 1; }
        }
      } else {
        // adjoint of the head: block 0 carries the means' rows, block 1 the log-stds'; slots 3 s + {0, 1, 2} = W.hi (x g.hi),
        // W.hi (x g.lo), W.lo (x g.hi) of the head pair of input slot (g, s)
        if (o < hid) {
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            const int row = pr_head_row(net, P.D, P.U, 4 * g + 2 * (e / 3) + blk);
            if (row >= 0 && row < n_out) { v[e] = W2[(size_t)row * hid + o]; piece[e] = (e % 3) == 2 ? 1 : 0; }
          }
        }
      }
    } else {
      // head (forward) / tail (adjoint), K-split over the waves: block 0 = the wave's tiles of slots 0 and 1,
      // block 1 = slot 2 and, on the wave that runs it, the LDS-resident tile
      const int r = fr - n_res - n_xt - n_l0;
      const int w = r >> 2, blk = (r >> 1) & 1, p = r & 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int ot = -1;
        if (blk == 0) ot = 4 * (e >> 2) + w;
        else if (e < 4) ot = 8 + w;
        else if (w == pr_xwave(net)) ot = PR_XT;
        const int k = ot >= 0 ? 16 * ot + 4 * g + (e & 3) : hid;
        piece[e] = p;
        if (k < hid) {
          if (dir == 0) {
            const int row = pr_head_row(net, P.D, P.U, m);
            if (row >= 0 && row < n_out) v[e] = W2[(size_t)row * hid + k];
          } else {
            // gradient with respect to network input i = 2 (m >> 2) + ((m >> 1) & 1), in accumulator register 2 s
            const int i = 2 * (m >> 2) + ((m >> 1) & 1);
            if ((m & 1) == 0 && i < n_in) v[e] = W0[(size_t)k * n_in + i];
          }
        }
      }
    }
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (f16 && piece[e] == 0 && P.wflag && !(fabsf(v[e]) <= 65504.f)) atomicMax(P.wflag, P.gen);
      h[e] = pr_piece(v[e], piece[e], f16);
    }
    uint4 o4;
    o4.x = h[0] | ((unsigned)h[1] << 16);
    o4.y = h[2] | ((unsigned)h[3] << 16);
    o4.z = h[4] | ((unsigned)h[5] << 16);
    o4.w = h[6] | ((unsigned)h[7] << 16);
    *reinterpret_cast<uint4*>(P.out + ((size_t)dn * PR_NET_FLOATS + (size_t)fr * PR_FRAG) + lane * 4) = o4;
  }
}

// ---------------------------------------------------------------------------
// MFMA with the A operand (weights) in the accumulator file or in a VGPR.  Inline asm: the compiler schedules the
// statement as one opaque instruction and pads nothing around it -- pr_mfma_fence() supplies the wait states
// between a chain's last MFMA and the first VALU read of its accumulator (8-pass XDL: 12 states), pr_mfma_open()
// those between a VALU write of an operand and the first MFMA of a group.
// ---------------------------------------------------------------------------
// NOP: two wait states in front of the instruction, INSIDE the statement.  A VGPR written by a VALU instruction may
// not be read as an MFMA operand in the next two issue slots (hipcc pads that for its own MFMAs, not for asm), and
// the compiler is free to schedule the VALU that builds an operand -- a cvt_pk of a fragment -- right in front of
// the statement that consumes it: measured as garbage head / tail products before this was here.  The big hidden
// layers need none (B operands come from ds_read, weights from the prologue).
template <bool F16, bool AG, bool NOP = false>
__device__ __forceinline__ void pr_mfma(f32x4& acc, const f32x4& w, const f32x4& b) {
  if constexpr (NOP) {
    if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
  } else if constexpr (F16) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
  } else {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
  }
}
// first product of a chain: C = 0 (inline constant), the accumulator is a pure output
template <bool F16, bool AG, bool NOP = false>
__device__ __forceinline__ void pr_mfma0(f32x4& acc, const f32x4& w, const f32x4& b) {
  if constexpr (NOP) {
    if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
  } else if constexpr (F16) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
  } else {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
  }
}
// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt: every stash store of
// the phase would be waited for at every barrier of the horizon loop, and nothing a barrier orders here goes through
// global memory.
__device__ __forceinline__ void pr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// One buffer descriptor over the whole workspace: every stash store of the sweeps is buffer_store v_data, v_lane_offset,
// srd, s_uniform_offset offset:imm -- no 64-bit address arithmetic in VGPRs, no scalar register pair per array.
typedef __amdgpu_buffer_rsrc_t pr_rsrc;
__device__ __forceinline__ pr_rsrc pr_make_rsrc(void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffff, 0x00020000);
}
// a value every lane holds alike, pinned into a scalar register: the soffset operand of a buffer access must be one, and
// hipcc wraps the access in a waterfall loop (readfirstlane / compare / saveexec per distinct value) whenever it cannot
// PROVE that -- loop-carried offsets of the horizon loop, for one (cdna_hip_programming.md T20)
__device__ __forceinline__ int pr_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Loads the compiler does not track (inline asm): the adjoint sweep fetches a step's per-row inputs a step (or half a
// step) before they are used, and hipcc's own s_waitcnt placement -- one counter for loads AND stores on gfx9, merged
// conservatively over the loop's back edge -- made the first use of a prefetched value wait for everything issued
// since, the loads requested a few instructions earlier included: a full memory round trip at the top of every step
// (3.8 k of a step's 10 k cycles).  With these the wait is written by hand: pr_landed<N>(values...) = s_waitcnt vmcnt(N)
// with N a LOWER bound of the memory operations issued after the loads in question (returns are in order), tied to the
// destination registers so that no use moves above it.  tools/check_inflight.py checks that nothing touches a
// destination in between (tests/test_isa_lint.py).
#ifdef PR_EXP_NOSTASH      // timing experiment only (tools/ubench): what a step costs without its stash stores
#define PR_EXP_STASH(x) ;
#else
#define PR_EXP_STASH(x) x
#endif
typedef int pr_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pr_i32x4 pr_rsrc_words(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  pr_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r[2] = -1;
  r[3] = 0x00020000;
  return r;
}
// (s_nop 4 in front: a scalar offset that a v_readfirstlane has just written needs 5 wait states before a vector-memory
//  instruction may read it, and the compiler's hazard recogniser does not look inside an asm statement)
__device__ __forceinline__ void pr_aload4_b8(unsigned (&d)[4], const pr_i32x4& srd, unsigned voff, int s0, int s1, int s2, int s3) {
  asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %4, %5, %6 offen\n\tbuffer_load_ubyte %1, %4, %5, %7 offen\n\t"
               "buffer_load_ubyte %2, %4, %5, %8 offen\n\tbuffer_load_ubyte %3, %4, %5, %9 offen"
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
               : "v"(voff), "s"(srd), "s"(s0), "s"(s1), "s"(s2), "s"(s3)
               : "memory");
}
__device__ __forceinline__ void pr_mfma_open() { asm volatile("s_nop 3"); }
__device__ __forceinline__ void pr_mfma_fence() { asm volatile("s_nop 7\n\ts_nop 7"); }

// Register-resident weights of one sweep direction: both networks' hidden->hidden layers, PR_SLOTS tiles per wave
// and network, PR_KB blocks x 2 pieces each = 84 fragments.  The first 64 go to the accumulator file ("a": all 256
// of its registers), the rest to VGPRs.
#define PR_NRESF (2 * PR_SLOTS * PR_KB * 2)
#define PR_NAG 62
struct RegW {
  f32x4 a[PR_NAG];
  f32x4 v[PR_NRESF - PR_NAG];
};
__host__ __device__ constexpr int pr_widx(int net, int slot, int kb, int p) { return ((net * PR_SLOTS + slot) * PR_KB + kb) * 2 + p; }

// (inline-asm loads, the accumulator-file fragments STRAIGHT into AGPRs: through the compiler they went to VGPRs first and
//  were copied over in batches of what the VGPR file had room for -- five serial memory round trips of the prologue.
//  All 84 requests go out back to back; one wait; the empty statements behind it hang every later use on the wait.)
template <int I>
__device__ __forceinline__ void pr_res_load(RegW& W, unsigned voff, const float* packed_dir, int wid) {
  typedef const PM_GLOBAL_ char* gcp;
  constexpr int net = I / (PR_SLOTS * PR_KB * 2), r = I % (PR_SLOTS * PR_KB * 2);
  gcp src = (gcp)(packed_dir + (size_t)net * PR_NET_FLOATS + PR_OFF_RES + ((size_t)wid * (PR_SLOTS * PR_KB * 2) + r) * PR_FRAG);
  if constexpr (I < PR_NAG) asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(W.a[I]) : "v"(voff), "s"(src) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W.v[I - PR_NAG]) : "v"(voff), "s"(src) : "memory");
}
template <int I>
__device__ __forceinline__ void pr_res_tie(RegW& W) {
  if constexpr (I < PR_NAG) asm volatile("" : "+a"(W.a[I]));
  else asm volatile("" : "+v"(W.v[I - PR_NAG]));
}
__device__ __forceinline__ void pr_load_resident(RegW& W, const float* packed_dir, int wid, int lane) {
  const unsigned voff = (unsigned)lane * 16u;
  pr_for<PR_NRESF>([&](auto ic) { pr_res_load<decltype(ic)::value>(W, voff, packed_dir, wid); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pr_for<PR_NRESF>([&](auto ic) { pr_res_tie<decltype(ic)::value>(W); });
}

// one hidden->hidden layer of network NET on this wave's resident tiles: B fragments from the LDS activation buffer,
// accumulators acc[slot][chain].  Chains: fp16 pieces -- 0 = the product with the weights' low piece, 1 = the rest
// (pmbrl_split.h); bf16 pieces -- the same split (the low-piece product is simply not rescaled).
// XT: this wave also runs the LDS-resident tile (weights read from LDS two blocks ahead) into accx.
template <int NET, bool F16, bool XT>
__device__ __forceinline__ void pr_hidden_layer(const RegW& W, const float* lds, const float* xtw, int lane,
                                                f32x4 (&acc)[PR_SLOTS][2], f32x4 (&accx)[2]) {
  const float* act = lds + PR_LDS_ACT + lane * 4;
  f32x4 b[2][2];        // B fragments of two blocks: one feeds the MFMAs, the next is in flight
  f32x4 xw[2][2];       // the LDS-resident tile's weights, same ring
  auto fetch = [&](int kb) {
    const int q = kb & 1;
    b[q][0] = *reinterpret_cast<const f32x4*>(act + (kb * 2 + 0) * PR_FRAG);
    b[q][1] = *reinterpret_cast<const f32x4*>(act + (kb * 2 + 1) * PR_FRAG);
    if constexpr (XT) {
      xw[q][0] = *reinterpret_cast<const f32x4*>(xtw + lane * 4 + (kb * 2 + 0) * PR_FRAG);
      xw[q][1] = *reinterpret_cast<const f32x4*>(xtw + lane * 4 + (kb * 2 + 1) * PR_FRAG);
    }
  };
  fetch(0);
  pr_for<PR_KB>([&](auto ic) {
    constexpr int kb = decltype(ic)::value;
    constexpr int q = kb & 1;
    if constexpr (kb + 1 < PR_KB) fetch(kb + 1);
    // order: every chain is touched once per round of PR_SLOTS (+1) MFMAs -- the dependent-MFMA latency is covered
    pr_for<PR_SLOTS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int I = pr_widx(NET, s, kb, 1);
      if constexpr (kb == 0) {
        if constexpr (I < PR_NAG) pr_mfma0<F16, true>(acc[s][0], W.a[I], b[q][0]);
        else pr_mfma0<F16, false>(acc[s][0], W.v[I - PR_NAG], b[q][0]);
      } else {
        if constexpr (I < PR_NAG) pr_mfma<F16, true>(acc[s][0], W.a[I], b[q][0]);
        else pr_mfma<F16, false>(acc[s][0], W.v[I - PR_NAG], b[q][0]);
      }
    });
    if constexpr (XT) {
      if constexpr (kb == 0) pr_mfma0<F16, false>(accx[0], xw[q][1], b[q][0]);
      else pr_mfma<F16, false>(accx[0], xw[q][1], b[q][0]);
    }
    pr_for<PR_SLOTS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int I = pr_widx(NET, s, kb, 0);
      if constexpr (kb == 0) {
        if constexpr (I < PR_NAG) pr_mfma0<F16, true>(acc[s][1], W.a[I], b[q][1]);
        else pr_mfma0<F16, false>(acc[s][1], W.v[I - PR_NAG], b[q][1]);
      } else {
        if constexpr (I < PR_NAG) pr_mfma<F16, true>(acc[s][1], W.a[I], b[q][1]);
        else pr_mfma<F16, false>(acc[s][1], W.v[I - PR_NAG], b[q][1]);
      }
    });
    if constexpr (XT) {
      if constexpr (kb == 0) pr_mfma0<F16, false>(accx[1], xw[q][0], b[q][1]);
      else pr_mfma<F16, false>(accx[1], xw[q][0], b[q][1]);
    }
    pr_for<PR_SLOTS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int I = pr_widx(NET, s, kb, 0);
      if constexpr (I < PR_NAG) pr_mfma<F16, true>(acc[s][1], W.a[I], b[q][0]);
      else pr_mfma<F16, false>(acc[s][1], W.v[I - PR_NAG], b[q][0]);
    });
    if constexpr (XT) pr_mfma<F16, false>(accx[1], xw[q][0], b[q][0]);
  });
  // the chains' last MFMAs are in flight: wait states before anything reads an accumulator, tied to the
  // accumulators so that no reader is scheduled above them
  if constexpr (XT)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                 "+v"(acc[2][1]), "+v"(accx[0]), "+v"(accx[1]));
  else
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                 "+v"(acc[2][1]));
}

// value of a tile from its two chains
template <bool F16>
__device__ __forceinline__ f32x4 pr_tile_value(const f32x4 (&c)[2]) {
  return pm_chains_sum<F16, 2>(c[0], c[1]);
}

// two packed 16-bit piece pairs of four fp32 values (hi.x = pieces of v0, v1; hi.y = of v2, v3)
template <bool F16>
__device__ __forceinline__ void pr_split(f32x4 h, pm_u32x2& hi, pm_u32x2& lo) {
  if constexpr (F16) {
    // two fp16 pieces: the residuals h - hi in one v_fma_mix_f32 each (it reads the fp16 half directly; pm_split4's form is
    // four conversions back and two packed subtractions) -- the same bits
    const unsigned ha = pm_pk_f16(h[0], h[1]), hb = pm_pk_f16(h[2], h[3]);
    float l0, l1, l2, l3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ha), "v"(h[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ha), "v"(h[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(hb), "v"(h[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(hb), "v"(h[3]));
    hi = pm_u32x2{ha, hb};
    lo = pm_u32x2{pm_pk_f16(l0, l1), pm_pk_f16(l2, l3)};
  } else {
    pm_u32x2 pc[2];
    pm_split4<2, F16>(h, pc);
    hi = pc[0];
    lo = pc[1];
  }
}

// ===========================================================================
// forward sweep
// ===========================================================================
// reciprocal to 1 ulp: v_rcp_f32 and one Newton step (the IEEE division is a dozen instructions on this part, and
// a single wave per SIMD pays every one of them in issue slots)
__device__ __forceinline__ float pr_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
}
// tanh through one exponential: absolute error ~1e-7 (what a = scale tanh(u) + bias needs; near zero the RELATIVE
// error is larger, which nothing here divides by)
// (|u| clamped to 15: tanh is +-1 to fp32 there already, and exp(2 u) overflows at u = 44 -- the reciprocal's Newton step
//  turns an infinity into a NaN: a policy that saturates hard, the double cart-pole shape late in the horizon, produced NaN
//  actions here until round 5)
// e^x for the clamped arguments of the two functions below (|x| <= 80): 2^(x log2 e) by v_exp_f32, the rounding error of the
// product x log2 e -- up to 2^-18 absolute at |x| = 80 -- put back through e^x (1 + eps ln 2).  Six instructions; libm's expf
// inlines to fifteen (range reduction by hand, ldexp, two range selects), four of them a step on a wave that is instruction issue
__device__ __forceinline__ float pr_exp(float x) {
  const float t = x * 1.44269504f;
  const float eps = __builtin_fmaf(x, 1.92596299e-8f, __builtin_fmaf(x, 1.44269504f, -t));
  const float p = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(p, eps * 0.693147181f, p);
}
// (v_med3_f32 answers min3 when an operand is a NaN: a NaN head output would leave here as tanh(-15) = -1, a finite action,
//  and the failed step would go unreported where the reference produces NaNs -- the NaN is put back by hand)
__device__ __forceinline__ float pr_tanh(float u) {
  const float r = 1.f - 2.f * pr_rcp(1.f + pr_exp(2.f * __builtin_amdgcn_fmed3f(u, -15.f, 15.f)));
  return u != u ? u : r;
}
// logistic function of x = ls - max_log_std through the same reciprocal: the exponent clamped for the same reason
// (sigma(-80) = 2e-35: zero to everything downstream)
// (both ways: pr_exp's error term is inf - inf at y = -inf)
__device__ __forceinline__ float pr_sigmoid_neg(float y) { return pr_rcp(1.f + pr_exp(__builtin_amdgcn_fmed3f(y, -80.f, 80.f))); }

// epilogue arithmetic of one hidden tile: h = max(v mf, 0) (mf = dropout multiplier >= 0), activity nibble, running
// maximum (fp16 range check), piece pairs.  Branch-free by construction: nothing here turns into an exec-masked region.
template <bool F16>
__device__ __forceinline__ void pr_tile_epilogue(f32x4 v, f32x4 mf, f32x4& h, unsigned& ab, unsigned& amax, pm_u32x2& hi,
                                                 pm_u32x2& lo) {
  const f32x4 t = v * mf;
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = fmaxf(t[r], 0.f);
  // (h >= +0: a unit is active iff its bit pattern is non-zero.  v_min_u32 by hand -- written as min(u, 1) the compiler makes a
  //  v_cmp_class and a v_cndmask of each: ten instructions a tile where seven do, in a kernel that is instruction issue)
  {
    unsigned x0, x1, x2, x3;
    asm("v_min_u32 %0, 1, %1" : "=v"(x0) : "v"(__float_as_uint(h[0])));
    asm("v_min_u32 %0, 1, %1" : "=v"(x1) : "v"(__float_as_uint(h[1])));
    asm("v_min_u32 %0, 1, %1" : "=v"(x2) : "v"(__float_as_uint(h[2])));
    asm("v_min_u32 %0, 1, %1" : "=v"(x3) : "v"(__float_as_uint(h[3])));
    unsigned a01, a23;
    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(a01) : "v"(x1), "v"(x0));
    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(a23) : "v"(x3), "v"(x2));
    asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(ab) : "v"(a23), "v"(a01));
  }
  // (h >= 0: the order of the bit patterns is the order of the values, a NaN's pattern is above every finite one's)
  amax = max(max(amax, max(__float_as_uint(h[0]), __float_as_uint(h[1]))), max(__float_as_uint(h[2]), __float_as_uint(h[3])));
  if constexpr (F16) {
    // two fp16 pieces: the residuals h - hi in one v_fma_mix_f32 each (it reads the fp16 half directly; pm_split4's form is
    // four conversions back and two packed subtractions) -- the same bits
    const unsigned ha = pm_pk_f16(h[0], h[1]), hb = pm_pk_f16(h[2], h[3]);
    float l0, l1, l2, l3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ha), "v"(h[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ha), "v"(h[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(hb), "v"(h[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(hb), "v"(h[3]));
    hi = pm_u32x2{ha, hb};
    lo = pm_u32x2{pm_pk_f16(l0, l1), pm_pk_f16(l2, l3)};
  } else {
    pm_u32x2 pc[2];
    pm_split4<2, F16>(h, pc);
    hi = pc[0];
    lo = pc[1];
  }
}

// Copy of the LDS-resident sections of one network's pack (first layer, 13th tile, head), ALL loads requested before the
// first store: written as three copy loops, every iteration waited for its own load -- 14 serial memory round trips a
// network, 36 k cycles (16 us) of prologue in each sweep kernel.  L0_BLOCKS: blocks per tile the destination keeps (the
// forward sweep keeps block 0 only of the packed [tile][2 blocks]).
template <int L0_BLOCKS>
__device__ __forceinline__ void pr_copy_net_to_lds(float* smem, int lds_l0, int lds_xt, int lds_head, const float* src, int tid) {
  constexpr int N_L0 = PR_NT * L0_BLOCKS * PR_FRAG / 4, N_XT = PR_XT_FLOATS / 4, N_HD = PR_HEAD_FLOATS / 4;
  constexpr int I_L0 = (N_L0 + PR_NTHR - 1) / PR_NTHR, I_XT = (N_XT + PR_NTHR - 1) / PR_NTHR, I_HD = (N_HD + PR_NTHR - 1) / PR_NTHR;
  f32x4 a[I_L0], b[I_XT], c[I_HD];
#pragma unroll
  for (int k = 0; k < I_L0; ++k) {
    const int i = min(tid + k * PR_NTHR, N_L0 - 1);
    a[k] = ldg4(src + PR_OFF_L0 + (L0_BLOCKS == 1 ? (i >> 6) * (2 * PR_FRAG) + (i & 63) * 4 : i * 4));
  }
#pragma unroll
  for (int k = 0; k < I_XT; ++k) b[k] = ldg4(src + PR_OFF_XT + min(tid + k * PR_NTHR, N_XT - 1) * 4);
#pragma unroll
  for (int k = 0; k < I_HD; ++k) c[k] = ldg4(src + PR_OFF_HEAD + min(tid + k * PR_NTHR, N_HD - 1) * 4);
#pragma unroll
  for (int k = 0; k < I_L0; ++k)
    if (tid + k * PR_NTHR < N_L0) *reinterpret_cast<f32x4*>(smem + lds_l0 + (tid + k * PR_NTHR) * 4) = a[k];
#pragma unroll
  for (int k = 0; k < I_XT; ++k)
    if (tid + k * PR_NTHR < N_XT) *reinterpret_cast<f32x4*>(smem + lds_xt + (tid + k * PR_NTHR) * 4) = b[k];
#pragma unroll
  for (int k = 0; k < I_HD; ++k)
    if (tid + k * PR_NTHR < N_HD) *reinterpret_cast<f32x4*>(smem + lds_head + (tid + k * PR_NTHR) * 4) = c[k];
}

// MMD: 0, or the state width of the instance that moment-matches the sampled states inside the sweep (pmbrl_reg_mm.h)
// TREE: the instance that carries the two-level statistics exchange (ONE group in more than 8 parts) -- its own: compiled
// into the shared instances its polling code cost them 12-21 more spilled scalar registers and 3-5 % of a sweep
template <bool PROF, int MMD = 0, bool TREE = false>
__global__ __launch_bounds__(PR_NTHR, 1) void pm_reg_fwd_kernel(const RegArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool F16 = true;
  constexpr int MMDc = MMD ? MMD : 2;
  typedef PM_GLOBAL_ uint8_t gu8;
  typedef PM_GLOBAL_ float gf32;
  typedef PM_GLOBAL_ char gch;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = pr_wg(A);
  if (wg < 0) return;
  const int row = lane & 15, g = lane >> 4;
  int row0, nvalid;
  pr_rows(A, wg, row0, nvalid);
  const bool rvalid = row < nvalid;
  const int D = A.D, U = A.U, B = A.B;
  const float* packed = A.packed;     // direction 0

  // (cycle stamps of the launch as a whole in row 0: slot 30 kernel entry, 31 kernel end; 28 / 29 the same instants on the
  //  constant 100 MHz clock -- the ratio is the shader clock the sweep actually ran at)
  if (PROF && wg == 0 && tid == 0) {
    A.prof[30] = (long long)__builtin_readcyclecounter();
    A.prof[28] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  // (first what comes from HBM in small pieces -- the lanes' constants, the mask rows: requested here, they land while the
  //  weights stream in; asked for afterwards they were three more serial round trips of the prologue)
  // ---- per-lane constants: input slot s (0, 1) of this lane group = network input 2 g + s
  float c_mx[2], c_isx[2], c_sy[2], c_my[2], c_zp[2], c_zd[2], c_psc[2], c_pbi[2];
  float x[2];                       // state dimensions 2 g, 2 g + 1 of this lane's row
  unsigned so_x[2], so_a[2];        // byte offsets of this lane's [t][row][dim] / [t][row][action] entries at t = 0
  bool ok_x[2], ok_a[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int i = 2 * g + s;
    const bool isd = i < D, isa = i >= D && i < D + U;
    const int ja = isa ? i - D : 0, id = isd ? i : 0;
    c_mx[s] = (isd || isa) ? A.mx[i] : 0.f;
    c_isx[s] = (isd || isa) ? A.iSx[i] : 0.f;
    c_sy[s] = isd ? A.Sy[id] : 0.f;
    c_my[s] = isd ? A.my[id] : 0.f;
    c_zp[s] = (isa && rvalid) ? A.zpol[(size_t)(row0 + row) * U + ja] : 0.f;
    c_zd[s] = (isd && rvalid) ? A.zdyn[(size_t)(row0 + row) * D + id] : 0.f;
    c_psc[s] = isa ? A.pscale[ja] : 0.f;
    c_pbi[s] = isa ? A.pbias[ja] : 0.f;
    x[s] = (isd && rvalid) ? A.x0[(size_t)(row0 + row) * D + id] : 0.f;
    ok_x[s] = isd && rvalid;
    ok_a[s] = isa && rvalid;
    so_x[s] = ((unsigned)(row0 + row) * D + id) * 4u;
    so_a[s] = ((unsigned)(row0 + row) * U + ja) * 4u;
  }
  f32x4 hbp, hbd;                   // head biases in accumulator-register order
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rp = pr_head_row(0, D, U, 4 * g + r), rd = pr_head_row(1, D, U, 4 * g + r);
    hbp[r] = rp >= 0 ? (A.pol_params + A.pol.b_off[2])[rp] : 0.f;
    hbd[r] = rd >= 0 ? (A.dyn_params + A.dyn.b_off[2])[rd] : 0.f;
  }
  unsigned mbits[2][2][PR_SLOTS + 1];      // (all sixteen requested before the first is used)
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const uint16_t* mrow = (n == 0 ? A.pol : A.dyn).mask[l] + (size_t)(row0 + (rvalid ? row : 0)) * PR_NT;
#pragma unroll
      for (int q = 0; q <= PR_SLOTS; ++q) mbits[n][l][q] = (unsigned)mrow[q < PR_SLOTS ? 4 * q + wid : PR_XT];
    }
  // ---- prologue: resident weights -> registers, the rest -> LDS
  RegW W;
  pr_load_resident(W, packed, wid, lane);
  if (PROF && wg == 0 && tid == 0) A.prof[27] = (long long)__builtin_readcyclecounter();
#pragma unroll
  for (int s = 0; s < 2; ++s)      // x_0 into the trajectory
    if (ok_x[s] && wid == 0) *(gf32*)((gch*)A.states + so_x[s]) = x[s];
  pr_copy_net_to_lds<1>(smem, PR_LDS_L0(0), PR_LDS_XT(0), PR_LDS_HEAD(0), packed, tid);
  pr_copy_net_to_lds<1>(smem, PR_LDS_L0(1), PR_LDS_XT(1), PR_LDS_HEAD(1), packed + PR_NET_FLOATS, tid);
  if (PROF && wg == 0 && tid == 0) A.prof[26] = (long long)__builtin_readcyclecounter();
  // activation buffer: zero, then the constant 1 the hidden-layer biases multiply (slot 4 of the last block, high plane)
  for (int i = tid; i < PR_KB * 2 * PR_FRAG; i += PR_NTHR) smem[PR_LDS_ACT + i] = 0.f;
  if (tid < 16) smem[PR_LDS_FLAG + tid] = 0.f;      // (the two-level exchange's tags: pmbrl_xch.h)
  // dropout multipliers of this lane's values, tile by tile (rows past the batch: zero -- their activations stay 0)
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const RegNet& N = n == 0 ? A.pol : A.dyn;
#pragma unroll
      for (int q = 0; q <= PR_SLOTS; ++q) {
        if (q == PR_SLOTS && wid != pr_xwave(n)) continue;
        const unsigned nib = rvalid ? ((mbits[n][l][q] >> (4 * g)) & 0xFu) : 0u;
        f32x4 mf;
#pragma unroll
        for (int r = 0; r < 4; ++r) mf[r] = ((nib >> r) & 1u) ? N.inv_keep[l] : 0.f;
        float* dst = q < PR_SLOTS ? smem + PR_LDS_MF + (size_t)(((wid * 2 + n) * 2 + l) * PR_SLOTS + q) * PR_FRAG
                                  : smem + PR_LDS_MFX + (size_t)(n * 2 + l) * PR_FRAG;
        *reinterpret_cast<f32x4*>(dst + lane * 4) = mf;
      }
    }
  __syncthreads();
  if (tid < 64) reinterpret_cast<unsigned*>(smem + PR_LDS_ACT + ((PR_KB - 1) * 2 + 0) * PR_FRAG + tid * 4)[2] = 0x3c00u;   // fp16 1.0, 0
  if (tid == 0 && A.wflag && *A.wflag == A.wgen) atomicMin(A.status, 0);
  if (PROF && wg == 0 && tid == 0) A.prof[25] = (long long)__builtin_readcyclecounter();

  // which input slots hold an action / a state dimension in ANY lane group (wave-uniform: the squash of a slot that
  // is an action nowhere is skipped)
  bool any_a[2], any_x[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    any_a[s] = false;
    any_x[s] = false;
    for (int gg = 0; gg < 4; ++gg) {
      const int i = 2 * gg + s;
      any_a[s] = any_a[s] || (i >= D && i < D + U);
      any_x[s] = any_x[s] || i < D;
    }
  }
  const float max_std_pol = expf(A.mls_pol), max_std_dyn = expf(A.mls_dyn);
  // moment matching: wave 0's state across the steps -- the reference point's column lane & 15 (the first one: the
  // group's first row, which every part can read) and the step's noise row / standardisation, requested a step ahead
  const int mm_gi = MMD ? wg / A.mm.parts : 0, mm_me = MMD ? wg - mm_gi * A.mm.parts : 0, mm_g0 = mm_gi * (MMD ? A.mm.M : 0);
  double mm_ref = 0.0;
  double* const mm_zh = reinterpret_cast<double*>(smem + PR_LDS_ZH);
  double* const mm_rec = reinterpret_cast<double*>(smem + PR_LDS_REC);
  if constexpr (MMD != 0) {
    if (wid == 0) mm_ref = row < MMD ? (double)A.x0[(size_t)mm_g0 * D + row] : 0.0;
    if (wid == 1) pr_mm_fwd_prep<MMDc>(A.mm, 0, mm_gi, mm_g0, row0, nvalid, mm_me, lane, mm_zh);
  }
  unsigned amax = 0u;               // bit pattern of the largest magnitude that went into an fp16 piece so far
  const bool xw_pol = wid == pr_xwave(0), xw_dyn = wid == pr_xwave(1);
  float* const act_w = smem + PR_LDS_ACT + lane * 4;     // this lane's 16 bytes of every B fragment
  const float* const mf_w = smem + PR_LDS_MF + (size_t)wid * (2 * 2 * PR_SLOTS * PR_FRAG) + lane * 4;
  const float* const mfx_w = smem + PR_LDS_MFX + lane * 4;
  const unsigned lane_st = ((4u * g) * 16u + row) * 4u;                // byte inside a stash block's tile
  const pr_rsrc srd = pr_make_rsrc(A.ws);
  // activity bits: one 32-bit word per lane, layer and step -- nibble q = the 4 features of this wave's tile in slot q
  // (q = 3: the LDS-resident tile, written by the wave that runs it), [step][workgroup][wave][lane] (PR_ABP_*): one
  // coalesced 256-byte store per wave and layer instead of a byte store per tile
  const unsigned vo_abp = (unsigned)lane * 4u;
  int so_abp = (wg * PR_NW + wid) * 256;
  const int abp_step = A.nwg * (PR_NW * 256);

  // one first layer: input B fragment from registers, one MFMA per tile, epilogue -> LDS (+ stashes)
  auto first_layer = [&](auto netc, const float (&in)[2], int so_ab, int so_st) {
    constexpr int NET = decltype(netc)::value;
    // B fragment: slots 3 s + {0, 1, 2} = in.hi, in.lo, in.hi 2^-11; slots 6, 7 = 1, 2^-11
    _Float16 hh[2], hl[2], hs[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      hh[s] = (_Float16)in[s];
      hl[s] = (_Float16)(in[s] - (float)hh[s]);
      hs[s] = hh[s] * (_Float16)PM_F16_LO_ISCALE;
      amax = max(amax, __float_as_uint(in[s]) & 0x7fffffffu);
#ifdef PR_EXP_AMAX_CODE
      if (!(fabsf(in[s]) <= 3.0e38f)) atomicMin(A.status, -((NET + 1) * 1000000 + lane * 1000 + s * 100 + wg));
#endif
    }
    const pm_f16x8 bf = {hh[0], hl[0], hs[0], hh[1], hl[1], hs[1], (_Float16)1.0f, (_Float16)PM_F16_LO_ISCALE};
    const float* l0w = smem + PR_LDS_L0(NET) + lane * 4;
    const bool xw = NET == 0 ? xw_pol : xw_dyn;
    f32x4 acc[PR_SLOTS + 1];
    const f32x4 bfv = __builtin_bit_cast(f32x4, bf);
    f32x4 wf[PR_SLOTS + 1];
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      wf[q] = *reinterpret_cast<const f32x4*>(l0w + (size_t)(4 * q + wid) * PR_FRAG);
    });
    wf[PR_SLOTS] = *reinterpret_cast<const f32x4*>(l0w + (size_t)PR_XT * PR_FRAG);
    f32x4 mfv[PR_SLOTS + 1];
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      mfv[q] = *reinterpret_cast<const f32x4*>(mf_w + (size_t)((NET * 2 + 0) * PR_SLOTS + q) * PR_FRAG);
    });
    mfv[PR_SLOTS] = *reinterpret_cast<const f32x4*>(mfx_w + (size_t)(NET * 2 + 0) * PR_FRAG);
    pr_mfma_open();
    pr_for<PR_SLOTS + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      pr_mfma0<F16, false, true>(acc[q], wf[q], bfv);       // (the LDS-resident tile: computed by every wave, used by one)
    });
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    unsigned abw = 0u;
    auto epi = [&](auto qc, int ot) {
      constexpr int q = decltype(qc)::value;
      f32x4 h;
      unsigned ab;
      pm_u32x2 hi, lo;
      pr_tile_epilogue<F16>(acc[q], mfv[q], h, ab, amax, hi, lo);
      float* dst = act_w + (size_t)((ot >> 1) * 2) * PR_FRAG + 2 * (ot & 1);
      *reinterpret_cast<pm_u32x2*>(dst) = hi;
      *reinterpret_cast<pm_u32x2*>(dst + PR_FRAG) = lo;
      abw |= ab << (4 * q);
      if constexpr (NET == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[r]), srd, lane_st + r * 64, pr_uni(so_st + ot * 1024), 0);)
      }
    };
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      epi(qc, 4 * q + wid);
    });
    if (xw) epi(std::integral_constant<int, PR_SLOTS>{}, PR_XT);
    PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(abw, srd, vo_abp, pr_uni(so_ab), 0);)
  };

  // hidden->hidden layer + head partial of network NET; leaves the summed head tile (+ bias) in `o`
  auto second_layer_and_head = [&](auto netc, int so_ab, int so_st, f32x4& o) {
    constexpr int NET = decltype(netc)::value;
    const bool xw = NET == 0 ? xw_pol : xw_dyn;
    f32x4 acc[PR_SLOTS][2], accx[2];
    // LDS operands of the epilogues and of the head: requested before the MFMAs (one wave per SIMD: nothing else
    // covers an LDS round trip taken right before its use)
    f32x4 mfv[PR_SLOTS + 1], hwv[2][2];
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      mfv[q] = *reinterpret_cast<const f32x4*>(mf_w + (size_t)((NET * 2 + 1) * PR_SLOTS + q) * PR_FRAG);
    });
    mfv[PR_SLOTS] = *reinterpret_cast<const f32x4*>(mfx_w + (size_t)(NET * 2 + 1) * PR_FRAG);
    if (xw) pr_hidden_layer<NET, F16, true>(W, smem, smem + PR_LDS_XT(NET), lane, acc, accx);
    else pr_hidden_layer<NET, F16, false>(W, smem, nullptr, lane, acc, accx);
    // (the head's weights: requested here, they land behind the epilogues)
    const float* hw = smem + PR_LDS_HEAD(NET) + (size_t)wid * (4 * PR_FRAG) + lane * 4;
#pragma unroll
    for (int blkk = 0; blkk < 2; ++blkk) {
      hwv[blkk][0] = *reinterpret_cast<const f32x4*>(hw + (blkk * 2 + 0) * PR_FRAG);
      hwv[blkk][1] = *reinterpret_cast<const f32x4*>(hw + (blkk * 2 + 1) * PR_FRAG);
    }
    // epilogues: activations of the wave's tiles as piece pairs (they ARE the head's B operand), stashes
    pm_u32x2 hi[PR_SLOTS + 1], lo[PR_SLOTS + 1];
    hi[PR_SLOTS] = lo[PR_SLOTS] = pm_u32x2{0u, 0u};
    unsigned abw = 0u;
    auto epi = [&](auto qc, f32x4 v, int ot) {
      constexpr int q = decltype(qc)::value;
      f32x4 h;
      unsigned ab;
      pr_tile_epilogue<F16>(v, mfv[q], h, ab, amax, hi[q], lo[q]);
      abw |= ab << (4 * q);
      if constexpr (NET == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[r]), srd, lane_st + r * 64, pr_uni(so_st + ot * 1024), 0);)
      }
    };
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      epi(qc, pr_tile_value<F16>(acc[q]), 4 * q + wid);
    });
    if (xw) epi(std::integral_constant<int, PR_SLOTS>{}, pr_tile_value<F16>(accx), PR_XT);
    PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(abw, srd, vo_abp, pr_uni(so_ab), 0);)
    // head, K-split: block 0 = slots 0, 1; block 1 = slot 2 and the LDS-resident tile (zeros elsewhere)
    f32x4 c0, c1, bh[2], bl[2];
#pragma unroll
    for (int blkk = 0; blkk < 2; ++blkk) {
      const pm_u32x2 ha = hi[2 * blkk], hb = hi[2 * blkk + 1], la = lo[2 * blkk], lb = lo[2 * blkk + 1];
      bh[blkk] = __builtin_bit_cast(f32x4, (pr_u32x4){ha[0], ha[1], hb[0], hb[1]});
      bl[blkk] = __builtin_bit_cast(f32x4, (pr_u32x4){la[0], la[1], lb[0], lb[1]});
    }
    pr_mfma_open();
    pr_mfma0<F16, false, true>(c0, hwv[0][1], bh[0]);
    pr_mfma0<F16, false, true>(c1, hwv[0][0], bl[0]);
    pr_mfma<F16, false, true>(c0, hwv[1][1], bh[1]);
    pr_mfma<F16, false, true>(c1, hwv[0][0], bh[0]);
    pr_mfma<F16, false, true>(c1, hwv[1][0], bl[1]);
    pr_mfma<F16, false, true>(c1, hwv[1][0], bh[1]);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1));
    float* part = smem + PR_LDS_PART + lane * 4;
    *reinterpret_cast<f32x4*>(part + wid * PR_FRAG) = pm_chains_sum<F16, 2>(c0, c1);
    pr_barrier();
    o = NET == 0 ? hbp : hbd;
#pragma unroll
    for (int w = 0; w < PR_NW; ++w) o += *reinterpret_cast<const f32x4*>(part + w * PR_FRAG);
  };

  // uniform 32-bit workspace offsets that advance with the step (one s_add each; the lanes' offsets are loop constants)
  int so_st0 = (int)A.actT[0] + wg * 1024, so_st1 = (int)A.actT[1] + wg * (PR_NT * 1024), so_st2 = (int)A.actT[2] + wg * (PR_NT * 1024);
  int so_td = (int)A.Td, so_tp = (int)A.Tp;
  gch* b_states = (gch*)A.states + (size_t)B * D * 4u;     // x_{t+1}
  int so_xt = (int)A.mm.xt_off;                            // x~_{t+1} (moment matching: the pre-mm sample), in the workspace
  gch* b_actions = (gch*)A.actions;
  const int st_step0 = A.nwg * 1024, st_step = A.nwg * (PR_NT * 1024);
  const unsigned x_step = (unsigned)B * D * 4u, a_step = (unsigned)B * U * 4u;

  for (int t = 0; t < A.H; ++t) {
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 0] = (long long)__builtin_readcyclecounter();
    // policy-input stash (feature-major [16][16] block: dimensions 0 .. 7 by wave 2, the zero rest by wave 3)
    if (wid == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[s]), srd, ((2 * g + s) * 16 + row) * 4, pr_uni(so_st0), 0);
    } else if (wid == 3) {
      __builtin_amdgcn_raw_buffer_store_b32(0u, srd, 512 + lane * 4, pr_uni(so_st0), 0);
      __builtin_amdgcn_raw_buffer_store_b32(0u, srd, 768 + lane * 4, pr_uni(so_st0), 0);
    }
    // ---- policy
    first_layer(std::integral_constant<int, 0>{}, x, (int)A.pol.abits[0] + so_abp, so_st1);
    pr_barrier();
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 1] = (long long)__builtin_readcyclecounter();
    f32x4 o;
    second_layer_and_head(std::integral_constant<int, 0>{}, (int)A.pol.abits[1] + so_abp, so_st2, o);
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 2] = (long long)__builtin_readcyclecounter();
    // ---- squash; dynamics input (normalised) in the same lane groups
    float xin[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v = x[s];
      if (any_a[s]) {
        const float mu = o[2 * s], ls = o[2 * s + 1];
        const float sg = pr_sigmoid_neg(A.mls_pol - ls);
        const float e = max_std_pol * sg;
        const float u = mu + c_zp[s] * e;
        const float a = c_psc[s] * pr_tanh(u) + c_pbi[s];
        if (wid == 1 && ok_a[s]) {
          *(gf32*)(b_actions + so_a[s]) = a;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c_zp[s] * e * (1.f - sg)), srd, so_a[s], pr_uni(so_tp), 0);
        }
        v = ok_a[s] ? a : v;
      }
      xin[s] = (v - c_mx[s]) * c_isx[s];
    }
    // ---- dynamics
    first_layer(std::integral_constant<int, 1>{}, xin, (int)A.dyn.abits[0] + so_abp, 0);
    pr_barrier();
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 3] = (long long)__builtin_readcyclecounter();
    second_layer_and_head(std::integral_constant<int, 1>{}, (int)A.dyn.abits[1] + so_abp, 0, o);
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 4] = (long long)__builtin_readcyclecounter();
    // ---- sample the next state
    float xs[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (any_x[s]) {
        const float mu = o[2 * s], ls = o[2 * s + 1];
        const float sg = pr_sigmoid_neg(A.mls_dyn - ls);
        const float e = max_std_dyn * c_sy[s] * sg;
        const float xn = x[s] + (mu * c_sy[s] + c_my[s] + c_zd[s] * e);
        // (wave 3 stores: the moment matching's chain runs on wave 0 right behind this, and reuses the registers a store of
        //  its own would still be reading -- the compiler waits for such a store with vmcnt(0), a write's round trip)
        if (wid == 3 && ok_x[s]) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c_zd[s] * e * (1.f - sg)), srd, so_x[s], pr_uni(so_td), 0);
          // (moment matching: the sample the reward sees goes to its own stash; x_{t+1} is its moment-matched twin)
          if constexpr (MMD != 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xn), srd, so_x[s], pr_uni(so_xt), 0);
        }
        x[s] = ok_x[s] ? xn : 0.f;
        xs[s] = x[s];
        // (x_{t+1} from the register that carries it through the next step: a store of a temporary is waited for -- vmcnt(0),
        //  a write's round trip -- as soon as the temporary's register is reused)
        if constexpr (MMD == 0) {
          if (wid == 3 && ok_x[s]) *(gf32*)(b_states + so_x[s]) = x[s];
        }
      }
    }
    if constexpr (MMD != 0) {
      // ---- moment matching of the group's sampled states (utils/rollout.py:20-29): wave 0, the others wait
      if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 6] = (long long)__builtin_readcyclecounter();
      // (the lane index laundered per step: hoisted out of the horizon loop, the chain's per-lane constants -- masks as
      //  doubles, addresses -- would sit in registers through the GEMM phases, which have none to spare)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      // (... and the moment matching's arguments re-read from the kernel-argument segment per step: a dozen pointers and
      //  counts held in scalar registers across the loop are what the loop's own uniform offsets were spilled for)
      pr_mm_kptr Qp = pr_mm_args();
      asm volatile("" : "+s"(Qp));
      const RegMM Q = pr_mm_load(Qp);      // (ONE batch of scalar loads, one wait -- through the pointer every use was a round trip of its own)
      // (TREE -- the two-level exchange, below: the barrier that frees the heads' partial-tile buffer for the helper waves'
      //  sums stands HERE, in front of the other waves' duties: behind them the chain's wave waited for the factor record
      //  to be filed, 1.5 k cycles a step)
      if constexpr (TREE) pr_barrier();
      if (wid == 1 && t + 1 < A.H) {
        // (idle otherwise) the next step's standardised noise rows
        pr_mm_fwd_prep<MMDc>(Q, t + 1, mm_gi, mm_g0, row0, nvalid, mm_me, ln, mm_zh + ((t + 1) & 1) * (16 * MMDc));
        if (PROF && wg == 0 && lane == 0) A.prof[(size_t)t * 32 + 13] = (long long)__builtin_readcyclecounter();
      }
      // (idle otherwise) the previous step's factor record for the adjoint sweep, from what wave 0 left behind that step's barrier
      if (wid == 2 && t > 0) {
        if (PROF && wg == 0 && lane == 0) A.prof[(size_t)t * 32 + 14] = (long long)__builtin_readcyclecounter();
        pr_mm_fwd_file<MMDc>(Q, t - 1, mm_gi, mm_me, ln, mm_rec);
        if (PROF && wg == 0 && lane == 0) A.prof[(size_t)t * 32 + 15] = (long long)__builtin_readcyclecounter();
      }
      // ONE group in more than 8 parts (the two-level exchange): the waves that are not the chain's poll a quarter of each
      // level's slots and leave the sums in LDS (pmbrl_xch.h) -- in the heads' partial-tile buffer (3 of its 4 KB), which is
      // dead from the dynamics head's last read to the next step's policy head; the barrier is what says that every wave HAS
      // read its head value (the forward sweep's LDS is full: 448 bytes free).  The tags live in words of their own.
      double* const xh = reinterpret_cast<double*>(smem + PR_LDS_PART);
      volatile unsigned* const xtag = reinterpret_cast<volatile unsigned*>(smem + PR_LDS_FLAG);
      if constexpr (TREE) {
        if (wid != 0) {
          const bool okh = pm_xch_tree_help<PR_MM_NVX(MMDc), 4>(Q.xch, Q.nwg, mm_gi * Q.parts, Q.parts, Q.fan, mm_me, Q.tag0 + (unsigned)(t + 1), wid, xh,
                                                  xtag, ln);
          if (!okh && lane == 0) atomicMin(A.status, t);
        }
      }
      double rec[3] = {0.0, 0.0, 0.0};
      if (wid == 0) {
        float xo[2];
        const bool ok = pr_mm_fwd_chain<MMDc, TREE>(Q, t, Q.tag0 + (unsigned)(t + 1), mm_gi, mm_me, mm_gi * Q.parts, nvalid, ln, xs, mm_ref,
                                               mm_zh + (t & 1) * (16 * MMDc), xo, rec, xh, xtag,
                                               (PROF && wg <= 1) ? A.prof + (size_t)t * 32 + 8 * wg : nullptr);
        if (!ok && lane == 0) atomicMin(A.status, t);
        *reinterpret_cast<f32x2*>(smem + PR_LDS_XB + row * 8 + 2 * g) = f32x2{xo[0], xo[1]};
      }
      pr_barrier();
      // (behind the barrier: wave 2 has read the previous step's hand-over)
      if (wid == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) mm_rec[i * 64 + ln] = rec[i];
      }
      const f32x2 xm = *reinterpret_cast<const f32x2*>(smem + PR_LDS_XB + row * 8 + 2 * g);
#pragma unroll
      for (int s = 0; s < 2; ++s) x[s] = ok_x[s] ? xm[s] : 0.f;
      // (the moment-matched x_{t+1} into the trajectory: by wave 3, from the registers that carry it -- stored by the chain's
      //  wave from its temporaries, that wave sat out the store's round trip at the next reuse of the register)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (wid == 3 && ok_x[s]) *(gf32*)(b_states + so_x[s]) = x[s];
      if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 7] = (long long)__builtin_readcyclecounter();
    }
    // fp16 pieces: a value beyond the format's range was rounded to infinity somewhere in this step (or earlier)
#ifdef PR_EXP_AMAX_CODE
    if (amax > 0x477fe000u) atomicMin(A.status, -1);
#else
    if (amax > 0x477fe000u) atomicMin(A.status, t);      // 65504
#endif
    // next step's bases
    so_abp += abp_step;
    so_st0 += st_step0; so_st1 += st_step; so_st2 += st_step;
    so_td += (int)x_step; so_tp += (int)a_step;
    b_states += x_step; b_actions += a_step; so_xt += (int)x_step;
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 5] = (long long)__builtin_readcyclecounter();
  }
  if constexpr (MMD != 0) {
    // the last step's factor record
    pr_barrier();
    if (wid == 2 && A.H > 0) {
      const RegMM Q = pr_mm_load(pr_mm_args());
      pr_mm_fwd_file<MMDc>(Q, A.H - 1, mm_gi, mm_me, lane, mm_rec);
    }
  }
  if (PROF && wg == 0 && tid == 0) {
    A.prof[31] = (long long)__builtin_readcyclecounter();
    A.prof[29] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// activity words -> the per-tile nibble bytes [step][row][tile][lane group] of pmbrl_fast.h
struct RegUnpackArgs {
  int B, H, nwg;
  int mm_on, M, parts, rpw;      // the workgroups' rows under moment matching (pr_rows)
  const unsigned* src[2][2];
  unsigned char* dst[2][2];
};
__global__ __launch_bounds__(PR_NTHR) void pm_reg_unpack_abits_kernel(const RegUnpackArgs U) {
  const int wg = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  int row0 = wg * 16, nvalid = min(16, U.B - wg * 16);
  if (U.mm_on) {
    const int gi = wg / U.parts, me = wg - gi * U.parts;
    row0 = gi * U.M + me * U.rpw;
    nvalid = min(U.rpw, U.M - me * U.rpw);
  }
  const int rowg = row0 + (lane & 15), g = lane >> 4;
  if ((lane & 15) >= nvalid) return;
  for (int n = 0; n < 2; ++n)
    for (int l = 0; l < 2; ++l) {
      const unsigned w = U.src[n][l][((size_t)t * U.nwg + wg) * PR_NTHR + tid];
      unsigned char* d = U.dst[n][l] + ((size_t)t * U.B + rowg) * (PR_NT * 4) + g;
#pragma unroll
      for (int q = 0; q < PR_SLOTS; ++q) d[(4 * q + wid) * 4] = (unsigned char)((w >> (4 * q)) & 15u);
      if (wid == pr_xwave(n)) d[PR_XT * 4] = (unsigned char)((w >> (4 * PR_SLOTS)) & 15u);
    }
}

// ===========================================================================
// adjoint sweep
// ===========================================================================
// Same workgroup shape, same partition of the tiles, the transposed weights resident (two bf16 pieces: gradients of any
// magnitude).  Per step, dynamics model first:
//   head adjoint   [g Sy | g Td] (two K = 32 blocks: the means' rows, the log-stds' rows) -> hidden width, two MFMAs a tile
//   x activity bits of the second hidden layer -> LDS -> barrier
//   V1^T           resident tiles -> x activity bits of the first hidden layer (stays in registers)
//   tail           V0, K-split over the waves from registers -> partial tiles -> barrier -> sum: gradient of [x | a]
// then the policy the same way, with the pre-activation gradients stashed for the dW GEMM (feature-major, what
// pm_dw_* read).  The activity nibbles of a step are fetched a step ahead; a nibble becomes its four multipliers
// {0, 1 / keep} through a 16-entry table in LDS (one ds_read_b128).
#define PRB_LDS_ACT 0
#define PRB_LDS_PART (PRB_LDS_ACT + PR_KB * 2 * PR_FRAG)
#define PRB_LDS_L0(net) (PRB_LDS_PART + PR_NW * PR_FRAG + (net) * (PR_L0P_FLOATS + PR_XT_FLOATS + PR_HEAD_FLOATS))
#define PRB_LDS_XT(net) (PRB_LDS_L0(net) + PR_L0P_FLOATS)
#define PRB_LDS_HEAD(net) (PRB_LDS_XT(net) + PR_XT_FLOATS)
#define PRB_LDS_LUT (PRB_LDS_L0(2))              // [net][layer][16 nibbles][4]
#define PRB_IN_COLS 32
#define PRB_LDS_IN (PRB_LDS_LUT + 2 * 2 * 16 * 4)    // [16 rows][PRB_IN_COLS]: the step's per-row inputs
#define PRB_LDS_XB (PRB_LDS_IN + 16 * PRB_IN_COLS)  // [16 rows][8]: dL/dx~ behind the moment matching's adjoint, wave 0 -> every wave
#define PRB_LDS_MMB (PRB_LDS_XB + 128)              // [2][4][64] doubles: the noise operand, wave 1 -> wave 0 (pmbrl_reg_mm.h)
#define PRB_LDS_MMY (PRB_LDS_MMB + 2 * 2 * PR_MM_BOP_DOUBLES)    // [2][3 NK][64] doubles: Y1 | L | L^-T, wave 2 -> wave 0
#define PRB_LDS_XH (PRB_LDS_MMY + 2 * 2 * PR_MM_YOP_DOUBLES(6))      // the two-level exchange's helper sums (pmbrl_xch.h): 6 KB
#define PRB_LDS_XTAG (PRB_LDS_XH + 2 * PM_XCH_HELP_DOUBLES(2))        // ... and their tags: [2 levels][3 waves] + 1 words
#define PRB_LDS_FLOATS (PRB_LDS_XTAG + 16)

// TREE: the instance that carries the two-level statistics exchange (ONE group in more than 8 parts) -- its own: compiled
// into the shared instances its polling code cost them 12-21 more spilled scalar registers and 3-5 % of a sweep
template <bool PROF, int MMD = 0, bool TREE = false>
__global__ __launch_bounds__(PR_NTHR, 1) void pm_reg_bwd_kernel(const RegArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool F16 = false;
  constexpr int MMDc = MMD ? MMD : 2;
  typedef PM_GLOBAL_ float gf32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = pr_wg(A);
  if (wg < 0) return;
  const int row = lane & 15, g = lane >> 4;
  int row0, nvalid;
  pr_rows(A, wg, row0, nvalid);
  const bool rvalid = row < nvalid;
  const int D = A.D, U = A.U, B = A.B;
  const int T0 = A.t0;
  int T1 = A.t1;
  if (A.nvalid) T1 = min(T1, __builtin_amdgcn_readfirstlane(*A.nvalid));
  const float* packed = A.packed + (size_t)2 * PR_NET_FLOATS;     // direction 1
  if (T1 <= T0) {
    for (int i = tid; i < nvalid * D; i += PR_NTHR) {
      if (A.grad_x0 && T0 == 0) A.grad_x0[(size_t)row0 * D + i] = 0.f;
      if (A.gx_out) A.gx_out[(size_t)row0 * D + i] = A.gx_in ? A.gx_in[(size_t)row0 * D + i] : 0.f;
    }
    return;
  }

  if (PROF && wg == 0 && tid == 0) {
    A.prof[30] = (long long)__builtin_readcyclecounter();
    A.prof[28] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  // (requested before the weights: they land while those stream in)
  // ---- per-lane constants: input slot s (0, 1) of this lane group = network input 2 g + s
  float c_isx[2], c_sy[2], c_psc[2], c_pbi[2];
  bool ok_x[2], ok_a[2];
  unsigned so_x[2], so_a[2];        // byte offsets of this lane's [t][row][dim] / [t][row][action] entries at t = 0
  float gx[2] = {0.f, 0.f};         // dL/dx_{t+1} of state dimensions 2 g, 2 g + 1 of this lane's row
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int i = 2 * g + s;
    const bool isd = i < D, isa = i >= D && i < D + U;
    const int ja = isa ? i - D : 0, id = isd ? i : 0;
    c_isx[s] = (isd || isa) ? A.iSx[i] : 0.f;
    c_sy[s] = isd ? A.Sy[id] : 0.f;
    c_psc[s] = isa ? A.pscale[ja] : 1.f;
    c_pbi[s] = isa ? A.pbias[ja] : 0.f;
    ok_x[s] = isd && rvalid;
    ok_a[s] = isa && rvalid;
    so_x[s] = ((unsigned)(row0 + row) * D + id) * 4u;
    so_a[s] = ((unsigned)(row0 + row) * U + ja) * 4u;
    if (A.gx_in && ok_x[s]) gx[s] = *(const gf32*)((const PM_GLOBAL_ char*)A.gx_in + so_x[s]);
  }
  // ---- prologue
  RegW W;
  pr_load_resident(W, packed, wid, lane);
  pr_copy_net_to_lds<2>(smem, PRB_LDS_L0(0), PRB_LDS_XT(0), PRB_LDS_HEAD(0), packed, tid);
  pr_copy_net_to_lds<2>(smem, PRB_LDS_L0(1), PRB_LDS_XT(1), PRB_LDS_HEAD(1), packed + PR_NET_FLOATS, tid);
  for (int i = tid; i < PR_KB * 2 * PR_FRAG; i += PR_NTHR) smem[PRB_LDS_ACT + i] = 0.f;
  if (tid < 16) smem[PRB_LDS_XTAG + tid] = 0.f;     // (the two-level exchange's tags: pmbrl_xch.h)
  if (tid < 64) {
    // nibble -> {0, 1 / keep} x 4
    const int n = tid >> 5, l = (tid >> 4) & 1, nib = tid & 15;
    const float ik = (n == 0 ? A.pol : A.dyn).inv_keep[l];
    f32x4 m;
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = ((nib >> r) & 1) ? ik : 0.f;
    *reinterpret_cast<f32x4*>(smem + PRB_LDS_LUT + tid * 4) = m;
  }
  __syncthreads();

  const bool xw_pol = wid == pr_xwave(0), xw_dyn = wid == pr_xwave(1);
  float* const act_w = smem + PRB_LDS_ACT + lane * 4;
  const float* const lut = smem + PRB_LDS_LUT;
  const unsigned lane_st = ((4u * g) * 16u + row) * 4u;
  const pr_rsrc srd = pr_make_rsrc(A.ws);
  const unsigned x_step = (unsigned)B * D * 4u, a_step = (unsigned)B * U * 4u;
  const int st_step = A.nwg * (PR_NT * 1024), st_step2 = A.nwg * 1024;
  int so_g0 = (int)A.gT[0] + ((T1 - 1) * A.nwg + wg) * (PR_NT * 1024);
  int so_g1 = (int)A.gT[1] + ((T1 - 1) * A.nwg + wg) * (PR_NT * 1024);
  int so_g2 = (int)A.gT[2] + ((T1 - 1) * A.nwg + wg) * 1024;

  // ---- per-row inputs of a step, staged through LDS.  The 16 rows of a step need dL/dr~, the reward Jacobian Jx | Ja,
  // Td | Tp and the action of every input slot: PRB_IN_COLS floats a row, the same for all four waves.  The 512 floats of
  // the stage are fetched by the workgroup ONCE, two a thread (thread e -> row e >> 5, column e & 31; every column of
  // every array at its own address: a 64-bit pointer a thread, stepped back by the array's step stride), a step ahead of
  // their use, parked in LDS between the step's third and fourth barrier, and read by the lanes that need them at the
  // top of the next step (a ds_read_b64 an array).  36 vector-memory loads a wave and step became 6 (with the activity
  // words below): the loads of four waves queued behind one another in the CU's one address unit were 2 k of a step's
  // 10 k cycles.
  //   column 0: dL/dr~;  2 + i: Jx[i] (i < D) or Ja[i - D];  10 + i: Td[i] or Tp[i - D];  18 + i: a[i - D]
  float* const stage = smem + PRB_LDS_IN;
  typedef const PM_GLOBAL_ char* gcp;
  gcp sp[2];
  unsigned sstep[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + k * PR_NTHR, r = e >> 5, c = e & 31;
    const long long grow = row0 + r;
    gcp q = (gcp)A.ws;
    unsigned st = 0u;
    if (r < nvalid) {
      const int i = c >= 18 ? c - 18 : (c >= 10 ? c - 10 : c - 2);
      if (c == 0) {
        q = (gcp)A.grad_rewards + ((long long)(T1 - 1) * B + grow) * 4;
        st = (unsigned)B * 4u;
      } else if (c >= 2 && c < 26 && i < D && c < 18) {
        q = (gcp)A.ws + (c < 10 ? A.Jx : A.Td) + ((long long)(T1 - 1) * B + grow) * D * 4 + i * 4;
        st = x_step;
      } else if (c >= 2 && c < 26 && i >= D && i < D + U) {
        q = (c >= 18 ? (gcp)A.actions : (gcp)A.ws + (c < 10 ? A.Ja : A.Tp)) + ((long long)(T1 - 1) * B + grow) * U * 4 + (i - D) * 4;
        st = a_step;
      }
    }
    sp[k] = q;
    sstep[k] = st;
  }
  float sv[2];
  auto stage_load = [&]() {
    asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(sv[0]), "=&v"(sv[1]) : "v"(sp[0]), "v"(sp[1]) : "memory");
  };
  auto stage_landed = [&](auto nc, float (&v)[2]) {
    constexpr int N = decltype(nc)::value;
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(v[0]), "+v"(v[1]) : "n"(N) : "memory");
  };
  auto stage_park = [&]() {
    stage[tid] = sv[0];
    stage[tid + PR_NTHR] = sv[1];
  };
  // activity words of a step's four hidden layers (pm_reg_fwd_kernel: one word a lane, nibble q = the wave's tile in slot
  // q): [0] dynamics layer 1, [1] dynamics layer 0, [2] policy layer 1, [3] policy layer 0
  const pr_i32x4 wsrd = pr_rsrc_words(A.ws);
  const unsigned vo_abp = (unsigned)lane * 4u;
  const int abp_step = A.nwg * (PR_NW * 256);
  int so_abp = ((T1 - 1) * A.nwg + wg) * (PR_NW * 256) + wid * 256;
  auto ab_load = [&](unsigned (&w)[4]) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %4, %5, %6 offen\n\tbuffer_load_dword %1, %4, %5, %7 offen\n\t"
                 "buffer_load_dword %2, %4, %5, %8 offen\n\tbuffer_load_dword %3, %4, %5, %9 offen"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(vo_abp), "s"(wsrd), "s"(pr_uni((int)A.dyn.abits[1] + so_abp)), "s"(pr_uni((int)A.dyn.abits[0] + so_abp)),
                   "s"(pr_uni((int)A.pol.abits[1] + so_abp)), "s"(pr_uni((int)A.pol.abits[0] + so_abp))
                 : "memory");
  };
  auto ab_landed = [&](auto nc, unsigned (&w)[4]) {
    constexpr int N = decltype(nc)::value;
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N) : "memory");
  };
  // what a lane reads back from the stage
  const float* const st_row = stage + row * PRB_IN_COLS;
  struct StepIn {
    float gr, jx[2], td[2];
  };
  auto read_in = [&](StepIn& I) {
    const f32x2 a = *reinterpret_cast<const f32x2*>(st_row + 2 + 2 * g), b = *reinterpret_cast<const f32x2*>(st_row + 10 + 2 * g);
    I.gr = rvalid ? st_row[0] : 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      I.jx[s] = ok_x[s] ? a[s] : 0.f;
      I.td[s] = ok_x[s] ? b[s] : 0.f;
    }
  };

  // head adjoint of network NET: inputs gm (x the means' rows), gl (x the log-stds' rows) of the lane's two slots;
  // result x activity of the second hidden layer -> LDS (+ stash)
  auto head_adjoint = [&](auto netc, const float (&gm)[2], const float (&gl)[2], const unsigned abw, int so_st) {
    constexpr int NET = decltype(netc)::value;
    // B fragments: slots 3 s + {0, 1, 2} = g.hi, g.lo, g.hi
    f32x4 bf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned short hh[2], hl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float v = k == 0 ? gm[s] : gl[s];
        const unsigned a = pm_pk_bf16(v, 0.f) & 0xffffu;
        hh[s] = (unsigned short)a;
        hl[s] = (unsigned short)(pm_pk_bf16(v - pm_bf_lo(a), 0.f) & 0xffffu);
      }
      const pr_u32x4 u = {(unsigned)hh[0] | ((unsigned)hl[0] << 16), (unsigned)hh[0] | ((unsigned)hh[1] << 16),
                          (unsigned)hl[1] | ((unsigned)hh[1] << 16), 0u};
      bf[k] = __builtin_bit_cast(f32x4, u);
    }
    const float* l0w = smem + PRB_LDS_L0(NET) + lane * 4;
    const bool xw = NET == 0 ? xw_pol : xw_dyn;
    f32x4 acc[PR_SLOTS + 1], wf[PR_SLOTS + 1][2], mfv[PR_SLOTS + 1];
    pr_for<PR_SLOTS + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int ot = q < PR_SLOTS ? 4 * q + wid : PR_XT;
      wf[q][0] = *reinterpret_cast<const f32x4*>(l0w + (size_t)(ot * 2 + 0) * PR_FRAG);
      wf[q][1] = *reinterpret_cast<const f32x4*>(l0w + (size_t)(ot * 2 + 1) * PR_FRAG);
      mfv[q] = *reinterpret_cast<const f32x4*>(lut + ((NET * 2 + 1) * 16 + ((abw >> (4 * q)) & 15u)) * 4);
    });
    pr_mfma_open();
    pr_for<PR_SLOTS + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      pr_mfma0<F16, false, true>(acc[q], wf[q][0], bf[0]);
    });
    pr_for<PR_SLOTS + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      pr_mfma<F16, false, true>(acc[q], wf[q][1], bf[1]);
    });
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    auto epi = [&](auto qc, int ot) {
      constexpr int q = decltype(qc)::value;
      const f32x4 gq = acc[q] * mfv[q];
      pm_u32x2 pc[2];
      pm_split4<2, false>(gq, pc);
      float* dst = act_w + (size_t)((ot >> 1) * 2) * PR_FRAG + 2 * (ot & 1);
      *reinterpret_cast<pm_u32x2*>(dst) = pc[0];
      *reinterpret_cast<pm_u32x2*>(dst + PR_FRAG) = pc[1];
      if constexpr (NET == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gq[r]), srd, lane_st + r * 64, pr_uni(so_st + ot * 1024), 0);)
      }
    };
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      epi(qc, 4 * q + wid);
    });
    if (xw) epi(std::integral_constant<int, PR_SLOTS>{}, PR_XT);
  };

  // transposed hidden->hidden layer + tail of network NET; leaves the summed tail tile in `o`
  // (`arrived`: called behind the MFMA loop -- where the step's staged inputs are parked)
  auto hidden_adjoint_and_tail = [&](auto netc, const unsigned abw, int so_st, f32x4& o, auto arrived) {
    constexpr int NET = decltype(netc)::value;
    const bool xw = NET == 0 ? xw_pol : xw_dyn;
    f32x4 acc[PR_SLOTS][2], accx[2];
    f32x4 mfv[PR_SLOTS + 1], hwv[2][2];
    if (xw) pr_hidden_layer<NET, F16, true>(W, smem, smem + PRB_LDS_XT(NET), lane, acc, accx);
    else pr_hidden_layer<NET, F16, false>(W, smem, nullptr, lane, acc, accx);
    arrived();
    pr_for<PR_SLOTS + 1>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      mfv[q] = *reinterpret_cast<const f32x4*>(lut + ((NET * 2 + 0) * 16 + ((abw >> (4 * q)) & 15u)) * 4);
    });
    // (the tail's weights: requested here, they land behind the epilogues)
    const float* hw = smem + PRB_LDS_HEAD(NET) + (size_t)wid * (4 * PR_FRAG) + lane * 4;
#pragma unroll
    for (int blkk = 0; blkk < 2; ++blkk) {
      hwv[blkk][0] = *reinterpret_cast<const f32x4*>(hw + (blkk * 2 + 0) * PR_FRAG);
      hwv[blkk][1] = *reinterpret_cast<const f32x4*>(hw + (blkk * 2 + 1) * PR_FRAG);
    }
    pm_u32x2 hi[PR_SLOTS + 1], lo[PR_SLOTS + 1];
    hi[PR_SLOTS] = lo[PR_SLOTS] = pm_u32x2{0u, 0u};
    auto epi = [&](auto qc, f32x4 v, int ot) {
      constexpr int q = decltype(qc)::value;
      const f32x4 gq = v * mfv[q];
      pm_u32x2 pc[2];
      pm_split4<2, false>(gq, pc);
      hi[q] = pc[0];
      lo[q] = pc[1];
      if constexpr (NET == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          PR_EXP_STASH(__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gq[r]), srd, lane_st + r * 64, pr_uni(so_st + ot * 1024), 0);)
      }
    };
    pr_for<PR_SLOTS>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      epi(qc, pr_tile_value<F16>(acc[q]), 4 * q + wid);
    });
    if (xw) epi(std::integral_constant<int, PR_SLOTS>{}, pr_tile_value<F16>(accx), PR_XT);
    f32x4 c0, c1, bh[2], bl[2];
#pragma unroll
    for (int blkk = 0; blkk < 2; ++blkk) {
      const pm_u32x2 ha = hi[2 * blkk], hb = hi[2 * blkk + 1], la = lo[2 * blkk], lb = lo[2 * blkk + 1];
      bh[blkk] = __builtin_bit_cast(f32x4, (pr_u32x4){ha[0], ha[1], hb[0], hb[1]});
      bl[blkk] = __builtin_bit_cast(f32x4, (pr_u32x4){la[0], la[1], lb[0], lb[1]});
    }
    pr_mfma_open();
    pr_mfma0<F16, false, true>(c0, hwv[0][1], bh[0]);
    pr_mfma0<F16, false, true>(c1, hwv[0][0], bl[0]);
    pr_mfma<F16, false, true>(c0, hwv[1][1], bh[1]);
    pr_mfma<F16, false, true>(c1, hwv[0][0], bh[0]);
    pr_mfma<F16, false, true>(c1, hwv[1][0], bl[1]);
    pr_mfma<F16, false, true>(c1, hwv[1][0], bh[1]);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0), "+v"(c1));
    float* part = smem + PRB_LDS_PART + lane * 4;
    *reinterpret_cast<f32x4*>(part + wid * PR_FRAG) = c0 + c1;
    pr_barrier();
    o = *reinterpret_cast<const f32x4*>(part);
#pragma unroll
    for (int w = 1; w < PR_NW; ++w) o += *reinterpret_cast<const f32x4*>(part + w * PR_FRAG);
  };

  // moment matching: wave 0's inputs of the adjoint chain, requested a step ahead (pmbrl_reg_mm.h)
  const int mm_gi = MMD ? wg / A.mm.parts : 0, mm_me = MMD ? wg - mm_gi * A.mm.parts : 0, mm_g0 = mm_gi * (MMD ? A.mm.M : 0);
  double* const mm_bop = reinterpret_cast<double*>(smem + PRB_LDS_MMB);
  double* const mm_yop = reinterpret_cast<double*>(smem + PRB_LDS_MMY);
  if constexpr (MMD != 0) {
    if (wid == 1) pr_mm_bwd_prep_noise<MMDc>(A.mm, T1 - 1, mm_gi, mm_g0, row0, nvalid, lane, mm_bop + ((T1 - 1) & 1) * PR_MM_BOP_DOUBLES);
    if (wid == 2) pr_mm_bwd_prep_factor<MMDc>(A.mm, B, T1 - 1, mm_gi, row0, nvalid, lane, mm_yop + ((T1 - 1) & 1) * PR_MM_YOP_DOUBLES(MMDc));
  }
  // prologue: the last step's inputs and activity words
  unsigned abc[4], abn[4];
  stage_load();
  ab_load(abc);
  stage_landed(std::integral_constant<int, 0>{}, sv);
  ab_landed(std::integral_constant<int, 0>{}, abc);
  stage_park();
  __syncthreads();
  for (int t = T1 - 1; t >= T0; --t) {
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 0] = (long long)__builtin_readcyclecounter();
    StepIn In;
    read_in(In);
    if constexpr (MMD != 0) {
      // ---- adjoint of the moment matching that produced x_{t+1}: dL/dx_{t+1} -> dL/dx~ (wave 0; the others wait).  The
      // horizon's last step carries no gradient yet unless one was handed in (every part of a group skips it alike)
      const bool do_chain = t < T1 - 1 || A.gx_in;
      int ln = lane;      // (laundered per step, the arguments re-read per step: see the forward sweep)
      asm volatile("" : "+v"(ln));
      pr_mm_kptr Qp = pr_mm_args();
      asm volatile("" : "+s"(Qp));
      const RegMM Q = pr_mm_load(Qp);      // (ONE batch of scalar loads, one wait -- through the pointer every use was a round trip of its own)
      if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 6] = (long long)__builtin_readcyclecounter();
      if (wid == 1 && t > T0) {
        // (idle otherwise) what step t - 1's chain needs that does not wait for the recursion: its noise operand ...
        pr_mm_bwd_prep_noise<MMDc>(Q, t - 1, mm_gi, mm_g0, row0, nvalid, ln, mm_bop + ((t - 1) & 1) * PR_MM_BOP_DOUBLES);
        if (PROF && wg == 0 && lane == 0) A.prof[(size_t)t * 32 + 13] = (long long)__builtin_readcyclecounter();
      } else if (wid == 2 && t > T0) {
        // ... and, from the forward sweep's factor, Y1 = L^-1 Delta^T and the factor in operand layout
        pr_mm_bwd_prep_factor<MMDc>(Q, B, t - 1, mm_gi, row0, nvalid, ln, mm_yop + ((t - 1) & 1) * PR_MM_YOP_DOUBLES(MMDc));
        if (PROF && wg == 0 && lane == 0) A.prof[(size_t)t * 32 + 14] = (long long)__builtin_readcyclecounter();
      }
      if (do_chain) {
        double* const xh = reinterpret_cast<double*>(smem + PRB_LDS_XH);
        volatile unsigned* const xtag = reinterpret_cast<volatile unsigned*>(smem + PRB_LDS_XTAG);
        if constexpr (TREE) {
          // (the two-level exchange: the other waves poll a quarter of each level's slots -- see the forward sweep)
          if (wid != 0) {
            const bool okh = pm_xch_tree_help<PR_MM_NVX(MMDc), 4>(Q.xch, Q.nwg, mm_gi * Q.parts, Q.parts, Q.fan, mm_me, Q.tag0 + (unsigned)(T1 - t), wid, xh,
                                                    xtag, ln);
            if (!okh && lane == 0 && A.status) atomicMax(A.status, 1);
          }
        }
        if (wid == 0) {
          float go[(MMDc + 3) / 4];
          const bool ok = pr_mm_bwd_chain<MMDc, TREE>(Q, Q.tag0 + (unsigned)(T1 - t), mm_me, mm_gi * Q.parts, nvalid, ln, gx,
                                                 mm_bop + (t & 1) * PR_MM_BOP_DOUBLES, mm_yop + (t & 1) * PR_MM_YOP_DOUBLES(MMDc), go,
                                                 xh, xtag, (PROF && wg <= 1) ? A.prof + (size_t)t * 32 + 8 * wg : nullptr);
          if (!ok && lane == 0 && A.status) atomicMax(A.status, 1);
#pragma unroll
          for (int rr = 0; rr < (MMDc + 3) / 4; ++rr) smem[PRB_LDS_XB + row * 8 + g + 4 * rr] = go[rr];
        }
        pr_barrier();
        const f32x2 gm = *reinterpret_cast<const f32x2*>(smem + PRB_LDS_XB + row * 8 + 2 * g);
#pragma unroll
        for (int s = 0; s < 2; ++s) gx[s] = ok_x[s] ? gm[s] : 0.f;
        if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 7] = (long long)__builtin_readcyclecounter();
      }
    }
    // the inputs of step t - 1: on their way while this step computes (the last step of the range asks for its own once
    // more: the count of operations in flight is the same at every step)
    if (t > T0) {
#pragma unroll
      for (int k = 0; k < 2; ++k) sp[k] -= sstep[k];
      so_abp -= abp_step;
    }
    stage_load();
    ab_load(abn);
    // ---- dynamics model: gradient of the sampled state, through the head
    float gxn[2], gm[2], gl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float gg = gx[s] + In.gr * In.jx[s];
      gxn[s] = gg;
      gm[s] = gg * c_sy[s];
      gl[s] = gg * In.td[s];
    }
    head_adjoint(std::integral_constant<int, 1>{}, gm, gl, abc[0], 0);
    pr_barrier();
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 1] = (long long)__builtin_readcyclecounter();
    f32x4 o;
    hidden_adjoint_and_tail(std::integral_constant<int, 1>{}, abc[1], 0, o, []() {});
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 2] = (long long)__builtin_readcyclecounter();
    // ---- gradient of [x | a] (normalised inputs): state part -> gxn, action part -> through the squashing
    float pm_[2], pl_[2];
    {
      const f32x2 ja = *reinterpret_cast<const f32x2*>(st_row + 2 + 2 * g), tp = *reinterpret_cast<const f32x2*>(st_row + 10 + 2 * g),
                  ac = *reinterpret_cast<const f32x2*>(st_row + 18 + 2 * g);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float tail = o[2 * s] * c_isx[s];
        gxn[s] += ok_x[s] ? tail : 0.f;
        const float ga = In.gr * (ok_a[s] ? ja[s] : 0.f) + tail;
        const float th = ((ok_a[s] ? ac[s] : c_pbi[s]) - c_pbi[s]) * pr_rcp(c_psc[s]);
        const float gu = ok_a[s] ? ga * c_psc[s] * (1.f - th * th) : 0.f;
        pm_[s] = gu;
        pl_[s] = ok_a[s] ? gu * tp[s] : 0.f;
      }
    }
    // head-gradient stash of the policy ([16][16] block: row j = d mean_j, row U + j = d log-std_j, zero below)
    if (wid == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = 2 * g + s - D;
        if (j >= 0 && j < U) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pm_[s]), srd, (j * 16 + row) * 4, pr_uni(so_g2), 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pl_[s]), srd, ((U + j) * 16 + row) * 4, pr_uni(so_g2), 0);
        }
      }
    } else if (wid == 3) {
      for (int k = 2 * U * 16 + lane; k < 256; k += 64) __builtin_amdgcn_raw_buffer_store_b32(0u, srd, k * 4, pr_uni(so_g2), 0);
    }
    head_adjoint(std::integral_constant<int, 0>{}, pm_, pl_, abc[2], so_g1);
    pr_barrier();
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 3] = (long long)__builtin_readcyclecounter();
    // (between the step's third and fourth barrier: every lane has read this step's inputs, none reads the next one's
    //  before the fourth.  Behind the two staged loads: the 4 activity-word loads and at least the 12 stash stores of the
    //  policy's head adjoint)
    hidden_adjoint_and_tail(std::integral_constant<int, 0>{}, abc[3], so_g0, o, [&]() {
#ifdef PR_EXP_NOSTASH
      stage_landed(std::integral_constant<int, 0>{}, sv);
#else
      stage_landed(std::integral_constant<int, 16>{}, sv);
#endif
      stage_park();
    });
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 4] = (long long)__builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < 2; ++s) gx[s] = ok_x[s] ? gxn[s] + o[2 * s] : 0.f;
    so_g0 -= st_step; so_g1 -= st_step; so_g2 -= st_step2;
    // the next step's activity words: requested a step ago; behind them this step issued at least the 24 stash stores of
    // the policy's two hidden layers (3 tiles x 4 registers each, every wave)
#ifdef PR_EXP_NOSTASH
    ab_landed(std::integral_constant<int, 0>{}, abn);
#else
    ab_landed(std::integral_constant<int, 24>{}, abn);
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) abc[k] = abn[k];
    if (PROF && wg == 0 && tid == 0) A.prof[(size_t)t * 32 + 5] = (long long)__builtin_readcyclecounter();
  }
  if (PROF && wg == 0 && tid == 0) {
    A.prof[31] = (long long)__builtin_readcyclecounter();
    A.prof[29] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (wid == 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (A.grad_x0 && T0 == 0 && ok_x[s]) *(gf32*)((PM_GLOBAL_ char*)A.grad_x0 + so_x[s]) = gx[s];
      if (A.gx_out && ok_x[s]) *(gf32*)((PM_GLOBAL_ char*)A.gx_out + so_x[s]) = gx[s];
    }
  }
}
