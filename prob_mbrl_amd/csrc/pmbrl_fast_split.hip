// Latency-optimised sweep kernels on split bf16 / fp16 operands (pmbrl_split.h): instantiations, attribute
// setup and launch dispatch.  Compiled once per precision: -DPM_SPLIT_PR=1 (three bf16 pieces forward) and
// -DPM_SPLIT_PR=2 (two fp16 pieces forward); the adjoint uses two bf16 pieces in both.
#include <cstdio>
#include <cstdlib>
#include "pmbrl_host.h"
#include "pmbrl_mm.h"
#include "pmbrl_rollout.h"
#include "pmbrl_fast.h"

#ifndef PM_SPLIT_PR
#error "compile with -DPM_SPLIT_PR=1 or 2"
#endif

template <int RT, int CA, int CB, int PR>
static int set_attr_split(size_t lds) {
  if constexpr (RT < 4) {   // (64-row workgroups: the shape-specialised instances below only)
  const void* fns[] = {
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_LEAN, PfShapeAny, PR>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_LEAN, PfShapeAny, PR>),
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_EXT, PfShapeAny, PR>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_EXT, PfShapeAny, PR>),
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_MM, PfShapeAny, PR>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_MM, PfShapeAny, PR>)};
  for (const void* f : fns)
    HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if constexpr (RT == 1) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_MMG, PfShapeAny, PR>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_MMG, PfShapeAny, PR>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
#define PM_FAST_SHAPED(RTV, CAV, CBV, VARV, DV, UV, LDV, NLV, NTV)                                          \
  if (RT == RTV && CA == CAV && CB == CBV) {                                                             \
    HIPCHK(hipFuncSetAttribute(                                                                          \
        reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>, PR>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                          \
    HIPCHK(hipFuncSetAttribute(                                                                          \
        reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>, PR>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                          \
  }
  PM_SPLIT_SHAPED_CASES((PR == 2 ? 240 : 360))
#undef PM_FAST_SHAPED
  if constexpr (PR == 2 && RT == 1 && CA == 4 && CB == 3) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd_fast<1, 4, 3, PF_VAR_MM, PfShapeTree<4, 1, 240, 3, 13>, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd_fast<1, 4, 3, PF_VAR_MM, PfShapeTree<4, 1, 240, 3, 13>, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  return 0;
}

// split-bf16 precision (pmbrl_split.h): same variants and shape specialisation, PR = 1
template <int RT, int CA, int CB, int PR>
static void launch_split(const pmbrl_plan* p, const RolloutArgs& A0, hipStream_t s, bool fwd) {
  const int var = fast_variant(RT, A0);
  const dim3 g(A0.launch_wg > 0 ? A0.launch_wg : p->nwg), b(PF_NT);
  RolloutArgs A = A0;
  // Register-resident first tiles (pmbrl_fast.h, resident_tile_s): the 16-row plain variants keep output
  // tile `wave` of the sweep's first streamed layer in registers when a tile is exactly one stage pair and
  // two pieces wide; the stream table then starts at tile 8 of that layer.
  {
    StreamDesc& sd = fwd ? A.sd_fwd : A.sd_bwd;
    const int np = (fwd && PR != 2) ? 3 : 2;
    if (RT == 1 && CA + CB == 7 && np == 2 && (var == PF_VAR_LEAN || var == PF_VAR_EXT) && sd.n >= 1 &&
        sd.n_kb[0] == 7 * np && sd.n_ot[0] > 8) {
      A.res_tiles = 8;
      A.res_w = sd.wf[0];
      sd.wf[0] += (size_t)8 * sd.n_kb[0] * 256;
      sd.n_ot[0] -= 8;
    }
    // LDS-resident tiles (lds_tile_s): tiles 0..7 of the SECOND streamed layer, where a shape-specialised 16-row
    // instance will run (plain variants and in-kernel moment matching; the plan reserved the LDS)
    if (PR == 2 && RT == 1 && CA + CB == 7 && np == 2 && (var == PF_VAR_LEAN || var == PF_VAR_EXT || var == PF_VAR_MM)) {
      bool shaped = false;
#define PM_FAST_SHAPED(RTV, CAV, CBV, VARV, DV, UV, LDV, NLV, NTV)                                             \
  if (RT == RTV && CA == CAV && CB == CBV && var == VARV && A.D == DV && A.U == UV && A.LD == LDV &&        \
      A.pol.nl == NLV && A.dyn.nl == NLV && hidden_tiles(A) == NTV && NLV == 3 && NTV > 8)                   \
    shaped = true;
      PM_SPLIT_SHAPED_CASES((PR == 2 ? 240 : 360))
#undef PM_FAST_SHAPED
      if (shaped && !(A.flags & PMBRL_FLAG_NO_SHAPED)) {
        if (p->wlds_off > 0 && sd.n == 2 && sd.n_kb[1] == 7 * np && sd.n_ot[1] > 8) {
          A.lds_w = sd.wf[1];
          A.wlds_off = p->wlds_off;
          A.lds_last_lanes = p->lds_last_lanes;
          sd.wf[1] += (size_t)8 * sd.n_kb[1] * 256;
          sd.n_ot[1] -= 8;
        } else {
          A.flags |= PMBRL_FLAG_NO_SHAPED;      // (those instances count on the LDS tiles: the generic one then)
        }
      }
    }
  }
#define PM_FAST_SHAPED(RTV, CAV, CBV, VARV, DV, UV, LDV, NLV, NTV)                                             \
  if (RT == RTV && CA == CAV && CB == CBV && var == VARV && A.D == DV && A.U == UV && A.LD == LDV &&        \
      A.pol.nl == NLV && A.dyn.nl == NLV && hidden_tiles(A) == NTV) {                                        \
    if (fwd)                                                                                                \
      hipLaunchKernelGGL((pm_rollout_fwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>, PR>), g, b,    \
                         p->lds_bytes, s, A);                                                               \
    else                                                                                                    \
      hipLaunchKernelGGL((pm_rollout_bwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>, PR>), g, b,    \
                         p->lds_bytes, s, A);                                                               \
    return;                                                                                                 \
  }
  // groups split over more than 8 workgroups (the plan chose that only where this instance exists): two-level exchange
  if constexpr (PR == 2 && RT == 1 && CA == 4 && CB == 3) {
    if (A.mm_mode == 1 && A.mm_parts > 8) {
      if (var != PF_VAR_MM || A.D != 4 || A.U != 1 || A.LD != 240 || A.pol.nl != 3 || A.dyn.nl != 3 || hidden_tiles(A) != 13 ||
          (A.flags & PMBRL_FLAG_NO_SHAPED) || !A.xch) {
        fprintf(stderr, "pmbrl: no instance for a moment-matching group split over %d workgroups on this shape\n", A.mm_parts);
        abort();
      }
      if (fwd)
        hipLaunchKernelGGL((pm_rollout_fwd_fast<1, 4, 3, PF_VAR_MM, PfShapeTree<4, 1, 240, 3, 13>, 2>), g, b, p->lds_bytes, s, A);
      else
        hipLaunchKernelGGL((pm_rollout_bwd_fast<1, 4, 3, PF_VAR_MM, PfShapeTree<4, 1, 240, 3, 13>, 2>), g, b, p->lds_bytes, s, A);
      return;
    }
  }
  if (!(A.flags & PMBRL_FLAG_NO_SHAPED)) {
    PM_SPLIT_SHAPED_CASES((PR == 2 ? 240 : 360))
  }
#undef PM_FAST_SHAPED
  if constexpr (RT >= 4) {
    // unreachable: pmbrl_plan_create selects 64-row split workgroups for the specialised shape only
    fprintf(stderr, "pmbrl: no 64-row split-precision instance for this shape\n");
    abort();
  } else {
#define PM_LAUNCH_VAR(K, V) hipLaunchKernelGGL((K<RT, CA, CB, V, PfShapeAny, PR>), g, b, p->lds_bytes, s, A)
  if constexpr (RT == 1) {
    if (var == PF_VAR_MMG) {
      if (fwd) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_MMG);
      else PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_MMG);
      return;
    }
  }
  if (fwd) {
    if (var == PF_VAR_MM) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_MM);
    else if (var == PF_VAR_EXT) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_EXT);
    else PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_LEAN);
  } else {
    if (var == PF_VAR_MM) PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_MM);
    else if (var == PF_VAR_EXT) PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_EXT);
    else PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_LEAN);
  }
#undef PM_LAUNCH_VAR
  }
}


#define PM_CAT2(a, b, c) a##b##c
#define PM_CAT(a, b, c) PM_CAT2(a, b, c)
int PM_CAT(pm_fast_split, PM_SPLIT_PR, _set_attr)(const pmbrl_plan* p) {
  int rc2 = 0;
#define PM_SPLIT_CASE(RTV, CAV, CBV) \
  if (p->RT == RTV && p->CA == CAV && p->CB == CBV) rc2 = set_attr_split<RTV, CAV, CBV, PM_SPLIT_PR>(p->lds_bytes);
  PM_SPLIT_CASES
#undef PM_SPLIT_CASE
  return rc2;
}
void PM_CAT(pm_fast_split, PM_SPLIT_PR, _launch)(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd) {
#define PM_SPLIT_CASE(RTV, CAV, CBV) \
  if (p->RT == RTV && p->CA == CAV && p->CB == CBV) return launch_split<RTV, CAV, CBV, PM_SPLIT_PR>(p, A, s, fwd);
  PM_SPLIT_CASES
#undef PM_SPLIT_CASE
}

template <int CA, int CB>
static int mmg_occ_split(size_t lds) {
  int a = 0, b = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, pm_rollout_fwd_fast<1, CA, CB, PF_VAR_MMG, PfShapeAny, PM_SPLIT_PR>, PF_NT, lds) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, pm_rollout_bwd_fast<1, CA, CB, PF_VAR_MMG, PfShapeAny, PM_SPLIT_PR>, PF_NT, lds) != hipSuccess) return 0;
  return a < b ? a : b;
}
int PM_CAT(pm_fast_split, PM_SPLIT_PR, _mmg_blocks_per_cu)(const pmbrl_plan* p) {
  if (p->RT != 1) return 0;
#define PM_SPLIT_CASE(RTV, CAV, CBV) \
  if (RTV == 1 && p->CA == CAV && p->CB == CBV) return mmg_occ_split<CAV, CBV>(p->lds_bytes);
  PM_SPLIT_CASES
#undef PM_SPLIT_CASE
  return 0;
}
