// Latency-optimised sweep kernels, exact fp32 MFMA: instantiations, attribute setup and launch dispatch.
#include "pmbrl_host.h"
#include "pmbrl_mm.h"
#include "pmbrl_rollout.h"
#include "pmbrl_fast.h"

template <int RT, int CA, int CB>
static int set_attr_fast(size_t lds) {
  const void* fns[] = {
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_LEAN>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_LEAN>),
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_EXT>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_EXT>),
      reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_MM>),
      reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_MM>)};
  for (const void* f : fns)
    HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if constexpr (RT == 1) {     // the one-launch form of mm_mode 3 exists for 16-row workgroups
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RT, CA, CB, PF_VAR_MMG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RT, CA, CB, PF_VAR_MMG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
#define PM_FAST_SHAPED(RTV, CAV, CBV, VARV, DV, UV, LDV, NLV, NTV)                                          \
  if (RT == RTV && CA == CAV && CB == CBV) {                                                             \
    HIPCHK(hipFuncSetAttribute(                                                                          \
        reinterpret_cast<const void*>(&pm_rollout_fwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                          \
    HIPCHK(hipFuncSetAttribute(                                                                          \
        reinterpret_cast<const void*>(&pm_rollout_bwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                          \
  }
  PM_FAST_SHAPED_CASES
#undef PM_FAST_SHAPED
  return 0;
}

template <int RT, int CA, int CB>
static void launch_fast(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd) {
  const int var = fast_variant(RT, A);
  const bool mm = var == PF_VAR_MM, ext = var == PF_VAR_EXT, mmg = var == PF_VAR_MMG;
  const dim3 g(A.launch_wg > 0 ? A.launch_wg : p->nwg), b(PF_NT);
  // a shape-specialised instantiation if there is one for this plan
#define PM_FAST_SHAPED(RTV, CAV, CBV, VARV, DV, UV, LDV, NLV, NTV)                                             \
  if (RT == RTV && CA == CAV && CB == CBV && var == VARV && A.D == DV && A.U == UV && A.LD == LDV &&        \
      A.pol.nl == NLV && A.dyn.nl == NLV && hidden_tiles(A) == NTV) {                                                                \
    if (fwd)                                                                                                \
      hipLaunchKernelGGL((pm_rollout_fwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>>), g, b,       \
                         p->lds_bytes, s, A);                                                               \
    else                                                                                                    \
      hipLaunchKernelGGL((pm_rollout_bwd_fast<RTV, CAV, CBV, VARV, PfShape<DV, UV, LDV, NLV, NTV>>), g, b,       \
                         p->lds_bytes, s, A);                                                               \
    return;                                                                                                 \
  }
  if (!(A.flags & PMBRL_FLAG_NO_SHAPED)) {
    PM_FAST_SHAPED_CASES
  }
#undef PM_FAST_SHAPED
#define PM_LAUNCH_VAR(K, V) hipLaunchKernelGGL((K<RT, CA, CB, V>), g, b, p->lds_bytes, s, A)
  if constexpr (RT == 1) {
    if (mmg) {
      if (fwd) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_MMG);
      else PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_MMG);
      return;
    }
  }
  if (fwd) {
    if (mm) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_MM);
    else if (ext) PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_EXT);
    else PM_LAUNCH_VAR(pm_rollout_fwd_fast, PF_VAR_LEAN);
  } else {
    if (mm) PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_MM);
    else if (ext) PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_EXT);
    else PM_LAUNCH_VAR(pm_rollout_bwd_fast, PF_VAR_LEAN);
  }
#undef PM_LAUNCH_VAR
}

int pm_fast_f32_set_attr(const pmbrl_plan* p) {
  int rc2 = 0;
#define PM_FAST_CASE(RTV, CAV, CBV) \
  if (p->RT == RTV && p->CA == CAV && p->CB == CBV) rc2 = set_attr_fast<RTV, CAV, CBV>(p->lds_bytes);
  PM_FAST_CASES
#undef PM_FAST_CASE
  return rc2;
}
void pm_fast_f32_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd) {
#define PM_FAST_CASE(RTV, CAV, CBV) \
  if (p->RT == RTV && p->CA == CAV && p->CB == CBV) return launch_fast<RTV, CAV, CBV>(p, A, s, fwd);
  PM_FAST_CASES
#undef PM_FAST_CASE
}

template <int CA, int CB>
static int mmg_occ_f32(size_t lds) {
  int a = 0, b = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, pm_rollout_fwd_fast<1, CA, CB, PF_VAR_MMG>, PF_NT, lds) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, pm_rollout_bwd_fast<1, CA, CB, PF_VAR_MMG>, PF_NT, lds) != hipSuccess) return 0;
  return a < b ? a : b;
}
int pm_fast_f32_mmg_blocks_per_cu(const pmbrl_plan* p) {
  if (p->RT != 1) return 0;
#define PM_FAST_CASE(RTV, CAV, CBV) \
  if (RTV == 1 && p->CA == CAV && p->CB == CBV) return mmg_occ_f32<CAV, CBV>(p->lds_bytes);
  PM_FAST_CASES
#undef PM_FAST_CASE
  return 0;
}
