// General kernel family on split operands (pmbrl_gsplit.h: two fp16 pieces forward, two bf16 pieces in the
// adjoint): instantiations, attribute setup and launch dispatch.
#include "pmbrl_host.h"
#include "pmbrl_mm.h"
#include "pmbrl_rollout.h"

template <int RT>
static int set_attr_gs(size_t lds) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd<RT, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd<RT, 2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return 0;
}
int pm_general_split_set_attr(const pmbrl_plan* p) {
  if (p->inplace == 2) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd<4, 2, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd<4, 2, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    // (<4, 2, 3>: the wide layers with every 512-wide stash pre-split for the dW GEMM -- pmbrl_dw.h, pm_dw_wide_pre_kernel)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd<4, 2, 3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd<4, 2, 3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    return 0;
  }
  if (p->inplace) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_fwd<4, 2, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_rollout_bwd<4, 2, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    return 0;
  }
  switch (p->RT) {
    case 1: return set_attr_gs<1>(p->lds_bytes);
    case 2: return set_attr_gs<2>(p->lds_bytes);
    default: return set_attr_gs<4>(p->lds_bytes);
  }
}
template <int RT>
static void launch_gs(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd) {
  if (fwd) hipLaunchKernelGGL((pm_rollout_fwd<RT, 2>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
  else hipLaunchKernelGGL((pm_rollout_bwd<RT, 2>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
}
void pm_general_split_launch(const pmbrl_plan* p, const RolloutArgs& A, hipStream_t s, bool fwd) {
  if (p->inplace == 2 && p->dw_pre_mask) {
    if (fwd) hipLaunchKernelGGL((pm_rollout_fwd<4, 2, 3>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    else hipLaunchKernelGGL((pm_rollout_bwd<4, 2, 3>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    return;
  }
  if (p->inplace == 2) {
    if (fwd) hipLaunchKernelGGL((pm_rollout_fwd<4, 2, 2>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    else hipLaunchKernelGGL((pm_rollout_bwd<4, 2, 2>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    return;
  }
  if (p->inplace) {
    if (fwd) hipLaunchKernelGGL((pm_rollout_fwd<4, 2, 1>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    else hipLaunchKernelGGL((pm_rollout_bwd<4, 2, 1>), dim3(p->nwg), dim3(PM_NT), p->lds_bytes, s, A);
    return;
  }
  switch (p->RT) {
    case 1: launch_gs<1>(p, A, s, fwd); break;
    case 2: launch_gs<2>(p, A, s, fwd); break;
    default: launch_gs<4>(p, A, s, fwd); break;
  }
}
