// Register-resident sweep kernels (pmbrl_reg.h): weight packer, attribute setup and launch dispatch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pmbrl_host.h"
#include "pmbrl_reg.h"

// does this plan's shape fit the family?  (decided once, at plan creation)
// the shape alone (networks, widths, arithmetic): what the plan asks BEFORE it settles rows per workgroup and the split of
// the moment-matching groups, so that it can lay the groups out for this family
bool pm_reg_shape_ok(const pmbrl_plan* p, int prec1) {
  const pmbrl_config& c = p->cfg;
  if (const char* e = getenv("PMBRL_REG")) {
    if (atoi(e) == 0) return false;
  }
  if (!p->fast || prec1 != PMBRL_PREC_SPLIT_F16 || (c.flags & PMBRL_FLAG_NO_SHAPED)) return false;
  if (p->pol.nl != 3 || p->dyn.nl != 3) return false;
  const int hid = p->pol.dim[1];
  if (p->pol.dim[2] != hid || p->dyn.dim[1] != hid || p->dyn.dim[2] != hid) return false;
  if ((hid + 15) / 16 != PR_NT) return false;
  if (c.D + c.U > 8 || p->pol.dim[0] != c.D || p->dyn.dim[0] != c.D + c.U) return false;
  if (p->pol.dim[3] != 2 * c.U || p->dyn.dim[3] != 2 * c.D) return false;
  return true;
}
// ... and what its moment-matching instances take of a configuration (state width, flags, switches)
bool pm_reg_mm_shape_ok(const pmbrl_plan* p, int prec1) {
  const pmbrl_config& c = p->cfg;
  if (!pm_reg_shape_ok(p, prec1)) return false;
  if (!(c.flags & PMBRL_FLAG_MM_STATES) || (c.flags & PMBRL_FLAG_INFER_NS) || c.D < 4 || c.D > 6) return false;
  if (getenv("PMBRL_MM_XCH") && atoi(getenv("PMBRL_MM_XCH")) == 0) return false;
  if (getenv("PMBRL_REG_MM") && atoi(getenv("PMBRL_REG_MM")) == 0) return false;
  return true;
}

bool pm_reg_plan_ok(const pmbrl_plan* p) {
  const pmbrl_config& c = p->cfg;
  if (!pm_reg_shape_ok(p, p->prec) || p->RT != 1) return false;
  if (p->mm_mode != 0) {
    // moment matching inside the sweep (pmbrl_reg_mm.h): the states of groups that are ONE workgroup of <= 16 rows or
    // split over 2..8 of them with the statistics exchange; state widths 4..6.  (Moment matching of the rewards alone
    // leaves the sweep plain but lays the rows out by groups: the latency-optimised family's.)
    if (p->mm_mode != 1 || !(c.flags & PMBRL_FLAG_MM_STATES) || (c.flags & PMBRL_FLAG_INFER_NS)) return false;
    // (more than 8 parts: only with the two-level exchange -- mm_fan, ONE group over the batch)
    if (!pm_reg_mm_shape_ok(p, p->prec) || (p->mm_parts > 8) != (p->mm_fan != 0) || p->rows_per_wg > 16 || c.H >= 4096) return false;
    // (every wave of a workgroup polls four slots of a level: at most 16 members per collector, at most 16 collectors)
    if (p->mm_fan && (p->mm_fan > 16 || (p->mm_parts + p->mm_fan - 1) / p->mm_fan > 16 || c.D != 4)) return false;
    if (getenv("PMBRL_REG_TREE") && atoi(getenv("PMBRL_REG_TREE")) == 0 && p->mm_fan) return false;
    if (p->mm_parts <= 1 && p->rows_per_wg != p->M) return false;      // (several whole groups per workgroup: not here)
  }
  return true;
}

// one group over the batch with the collectors laid out per XCD (pmbrl.hip): 2 fan parts per XCD
bool pm_reg_tree_xcd(const pmbrl_plan* p) { return p->mm_fan && p->G == 1 && 16 * p->mm_fan >= p->mm_parts && p->mm_fan <= 16; }

size_t pm_reg_pack_bytes() { return (size_t)PR_PACK_FLOATS * sizeof(float); }

// state width of the moment-matching instance this plan's sweeps run on (0: the plain instance)
int pm_reg_mm_width(const pmbrl_plan* p) { return (p->reg && p->mm_mode == 1) ? p->cfg.D : 0; }

template <int MMD, bool TREE = false>
static int reg_set_attr_mm() {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_reg_fwd_kernel<false, MMD, TREE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PR_LDS_FLOATS * sizeof(float))));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_reg_fwd_kernel<true, MMD, TREE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PR_LDS_FLOATS * sizeof(float))));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_reg_bwd_kernel<false, MMD, TREE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PRB_LDS_FLOATS * sizeof(float))));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pm_reg_bwd_kernel<true, MMD, TREE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PRB_LDS_FLOATS * sizeof(float))));
  return 0;
}
int pm_reg_set_attr(const pmbrl_plan* p) {
  if (p->reg_mm == 4 && p->mm_fan) return reg_set_attr_mm<4, true>();
  switch (p->reg_mm) {
    case 4: return reg_set_attr_mm<4>();
    case 5: return reg_set_attr_mm<5>();
    case 6: return reg_set_attr_mm<6>();
    default: return reg_set_attr_mm<0>();
  }
}

static void reg_net(const pmbrl_plan* p, int net, const NetPlan& n, const NetDev& d, RegNet& r) {
  for (int l = 0; l < 3; ++l) {
    r.w_off[l] = (int)n.w_off[l];
    r.b_off[l] = (int)n.b_off[l];
  }
  r.n_in = n.dim[0];
  r.n_out = n.dim[3];
  for (int l = 0; l < 2; ++l) {
    r.mask[l] = d.mask[l];
    r.abits[l] = (unsigned)p->off_reg_ab[net][l];
    r.inv_keep[l] = d.inv_keep[l];
  }
}

static void reg_args(const pmbrl_plan* p, char* ws, const RolloutArgs& A, const float* packed, const float* pol_params,
                     const float* dyn_params, RegArgs& R) {
  memset(&R, 0, sizeof(R));
  R.B = A.B; R.H = A.H; R.D = A.D; R.U = A.U; R.nwg = A.nwg; R.hid = p->pol.dim[1];
  R.mls_pol = A.mls_pol; R.mls_dyn = A.mls_dyn;
  reg_net(p, 0, p->pol, A.pol, R.pol);
  reg_net(p, 1, p->dyn, A.dyn, R.dyn);
  R.packed = packed;
  R.pol_params = pol_params; R.dyn_params = dyn_params;
  R.x0 = A.x0; R.mx = A.mx; R.iSx = A.iSx; R.my = A.my; R.Sy = A.Sy; R.pscale = A.pscale; R.pbias = A.pbias;
  R.zpol = A.zpol; R.zdyn = A.zdyn;
  R.states = A.states; R.actions = A.actions;
  R.ws = ws; R.Tp = (unsigned)p->off_Tp; R.Td = (unsigned)p->off_Td; R.Jx = (unsigned)p->off_Jx; R.Ja = (unsigned)p->off_Ja;
  for (int l = 0; l < 3; ++l) { R.actT[l] = (unsigned)p->off_actT[l]; R.gT[l] = (unsigned)p->off_gT[l]; }
  R.status = A.status;
  R.wflag = A.wflag; R.wgen = A.wgen;
  R.grad_rewards = A.grad_rewards; R.grad_x0 = A.grad_x0; R.nvalid = A.nvalid;
  R.t0 = A.t0; R.t1 = A.t1;
  R.gx_in = A.gx_from_carry ? A.gx_carry : nullptr;
  R.gx_out = (A.gx_carry && A.t0 > 0) ? A.gx_carry : nullptr;
  R.prof = A.prof;
  memset(&R.mm, 0, sizeof(R.mm));
  if (p->reg_mm) {
    R.mm.on = 1;
    R.mm.M = p->M; R.mm.parts = p->mm_parts; R.mm.rpw = p->rows_per_wg; R.mm.groups = p->G;
    R.mm.Bg = A.Bg; R.mm.row_off = A.row_off; R.mm.flags = A.flags;
    R.mm.zmm = A.zmm;
    R.mm.ztab = A.mm_ztab;
    R.mm.mmfac = A.mmfac;
    R.mm.linv = reinterpret_cast<double*>(ws + p->off_reg_linv);
    R.mm.xt = A.xt;
    R.mm.xt_off = (unsigned)p->off_xt;
    R.mm.xch = A.xch;
    R.mm.fan = p->mm_fan; R.mm.nwg = p->nwg;
    // (a group in more parts than an XCD has CUs cannot sit on one: the two-level exchange crosses the fabric)
    //  -- unless it is the ONLY group and its collectors were laid out per XCD: pmbrl.hip, mm_fan; pr_wg, mode 2)
    const bool xcd_on = !(getenv("PMBRL_XCH_XCD") && atoi(getenv("PMBRL_XCH_XCD")) == 0);
    R.mm.xcd = (p->mm_parts > 1 && xcd_on) ? (p->mm_fan ? (pm_reg_tree_xcd(p) ? 2 : 0) : 1) : 0;
    R.mm.tag0 = (++p->xch_gen) << 12;      // (steps < 4096; the parity of a tag is the parity of its step: the two sets alternate)
    R.mm.inv_m = 1.0 / (double)p->M;
    R.mm.inv_m1 = 1.0 / (double)(p->M - 1);
  }
}

// this launch can go to the family: the whole horizon in one launch, nothing optional asked for (what the LEAN
// variant of pmbrl_fast.h serves)
bool pm_reg_can_run(const pmbrl_plan* p, const RolloutArgs& A, bool fwd) {
  if (!p->reg || (A.mm_mode != 0) != (p->reg_mm != 0)) return false;
  if (!fwd && !p->reg_bwd) return false;
  const bool ext = A.grad_states || A.grad_actions || A.agn || A.zpol_ss != 0 || A.zdyn_ss != 0 ||
                   (((A.flags & PMBRL_FLAG_MM_STATES) != 0) != (p->reg_mm != 0)) || A.t0 != 0 || A.t1 != A.H || A.gx_from_carry;
  if (p->reg_mm && p->mm_parts > 1 && !A.xch) return false;
  return !ext;
}

void pm_reg_pack_launch(const pmbrl_plan* p, char* ws, const float* pol_params, const float* dyn_params, int* wflag, int gen,
                        hipStream_t s, int* status_reset, const ZtabArgs* ztab) {
  RegPackArgs P;
  memset(&P, 0, sizeof(P));
  P.par[0] = pol_params; P.par[1] = dyn_params;
  const NetPlan* nets[2] = {&p->pol, &p->dyn};
  for (int n = 0; n < 2; ++n) {
    for (int l = 0; l < 3; ++l) { P.w_off[n][l] = (int)nets[n]->w_off[l]; P.b_off[n][l] = (int)nets[n]->b_off[l]; }
    P.n_in[n] = nets[n]->dim[0];
    P.n_out[n] = nets[n]->dim[3];
  }
  P.D = p->cfg.D; P.U = p->cfg.U; P.hid = p->pol.dim[1];
  P.out = reinterpret_cast<float*>(ws + p->off_reg_pack);
  P.wflag = wflag; P.gen = gen;
  P.status = status_reset;
  const int total = 4 * (PR_NET_FLOATS / PR_FRAG) * 64;
  P.n_pack_blocks = (total + 255) / 256;
  if (ztab) P.zt = *ztab;      // (memset above: n_blocks = 0 without one)
  if (P.zt.n_blocks > 0) hipLaunchKernelGGL(pm_reg_pack_kernel<true>, dim3(P.n_pack_blocks + P.zt.n_blocks), dim3(256), 0, s, P);
  else hipLaunchKernelGGL(pm_reg_pack_kernel<false>, dim3(P.n_pack_blocks), dim3(256), 0, s, P);
}

// the activity bits of the last forward sweep, from the family's words to the per-tile bytes the latency-optimised
// family's adjoint sweep reads (an adjoint call with optional outputs after a forward call this family served)
void pm_reg_unpack_abits(const pmbrl_plan* p, char* ws, hipStream_t s) {
  RegUnpackArgs U;
  U.B = p->cfg.B; U.H = p->cfg.H; U.nwg = p->nwg;
  U.mm_on = p->reg_mm ? 1 : 0; U.M = p->M; U.parts = p->mm_parts; U.rpw = p->rows_per_wg;
  const NetPlan* nets[2] = {&p->pol, &p->dyn};
  for (int n = 0; n < 2; ++n)
    for (int l = 0; l < 2; ++l) {
      U.src[n][l] = reinterpret_cast<const unsigned*>(ws + p->off_reg_ab[n][l]);
      U.dst[n][l] = reinterpret_cast<unsigned char*>(ws + nets[n]->abits[l]);
    }
  hipLaunchKernelGGL(pm_reg_unpack_abits_kernel, dim3(p->nwg, p->cfg.H), dim3(PR_NTHR), 0, s, U);
}

template <int MMD, bool TREE = false>
static void reg_launch_mm(int nwg, const RegArgs& R, hipStream_t s, bool fwd) {
  if (fwd) {
    if (R.prof) hipLaunchKernelGGL((pm_reg_fwd_kernel<true, MMD, TREE>), dim3(nwg), dim3(PR_NTHR), PR_LDS_FLOATS * sizeof(float), s, R);
    else hipLaunchKernelGGL((pm_reg_fwd_kernel<false, MMD, TREE>), dim3(nwg), dim3(PR_NTHR), PR_LDS_FLOATS * sizeof(float), s, R);
  } else {
    if (R.prof) hipLaunchKernelGGL((pm_reg_bwd_kernel<true, MMD, TREE>), dim3(nwg), dim3(PR_NTHR), PRB_LDS_FLOATS * sizeof(float), s, R);
    else hipLaunchKernelGGL((pm_reg_bwd_kernel<false, MMD, TREE>), dim3(nwg), dim3(PR_NTHR), PRB_LDS_FLOATS * sizeof(float), s, R);
  }
}

void pm_reg_launch(const pmbrl_plan* p, char* ws, const RolloutArgs& A, const float* pol_params, const float* dyn_params,
                   hipStream_t s, bool fwd) {
  ++p->reg_calls[fwd ? 0 : 1];
  RegArgs R;
  reg_args(p, ws, A, reinterpret_cast<const float*>(ws + p->off_reg_pack), pol_params, dyn_params, R);
  if (getenv("PMBRL_REG_DEBUG"))
    fprintf(stderr, "pm_reg_launch %s: off_mmfac %zu off_reg_pack %zu gT %zu %zu %zu actT %zu %zu %zu abits %zu %zu %zu %zu ws_bytes %zu\n", fwd ? "fwd" : "bwd",
            p->off_mmfac, p->off_reg_pack, p->off_gT[0], p->off_gT[1], p->off_gT[2], p->off_actT[0], p->off_actT[1], p->off_actT[2],
            p->off_reg_ab[0][0], p->off_reg_ab[0][1], p->off_reg_ab[1][0], p->off_reg_ab[1][1], p->ws_bytes);
  // (more groups than fit the chip at once: batches of whole groups, one launch each -- pmbrl_host.h, mm_gpb)
  // (groups' parts on one XCD -- pr_wg: hardware workgroups in blocks of 8 groups, the last block padded with workgroups
  //  that exit at once; a batch is a whole number of blocks)
  const int total = R.mm.xcd == 2 ? 8 * 2 * p->mm_fan : R.mm.xcd ? (p->G + 7) / 8 * 8 * p->mm_parts : p->nwg;
  int per = (p->reg_mm && p->mm_gpb > 0) ? p->mm_gpb * p->mm_parts : total;
  if (R.mm.xcd == 1 && per < total) per = (p->mm_gpb + 7) / 8 * 8 * p->mm_parts;
  for (int w0 = 0; w0 < total; w0 += per) {
    R.wg0 = w0;
    const int n = std::min(per, total - w0);
    if (p->reg_mm == 4 && p->mm_fan) { reg_launch_mm<4, true>(n, R, s, fwd); continue; }
    switch (p->reg_mm) {
      case 4: reg_launch_mm<4>(n, R, s, fwd); break;
      case 5: reg_launch_mm<5>(n, R, s, fwd); break;
      case 6: reg_launch_mm<6>(n, R, s, fwd); break;
      default: reg_launch_mm<0>(n, R, s, fwd); break;
    }
  }
}
