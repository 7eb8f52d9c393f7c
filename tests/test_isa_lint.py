"""The fast sweep kernels and the dW GEMM issue some loads through inline asm and wait for them
with explicit s_waitcnt (prob_mbrl_amd/csrc/pmbrl_fast.h, pmbrl_dw.h).  This compiles the
device code to ISA and checks that no instruction touches a register whose load is still in
flight (tools/check_inflight.py) -- a register-allocator copy or spill there would be a silent
data corruption that only shows on some shapes."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_no_instruction_touches_in_flight_registers(tmp_path):
    csrc = os.path.join(ROOT, 'prob_mbrl_amd', 'csrc')
    # the translation units that hold inline-asm loads, compiled to ISA side by side
    units = [('pmbrl.hip', []), ('pmbrl_fast_f32.hip', []), ('pmbrl_fast_split.hip', ['-DPM_SPLIT_PR=1']),
             ('pmbrl_fast_split.hip', ['-DPM_SPLIT_PR=2']), ('pmbrl_general_split.hip', [])]
    procs = []
    for k, (src, extra) in enumerate(units):
        out = str(tmp_path / ('u%d.s' % k))
        procs.append((out, subprocess.Popen([HIPCC, '-w', '--offload-arch=gfx950', '-O3', '-std=c++17',
                                             '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only'] + extra +
                                            [os.path.join(csrc, src), '-o', out], cwd=csrc)))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_inflight as CI
    names, total_loads, general_split = [], 0, 0
    for out, pr in procs:
        assert pr.wait() == 0
        funcs = CI.parse_functions(out)
        for n in funcs:
            # (the general family's split-operand instantiations -- template argument 2 -- stream their weights
            #  with inline-asm loads too: pmbrl_gsplit.h)
            #  (every instantiation with precision argument 2, the in-place 64-row ones -- <4, 2, 1>, <4, 2, 2> -- included)
            if 'fast' in n or 'pm_dw_kernel' in n or re.search(r'pm_rollout_(fwd|bwd)ILi\dELi2E', n):
                bad, nload, _ = CI.check_function(n, funcs[n])
                assert bad == 0, '%s: %d in-flight register violations' % (n, bad)
                total_loads += nload
                names.append(n)
                general_split += 'fast' not in n and 'pm_dw' not in n
    assert len(names) >= 19 + 16 + 6 + 8, len(names)     # fp32 instantiations + the dW kernel + the split-precision ones
    assert total_loads > 500
    # pm_rollout_fwd / bwd <1|2|4, 2, 0>, the in-place <4, 2, 1>, the in-place wide layers <4, 2, 2> (pmbrl_wide.h) and the
    # same with pre-split stashes <4, 2, 3> (round 6: pmbrl_dw.h, pm_dw_wide_pre_kernel)
    assert general_split == 12, general_split
    # register spills of the default-precision instances that run the cart-pole shapes: none without moment matching,
    # none with 25-row groups split over two 16-row workgroups (statistics exchange: the rows + flags form is no
    # longer compiled into them); the 32-row instance of the double cart-pole shape is bounded
    txt = open(procs[3][0]).read()          # pmbrl_fast_split.hip, PM_SPLIT_PR = 2
    spills = {}
    for blk in txt.split('- .agpr_count:')[1:]:
        spills[re.search(r'\.name:\s+(\S+)', blk).group(1)] = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
    shape = lambda d: 'PfShapeILi%dELi1ELi240ELi3ELi13EELi2EE' % d
    for direction in ('fwd', 'bwd'):
        for var, d in ((0, 4), (2, 4)):
            n = '_Z19pm_rollout_%s_fastILi1ELi4ELi3ELi%dE7%sv11RolloutArgs' % (direction, var, shape(d))
            # (the adjoint instance with moment matching parks one 64-bit pointer in its prologue, outside the step loop)
            assert spills[n] <= (2 if (direction, var) == ('bwd', 2) else 0), (n, spills[n])
        n = '_Z19pm_rollout_%s_fastILi2ELi4ELi3ELi2E7%sv11RolloutArgs' % (direction, shape(6))
        assert spills[n] <= 64, (n, spills[n])
    # the in-place wide layers (pmbrl_wide.h): what their K loop uses must stay in registers -- a scratch reload between
    # the ring's loads is waited for with vmcnt(0) and drains the ring (measured: +10 % on the forward sweep with ONE
    # reloaded pointer); the adjoint instance spills nothing, the forward instance a few kernel-scope values outside
    txt = open(procs[4][0]).read()          # pmbrl_general_split.hip
    for blk in txt.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        sp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
        if 'pm_rollout_bwdILi4ELi2ELi2E' in name or 'pm_rollout_bwdILi4ELi2ELi3E' in name:
            assert sp == 0, (name, sp)
        if 'pm_rollout_fwdILi4ELi2ELi2E' in name or 'pm_rollout_fwdILi4ELi2ELi3E' in name:
            assert sp <= 32, (name, sp)
    shutil.rmtree(str(tmp_path), ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_register_resident_kernels_fit_the_register_file(tmp_path):
    """The register-resident sweep family (csrc/pmbrl_reg.h) keeps 62 weight fragments in AGPRs and the rest in VGPRs
    at one wave per SIMD: a spill would put weights in scratch memory inside the step loop.  No vector spills, the
    accumulation registers actually used, and one workgroup of four waves per CU."""
    csrc = os.path.join(ROOT, 'prob_mbrl_amd', 'csrc')
    out = str(tmp_path / 'reg.s')
    subprocess.check_call([HIPCC, '-w', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                           '-S', '--cuda-device-only', os.path.join(csrc, 'pmbrl_reg.hip'), '-o', out], cwd=csrc)
    txt = open(out).read()
    seen = 0
    for blk in txt.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        if 'pm_reg_fwd_kernel' not in name and 'pm_reg_bwd_kernel' not in name:
            continue
        seen += 1
        agpr = int(blk.split()[0])
        vgpr = int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1))
        spills = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
        assert spills == 0, (name, spills)
        assert agpr >= 4 * 62 and vgpr <= 512, (name, agpr, vgpr)
        assert int(re.search(r'\.max_flat_workgroup_size:\s+(\d+)', blk).group(1)) == 256, name
    # forward and adjoint, each with and without the cycle-counter instrumentation, each plain and with the in-sweep moment
    # matching of state widths 4, 5, 6 (pmbrl_reg_mm.h: C3's and C4's shapes among them), and the four instances of width 4
    # that carry the two-level exchange of ONE group over the batch (mm_groups=None)
    assert seen == 20, seen
    shutil.rmtree(str(tmp_path), ignore_errors=True)
