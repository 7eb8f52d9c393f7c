#!/usr/bin/env python
"""Benchmark of the MC-PILCO hot path on MI355X.

One "step" = one full optimiser iteration of algorithms/mc_pilco.py:86-216 on a
synthetic Cartpole-shaped problem (BASELINE.json configs[1]: |x|=4, |u|=1,
2x200 nets, 100 particles x 25 dropout samples = 2500 rows, H=40):
  fused rollout forward -> discounted-return loss -> adjoint sweep -> dW GEMM ->
  [RCCL all-reduce of the flat policy gradient when N > 1] -> fused clip + Adam.
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU
(torch.distributed, backend nccl = RCCL); the only collective is the gradient
all-reduce.  --scaling weak (default): every rank owns its own 2500 rows of a
global batch of N*2500 (`value`); --scaling strong: the 100 x 25 = 2500 rows of BASELINE.json's
metric are divided over the N ranks (whole particle groups per rank, as evenly as they
divide).  For N > 1 BOTH curves are timed in the one invocation: `value` is the chosen one, the
other is reported next to it (`strong: {...}` / `weak: {...}`), and `rccl_ranks` is the rank
count the RCCL communicator itself reports.  For N = 1 the exact-fp32 path is timed on the same
problem next to the default arithmetic (`f32: {...}`).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA (the 2:1-sparsity figure is not used)
N_CUS = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cartpole_nomm')
    ap.add_argument('--rows-per-wg', type=int, default=0)
    ap.add_argument('--precision', default=None, choices=['f32', 'split', 'split_f16'],
                    help='arithmetic of the hidden-width GEMMs (default: the library default)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='N > 1: per-GPU rows fixed (weak) or the global 100 x 25 rows divided over the GPUs (strong)')
    ap.add_argument('--mm-global', action='store_true',
                    help='moment-matching configs: ONE group over the rows of all ranks (mm_groups=None, the '
                         "reference examples' default) -- per-step statistics exchange between the ranks; weak scaling only")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--replay', type=int, default=None, choices=[0, 1, 2],
                    help='pmbrl_plan_set_replay: 0 never replay repeated calls as hipGraphs, 1 the library default (the '
                         'one-launch-per-step forms), 2 every form')
    ap.add_argument('--no-f32-twin', action='store_true', help='N = 1: skip the exact-fp32 leg')
    ap.add_argument('--no-sclk', action='store_true', help='skip the shader-clock probe (three instrumented forward launches)')
    ap.add_argument('--no-fused-tail', action='store_true',
                    help='N = 1: separate loss / dW-reduce / norm / Adam launches (what N > 1 runs around its all-reduce)')
    ap.add_argument('--no-second-curve', action='store_true', help='N > 1: skip the other scaling curve')
    ap.add_argument('--no-other-configs', action='store_true',
                    help="N = 1, headline config: skip the short legs of BASELINE.json's other configurations (`other_configs`)")
    ap.add_argument('--cpu-only', action='store_true', help='only time the CPU baseline (no GPU needed)')
    ap.add_argument('--timing-steps', type=int, default=10)
    ap.add_argument('--repeats', type=int, default=0,
                    help='timed blocks of EXACTLY --steps steps each (every block bracketed by barrier + synchronize); '
                         '`value` is the median block.  0 (default): as many as give >= 0.25 s of device time, at least 25 '
                         'for blocks under 10 ms, at most 200')
    ap.add_argument('--pmc', action='store_true',
                    help='N = 1: first collect the HBM-traffic and MFMA counters of this build and configuration by re-running '
                         'under rocprofv3 --pmc (one pass per counter set, a few minutes), write profiles/pmc_<config>_<precision>.json, '
                         'then time as usual and report them as measured in this run')
    # debugging aids for the N>1 control flow on a box with ONE GPU (tests/test_gpu_api.py):
    # every rank on cuda:0, collectives over gloo.  Never used for a reported number.
    ap.add_argument('--transport', default='rccl', choices=['rccl', 'p2p'],
                    help='N > 1: gradient all-reduce / statistics exchange through RCCL (default) or the one-shot '
                         'peer-to-peer transport of pmbrl_p2p.hip (PMBRL_P2P=1; ranks on one node)')
    ap.add_argument('--dist-backend', default='nccl')
    ap.add_argument('--one-device', action='store_true')
    return ap.parse_args()


def cpu_baseline_bnn(dims, h, Xc, Yc, N, batch, iters):
    """CPU leg of tools/bench_bnn.py (the BNN training step): the oracle's step (torch
    autograd on the host + its Adam) on the same shapes; returns (iterations/s, threads used)."""
    from oracle import ref_torch as R
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    W = [(torch.randn(dims[l + 1], dims[l]) / np.sqrt(dims[l]), torch.zeros(dims[l + 1])) for l in range(3)]
    lp = [torch.full((h,), 1.1) for _ in range(2)]
    params = [t for pair in W for t in pair] + lp
    for p in params:
        p.requires_grad_(True)
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]

    def cpu_it(i):
        idx = torch.randint(0, N, (batch,))
        us = [torch.rand(batch, h) for _ in range(2)]
        hs = []
        for p in params:
            p.grad = None
        with torch.no_grad():
            for l in range(2):
                hs.append(torch.bernoulli(torch.full((batch, h), 0.7)))
        loss, _, _ = R.bnn_loss(W, lp, [0.1, 0.1], [0.5, 0.5], [1.0, 1.0], Xc[idx], Yc[idx], us, hs, N)
        loss.backward()
        with torch.no_grad():
            for p, mm, vv in zip(params, ms, vs):
                R.adam_step(p, p.grad, mm, vv, i + 1, 1e-4)

    for i in range(5):
        cpu_it(i)
    t0 = time.perf_counter()
    for i in range(iters):
        cpu_it(5 + i)
    return iters / (time.perf_counter() - t0), torch.get_num_threads()


def cpu_baseline(d, budget_s=24.0):
    """The oracle (a torch-CPU port of the reference's op sequence, incl. the
    discarded dynamics-weight gradients) timed on the host: full iteration =
    rollout + loss + backward + clip + Adam.  Bounded sample of the SAME workload."""
    from oracle import ref_torch as R
    torch.set_flush_denormal(True)   # the examples do (examples/deep_pilco_mm.py:68)
    B = d['x0'].shape[0]
    ncpu = os.cpu_count() or 1
    best = None
    tried = []
    for threads in sorted({1, min(8, ncpu), min(32, ncpu)}):
        torch.set_num_threads(threads)
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float32)
        params = R.policy_params(pol)
        ms = [torch.zeros_like(p) for p in params]
        vs = [torch.zeros_like(p) for p in params]

        def it(step):
            loss, g, _ = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, meta['maximize'],
                                     meta['mm_states'], meta['mm_rewards'], meta['mm_groups'],
                                     z_mm, z_rr, dyn_requires_grad=True)
            grads = [p.grad for p in params]
            _, grads = R.clip_grad_norm(grads, 1.0)
            with torch.no_grad():
                for p, gg, m, v in zip(params, grads, ms, vs):
                    R.adam_step(p, gg, m, v, step, 1e-4)

        it(1)
        it(2)                      # two warm-ups, then >= 10 timed iterations unless the budget runs out
        times = []
        t_start = time.perf_counter()
        step = 3
        while len(times) < 10 and (time.perf_counter() - t_start) < budget_s / 3:
            t0 = time.perf_counter()
            it(step)
            times.append(time.perf_counter() - t0)
            step += 1
        med = float(np.median(times))
        tried.append((threads, med, len(times)))
        if best is None or med < best[1]:
            best = (threads, med, len(times))
    threads, med, n = best
    one = [m for t, m, _ in tried if t == 1][0]
    return dict(value=B / med, unit='rollouts/s', cores=threads, kind='port',
                sample='full iterations (B=%d rows, H=%d) after 2 warm-ups, median per thread setting; tried %s' %
                       (B, int(d['H']), ', '.join('%dT:%.0fms(n=%d)' % (t, m * 1e3, k) for t, m, k in tried)),
                ms_per_step=med * 1e3, host_cpus=ncpu,
                one_thread=dict(value=B / one, ms_per_step=one * 1e3))   # the reference's default (examples/deep_pilco_mm.py:21,65)


def shard_rows(d, lo, hi):
    """Rows [lo, hi) (whole moment-matching groups) of the global problem d as one rank's problem."""
    B = d['x0'].shape[0]
    G = int(d['mm_groups'])
    out = dict(d)
    for k, v in d.items():
        if isinstance(v, (str, bool, int, float)):
            continue
        v = np.asarray(v)
        if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and v.ndim == 2 and v.shape[0] == B):
            out[k] = v[lo:hi]
    out['mm_groups'] = (hi - lo) // (B // G) if G else 0
    return out


class Leg:
    """One timed configuration: an engine, its resident inputs and the optimiser state."""

    def __init__(self, a, d, dev, Bg, row_offset, precision, mm_span, world, allreduce):
        from prob_mbrl_amd import engine as E
        from prob_mbrl_amd import problem as PB
        self.E, self.world, self.allreduce = E, world, allreduce
        self.d, self.B, self.Bg, self.H = d, d['x0'].shape[0], Bg, int(d['H'])
        self.eng, self.args, _ = PB.engine_from_problem(d, dev, rows_per_wg_hint=a.rows_per_wg, B_global=Bg,
                                                        row_offset=row_offset, precision=precision, mm_span=mm_span)
        if mm_span:
            import torch.distributed as dist
            self.eng.attach_collective(dist.group.WORLD)
        if a.replay is not None:
            self.eng.set_replay(a.replay)
        self.gw = torch.tensor(PB.loss_weights(d, Bg)[:, :self.B].copy(), device=dev)
        self.params = self.args['pol_flat'].clone()
        self.params0 = self.params.clone()
        self.args['pol_flat'] = self.params
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.loss_buf = torch.zeros(1, device=dev)
        self.n = 0
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # one process: the iteration is two library calls (pmbrl_rollout_fwd, pmbrl_rollout_bwd_adam: adjoint, dW, the
        # loss on the way of the gradient reduction -- pmbrl_adam::loss_out_d --, clip + Adam); with a gradient all-reduce
        # between the dW reduction and the optimiser (N > 1) the separate calls
        self.fused = world == 1 and not a.no_fused_tail
        if self.fused:
            self.adam = dict(params=self.params, exp_avg=self.m, exp_avg_sq=self.v, step=self.step_dev, lr=1e-4,
                             betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, loss_out=self.loss_buf)

    def step(self):
        self.n += 1
        if self.fused:
            self.eng.forward(**self.args)
            self.eng.backward(self.gw, adam=self.adam)
            return
        _, _, R = self.eng.forward(**self.args)
        self.eng.weighted_sum(R, self.gw, out=self.loss_buf)
        g, _, _ = self.eng.backward(self.gw)
        if self.world > 1:
            self.allreduce(g)
        self.E.clip_adam(self.params, g, self.m, self.v, self.n, 1e-4, max_norm=1.0)

    def restart(self):
        self.params.copy_(self.params0)
        self.m.zero_()
        self.v.zero_()
        self.step_dev.zero_()
        self.n = 0

    def sync(self):
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def timed(self, steps, warmup, dev, repeats=0):
        """W untimed steps, then blocks of EXACTLY `steps` steps, each between barrier + synchronize, max over ranks per
        block.  Returns (median block seconds, [all block seconds]).  One 20-step block of the metric's shape is 10 ms of
        device time -- the clock the part settles at moves a single sample by +-3 %; the median of >= 25 blocks does not."""
        for _ in range(warmup):
            self.step()
        blocks = []
        n_blocks = max(1, repeats)
        while len(blocks) < n_blocks:
            # every block is the SAME work: the optimiser restarts from the initial parameters (a few hundred Adam steps on
            # a synthetic problem would otherwise walk the policy to where a group's particles collapse and the moment
            # matching loses its pivot -- a truncated horizon, which is less work per step); outside the timed region
            self.restart()
            self.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.sync()
            dt = time.perf_counter() - t0
            if self.world > 1:
                import torch.distributed as dist
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            blocks.append(dt)
            if repeats == 0 and len(blocks) == 1:
                # every rank derives the same count from the (max-reduced) first block
                n_blocks = int(min(200, max(25 if dt < 0.010 else 3, np.ceil(0.25 / max(dt, 1e-6)))))
        assert self.eng.valid_steps() == self.H, 'numerical failure inside the benchmark rollout'
        assert bool(torch.isfinite(self.params).all()) and bool(torch.isfinite(self.loss_buf).all())
        if self.fused:      # every optimiser step was taken (the device-side counter says so)
            assert int(self.step_dev.item()) == self.n, (int(self.step_dev.item()), self.n)
        return float(np.median(blocks)), blocks

    def kernel_ms(self, n):
        """Per-kernel durations (HIP events on the launch stream), outside the timed region.  Every rank runs these
        steps (step() contains the gradient all-reduce)."""
        self.eng.set_timing(True)
        acc = {}
        for _ in range(n):
            self.step()
            for k, ms in self.eng.read_timing().items():
                if ms >= 0:
                    acc.setdefault(k, []).append(ms)
        self.eng.set_timing(False)
        self.sync()
        return {k: float(np.mean(vv)) for k, vv in acc.items()}

    def roofline(self, timings):
        """The dominant kernel against the dense MFMA peak of the instruction it issues: exact fp32 MFMA, or fp16 / bf16
        MFMA at THREE instructions per fp32-equivalent product (two-piece split operands; the three-piece bf16
        forward of 'split' issues six)."""
        from prob_mbrl_amd import problem as PB
        eng, d, B, H = self.eng, self.d, self.B, self.H
        _, Pm, Fm = PB.algorithmic_flops_per_rollout(d)
        # algorithmic flops per launch of each kernel (one launch covers B rows x H steps)
        kflops = dict(fwd=2.0 * H * B * (Pm + Fm), bwd=2.0 * H * B * (Pm + Fm), dw=2.0 * H * B * Pm)
        dom = max(kflops, key=lambda k: timings.get(k, 0.0))
        if eng.info.get('dw_pipe', 1) > 1 and dom == 'bwd':
            # the adjoint timer then spans several launches of the sweep (with the dW GEMM running behind them on a
            # second stream) and is not one kernel's duration: price the one-launch forward sweep instead
            dom = 'fwd'
        achieved = kflops[dom] / (timings[dom] * 1e-3) / 1e12
        kname = {'fwd': 'pm_rollout_fwd', 'bwd': 'pm_rollout_bwd', 'dw': 'pm_dw_kernel'}[dom]
        if eng.info.get('fast') and dom != 'dw':
            kname += '_fast'
        reg = bool(eng.info.get('reg')) and dom != 'dw' and not (dom == 'bwd' and os.environ.get('PMBRL_REG_BWD') == '0')
        if reg:         # the register-resident family (csrc/pmbrl_reg.h): weights live in the four waves' registers
            kname = 'pm_reg_%s_kernel' % dom
        prec = eng.info['precision']
        mfma_per_product = {('f32', 'fwd'): 1, ('f32', 'bwd'): 1, ('split', 'fwd'): 6, ('split', 'bwd'): 3,
                            ('split_f16', 'fwd'): 3, ('split_f16', 'bwd'): 3}.get((prec, dom), 1)
        if prec == 'f32' or dom == 'dw':
            peak, peak_note = PEAK_F32_MFMA_TFLOPS, 'v_mfma_f32_16x16x4_f32 dense peak'
        else:
            peak = PEAK_F16_MFMA_TFLOPS / mfma_per_product
            peak_note = ('v_mfma_f32_16x16x32_%s dense peak (2500 TFLOP/s) / %d MFMAs per fp32-equivalent product' %
                         ('f16' if (prec == 'split_f16' and dom == 'fwd') else 'bf16', mfma_per_product))
        n_wg = eng.info['n_wg']
        # what binds the sweep: the latency-optimised family is one sequential chain of H steps per workgroup (DESIGN.md
        # 4) -- the matrix pipe is not the limiter at any size it serves; the general family (wide networks, >= 2
        # workgroups per CU) is priced as MFMA-bound
        latency = bool(eng.info.get('fast')) and dom != 'dw'
        r = dict(bound='latency' if latency else 'mfma', kernel=kname, achieved=achieved, peak=peak, unit='TFLOP/s',
                 frac=achieved / peak, peak_is=peak_note, frac_of_f32_mfma_peak=achieved / PEAK_F32_MFMA_TFLOPS,
                 binding=(('latency: H sequential steps per workgroup, %d of %d CUs occupied; ' % (min(n_wg, N_CUS), N_CUS)) +
                          ('weights register-resident, one wave per SIMD: instruction issue inside a step' if reg else
                           'weight stream L2->CU inside a step')) if latency else
                         'matrix pipe / weight fetch L2->CU: %d workgroups over %d CUs' % (n_wg, N_CUS),
                 flops_per_launch=kflops[dom], avg_launch_ms=timings[dom])
        return r, kname


def sclk_mhz(leg):
    """Shader clock the part ran the sweep at, right after the timed region: cycle counter against the constant 100 MHz
    clock, both stamped by workgroup 0 at entry and exit of ONE instrumented forward launch (register-resident family
    only: slots 28..31 of pmbrl_plan_set_prof's buffer).  None where the running kernels carry no such stamps."""
    import ctypes as C
    from prob_mbrl_amd import _lib
    try:
        eng = leg.eng
        if not eng.info.get('reg'):
            return None
        H = eng.H
        pf = torch.zeros(H * 32, dtype=torch.int64, device=leg.params.device)
        pb = torch.zeros(H * 32, dtype=torch.int64, device=leg.params.device)
        _lib.check(eng.lib.pmbrl_plan_set_prof(eng.plan, C.c_void_p(pf.data_ptr()), C.c_void_p(pb.data_ptr())), 'prof')
        for _ in range(3):
            eng.forward(**leg.args)
        torch.cuda.synchronize()
        _lib.check(eng.lib.pmbrl_plan_set_prof(eng.plan, C.c_void_p(0), C.c_void_p(0)), 'prof')
        a = pf.cpu().numpy().reshape(H, 32)
        cyc, us = float(a[0, 31] - a[0, 30]), float(a[0, 29] - a[0, 28]) / 100.0
        return round(cyc / us, 1) if us > 0 and cyc > 0 else None
    except Exception:      # noqa: BLE001
        return None


def build_id():
    from prob_mbrl_amd import _lib
    return _lib.load().pmbrl_build_id().decode()


def pmc_path(config, prec):
    return os.path.join(ROOT, 'profiles', 'pmc_%s_%s.json' % (config, prec))


def collect_pmc(a, prec):
    """bench.py --pmc: the hardware counters of THIS build on THIS configuration, taken by re-running this script
    under `rocprofv3 --pmc <set>` once per counter set (counters are never collected together with a trace; FETCH_SIZE
    and WRITE_SIZE do not fit one pass -- MI355X_MICROARCH.md).  Per kernel, averaged over the launches of a pass:
    HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies a 128-byte read request at 64 bytes; KiB units), MFMA busy
    per cent, flops issued = MFMA_MOPS x 512.  Written to profiles/pmc_<config>_<precision>.json with the library's
    build id; returns the dict."""
    import csv
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    sets = [['FETCH_SIZE'], ['WRITE_SIZE'], ['MfmaUtil'], ['SQ_INSTS_VALU_MFMA_MOPS_F16', 'SQ_INSTS_VALU_MFMA_MOPS_BF16']]
    steps = 3 if a.config.startswith('stress') else 6
    warm, tsteps, reps = 2, 1, 1
    # bench iterations of one pass: warm-up + the timed block(s) + the timing steps; --replay 0: every launch is a kernel
    # dispatch the counters see, never a node of a replayed hipGraph
    iters = warm + steps * reps + tsteps
    inner = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', a.config, '--steps', str(steps), '--warmup', str(warm),
             '--repeats', str(reps), '--replay', '0', '--no-sclk', '--no-cpu-baseline', '--no-f32-twin', '--no-other-configs',
             '--timing-steps', str(tsteps)]
    if a.precision:
        inner += ['--precision', a.precision]
    if a.rows_per_wg:
        inner += ['--rows-per-wg', str(a.rows_per_wg)]
    env = dict(os.environ, TMPDIR='/tmp')
    acc = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
    for cs in sets:
        tmp = tempfile.mkdtemp(prefix='pmbrl_pmc_', dir='/tmp')
        try:
            r = subprocess.run(['rocprofv3', '--pmc'] + cs + ['--output-format', 'csv', '-d', tmp, '--'] + inner,
                               cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp) for f in fs if f.endswith('counter_collection.csv')]
            if r.returncode != 0 or not files:
                raise RuntimeError('rocprofv3 --pmc %s failed (rc %d): %s' % (' '.join(cs), r.returncode, r.stderr.decode()[-300:]))
            for fpath in files:
                for row in csv.DictReader(open(fpath)):
                    name = row['Kernel_Name'].replace('void ', '').split('(')[0].split('<')[0]
                    if name.startswith('pm_'):
                        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    mean = lambda v: sum(v) / len(v) if v else None
    kernels = {}
    for k, c in acc.items():
        f, w = mean(c.get('FETCH_SIZE')), mean(c.get('WRITE_SIZE'))
        mops = (mean(c.get('SQ_INSTS_VALU_MFMA_MOPS_F16')) or 0.0) + (mean(c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16')) or 0.0)
        kernels[k] = dict(FETCH_SIZE_KB=f, WRITE_SIZE_KB=w,
                          hbm_bytes_per_launch=(2.0 * (f or 0.0) + (w or 0.0)) * 1024.0 if (f is not None or w is not None) else None,
                          mfma_util_pct=mean(c.get('MfmaUtil')), flops_issued_per_launch=mops * 512.0,
                          launches=len(c.get('FETCH_SIZE', [])), iterations_per_pass=iters,
                          # (the sweeps of the one-launch-per-step forms are many launches under one timer)
                          launches_per_iteration=len(c.get('FETCH_SIZE', [])) / float(iters))
    out = dict(build_id=build_id(), config=a.config, precision=prec, command=' '.join(inner[1:]),
               note='rocprofv3 --pmc, one pass per counter set: %s; FETCH_SIZE doubled (gfx950: 128-byte requests tallied at '
                    '64 bytes), WRITE_SIZE as reported (KiB); means over the launches of a pass' % '; '.join(' '.join(c) for c in sets),
               kernels=kernels)
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    json.dump(out, open(pmc_path(a.config, prec), 'w'), indent=1)
    return out


def pmc_by_timer(pmc, timings, top=3):
    """Achieved HBM GB/s of the `top` longest phases of an iteration: the counter file's per-kernel HBM bytes (x launches per
    iteration), grouped under the library timer that brackets the kernel, over that timer's HIP-event duration in THIS run."""
    def timer_of(k):
        if 'dw_reduce' in k:
            return 'dw_reduce'
        if 'pm_dw' in k:
            return 'dw'
        if 'reward' in k:
            return 'reward'
        if 'fwd' in k:
            return 'fwd'
        if 'bwd' in k:
            return 'bwd'
        if 'pack' in k:
            return 'pack'
        return None
    acc = {}
    for k, c in pmc['kernels'].items():
        t = timer_of(k)
        if t is None or c.get('hbm_bytes_per_launch') is None:
            continue
        e = acc.setdefault(t, dict(bytes=0.0, kernels=[]))
        e['bytes'] += c['hbm_bytes_per_launch'] * max(1.0, round(c.get('launches_per_iteration') or 1.0))
        e['kernels'].append(k)
    out = {}
    for t in sorted((t for t in acc if timings.get(t, 0.0) > 0.0), key=lambda t: -timings[t])[:top]:
        gbps = acc[t]['bytes'] / (timings[t] * 1e-3) / 1e9
        out[t] = dict(kernels=sorted(acc[t]['kernels']), hbm_bytes_per_iteration=acc[t]['bytes'], ms=round(timings[t], 4),
                      hbm_gbps=round(gbps, 1), frac_of_8tbps=round(gbps / 8000.0, 4))
    return out


def pmc_lookup(config, prec, kname, world, fresh=None):
    """Counters of the dominant kernel: from this invocation's own --pmc passes (`fresh`), else from the committed
    profiles/pmc_<config>_<precision>.json -- but ONLY if that file was measured on the build that is running
    (pmbrl_build_id): a measurement of other code is not reported."""
    if world != 1:
        return None
    src, pmc = None, fresh
    if pmc is None:
        try:
            pmc = json.load(open(pmc_path(config, prec)))
            src = os.path.relpath(pmc_path(config, prec), ROOT)
        except Exception:
            return dict(available=False, reason='no profiles/pmc_%s_%s.json (run bench.py --pmc)' % (config, prec))
        if pmc.get('build_id') != build_id():
            return dict(available=False, source=src, reason='measured on build %s, this is build %s (re-run bench.py --pmc)' %
                        (pmc.get('build_id'), build_id()))
    k = pmc['kernels'].get(kname)
    if k is None:
        return dict(available=False, source=src, reason='kernel %s not in the measurement' % kname)
    return dict(available=True, measured_in_run=fresh is not None, source=src, build_id=pmc['build_id'], _all=pmc, **k)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one per GPU) under
    torch.distributed.run, rendezvous on 127.0.0.1 at a free port.  The children inherit stdout / stderr (rank 0 prints
    the one JSON line); returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# what the weak curve should look like if the one exposed 163 KiB all-reduce per iteration costs what a
# ring over xGMI is expected to cost (2 (N - 1) hops of 2-3 us) -- kept in the line so that the record can be checked
# against it
PREDICTED_WEAK = {2: dict(ms_per_step=0.465, value=10.7e6, efficiency=0.95), 4: dict(ms_per_step=0.475, value=21.0e6, efficiency=0.93),
                  8: dict(ms_per_step=0.495, value=40.4e6, efficiency=0.89)}


def other_configs(a, dev):
    """Short timed legs of the configurations the headline line does not time: {name: {value, ms_per_step, ...}}."""
    import copy
    from prob_mbrl_amd import problem as PB
    out = {}
    for name, cfg, groups, steps, reps in (('cartpole_mm', 'cartpole_mm', None, 20, 5),
                                           ('cartpole_mm_g1', 'cartpole_mm', 0, 20, 5),
                                           ('dcartpole_mm', 'dcartpole_mm', None, 10, 5),
                                           ('stress32', 'stress32', None, 3, 2)):
        b = copy.copy(a)
        b.config, b.no_fused_tail = cfg, False
        d = dict(PB.synthetic_problem(cfg, seed=0, data_seed=0))
        if groups is not None:
            d['mm_groups'] = np.asarray(groups)
        try:
            leg = Leg(b, d, dev, d['x0'].shape[0], 0, a.precision, None, 1, None)
            dt, blocks = leg.timed(steps, 2, dev, reps)
            t = leg.kernel_ms(2)
            out[name] = dict(value=leg.Bg * steps / dt, unit='rollouts/s', ms_per_step=dt / steps * 1e3, steps=steps,
                             timed_blocks=len(blocks), value_min=leg.Bg * steps / max(blocks), value_max=leg.Bg * steps / min(blocks),
                             rows=leg.B, horizon=leg.H, mm_groups=int(d['mm_groups']), precision=leg.eng.info['precision'],
                             workgroups=leg.eng.info['n_wg'], rows_per_wg=leg.eng.info['rows_per_wg'],
                             register_resident=bool(leg.eng.info.get('reg')), kernel_ms={k: round(v, 4) for k, v in t.items()})
            del leg
        except Exception as e:      # noqa: BLE001  (a leg that cannot run must not take the headline line with it)
            out[name] = dict(error=str(e)[:200])
        torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    from prob_mbrl_amd import problem as PB
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.cpu_only:
        d = PB.synthetic_problem(a.config, seed=0, data_seed=0)
        print(json.dumps(cpu_baseline(d)))
        return
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher -- one process per GPU through
        # torch.distributed.run on this node, rank 0's JSON line comes out of this process's stdout
        sys.exit(self_launch(a.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=a.dist_backend, rank=rank, world_size=world)
    assert world == a.gpus, ('WORLD_SIZE=%d but --gpus %d: start as `python bench.py --gpus N` (launches itself) or under '
                             'torch.distributed.run --nproc-per-node N' % (world, a.gpus))
    dev = torch.device('cuda:%d' % local_rank)
    torch.cuda.set_device(dev)

    allreduce, rccl_ranks = None, None
    transport_note = None
    if world > 1:
        from prob_mbrl_amd.distributed import get_comm, get_p2p, grad_allreduce
        if a.transport == 'p2p':
            # the one-shot peer-to-peer transport needs every rank to map every peer's buffer (hipIpcOpenMemHandle /
            # peer access): if ANY rank cannot, all of them fall back to RCCL together and the line says so
            os.environ['PMBRL_P2P'] = '1'
            ok = 1
            try:
                get_p2p(None, dev)
            except Exception as e:      # noqa: BLE001
                ok, transport_note = 0, 'p2p unavailable on rank %d (%s)' % (rank, str(e)[:120])
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if a.dist_backend == 'nccl' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                os.environ['PMBRL_P2P'] = '0'
                from prob_mbrl_amd import distributed as _D
                _D._P2PS.clear()
                a.transport = 'rccl'
                transport_note = transport_note or 'p2p unavailable on another rank'
        allreduce = grad_allreduce(None, dev)      # RCCL through the C ABI (or the p2p kernel), on the compute stream
        comm = get_comm(None, dev)
        rccl_ranks = comm.count() if comm is not None else None   # as the communicator reports it (ncclCommCount)
        # start-up self-check, before anything is timed: a known vector through the transport the timed steps will use
        # must come back as its sum over the ranks, bit for bit the same on every rank, and the communicator must have
        # seen all N ranks
        n_chk = 41602      # the flat policy gradient of the metric's shape (163 KiB)
        vec = (torch.arange(n_chk, device=dev, dtype=torch.float32) % 97 + 1.0) * float(rank + 1)
        allreduce(vec)
        torch.cuda.synchronize()
        want = (torch.arange(n_chk, device=dev, dtype=torch.float32) % 97 + 1.0) * float(world * (world + 1) // 2)
        assert torch.equal(vec, want), 'gradient all-reduce self-check failed on rank %d (transport %s)' % (rank, a.transport)
        if a.dist_backend == 'nccl' and not a.one_device and a.transport == 'rccl':
            assert rccl_ranks == world, 'RCCL communicator reports %r ranks, launched %d' % (rccl_ranks, world)
        from prob_mbrl_amd.distributed import p2p_check
        p2p_check(None, dev)

    def make_leg(scaling, precision):
        """weak: every rank brings its own rows of a global batch of N x rows; strong: the rows of BASELINE.json's
        metric (one global problem, the same for every N) dealt to the ranks in whole moment-matching groups, as evenly
        as they divide."""
        mm_span = None
        if scaling == 'strong' and world > 1:
            from prob_mbrl_amd.distributed import shard_bounds
            dg = PB.synthetic_problem(a.config, seed=0, data_seed=0)
            Bg = dg['x0'].shape[0]
            lo, hi = shard_bounds(Bg, int(dg['mm_groups']) or None, world, rank)
            return Leg(a, shard_rows(dg, lo, hi), dev, Bg, lo, precision, None, world, allreduce)
        d = PB.synthetic_problem(a.config, seed=0, data_seed=rank)
        B, H = d['x0'].shape[0], int(d['H'])
        Bg = B * world
        if a.mm_global:
            assert bool(d['mm_states']) and scaling == 'weak', '--mm-global: a moment-matching config, weak scaling'
            d = dict(d)
            d['mm_groups'] = 0
            if world > 1:
                mm_span = (Bg, rank * B, world, rank)
        if world > 1 and bool(d['mm_states']):
            # one cyclic noise buffer over the GLOBAL rows (utils/rollout.py:53-59), the same on every rank
            d = dict(d)
            dz = PB.synthetic_problem(a.config, seed=0, data_seed=0, P=PB.CONFIGS[a.config]['P'] * world)
            d['z_mm'], d['z_rr'] = dz['z_mm'], dz['z_rr']
        return Leg(a, d, dev, Bg, rank * B, precision, mm_span, world, allreduce)

    fresh_pmc = None
    if a.pmc and world == 1:
        from prob_mbrl_amd import engine as _E
        fresh_pmc = collect_pmc(a, a.precision or _E.get_precision())
    primary = a.scaling if world > 1 else 'weak'
    leg = make_leg(primary, a.precision)
    dt, blocks = leg.timed(a.steps, a.warmup, dev, a.repeats)
    timings = leg.kernel_ms(a.timing_steps)
    eng, d, B, Bg, H = leg.eng, leg.d, leg.B, leg.Bg, leg.H
    prec = eng.info['precision']

    extra = {}
    if world > 1 and not a.no_second_curve and not a.mm_global:
        # the other scaling curve in the same invocation, timed the same way
        other = 'strong' if primary == 'weak' else 'weak'
        leg2 = make_leg(other, a.precision)
        dt2, blocks2 = leg2.timed(a.steps, a.warmup, dev, a.repeats)
        rows = torch.tensor([leg2.B], device=dev, dtype=torch.int64)
        lst = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(lst, rows)
        extra[other] = dict(value=leg2.Bg * a.steps / dt2, unit='rollouts/s', ms_per_step=dt2 / a.steps * 1e3,
                            value_min=leg2.Bg * a.steps / max(blocks2), value_max=leg2.Bg * a.steps / min(blocks2),
                            timed_blocks=len(blocks2),
                            global_rows=leg2.Bg, rows_per_gpu=[int(x.item()) for x in lst], steps=a.steps,
                            warmup=a.warmup, workgroups_rank0=leg2.eng.info['n_wg'],
                            rows_per_wg=leg2.eng.info['rows_per_wg'])
        del leg2
    if world == 1 and prec != 'f32' and not a.no_f32_twin:
        # the exact-fp32 MFMA path on the same problem in the same invocation (the reference's arithmetic)
        leg3 = make_leg('weak', 'f32')
        dt3, blocks3 = leg3.timed(a.steps, a.warmup, dev, a.repeats)
        t3 = leg3.kernel_ms(a.timing_steps)
        r3, _ = leg3.roofline(t3)
        extra['f32'] = dict(value=leg3.Bg * a.steps / dt3, unit='rollouts/s', ms_per_step=dt3 / a.steps * 1e3,
                            value_min=leg3.Bg * a.steps / max(blocks3), value_max=leg3.Bg * a.steps / min(blocks3),
                            timed_blocks=len(blocks3),
                            steps=a.steps, warmup=a.warmup, dtype='f32 (v_mfma_f32_16x16x4_f32)',
                            kernel_ms={k: round(vv, 4) for k, vv in t3.items()}, roofline=r3)
        del leg3

    if world == 1 and a.config == 'cartpole_nomm' and not a.no_other_configs and not a.rows_per_wg:
        # BASELINE.json's other configurations in the SAME invocation (short legs, same timing protocol: blocks between
        # barrier + synchronize, median): whoever times this script times them too -- until round 6 only the headline
        # configuration had a number the builder had not run himself.  C3 = cartpole_mm (25-row groups), its
        # mm_groups=None form (ONE 2 500-row group, the reference examples' default), C4 = dcartpole_mm, C5 = stress32.
        extra['other_configs'] = other_configs(a, dev)

    if rank == 0:
        flops_rollout, Pm, Fm = PB.algorithmic_flops_per_rollout(d)
        roof, kname = leg.roofline(timings)
        pm = pmc_lookup(a.config, prec, kname, world, fresh_pmc)
        if pm and pm.get('available'):
            traffic = pm['hbm_bytes_per_launch']
            roof.update(traffic=traffic, traffic_measured_in_run=pm['measured_in_run'], traffic_source=pm['source'] or 'this run (--pmc)',
                        traffic_build_id=pm['build_id'],
                        launches_under_timer=max(1.0, round(pm.get('launches_per_iteration') or 1.0)),
                        hbm_gbps=(traffic * max(1.0, round(pm.get('launches_per_iteration') or 1.0)) /
                                  (roof['avg_launch_ms'] * 1e-3) / 1e9) if traffic else None,
                        hbm_frac_of_8tbps=(traffic * max(1.0, round(pm.get('launches_per_iteration') or 1.0)) /
                                           (roof['avg_launch_ms'] * 1e-3) / 8e12) if traffic else None,
                        mfma_counters=dict(mfma_util_pct=pm['mfma_util_pct'], flops_issued_per_launch=pm['flops_issued_per_launch'],
                                           measured_in_run=pm['measured_in_run']),
                        hbm_top3=pmc_by_timer(pm['_all'], timings))
        else:
            roof.update(traffic=None, traffic_measured_in_run=False, traffic_source=None,
                        traffic_unavailable=(pm or {}).get('reason', 'N > 1'), mfma_counters=None)
        dtype = {'f32': 'f32', 'split': 'f32 via split bf16 MFMA (3 pieces fwd / 2 adjoint), fp32 accumulate',
                 'split_f16': 'f32 via split fp16 (fwd, 2 pieces) / bf16 (adjoint, 2 pieces) MFMA, fp32 accumulate'}[prec]
        out = dict(
            metric='particle_rollouts_per_sec', value=Bg * a.steps / dt, unit='rollouts/s',
            n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
            higher_is_better=True, scaling=primary, vs_baseline=None, dtype=dtype,
            data='synthetic',
            # `value` = the MEDIAN of `timed_blocks` blocks of exactly `steps` steps each (every block bracketed by barrier +
            # synchronize, max over ranks); the spread is the clock the part settles at
            value_min=Bg * a.steps / max(blocks), value_max=Bg * a.steps / min(blocks), timed_blocks=len(blocks),
            timed_seconds=float(sum(blocks)), sclk_mhz=None if a.no_sclk else sclk_mhz(leg),
            config=dict(workload='%s: D=%d U=%d pol=%s dyn=%s rows/GPU=%d (%s) H=%d mm=%s; full '
                                 'iteration = rollout fwd + loss + adjoint + dW + %sclip + Adam' %
                                 (a.config, d['x0'].shape[1], d['pol_z'].shape[1],
                                  PB.layer_dims(d, 'pol'), PB.layer_dims(d, 'dyn'), B,
                                  'particles x samples', H, bool(d['mm_states']),
                                  'RCCL all-reduce + ' if world > 1 else ''),
                        rows_per_gpu=B, global_rows=Bg, horizon=H, parallelism='dp%d' % world,
                        precision=prec, rows_per_wg=eng.info['rows_per_wg'], workgroups=eng.info['n_wg'],
                        cu_occupancy='%d of %d CUs hold a workgroup' % (min(eng.info['n_wg'], N_CUS), N_CUS),
                        mm_mode=eng.info['mm_mode'], mm_grid=eng.info.get('mm_grid', 0),
                        **({'mm_groups': 'one group over the rows of all ranks'} if a.mm_global else {}),
                        adjoint_sweep_launches=eng.info.get('dw_pipe', 1), replay=dict(zip(('on', 'fwd_calls_replayed', 'adjoint_calls_replayed'), (eng.info.get('replay', 0),) + eng.replay_count())), **({'debug_one_device': True} if a.one_device else {})),
            algorithmic_gflop_per_step=flops_rollout * B / 1e9,
            algorithmic_tflops=flops_rollout * Bg * a.steps / dt / 1e12,
            kernel_ms={k: round(vv, 4) for k, vv in timings.items()},
            roofline=roof)
        if world > 1:
            out['scaling_note'] = ('`value` is the %s curve: ' % primary) + (
                'every rank owns its own %d rows (global batch N x %d = %d rows); the STRONG curve -- the 100 x 25 = 2 500 rows '
                'BASELINE.json\'s metric names, dealt over the N ranks -- is under `strong` and is flat by construction: a '
                'sweep is %d sequential steps whatever the row count per GPU' % (B, B, Bg, H) if primary == 'weak' else
                'the %d rows of the metric dealt over the ranks; the weak curve (N x 2 500 rows) is under `weak`' % Bg)
            if a.config == 'cartpole_nomm' and world in PREDICTED_WEAK:
                out['predicted'] = dict(weak=PREDICTED_WEAK[world], source='DESIGN.md section 5 (before any multi-GPU run)')
            out['allreduce_selfcheck'] = 'passed: %d floats summed over %d ranks, bit-identical to the closed form' % (41602, world)
            # (always present in the parsed line: None = the transport asked for is the one that ran)
            out['transport_fallback'] = (transport_note + ' -> RCCL') if transport_note else None
            out['transport'] = a.transport
            out['rccl_ranks'] = rccl_ranks
            # the first 200 characters of config.workload are what a truncating reader keeps: which curve `value` is, the
            # global row count, the other curve's value and what the communicator saw go FIRST
            oth = extra.get('strong' if primary == 'weak' else 'weak')
            head = 'N=%d %s: global_rows=%d (%s), rccl_ranks=%s, transport=%s%s; %s; ' % (
                world, primary.upper(), Bg,
                ('%d x %d rows/GPU' % (world, B)) if primary == 'weak' else 'the metric\'s rows dealt over the ranks',
                rccl_ranks, a.transport, ' (FELL BACK from p2p)' if transport_note else '',
                ('%s curve (global_rows=%d) = %.4g rollouts/s' % ('STRONG' if primary == 'weak' else 'WEAK', oth['global_rows'], oth['value']))
                if oth else 'other curve not timed')
            out['config']['workload'] = head + out['config']['workload']
            out['collective'] = ('one-shot peer-to-peer all-reduce (pmbrl_p2p.hip, IPC-mapped slots) on the compute stream'
                                 if a.transport == 'p2p' else
                                 'RCCL ncclAllReduce through the C ABI on the compute stream' if rccl_ranks else
                                 'torch.distributed all_reduce (backend %s)' % a.dist_backend)
        out.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(d)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
