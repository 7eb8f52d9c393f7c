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
global batch of N*2500; --scaling strong: the 100 x 25 = 2500 rows of BASELINE.json's
metric are divided over the N ranks (whole particle groups per rank).

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
    ap.add_argument('--cpu-only', action='store_true', help='only time the CPU baseline (no GPU needed)')
    ap.add_argument('--timing-steps', type=int, default=10)
    # debugging aids for the N>1 control flow on a box with ONE GPU (tests/test_gpu_api.py):
    # every rank on cuda:0, collectives over gloo.  Never used for a reported number.
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
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=a.dist_backend, rank=rank, world_size=world)
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % a.gpus
    dev = torch.device('cuda:%d' % local_rank)
    torch.cuda.set_device(dev)

    from prob_mbrl_amd import engine as E
    if a.scaling == 'strong' and world > 1:
        # the SAME global problem for every N: its particle groups are dealt to the ranks
        cfg = PB.CONFIGS[a.config]
        assert cfg['P'] % world == 0, 'strong scaling: the particle count must divide by the GPU count'
        dg = PB.synthetic_problem(a.config, seed=0, data_seed=0)
        d = PB.shard_problem(dg, rank, world)
    else:
        d = PB.synthetic_problem(a.config, seed=0, data_seed=rank)
    B = d['x0'].shape[0]
    H = int(d['H'])
    Bg = B * world
    mm_span = None
    if a.mm_global:
        assert bool(d['mm_states']) and a.scaling == 'weak', '--mm-global: a moment-matching config, weak scaling'
        d = dict(d)
        d['mm_groups'] = 0
        if world > 1:
            mm_span = (Bg, rank * B, world, rank)
    if world > 1 and bool(d['mm_states']) and a.scaling == 'weak':
        # one cyclic noise buffer over the GLOBAL rows (utils/rollout.py:53-59), the same on every rank
        d = dict(d)
        gen = np.random.default_rng(12345)
        d['z_mm'] = gen.standard_normal((H + Bg, d['x0'].shape[1])).astype(np.float32)
        d['z_rr'] = gen.standard_normal((H + Bg, 1)).astype(np.float32)
    eng, args, _ = PB.engine_from_problem(d, dev, rows_per_wg_hint=a.rows_per_wg, B_global=Bg,
                                          row_offset=rank * B, precision=a.precision, mm_span=mm_span)
    if mm_span:
        eng.attach_collective(dist.group.WORLD)
    gw = torch.tensor(PB.loss_weights(d, Bg)[:, :B].copy(), device=dev)
    params = args['pol_flat'].clone()
    args['pol_flat'] = params
    m = torch.zeros_like(params)
    v = torch.zeros_like(params)
    loss_buf = torch.zeros(1, device=dev)
    state = dict(step=0)

    if world > 1:
        from prob_mbrl_amd.distributed import grad_allreduce
        allreduce = grad_allreduce(None, dev)      # RCCL through the C ABI, on the compute stream

    def step():
        state['step'] += 1
        _, _, R = eng.forward(**args)
        eng.weighted_sum(R, gw, out=loss_buf)
        g, _, _ = eng.backward(gw)
        if world > 1:
            allreduce(g)
        E.clip_adam(params, g, m, v, state['step'], 1e-4, max_norm=1.0)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert eng.valid_steps() == H, 'numerical failure inside the benchmark rollout'
    assert bool(torch.isfinite(params).all()) and bool(torch.isfinite(loss_buf).all())

    # ---- per-kernel durations (HIP events on the launch stream), outside the timed region.
    # Every rank runs these steps (step() contains the gradient all-reduce); rank 0 reports its own.
    eng.set_timing(True)
    acc = {}
    for _ in range(a.timing_steps):
        step()
        for k, ms in eng.read_timing().items():
            if ms >= 0:
                acc.setdefault(k, []).append(ms)
    eng.set_timing(False)
    timings = {k: float(np.mean(vv)) for k, vv in acc.items()}
    sync()

    if rank == 0:
        flops_rollout, Pm, Fm = PB.algorithmic_flops_per_rollout(d)
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
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes);
        # only valid for the configuration they were collected on
        traffic, traffic_src = None, None
        prec = eng.info['precision']
        try:
            src = 'profiles/r02_pmc_traffic_%s.json' % prec
            pmc = json.load(open(os.path.join(ROOT, src)))
            if a.config == 'cartpole_nomm' and world == 1 and kname in pmc['kernels']:
                traffic = pmc['kernels'][kname]['hbm_bytes_per_launch']
                traffic_src = src + ' (rocprofv3 --pmc passes of this command, not measured in this run)'
        except Exception:
            traffic = None
        # peak the dominant kernel is priced against: the dense MFMA peak of the instruction it issues --
        # exact fp32 MFMA, or fp16 / bf16 MFMA at THREE instructions per fp32-equivalent product (two-piece
        # split operands; the three-piece bf16 forward of 'split' issues six)
        mfma_per_product = {('f32', 'fwd'): 1, ('f32', 'bwd'): 1, ('split', 'fwd'): 6, ('split', 'bwd'): 3,
                            ('split_f16', 'fwd'): 3, ('split_f16', 'bwd'): 3}.get((prec, dom), 1)
        if prec == 'f32' or dom == 'dw':
            peak, peak_note = PEAK_F32_MFMA_TFLOPS, 'v_mfma_f32_16x16x4_f32 dense peak'
        else:
            peak = PEAK_F16_MFMA_TFLOPS / mfma_per_product
            peak_note = ('v_mfma_f32_16x16x32_%s dense peak (2500 TFLOP/s) / %d MFMAs per fp32-equivalent product' %
                         ('f16' if (prec == 'split_f16' and dom == 'fwd') else 'bf16', mfma_per_product))
        dtype = {'f32': 'f32', 'split': 'f32 via split bf16 MFMA (3 pieces fwd / 2 adjoint), fp32 accumulate',
                 'split_f16': 'f32 via split fp16 (fwd, 2 pieces) / bf16 (adjoint, 2 pieces) MFMA, fp32 accumulate'}[prec]
        out = dict(
            metric='particle_rollouts_per_sec', value=Bg * a.steps / dt, unit='rollouts/s',
            n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
            higher_is_better=True, scaling=a.scaling if world > 1 else 'weak', vs_baseline=None, dtype=dtype,
            data='synthetic',
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
                        adjoint_sweep_launches=eng.info.get('dw_pipe', 1), **({'debug_one_device': True} if a.one_device else {})),
            algorithmic_gflop_per_step=flops_rollout * B / 1e9,
            algorithmic_tflops=flops_rollout * Bg * a.steps / dt / 1e12,
            kernel_ms={k: round(vv, 4) for k, vv in timings.items()},
            roofline=dict(bound='mfma', kernel=kname,
                          achieved=achieved, peak=peak, unit='TFLOP/s',
                          frac=achieved / peak, traffic=traffic,
                          traffic_source=traffic_src,
                          peak_is=peak_note, frac_of_f32_mfma_peak=achieved / PEAK_F32_MFMA_TFLOPS,
                          # what binds this kernel at this size is not the matrix pipe: per-workgroup latency
                          # of H sequential steps on %d of 256 CUs, and the weight stream L2 -> CU
                          binding='latency: H sequential steps per workgroup, %d of %d CUs occupied; '
                                  'weight stream L2->CU inside a step' % (min(eng.info['n_wg'], N_CUS), N_CUS),
                          flops_per_launch=kflops[dom], avg_launch_ms=timings[dom]))
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(d)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
