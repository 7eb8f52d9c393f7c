import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import common
from oracle import ref_torch as R
from prob_mbrl_amd import problem as PB
DEV = torch.device('cuda:0')

def part_a():
    d = dict(common.load('mmg_h40'))
    H, B = int(d['H']), d['x0'].shape[0]
    k = 3
    for key in ('pol_W0', 'dyn_W0'):
        W = np.asarray(d[key]).copy(); W[:, k] = 0.0; d[key] = W
    for drift in (0.0, 1e3, 1e10, 5e37):
        my = np.asarray(d['dyn_my'], dtype=np.float32).copy(); my[k] = drift; d['dyn_my'] = my
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
        l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, H, gamma, True, True, True, meta['mm_groups'], z_mm, z_rr, n_steps=6)
        S64 = torch.stack(S64).detach().numpy()
        for generic in (False, True):
            for prec in ('f32', 'split_f16'):
                eng, args, _ = common.engine_from_fixture(d, DEV, force_generic=generic, precision=prec)
                S, A, Rw = eng.forward(**args)
                n = eng.valid_steps()
                Sd = S[:7].cpu().numpy()
                errs = [common.rel(Sd[t][:, :3], S64[t][:, :3]) for t in range(7)]
                print('drift %g generic %d prec %s n=%d mm_mode=%d errs %s' % (drift, generic, prec, n, eng.info['mm_mode'], ' '.join('%.1e' % e for e in errs)), flush=True)

def part_b():
    for P in (8,):
        d = dict(PB.synthetic_problem('stress32_mm', seed=0, data_seed=0, P=P, S=64))
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
        torch.set_num_threads(16)
        l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True, P, z_mm, z_rr)
        S64 = torch.stack(S64).detach().numpy()
        print('oracle |S| per step', ' '.join('%.2f' % np.abs(S64[t]).max() for t in range(0, 101, 10)))
        for prec in ('f32', 'split_f16'):
            eng, args, _ = PB.engine_from_problem(d, DEV, precision=prec)
            S, A, Rw = eng.forward(**args)
            n = eng.valid_steps()
            Sd = S.cpu().numpy()
            print('C5mm prec %s n=%d mm_mode %d rows/wg %d' % (prec, n, eng.info['mm_mode'], eng.info['rows_per_wg']))
            print(' errs', ' '.join('%.1e' % common.rel(Sd[t], S64[t]) for t in range(0, min(n, 100) + 1, 5)), flush=True)
            gw = torch.tensor(PB.loss_weights(d, 512), device=DEV)
            if n == 100:
                g = eng.backward(gw)[0].cpu().numpy()
                print(' grad err %.2e' % common.rel(g, g64.numpy()))

def part_c():
    d = dict(PB.synthetic_problem('stress32', seed=0, data_seed=0))
    eng, args, _ = PB.engine_from_problem(d, DEV)
    S, A, Rw = eng.forward(**args)
    print('C5 full n=%d rows/wg %d ws %.1f GB' % (eng.valid_steps(), eng.info['rows_per_wg'], eng.ws_bytes / 1e9), flush=True)
    gw = torch.tensor(PB.loss_weights(d, 16384), device=DEV)
    g = eng.backward(gw)[0].cpu().numpy().copy()
    m = torch.zeros_like(gw); m[:, :512] = gw[:, :512]
    g1 = eng.backward(m)[0].cpu().numpy().copy()
    g2 = eng.backward(gw - m)[0].cpu().numpy().copy()
    print('linearity', common.rel(g1 + g2, g), flush=True)



def part_d():
    """C5 no-mm: first 512 rows of the full 16384-row problem; engine f32 / split_f16 vs fp64 oracle; fp32 oracle floor"""
    import time
    from tests.test_gpu_full_size import _sub_rows, _oracle_on_first_rows
    for cfg in ('stress32_mm',):
        d = dict(PB.synthetic_problem(cfg, seed=0, data_seed=0))
        t = time.time()
        S64, l64, g64 = _oracle_on_first_rows(d, 512)
        print(cfg, 'oracle64 %.1fs |S|max %.1f' % (time.time() - t, np.abs(S64).max()), flush=True)
        e = _sub_rows(d, 512)
        try:
            x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(e, torch.float32)
            orig = R.get_z_rnd
            R.get_z_rnd = lambda z, i, m: z[torch.arange(i, i + m) % 16384]
            l32, g32, (S32, _, _) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'], meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
            R.get_z_rnd = orig
            print(' fp32 oracle floor: grad %.2e states %.2e' % (common.rel(g32.numpy() * 512 / 16384, g64), common.rel(torch.stack(S32).detach().numpy(), S64)), flush=True)
        except Exception as ex:
            R.get_z_rnd = orig
            print(' fp32 oracle fails', str(ex)[:100])
        for prec in ('f32', 'split_f16'):
            for hint in (16, 32):
                eng, args, _ = PB.engine_from_problem(e, DEV, B_global=16384, row_offset=0, precision=prec, rows_per_wg_hint=hint)
                S, A, Rw = eng.forward(**args)
                n = eng.valid_steps()
                gw = torch.tensor(PB.loss_weights(d, 16384)[:, :512].copy(), device=DEV)
                g = eng.backward(gw)[0].cpu().numpy()
                H = int(d['H'])
                print(' %s rows/wg %d mm_mode %d n=%d: states %.2e (t=20: %.2e, t=50: %.2e) grad %.2e' % (
                    prec, eng.info['rows_per_wg'], eng.info['mm_mode'], n, common.rel(S.cpu().numpy()[:n+1], S64[:n+1]),
                    common.rel(S.cpu().numpy()[20], S64[20]), common.rel(S.cpu().numpy()[50], S64[50]),
                    common.rel(g, g64) if n == H else -1), flush=True)


for p in sys.argv[1:]:
    globals()['part_' + p]()
