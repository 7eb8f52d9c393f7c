"""CPU study: which of the three GEMM classes of an iteration -- forward layers, adjoint dX chain, weight-gradient
GEMM -- sets the policy-gradient error of the split-operand arithmetic at the C5 shape (3 x 512, D = 32, H = 100)?
Each is emulated with split operands on its own (tools/split_precision_study.py's emulation of the matrix core:
piece products exact, fp32 accumulation) while the other two stay plain fp32, on 64 rows of the stress32 problem.

What it shows (profiles/r03_c5_precision_study.txt): the error moves ONLY with the forward's arithmetic, and in
jumps -- the same 3.9e-4 for every variant that rounds a particular pre-activation to the other side of zero than
fp64 does, 1e-5 .. 4e-7 for the variants that do not.  At this size the gradient error of any fp32-class arithmetic
is a count of ReLU units whose pre-activation changes sign under rounding, not a smooth function of the piece
count of the adjoint or the dW GEMM.

    python tools/c5_precision_study.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import split_precision_study as S  # noqa: E402
from prob_mbrl_amd.problem import synthetic_problem  # noqa: E402


def combo(fwd, dx, dw):
    mms = {role: (S.make_mm(*spec, (role,)) if spec else None) for role, spec in (('fwd', fwd), ('dx', dx), ('dw', dw))}

    def mm(a, b, role='fwd'):
        f = mms.get(role)
        if f is None:
            return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
        return f(a, b, role)
    return mm


def main():
    lines = ['C5 shape, 64 rows, H = 100: one GEMM class at a time on split operands, the rest plain fp32; vs the fp64 oracle',
             '(the gradient column is a count of ReLU units that change sign under rounding -- it moves in jumps; the',
             ' states column is the smooth measure of the forward arithmetic)']
    for seed in (0, 1):
        prob = synthetic_problem('stress32', seed=seed, data_seed=seed, P=1, S=64, H=100)
        L64, S64, g64 = S.run(prob, np.float64)
        lines.append('seed %d' % seed)
        lines.append('%-54s %10s %10s %10s' % ('arithmetic (forward | adjoint dX | dW)', 'loss', 'states', 'grad'))
        main_one(prob, L64, S64, g64, lines)
    with open(os.path.join(ROOT, 'profiles', 'r04_c5_precision_study.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


def main_one(prob, L64, S64, g64, lines):
    F16x2, BFx2, BFx3, F16x2o2 = ('f16', 2, 2, 1), ('bf16', 2, 2, 1), ('bf16', 3, 3, 2), ('f16', 2, 2, 2)
    F16x2s = ('f16s', 2, 2, 1)
    for name, mm in [
            ('fp32 | fp32 | fp32', S.make_mm('bf16', 1, 1, 0, ())),
            ('f16x2 | bf16x2 | bf16x2   (the default)', combo(F16x2, BFx2, BFx2)),
            ('f16x2 | fp32 | fp32', combo(F16x2, None, None)),
            ('f16x2, w.lo * 2^11 | fp32 | fp32', combo(F16x2s, None, None)),
            ('f16x2, w.lo * 2^11 | bf16x2 | bf16x2  (r04 default)', combo(F16x2s, BFx2, BFx2)),
            ('fp32 | bf16x2 | fp32', combo(None, BFx2, None)),
            ('fp32 | fp32 | bf16x2', combo(None, None, BFx2)),
            ('f16x2 + lo.lo (4 MFMA) | fp32 | fp32', combo(F16x2o2, None, None)),
            ('bf16x3 (6 MFMA) | fp32 | fp32', combo(BFx3, None, None))]:
        L, St, g = S.run(prob, np.float32, mm)
        lines.append('%-54s %10.2e %10.2e %10.2e' % (name, abs(L - L64) / abs(L64), S.rel(St, S64), S.rel(g, g64)))
        print(lines[-1], flush=True)


if __name__ == '__main__':
    main()
