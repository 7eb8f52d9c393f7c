"""CPU study (no GPU needed): what does running the hidden->hidden products of the rollout on the
bf16 / fp16 matrix cores with SPLIT operands cost in accuracy?

Every fp32 operand v is written as a sum of n low-precision pieces (v = p1 + p2 + ... exactly or
nearly so) and the product a.b is replaced by the sum of the piece products whose combined order is
small enough; the piece products are exact in fp32 and accumulate in fp32, which is what
v_mfma_f32_16x16x32_bf16 / _f16 do.  The emulation replaces oracle.adjoint_np._mm for the products
named in `roles` and reports, against the fp64 oracle on the same problem,
    loss rel. error, trajectory rel-L2, policy-gradient rel-L2
next to the plain-fp32 numbers (the noise floor of the reference's own arithmetic).

    python tools/split_precision_study.py            # table on stdout, also profiles/r02_split_precision_study.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import adjoint_np as AN  # noqa: E402
from prob_mbrl_amd.problem import synthetic_problem  # noqa: E402


def bf16_trunc(v):
    return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_rne(v):
    u = v.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(v, n, kind):
    """n pieces of fp32 array v (C-contiguous fp32).  kind 'f16s': fp16 pieces with the LOW piece taken as
    fp16((v - hi) * 2^11) / 2^11 -- the scaled low piece of the device's weights (pmbrl_split.h, PM_F16_LO_SCALE):
    out of fp16's subnormal range, so it keeps its eleven bits for |v| < 0.125 too."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    out, r = [], v
    if kind == 'f16s':
        assert n == 2
        hi = v.astype(np.float16).astype(np.float32)
        lo = ((v - hi).astype(np.float32) * np.float32(2048.0)).astype(np.float16).astype(np.float32) / np.float32(2048.0)
        return [hi, lo.astype(np.float32)]
    for _ in range(n):
        if kind == 'bf16t':
            p = bf16_trunc(r)
        elif kind == 'bf16':
            p = bf16_rne(r)
        elif kind == 'f16':
            p = r.astype(np.float16).astype(np.float32)
        else:
            raise ValueError(kind)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def make_mm(kind, na, nb, max_order, roles, min_k=32):
    """a (activations / gradients) in na pieces, b (weights, or the second stash for 'dw') in nb
    pieces; piece products (i, j) with i + j <= max_order are kept (0-based orders)."""
    def mm(a, b, role='fwd'):
        if role not in roles or a.shape[1] < min_k:
            return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)
        # ('f16s': only the weights -- the second operand -- carry the scaled low piece)
        pa, pb = split(a, na, 'f16' if kind == 'f16s' else kind), split(b, nb, kind)
        acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
        # smallest terms first, like an accumulator chain that ends with the leading product
        pairs = sorted(((i, j) for i in range(na) for j in range(nb) if i + j <= max_order),
                       key=lambda ij: -(ij[0] + ij[1]))
        for i, j in pairs:
            acc = (acc + pa[i] @ pb[j]).astype(np.float32)
        return acc
    return mm


def run(prob, dtype, mm=None):
    P = AN.Problem(prob, dtype=dtype)
    old = AN._mm
    if mm is not None:
        AN._mm = mm
    try:
        st = AN.forward(P)
        g, gx0, _ = AN.backward(P, st)
        return AN.loss(P, st), np.stack(st['states']), g
    finally:
        AN._mm = old


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


MODES = [
    # name, kind, pieces(a), pieces(b), max order, MFMAs per fp32-equivalent K=32 block
    ('bf16 x1 (plain)', 'bf16', 1, 1, 0, 1),
    ('bf16 x2 (3 mfma)', 'bf16', 2, 2, 1, 3),
    ('bf16 x2 trunc', 'bf16t', 2, 2, 1, 3),
    ('bf16 a3.w2 (5 mfma)', 'bf16', 3, 2, 2, 5),
    ('bf16 x3 (6 mfma)', 'bf16', 3, 3, 2, 6),
    ('bf16 x3 trunc', 'bf16t', 3, 3, 2, 6),
    ('f16 x1 (plain)', 'f16', 1, 1, 0, 1),
    ('f16 x2 (3 mfma)', 'f16', 2, 2, 1, 3),
    ('f16 x2 (4 mfma)', 'f16', 2, 2, 2, 4),
    ('f16 x2, w.lo * 2^11', 'f16s', 2, 2, 1, 3),
]


def main():
    cases = [
        ('cartpole_nomm 16x4 H=40', synthetic_problem('cartpole_nomm', seed=0, P=16, S=4, H=40)),
        ('cartpole_mm 4x25 H=40', synthetic_problem('cartpole_mm', seed=1, P=4, S=25, H=40)),
        ('dcartpole_mm 2x50 H=60', synthetic_problem('dcartpole_mm', seed=2, P=2, S=50, H=60)),
        ('stress32 4x8 H=30', synthetic_problem('stress32', seed=3, P=4, S=8, H=30)),
    ]
    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)

    out('split-operand matrix-core arithmetic vs the fp64 oracle (emulated on the CPU)')
    out('roles: fwd+dx = forward layers and adjoint chain with K >= 32; +dw = also the weight-gradient GEMM')
    for cname, prob in cases:
        L64, S64, g64 = run(prob, np.float64)
        out('')
        out('%s   (|g| = %.3e)' % (cname, np.linalg.norm(g64)))
        out('%-24s %-8s %12s %12s %12s' % ('arithmetic', 'roles', 'loss', 'states', 'grad'))
        L, S, g = run(prob, np.float32, make_mm('bf16', 1, 1, 0, ()))
        out('%-24s %-8s %12.2e %12.2e %12.2e' % ('fp32 (reference math)', '-', abs(L - L64) / abs(L64), rel(S, S64), rel(g, g64)))
        for name, kind, na, nb, mo, _ in MODES:
            for roles in (('fwd', 'dx'), ('fwd', 'dx', 'dw')):
                L, S, g = run(prob, np.float32, make_mm(kind, na, nb, mo, roles))
                out('%-24s %-8s %12.2e %12.2e %12.2e' % (name, '+'.join(r for r in roles if r != 'fwd') if len(roles) > 1 else 'fwd',
                                                      abs(L - L64) / abs(L64), rel(S, S64), rel(g, g64)))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles',
                        'r02_split_precision_study.txt')
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
