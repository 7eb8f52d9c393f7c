"""CPU: the statistics-exchange statement of moment matching over ranks (oracle/span_np.py, the algorithm of
prob_mbrl_amd/csrc/pmbrl_mmx.h) is mm_resample_ over the concatenated rows (oracle/adjoint_np.mm_forward /
mm_backward, pinned to the reference by tests/test_oracle_golden.py) -- forward and adjoint, unequal parts,
infer_noise_variables, and the nearly-constant 1 x 1 case in which raw second moments cancel to noise."""
import numpy as np
import pytest

from oracle import adjoint_np as A
from oracle import span_np as S


def _split(x, parts):
    out, o = [], 0
    for n in parts:
        out.append(x[o:o + n])
        o += n
    return out


@pytest.mark.parametrize('d,parts,infer_ns', [(4, (13, 12), False), (6, (20, 17, 13), False), (1, (7, 30, 3, 10), False),
                                             (5, (9, 15), True), (32, (40, 60), False)])
def test_span_statement_equals_single_process(d, parts, infer_ns):
    rng = np.random.default_rng(d * 100 + len(parts))
    M = sum(parts)
    s = rng.standard_normal((M, d)) @ rng.standard_normal((d, d)) * 0.3 + 5.0 * rng.standard_normal(d)
    z = rng.standard_normal((M, d))
    g = rng.standard_normal((M, d))
    out, cache = A.mm_forward(s, z, infer_ns)
    gs = A.mm_backward(g, cache)
    outs, cache_s = S.mm_forward_span(_split(s, parts), _split(z, parts), infer_ns)
    gss = S.mm_backward_span(_split(g, parts), cache_s)
    assert np.allclose(np.concatenate(outs), out, rtol=1e-10, atol=1e-10)
    assert np.allclose(np.concatenate(gss), gs, rtol=1e-8, atol=1e-10)


def test_centred_combination_survives_nearly_equal_rewards():
    """Rewards that differ in the 9th digit (a converged cart-pole): variance 1e-18 next to a mean of 1.  The slots
    carry centred moments, so the combined variance is exact to fp64 relative precision; sum r^2 - M mean^2 is not."""
    rng = np.random.default_rng(7)
    parts = (25, 25, 25, 25)
    r = (1.0 - 1e-9 * rng.random((100, 1)))
    z = rng.standard_normal((100, 1))
    M, mean, Sg, _, _ = S.combine([S.slot(a, b) for a, b in zip(_split(r, parts), _split(z, parts))])
    direct = np.var(r[:, 0], ddof=1) + 1e-12
    assert abs(Sg[0, 0] - direct) <= 1e-9 * direct
    raw = ((r * r).sum() - M * mean[0] ** 2) / (M - 1)
    assert abs(raw - np.var(r[:, 0], ddof=1)) > 1e-3 * np.var(r[:, 0], ddof=1)      # what the slots avoid
