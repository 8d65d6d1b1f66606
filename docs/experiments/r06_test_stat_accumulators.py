"""The order-independent statistics accumulators (include/sdt_hip.h "Statistics accumulators", csrc/common.h sdt_stat_add / sdt_stat_get) restated in
Python floats (IEEE doubles, round to nearest: the arithmetic of the device code): a partial is cut EXACTLY at the window boundaries, every window
addition is exact, so the four doubles of a statistic do not depend on the order of arrival -- the property that makes the normalisation statistics
deterministic by construction (reference: main.py:37-38, cudnn.deterministic = True).  CPU test of the format; the kernels are checked against
float64 sums and for bitwise repeatability in tests/test_ops_gpu.py."""
import random
from fractions import Fraction

import numpy as np

MAGIC = [1.5 * 2.0 ** (b + 52) for b in (-84, -48, -12, 24)]  # (x + m) - m: x to the nearest multiple of 2^b
BASE = (-84, -48, -12, 24)


def pieces(x):
    """sdt_stat_add: the pieces of x for windows 0..3 (None: |x| >= 2^60 or non-finite -> the statistic becomes NaN)."""
    if not (abs(x) < 2.0 ** 60):
        return None
    p3 = (x + MAGIC[3]) - MAGIC[3]
    r3 = x - p3
    p2 = (r3 + MAGIC[2]) - MAGIC[2]
    r2 = r3 - p2
    p1 = (r2 + MAGIC[1]) - MAGIC[1]
    r1 = r2 - p1
    p0 = (r1 + MAGIC[0]) - MAGIC[0]
    return [p0, p1, p2, p3], r1 - p0


def value(w):
    """sdt_stat_get"""
    return ((w[3] + w[2]) + w[1]) + w[0]


def _partials(rng, n, kind):
    scale = 10.0 ** rng.uniform(-22, 15)
    if kind == "f32":  # a conv epilogue's fp32 partial sums
        return [float(np.float32(rng.gauss(0, 1) * scale * 10.0 ** rng.uniform(-3, 3))) for _ in range(n)]
    return [sum(float(np.float32(rng.gauss(0, 1) * scale)) for _ in range(4)) for _ in range(n)]  # a workgroup's fp64 sum of four fp32 numbers


def test_a_partial_is_cut_exactly_and_every_piece_fits_its_window():
    rng = random.Random(1)
    for kind in ("f32", "f64"):
        for x in _partials(rng, 2000, kind) + [0.0, -0.0, 2.0 ** 59, -(2.0 ** 60 - 2.0 ** 8), 2.0 ** -84, 2.0 ** -86, 3.0 * 2.0 ** -85]:
            ps, dropped = pieces(x)
            assert Fraction(x) == sum(Fraction(p) for p in ps) + Fraction(dropped)
            assert abs(dropped) <= 2.0 ** -85
            for p, b in zip(ps, BASE):
                assert (Fraction(p) / Fraction(2) ** b).denominator == 1 and abs(p) <= 2.0 ** (b + 36), (x, p, b)
    assert pieces(2.0 ** 60) is None and pieces(float("inf")) is None and pieces(float("nan")) is None


def test_window_sums_are_exact_and_independent_of_the_arrival_order():
    rng = random.Random(2)
    for trial in range(400):
        vals = _partials(rng, rng.randint(1, 400), "f32" if trial % 2 else "f64")

        def accumulate(order):
            w = [0.0] * 4
            for k in order:
                for j, p in enumerate(pieces(vals[k])[0]):
                    if p != 0.0:
                        before = Fraction(w[j])
                        w[j] += p
                        assert Fraction(w[j]) == before + Fraction(p)  # no rounding: the addition is associative
            return w

        order = list(range(len(vals)))
        w1 = accumulate(order)
        rng.shuffle(order)
        assert accumulate(order) == w1
        exact = float(sum(Fraction(v) for v in vals))
        assert abs(value(w1) - exact) <= 4e-16 * max(abs(x) for x in w1 + [exact]) + len(vals) * 2.0 ** -85


def test_the_bound_on_the_number_of_pieces():
    """2^16 pieces of the largest magnitude a window takes still sum exactly (the documented limit of a statistic)."""
    for b in BASE:
        p = 2.0 ** (b + 36)
        s = p * 2 ** 16
        assert Fraction(s) == Fraction(p) * 2 ** 16 and s + 2.0 ** b != s  # the unit is still resolved: < 2^53 units
