"""Host-side checks of arithmetic identities the kernels rely on (no GPU, no library)."""
from fractions import Fraction

import numpy as np


def _rn32(x):
    """Fraction -> the nearest float32 (ties to even), as a Fraction: one rounding, no detour through float64."""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    if Fraction(2) ** (e + 1) <= x:
        e += 1
    e = max(e, -126)                       # subnormals share the exponent of the smallest normal
    ulp = Fraction(2) ** (e - 23)
    q = x / ulp
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (n & 1)):
        n += 1
    return s * n * ulp


def test_reciprocal_plus_one_fma_step_equals_ieee_division_for_this_network():
    """lookup_fused.hip evaluates the reference's 2*pos/(S-1) (utils.py:61-67 via grid_sample's normalisation) as
    q0 = t * RN(1/b); q = fma(fma(-q0, b, t), RN(1/b), q0). That equals RN(t/b) only if q0 is within an ulp of the quotient
    (ADVICE r3): checked here exhaustively for the divisors the four pyramid levels have at C1 (20x64) and C2 (47x154), over a
    1/64-pixel lattice of positions from -16 to S+16 — exact rational arithmetic, one rounding per operation."""
    sizes = set()
    for (h, w) in ((47, 154), (20, 64)):
        for l in range(4):
            sizes.add(h - 1); sizes.add(w - 1)
            h //= 2; w //= 2
    bad = 0
    total = 0
    for b_int in sorted(sizes):
        b = Fraction(b_int)
        rcp = _rn32(1 / b)
        for k in range(-16 * 64, (b_int + 1 + 16) * 64 + 1):
            pos = Fraction(k, 64)                       # exactly representable in fp32
            t = _rn32(2 * pos)
            q0 = _rn32(t * rcp)
            r = _rn32(-q0 * b + t)                      # fma: one rounding
            q = _rn32(r * rcp + q0)                     # fma: one rounding
            total += 1
            if q != _rn32(t / b):
                bad += 1
    assert total > 50000 and bad == 0, (bad, total)


def test_rn32_helper_matches_numpy():
    rng = np.random.default_rng(0)
    for v in rng.standard_normal(200) * 10.0 ** rng.integers(-20, 20, 200):
        assert float(_rn32(Fraction(float(v)))) == float(np.float32(v))
