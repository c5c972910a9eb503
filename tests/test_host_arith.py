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


def test_reciprocal_fma_division_for_any_map_size_up_to_256():
    """the check above covers the divisors of C1 (160x512 -> 20x64) and C2 (376x1232 = pipeline.SLAM_SIZE -> 47x154) —
    the two geometries the product runs. The kernel takes any H, W: every divisor S - 1 for map sizes S = 2 .. 256 (pyramid
    levels of images up to 2048 pixels), on a 1/8-pixel lattice from -16 to S + 16 plus the 1/64 neighbours of every integer
    position (where a wrong last bit would move a floor)."""
    bad = total = 0
    for b_int in range(1, 256):
        b = Fraction(b_int)
        rcp = _rn32(1 / b)
        ks = set(range(-16 * 64, (b_int + 1 + 16) * 64 + 1, 8))
        for i in range(-16, b_int + 18):
            ks.update((64 * i - 1, 64 * i + 1))
        for k in ks:
            t = _rn32(2 * Fraction(k, 64))
            q0 = _rn32(t * rcp)
            q = _rn32(_rn32(-q0 * b + t) * rcp + q0)
            total += 1
            bad += q != _rn32(t / b)
    assert total > 300000 and bad == 0, (bad, total)


def _chain32(c, size):
    """The reference's coordinate round trip for one axis in fp32, one rounding per operation (corr.py:43-49: 2 pos / (S - 1) - 1;
    grid_sample, align_corners=True: ((g + 1) / 2) (S - 1))."""
    f = np.float32
    g = f(2) * c / f(size - 1) - f(1)
    return (g + f(1)) / f(2) * f(size - 1)


def _sample1d(row, u):
    """1-D linear interpolation with zero padding at positions u (fp32), as grid_sample forms it: weights from u - floor(u)."""
    x0 = np.floor(u)
    w = (u - x0).astype(np.float32)
    i0 = x0.astype(np.int64)
    pad = np.concatenate([[0.0], row, [0.0]]).astype(np.float32)
    a = pad[np.clip(i0 + 1, 0, len(row) + 1)] * ((i0 >= 0) & (i0 < len(row)))
    b = pad[np.clip(i0 + 2, 0, len(row) + 1)] * ((i0 + 1 >= 0) & (i0 + 1 < len(row)))
    return (np.float32(1) - w) * a.astype(np.float32) + w * b.astype(np.float32)


def test_one_fractional_part_per_window_stays_within_a_few_ulp_of_the_per_offset_chains():
    """Contract of lookup_fused.hip's sampling unit: the reference rounds the normalise / denormalise round trip
    separately for each of the 9 offsets of an axis (bilinear_sampler, utils.py:63-70); the kernel evaluates the chain ONCE, for
    offset 0, and uses floor(u0) + d and frac(u0) for every offset. The two positions differ by at most a few ulp of the
    coordinate, bilinear interpolation is continuous (also across an integer, where the two floors may differ), so the samples
    differ by at most |du| x the largest step between neighbouring cells. Checked in fp32 emulation on near-integer coordinates
    (exact integers, +-1 ulp, +-1e-6, +-1e-4), on ordinary ones, and outside the map, for every pyramid size of C1 / C2 and for
    odd sizes: |sample_shared - sample_per_offset| <= 4 ulp(S) x max step, per axis."""
    rng = np.random.default_rng(5)
    f = np.float32
    worst = 0.0
    for size in (154, 77, 38, 19, 47, 23, 11, 5, 64, 32, 16, 8, 20, 10, 200, 97):
        row = rng.uniform(-1, 1, size).astype(np.float32)
        step = float(np.abs(np.diff(np.concatenate([[0.0], row, [0.0]]))).max())
        ints = np.arange(-6, size + 6, dtype=np.float32)
        cs = [ints, np.nextafter(ints, f(1e9)), np.nextafter(ints, f(-1e9)), ints + f(1e-6), ints - f(1e-6), ints + f(1e-4),
              ints - f(1e-4), rng.uniform(-6, size + 6, 4000).astype(np.float32)]
        c = np.concatenate(cs).astype(np.float32)
        u0 = _chain32(c, size)
        fl0 = np.floor(u0)
        fr0 = (u0 - fl0).astype(np.float32)
        ulp = float(np.spacing(f(size)))
        for d in range(-4, 5):
            ref = _sample1d(row, _chain32(c + f(d), size))
            shared_pos = fl0 + f(d)                                   # the kernel: integer origin + d, ONE fractional part
            i0 = shared_pos.astype(np.int64)
            pad = np.concatenate([[0.0], row, [0.0]]).astype(np.float32)
            a = pad[np.clip(i0 + 1, 0, size + 1)] * ((i0 >= 0) & (i0 < size))
            b = pad[np.clip(i0 + 2, 0, size + 1)] * ((i0 + 1 >= 0) & (i0 + 1 < size))
            got = (f(1) - fr0) * a.astype(np.float32) + fr0 * b.astype(np.float32)
            err = float(np.abs(got - ref).max())
            worst = max(worst, err / (ulp * step))
            assert err <= 4.0 * ulp * step + 1e-7, (size, d, err, ulp * step)
    assert worst > 0.0   # the two forms DO differ in the last bits: this is a bound, not an identity
