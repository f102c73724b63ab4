"""Pins for the two pieces of arithmetic this build introduces in place of third-party code:
the counter-based RNG (replaces rand 0.8.5 / rand_chacha 0.3.1, Cargo.lock:34-62, which the reference
seeds from the OS and therefore cannot be matched) and rt_math (replaces the platform libm)."""
import ctypes as C

import numpy as np


def test_philox4x32_10_random123_known_answers(orc):
    """Known-answer vectors published with Random123 (Salmon et al., SC'11) for philox4x32-10."""
    lib = orc.load()
    cases = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in cases:
        c = (C.c_uint32 * 4)(*ctr)
        lib.oracle_philox4x32_10(c, key[0], key[1])
        assert tuple(c) == want


def test_splitmix64_known_answers(orc):
    """SplitMix64 (scene-construction stream), seed 1234567: the sequence quoted in the xoshiro seeding notes."""
    lib = orc.load()
    st = C.c_uint64(1234567)
    got = [lib.oracle_splitmix64_next(C.byref(st)) for _ in range(5)]
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431, 16408922859458223821]


def _stream(orc, seed, pixel, sample, n):
    out = np.empty(n)
    orc.load().oracle_sample_stream(seed, pixel, sample, n, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_sample_streams(orc):
    a = _stream(orc, 1, 10, 3, 4096)
    assert ((a >= 0.0) & (a < 1.0)).all()
    assert np.array_equal(a, _stream(orc, 1, 10, 3, 4096))            # pure function of (seed, pixel, sample)
    assert (a * 2.0 ** 53 == np.floor(a * 2.0 ** 53)).all()           # 53-bit grid: (u64 >> 11) * 2^-53
    for other in (_stream(orc, 2, 10, 3, 4096), _stream(orc, 1, 11, 3, 4096), _stream(orc, 1, 10, 4, 4096)):
        assert not np.array_equal(a, other) and abs(np.corrcoef(a, other)[0, 1]) < 0.06
    assert abs(a.mean() - 0.5) < 0.02 and abs(a.var() - 1 / 12) < 0.01
    # first draws across neighbouring pixels / samples are uniform too (inter-stream quality)
    firsts = np.array([_stream(orc, 1, p, s, 1)[0] for p in range(64) for s in range(32)])
    assert abs(firsts.mean() - 0.5) < 0.02 and abs(firsts.var() - 1 / 12) < 0.01
    hist, _ = np.histogram(firsts, bins=16, range=(0, 1))
    assert hist.min() > 80 and hist.max() < 180  # 2048 draws, 128 expected per bin


def _ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    d = np.abs(a.view(np.int64) - b.view(np.int64))  # same-sign neighbours: exact in int64
    return np.where(same, 0, d)


def test_rt_math_against_libm(orc):
    """rt_math is specified as 'a libm': check <= 2 ulp against numpy (glibc) over the ranges the path uses."""
    rng = np.random.default_rng(11)
    n = 400000
    x = np.concatenate([rng.uniform(-1, 1, n), rng.uniform(-200, 200, n), rng.uniform(-6e4, 6e4, n)])  # 10*p, |p| <= 5000
    assert _ulp_diff(orc.rt_math("sin", x), np.sin(x)).max() <= 2
    assert _ulp_diff(orc.rt_math("cos", x), np.cos(x)).max() <= 2
    t = rng.uniform(-1.4, 1.4, n)
    assert _ulp_diff(orc.rt_math("tan", t), np.tan(t)).max() <= 3
    u = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 1e-12, n)])
    u = u[u > 0]
    assert _ulp_diff(orc.rt_math("log", u), np.log(u)).max() <= 2
    a = np.concatenate([rng.uniform(-1, 1, n), 1 - rng.uniform(0, 1e-8, 1000), -1 + rng.uniform(0, 1e-8, 1000)])
    assert _ulp_diff(orc.rt_math("acos", a), np.arccos(a)).max() <= 2
    yy, xx = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    assert _ulp_diff(orc.rt_math("atan2", yy, xx), np.arctan2(yy, xx)).max() <= 2
    assert np.array_equal(orc.rt_math("sqrt", np.abs(x)), np.sqrt(np.abs(x)))


def test_rt_math_special_values(orc):
    inf, nan = float("inf"), float("nan")
    assert orc.rt_math("log", [0.0])[0] == -inf                       # ln(0): the medium's free path becomes +inf -> miss
    assert orc.rt_math("log", [1.0])[0] == 0.0
    assert np.isnan(orc.rt_math("log", [-1.0])[0])
    assert orc.rt_math("acos", [1.0])[0] == 0.0
    assert orc.rt_math("acos", [-1.0])[0] == np.pi
    assert np.isnan(orc.rt_math("acos", [1.0000000000000002])[0])     # |x| > 1 -> NaN, as libm
    assert orc.rt_math("atan2", [0.0], [-1.0])[0] == np.pi
    assert orc.rt_math("atan2", [-0.0], [-1.0])[0] == -np.pi
    assert orc.rt_math("atan2", [1.0], [0.0])[0] == np.pi / 2
    assert orc.rt_math("sin", [0.0])[0] == 0.0 and orc.rt_math("cos", [0.0])[0] == 1.0
    assert np.isnan(orc.rt_math("sin", [inf])[0]) and np.isnan(orc.rt_math("cos", [nan])[0])


def test_sign_of_sine_is_exact(orc):
    """rt_sin_sign (the checker texture's sign test without the polynomials) must equal the sign of rt_sin
    everywhere, including next to multiples of pi/2, at zero, and for non-finite input (2 = NaN)."""
    rng = np.random.default_rng(11)
    k = rng.integers(-200000, 200000, 400000).astype(np.float64)
    near = k * (np.pi / 2)
    near = np.concatenate([near, np.nextafter(near, np.inf), np.nextafter(near, -np.inf),
                           near * (1 + rng.uniform(-1e-12, 1e-12, near.size))])
    x = np.concatenate([rng.uniform(-1, 1, 200000), rng.uniform(-60000, 60000, 400000), near,
                        [0.0, -0.0, 1e-300, -1e-300, 0.7853981633974483, -0.7853981633974483, 1.5e6, -1.5e6,
                         np.inf, -np.inf, np.nan, 1e301]])
    s = orc.rt_math("sin", x)
    want = np.where(np.isnan(s), 2.0, np.sign(s))
    got = orc.rt_math("sin_sign", x)
    assert np.array_equal(got, want), "%d mismatches" % int((got != want).sum())
