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


def test_gen_range_pm1_shortcut_is_exact():
    """core/rng.hpp: rng_range_pm1 builds gen_range(-1.0..1.0) as bits(0x400.. | u >> 12) + -3 instead of rand 0.8.5's
    value1_2 * scale + offset = bits(0x3FF.. | u >> 12) * 2 + -3 (UniformFloat::sample_single; vec3.rs:288-294 draws it three
    times per attempt).  Every step of either form is exact, so the bits agree -- here on 4M random mantissas and the edges."""
    import numpy as np
    rs = np.random.RandomState(7)
    m = rs.randint(0, 1 << 52, size=1 << 22, dtype=np.int64).astype(np.uint64)
    m = np.concatenate([m, np.array([0, 1, (1 << 52) - 1, 1 << 51, (1 << 51) - 1, (1 << 51) + 1], dtype=np.uint64)])
    a = (m | np.uint64(0x3FF0000000000000)).view(np.float64) * 2.0 + -3.0
    b = (m | np.uint64(0x4000000000000000)).view(np.float64) + -3.0
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert a.min() == -1.0 and a.max() < 1.0
    m32 = np.arange(1 << 23, dtype=np.uint32)  # every float mantissa (the f32 fast mode)
    a32 = (m32 | np.uint32(0x3F800000)).view(np.float32) * np.float32(2.0) + np.float32(-3.0)
    b32 = (m32 | np.uint32(0x40000000)).view(np.float32) + np.float32(-3.0)
    assert np.array_equal(a32.view(np.uint32), b32.view(np.uint32))


def test_stream_against_an_independent_python_restatement(orc):
    """The whole stream definition of core/rng.hpp written out again in plain Python integers: Philox4x32-10 keyed by the render
    seed over (pixel_lo, pixel_hi, sample, 0) -> the 128-bit state of xoroshiro128+ (a = 24, b = 16, c = 37) -> gen::<f64>() =
    (u64 >> 11) * 2^-53.  Hand-checked anchor of the generator itself: from the state (1, 2) the outputs are 3 and 0x6001030003."""
    M64 = (1 << 64) - 1

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & M64

    def xoroshiro128plus(s0, s1):
        while True:
            yield (s0 + s1) & M64
            s1 ^= s0
            s0, s1 = rotl(s0, 24) ^ s1 ^ ((s1 << 16) & M64), rotl(s1, 37)

    g = xoroshiro128plus(1, 2)
    assert next(g) == 3 and next(g) == 0x6001030003

    def philox(c, k0, k1):
        c = list(c)
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
            k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
        return c

    for seed, pixel, sample in ((1, 10, 3), (0xDEADBEEF12345678, (1 << 33) + 5, 4_000_000_000), (0, 0, 0)):
        c = philox((pixel & 0xFFFFFFFF, pixel >> 32, sample, 0), seed & 0xFFFFFFFF, seed >> 32)
        s0, s1 = c[0] | (c[1] << 32), c[2] | (c[3] << 32)
        assert s0 | s1
        gen = xoroshiro128plus(s0, s1)
        want = np.array([(next(gen) >> 11) * 2.0 ** -53 for _ in range(64)])
        assert np.array_equal(_stream(orc, seed, pixel, sample, 64), want)
