"""Accuracy of include/rpt_strict_math.h (the bit-reproducible libm stand-in) against
float64 references and against glibc's f32 functions.  CPU only."""
import numpy as np


def ulp_err(got, ref64):
    """|got - ref| in units of ulp(float32(ref))."""
    got = got.astype(np.float64)
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got - ref64) / ulp


def test_sincos_accuracy_on_path_domain(oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 2_000_000), (np.arange(1 << 16) / float(1 << 16) * 6.2831855)]).astype(np.float32)
    s, c = oracle.math(0, x), oracle.math(1, x)
    es = ulp_err(s, np.sin(x.astype(np.float64)))
    ec = ulp_err(c, np.cos(x.astype(np.float64)))
    # near the zeros of sin/cos the error is measured against a tiny ulp: bound the absolute error there
    big = np.abs(np.sin(x.astype(np.float64))) > 1e-3
    assert es[big].max() <= 1.5, es[big].max()
    big = np.abs(np.cos(x.astype(np.float64))) > 1e-3
    assert ec[big].max() <= 1.5, ec[big].max()
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.2e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 1.2e-7


def test_sincos_wider_range(oracle):
    rng = np.random.default_rng(2)
    x = rng.uniform(-3e4, 3e4, 1_000_000).astype(np.float32)
    assert np.abs(oracle.math(0, x) - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(oracle.math(1, x) - np.cos(x.astype(np.float64))).max() < 3e-7


def test_sincos_specials(oracle):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan], dtype=np.float32)
    s, c = oracle.math(0, x), oracle.math(1, x)
    assert s[0] == 0 and s[1] == 0 and c[0] == 1 and c[1] == 1
    assert np.isnan(s[2:]).all() and np.isnan(c[2:]).all()


def test_tan(oracle):
    x = np.linspace(0.01, 1.5, 100000).astype(np.float32)
    e = ulp_err(oracle.math(100, x), np.tan(x.astype(np.float64)))
    assert e.max() <= 3.0, e.max()
    # the value the camera uses: tan(radians(80) / 2)
    half = np.float32(80.0) * np.float32(np.float32(np.pi) / np.float32(180.0)) * np.float32(0.5)
    assert abs(float(oracle.math(100, np.array([half], dtype=np.float32))[0]) - np.tan(np.float64(half))) < 1e-7


def test_log2_is_almost_always_correctly_rounded(oracle):
    rng = np.random.default_rng(3)
    bits = rng.integers(0x00000001, 0x7f7fffff, size=3_000_000, dtype=np.uint32)      # every positive finite f32
    x = bits.view(np.float32)
    got = oracle.math(2, x)
    ref = np.log2(x.astype(np.float64))
    e = ulp_err(got, ref)
    assert e.max() <= 0.5001, e.max()
    assert (got == ref.astype(np.float32)).mean() > 0.999999


def test_log2_specials(oracle):
    x = np.array([0.0, -0.0, 1.0, np.inf, -1.0, np.nan, 2.0, 0.25, 1e-45], dtype=np.float32)
    g = oracle.math(2, x)
    assert g[0] == -np.inf and g[1] == -np.inf and g[2] == 0 and g[3] == np.inf
    assert np.isnan(g[4]) and np.isnan(g[5]) and g[6] == 1 and g[7] == -2
    assert abs(g[8] - np.log2(np.float64(np.float32(1e-45)))) < 1e-4


def test_pow_accuracy(oracle):
    rng = np.random.default_rng(4)
    n = 1_000_000
    x = np.concatenate([rng.uniform(0.0, 2.0, n), rng.uniform(1e-6, 1e-2, n), np.exp(rng.uniform(-80, 80, n))]).astype(np.float32)
    y = np.concatenate([rng.choice([2.2, 0.5, 0.4545], n), rng.uniform(0, 1, n), rng.uniform(-1.5, 1.5, n)]).astype(np.float32)
    got = oracle.math(3, x, y)
    with np.errstate(over="ignore", under="ignore"):
        ref = np.power(x.astype(np.float64), y.astype(np.float64))
    fin = np.isfinite(ref) & (ref > 1e-37) & (ref < 3e38)
    e = ulp_err(got[fin], ref[fin])
    assert e.max() <= 0.501, e.max()


def test_pow_specials_follow_c99(oracle):
    inf, nan = np.inf, np.nan
    cases = [(0.0, 0.5, 0.0), (-0.0, 0.5, 0.0), (0.0, -1.0, inf), (-0.0, -1.0, -inf), (-0.0, -2.0, inf), (0.0, 0.0, 1.0),
             (nan, 0.0, 1.0), (1.0, nan, 1.0), (1.0, inf, 1.0), (-1.0, inf, 1.0), (-1.0, -inf, 1.0), (2.0, inf, inf), (0.5, inf, 0.0),
             (2.0, -inf, 0.0), (0.5, -inf, inf), (inf, 2.0, inf), (inf, -2.0, 0.0), (-inf, 3.0, -inf), (-inf, 2.0, inf),
             (-inf, -3.0, -0.0), (-8.0, 3.0, -512.0), (-8.0, 2.0, 64.0), (-8.0, 0.5, nan), (-8.0, -1.0, -0.125), (nan, 1.0, nan),
             (2.0, nan, nan), (4.0, 0.5, 2.0), (2.0, 200.0, inf), (2.0, -200.0, 0.0), (3.0, 16777217.0, inf), (-3.0, 16777216.0, inf)]
    x = np.array([c[0] for c in cases], dtype=np.float32)
    y = np.array([c[1] for c in cases], dtype=np.float32)
    want = np.array([c[2] for c in cases], dtype=np.float32)
    got = oracle.math(3, x, y)
    for i, c in enumerate(cases):
        if np.isnan(want[i]):
            assert np.isnan(got[i]), c
        else:
            assert got[i] == want[i] and np.signbit(got[i]) == np.signbit(want[i]), (c, got[i])


def test_strict_math_vs_glibc(oracle, oracle_libm):
    """The strict functions and glibc's differ by at most 1 ulp on the path's domain, and
    agree exactly almost everywhere (glibc's powf/log2f are themselves < 1 ulp)."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    phi = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    for fn in (0, 1):
        a, b = oracle.math(fn, phi), oracle_libm.math(fn, phi)
        assert np.abs(a.astype(np.float64) - b).max() < 1.3e-7
        assert (a == b).mean() > 0.80
    x = rng.uniform(0.4, 1.05, n).astype(np.float32)
    y = np.full(n, 2.2, dtype=np.float32)
    a, b = oracle.math(3, x, y), oracle_libm.math(3, x, y)
    assert (a == b).mean() > 0.999      # glibc powf itself is not correctly rounded in ~0.06 % of cases
    a2 = rng.uniform(1e-6, 1e-2, n).astype(np.float32)
    assert (oracle.math(2, a2) == oracle_libm.math(2, a2)).mean() > 0.999


def test_build_flavours(oracle, oracle_libm, oracle_opcount):
    assert oracle.build_info() == "oracle: strict math"
    assert oracle_libm.build_info() == "oracle: glibc libm"
    assert oracle_opcount.build_info() == "oracle: opcount"


def test_exp_and_log_accuracy_and_special_cases(oracle, oracle_libm):
    """rpt_expf / rpt_logf (used only by the project-defined participating media): <= 0.5001 ulp against f64, Rust's
    f32::exp / f32::ln special cases, and agreement with glibc in all but a sliver of inputs."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-90, 90, 400000), rng.uniform(-1, 1, 200000), rng.uniform(-1e-3, 1e-3, 50000)]).astype(np.float32)
    got = oracle.math(7, x)
    ref = np.exp(x.astype(np.float64))
    ok = np.isfinite(got) & (got > 1e-37)                                  # (below: f32 subnormals, ulp is not relative there)
    assert ulp_err(got[ok], ref[ok]).max() <= 0.5001 + 1e-4
    y = np.concatenate([rng.uniform(1e-30, 1e30, 200000), rng.uniform(0.5, 2.0, 300000), 2.0 ** rng.uniform(-126, 127, 100000)]).astype(np.float32)
    got = oracle.math(8, y)
    ref = np.log(y.astype(np.float64))
    nz = np.abs(ref) > 1e-30
    assert ulp_err(got[nz], ref[nz]).max() <= 0.5001 + 1e-4
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 100.0, -110.0], dtype=np.float32)
    e = oracle.math(7, sp)
    assert e[0] == 1.0 and e[1] == 1.0 and np.isinf(e[2]) and e[3] == 0.0 and np.isnan(e[4]) and np.isinf(e[7]) and e[8] == 0.0
    l = oracle.math(8, sp)
    assert l[0] == -np.inf and l[1] == -np.inf and l[2] == np.inf and np.isnan(l[3]) and np.isnan(l[4]) and l[5] == 0.0 and np.isnan(l[6])
    assert (oracle.math(7, x) == oracle_libm.math(7, x)).mean() > 0.99
    assert (oracle.math(8, y) == oracle_libm.math(8, y)).mean() > 0.99


def test_pow_with_the_logarithm_handed_in_is_pow(oracle):
    """rpt_powf_log2x(x, rpt_log2_core(x), y) — what the library's material tables call for a roughness whose logarithm a row keeps
    (csrc/dev_bsdf.h, mat_cc_cos_theta) — is rpt_powf(x, y) bit for bit: random operands, and every special case of the C99 table."""
    rng = np.random.default_rng(9)
    n = 400_000
    x = np.concatenate([rng.uniform(0.0, 2.0, n), rng.uniform(1e-6, 1e-2, n), np.exp(rng.uniform(-80, 80, n)), -rng.uniform(0.0, 4.0, n // 8)]).astype(np.float32)
    y = np.concatenate([rng.uniform(0, 1, n), rng.uniform(-1.5, 1.5, n), rng.choice([2.2, 0.5, 0.4545, 3.0, -2.0], n), rng.choice([2.0, 3.0, 0.5], n // 8)]).astype(np.float32)
    inf, nan = np.inf, np.nan
    sx = np.array([0.0, -0.0, 1.0, inf, -inf, nan, 1e-45, 3.4e38, -1.0, 2.0, 0.5], dtype=np.float32)
    sy = np.array([0.0, -0.0, 1.0, inf, -inf, nan, 0.5, -1.0, 3.0, 200.0, -200.0, 16777217.0], dtype=np.float32)
    gx, gy = np.meshgrid(sx, sy)
    x = np.concatenate([x, gx.ravel()])
    y = np.concatenate([y, gy.ravel()])
    a = oracle.math(3, x, y)
    b = oracle.math(10, x, y)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
