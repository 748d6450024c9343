"""Statistical self-checks of the CPU oracle.  The reference cannot be run here and its RNG is
unseedable, so the oracle cannot be diffed against it; these checks catch transcription errors
that change the estimator (wrong pdf, wrong lobe weight, lost energy).  CPU only."""
import numpy as np


def _uniform_sphere(rng, n):
    z = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)


def _material(oracle, **kw):
    names = ["r", "g", "b", "er", "eg", "eb", "anisotropic", "metallic", "roughness", "subsurface", "specular_tint", "sheen",
             "sheen_tint", "clearcoat", "clearcoat_gloss", "spec_trans", "ior"]
    m = oracle.material_defaults()
    for k, v in kw.items():
        m[names.index(k)] = v
    return m


def test_rng_is_uniform_and_uncorrelated(oracle):
    draws = np.concatenate([oracle.rng_f32(1, f, p, 16) for f in range(40) for p in range(0, 4000, 7)])
    n = draws.size
    assert abs(draws.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n)
    assert abs(draws.var() - 1 / 12) < 1e-3
    hist = np.bincount((draws * 64).astype(int), minlength=64)
    chi2 = ((hist - n / 64) ** 2 / (n / 64)).sum()
    assert chi2 < 130                                       # 63 dof: P(chi2 > 130) ~ 1e-6
    x = draws.reshape(-1, 16)
    lag1 = np.corrcoef(x[:, :-1].ravel(), x[:, 1:].ravel())[0, 1]
    assert abs(lag1) < 0.01
    # neighbouring pixels and frames are independent streams
    a = np.array([oracle.rng_f32(1, 0, p, 1)[0] for p in range(5000)])
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.05
    b = np.array([oracle.rng_f32(1, f, 123, 1)[0] for f in range(5000)])
    assert abs(np.corrcoef(b[:-1], b[1:])[0, 1]) < 0.05


def test_bsdf_pdf_integrates_to_one(oracle):
    """disney_eval's pdf (sum of lobe pdfs times lobe weights, tracer.rs:601-623) is a density over
    the sphere: its integral is <= 1 (= 1 up to the part of the VNDF lobe that reflects below the
    horizon).  Materials without clearcoat: the clearcoat lobe's GTR1 carries the log2 quirk."""
    rng = np.random.default_rng(0)
    n = 60000
    L = _uniform_sphere(rng, n)
    nrm = np.array([0, 0, 1], dtype=np.float32)
    for kw in (dict(r=0.8, g=0.6, b=0.3, roughness=0.5), dict(r=0.9, g=0.9, b=0.9, roughness=0.3, metallic=1.0),
               dict(r=0.25, g=0.25, b=0.25, roughness=1.0)):
        m = _material(oracle, **kw)
        for v in (np.array([0, 0, 1.0]), np.array([0.6, 0, 0.8]), np.array([0.0, 0.95, 0.3122499])):
            pdf = np.array([oracle.disney_eval(m, 1 / 1.45, v.astype(np.float32), nrm, L[i])[3] for i in range(n)], dtype=np.float64)
            integral = 4 * np.pi * pdf.mean()
            err = 4 * np.pi * pdf.std() / np.sqrt(n)
            assert integral < 1.0 + 4 * err + 0.01, (kw, v, integral)
            assert integral > 0.80, (kw, v, integral)


def test_sampling_matches_evaluation(oracle):
    """E[f/pdf] over disney_sample equals the integral of disney_eval's f over the sphere
    (both include the cosine): the sampled lobe's f over (lobe weight x lobe pdf), summed over the
    lobe choice, is the full BSDF integral.  Checked for diffuse+specular and for a rough metal."""
    rng = np.random.default_rng(1)
    nrm = np.array([0, 0, 1], dtype=np.float32)
    zero = np.zeros(3, dtype=np.float32)
    for kw in (dict(r=0.8, g=0.6, b=0.3, roughness=0.6), dict(r=0.9, g=0.7, b=0.5, roughness=0.4, metallic=1.0)):
        m = _material(oracle, **kw)
        v = np.array([0.5, 0.1, np.sqrt(1 - 0.26)], dtype=np.float32)
        n = 60000
        L = _uniform_sphere(rng, n)
        f_eval = np.array([oracle.disney_eval(m, 1 / 1.45, v, nrm, L[i])[:3] for i in range(n)], dtype=np.float64)
        albedo_eval = 4 * np.pi * f_eval.mean(axis=0)
        err_eval = 4 * np.pi * f_eval.std(axis=0) / np.sqrt(n)
        est = []
        for i in range(n):
            o = oracle.disney_sample(m, 1 / 1.45, v, nrm, zero, 12345, i, 0)
            est.append(o[:3] / o[6] if o[6] > 0 else np.zeros(3))
        est = np.array(est, dtype=np.float64)
        albedo_samp = est.mean(axis=0)
        err_samp = est.std(axis=0) / np.sqrt(n)
        tol = 4 * np.sqrt(err_eval ** 2 + err_samp ** 2) + 0.01
        assert np.all(np.abs(albedo_eval - albedo_samp) < tol), (kw, albedo_eval, albedo_samp, tol)
        assert np.all(albedo_samp < 1.02)                    # a BSDF does not create energy


def test_disney_sample_draw_counts(oracle):
    """Draw order is part of the spec: 2 draws for the diffuse and clearcoat lobes, 3 for the
    specular one (tracer.rs:446-447, 534)."""
    nrm = np.array([0, 0, 1], dtype=np.float32)
    v = np.array([0.3, 0.2, np.sqrt(1 - 0.13)], dtype=np.float32)
    zero = np.zeros(3, dtype=np.float32)
    metal = _material(oracle, r=1, g=1, b=1, roughness=0.05, metallic=1.0)          # diffuse weight 0 -> always specular
    assert all(oracle.disney_sample(metal, 1 / 1.45, v, nrm, zero, 7, i, 0)[7] == 3 for i in range(50))
    matte = _material(oracle, r=0.25, g=0.25, b=0.25, roughness=1.0)
    counts = [oracle.disney_sample(matte, 1 / 1.45, v, nrm, zero, 7, i, 0)[7] for i in range(400)]
    assert set(counts) == {2.0, 3.0} and counts.count(2.0) > counts.count(3.0)


def test_furnace(oracle, rpt):
    """Constant environment, rough dielectric sphere, no lights.
    (1) black environment -> exactly black: nothing creates energy;
    (2) white environment, albedo 0.5 -> the sphere's centre shows the BSDF's directional albedo:
        0.5 diffuse plus the specular lobe, which the reference does NOT take out of the diffuse
        one (tracer.rs:365 has no (1 - F) factor), so a little above 0.5 and well below 1;
    (3) white environment, albedo 1 -> ~1.05 in the centre for the same reason (not <= 1)."""
    def sphere_scene(env, albedo):
        s = rpt.Scene()
        s.background = dict(kind=rpt._abi.RPT_BG_CONSTANT, colour_a=(env, env, env), colour_b=(0, 0, 0), gamma=2.2, scale=1.0)
        s.materials = [rpt.Material(rgb=(albedo, albedo, albedo), roughness=1.0)]
        s.spheres = [((0.0, 0.0, 0.0), 1.0, 0)]
        s.max_depth = 8
        return s
    black = oracle.render(sphere_scene(0.0, 1.0).describe(), 32, 24, 8, seed=5)[..., :3]
    assert np.all(black == 0.0)
    grey = oracle.render(sphere_scene(1.0, 0.5).describe(), 48, 36, 64, seed=5)[..., :3]
    assert not np.isnan(grey).any()
    assert 0.5 < grey[12:24, 18:30].mean() < 0.62
    assert np.all(grey[0] == 1.0)                            # outside the sphere: the environment itself
    white = oracle.render(sphere_scene(1.0, 1.0).describe(), 48, 36, 64, seed=5)[..., :3]
    assert 1.0 < white[12:24, 18:30].mean() < 1.12 and white.max() < 1.6


def test_independent_seeds_converge_to_the_same_image(oracle):
    d = oracle.scene_analytical()
    a = oracle.render(d, 64, 48, 96, seed=11)[..., :3]
    b = oracle.render(d, 64, 48, 96, seed=12)[..., :3]
    assert not np.array_equal(a, b)
    assert abs(a.mean() - b.mean()) < 0.01 * a.mean()
    blk = lambda x: x.reshape(6, 8, 8, 8, 3).mean(axis=(1, 3))
    assert np.abs(blk(a) - blk(b)).max() < 0.08


def test_strict_and_glibc_oracles_agree_statistically(oracle, oracle_libm):
    """The strict libm stand-in vs glibc: the same image up to ~1e-6 except for the rare samples
    where a 1-ulp difference flips a branch (SURVEY.md §7 'Parity definition')."""
    d = oracle.scene_analytical()
    a = oracle.render(d, 96, 72, 16, seed=3)[..., :3].astype(np.float64)
    b = oracle_libm.render(oracle_libm.scene_analytical(), 96, 72, 16, seed=3)[..., :3].astype(np.float64)
    diff = np.abs(a - b)
    assert np.median(diff) < 1e-6
    assert (diff > 1e-4).mean() < 0.01
    assert abs(a.mean() - b.mean()) < 1e-4


def test_opcount_build_matches_and_counts(oracle, oracle_opcount):
    d = oracle.scene_analytical()
    c = oracle_opcount.opcount(oracle_opcount.scene_analytical(), 60, 45, 2, seed=1)
    n = 60 * 45 * 2
    flops = (c["add"] + c["mul"] + c["div"] + c["sqrt"]) / n
    assert 500 < flops < 5000                                # SURVEY.md §8d estimated ~1.3 kflop
    assert 4 < c["transc"] / n < 100                        # sin+cos count as two, pow/log2/tan as one each
    a = oracle.render(d, 60, 45, 2, seed=1)
    b = oracle_opcount.render(oracle_opcount.scene_analytical(), 60, 45, 2, seed=1, threads=1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
