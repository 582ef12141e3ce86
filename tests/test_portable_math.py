"""Accuracy of the portable log/cos/pow used by the kernels' exact-math mode, measured through
the oracle build that includes the same header (csrc/pt_portable_math.h).  CPU only."""
import math

import numpy as np

F = np.float32


def ulp_err(got, want64):
    want32 = np.float32(want64)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(ulp, 1e-45)


def test_random_normal_portable_vs_float64(oracle_shared, oracle):
    seeds = np.random.default_rng(1).integers(1, 2 ** 32, 5000)
    worst = 0.0
    for s in seeds:
        s = int(s)
        u1 = float(oracle.random_value(s))
        u2 = float(oracle.random_value((s * 7462) & 0xFFFFFFFF))
        if u1 <= 0.0:
            continue
        want = math.sqrt(-2.0 * math.log(u1)) * math.cos(float(F(2.0 * 3.14159)) * u2)
        got = oracle_shared.random_normal(s)
        worst = max(worst, abs(got - want))
    assert worst < 2e-6          # absolute: values are O(1); cos near its zeros dominates the error


def test_environment_light_portable_vs_libm(oracle_shared, oracle):
    """pow via exp2(y*log2 x): relative error <= 2e-5 at exponent 200, far inside the radiance tolerance."""
    p = oracle.default_params()
    ps = oracle_shared.default_params()
    rng = np.random.default_rng(2)
    for _ in range(2000):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        a = oracle.environment_light(p, d)
        b = oracle_shared.environment_light(ps, d)
        np.testing.assert_allclose(b, a, rtol=5e-5, atol=1e-7)


def test_special_values(oracle_shared):
    # seed 0: log(0) = -inf -> inf; NaN direction, exactly as with libm
    assert math.isinf(oracle_shared.random_normal(0))
    assert np.isnan(oracle_shared.random_direction(0)).all()
    p = oracle_shared.default_params()
    # pow(0, 0.35) = 0 at the horizon (smoothstep = 0), pow(1, 0.35) = 1 at the zenith
    np.testing.assert_array_equal(oracle_shared.environment_light(p, [1, 0, 0])[:3],
                                  np.array([0.5, 0.7, 1.0], F))
    z = oracle_shared.environment_light(p, [0, 1, 0])
    np.testing.assert_allclose(z, [0.1, 0.25, 1.0], rtol=1e-6)
