"""K8, two-band screen synthesis (``aog_set_screen_method``, default): deterministic acceptance on the host, in float64.

``layer.reset()`` (AO_env.py:76-77) draws a zero-mean stationary Gaussian field; such a field is its covariance function.  The
literal method's covariance C16(r) and the two-band method's C_fast(r) are both finite cosine sums over discrete spectra, so they are
compared EXACTLY (no Monte Carlo) at every lag of the N x N pupil.  The Monte-Carlo tests of the device output live in
``tests/test_gpu_parity.py``."""
import numpy as np
import pytest

from adaptive_optics_gym_amd.atmosphere_host import (literal_covariance, screen_numpy, screen_twoband_numpy, twoband_covariance,
                                                     twoband_high_weight)

L0 = 10.0
D = 0.5


@pytest.mark.parametrize("N,q", [(32, 16), (64, 16), (60, 16), (128, 16), (240, 16), (256, 16), (64, 8), (128, 4)])   # 240: the reference's pupil
def test_twoband_covariance_equals_literal_covariance_at_every_lag(N, q):
    delta = D / N
    lit = literal_covariance(N, delta, L0, q)
    two = twoband_covariance(N, delta, L0, q)
    assert lit.shape == two.shape == (2 * N - 1, 2 * N - 1)
    c0 = lit[N - 1, N - 1]
    # bound asked for: 1e-3 C(0); achieved: ~2e-5 at hcipy's oversampling of 16 (coarser fine grids put more variance next to the band edge)
    tol = 1e-4 if q >= 16 else 1e-3
    assert np.abs(two - lit).max() <= tol * c0
    # the structure function D(r) = 2 (C(0) - C(r)) is what the optics sees (piston never reaches an output): relative bound at every lag
    d_lit = 2 * (c0 - lit)
    d_two = 2 * (two[N - 1, N - 1] - two)
    mask = np.ones_like(d_lit, dtype=bool)
    mask[N - 1, N - 1] = False
    assert (np.abs(d_two - d_lit)[mask] / d_lit[mask]).max() <= (1e-3 if q >= 16 else 5e-3)
    # symmetric, and a valid (non-negative) variance split
    np.testing.assert_allclose(two, two[::-1, ::-1], rtol=0, atol=1e-9 * c0)
    w = twoband_high_weight(np.linspace(0, (3 * q) ** 2, 1000), q)
    assert w.min() >= 0 and w.max() <= 1 and w[0] == 0 and w[-1] == 1 and np.all(np.diff(w) >= 0)


def test_twoband_host_sampler_matches_its_covariance_and_the_literal_generator():
    """The float64 host sampler of the model (what the device kernels are written against) reproduces the literal covariance: variance of
    pixel differences at short and long lags, against the exact values and against screens of the reference-faithful generator."""
    N, q, n_scr = 32, 16, 600
    delta = D / N
    rng = np.random.RandomState(11)
    two = np.stack([screen_twoband_numpy(N, delta, 1.0, L0, rng, q) for _ in range(n_scr)])
    lit_cov = literal_covariance(N, delta, L0, q)
    c0 = lit_cov[N - 1, N - 1]
    for r, rtol in ((1, 0.03), (3, 0.05), (8, 0.10), (16, 0.15)):
        exact = 2 * (c0 - lit_cov[N - 1, N - 1 + r])
        np.testing.assert_allclose(np.mean((two[:, :, r:] - two[:, :, :-r]) ** 2), exact, rtol=rtol)
        np.testing.assert_allclose(np.mean((two[:, r:, :] - two[:, :-r, :]) ** 2), exact, rtol=rtol)
    lit = np.stack([screen_numpy(N, delta, 1.0, L0, np.random.RandomState(100 + i), q) for i in range(40)])
    np.testing.assert_allclose(np.mean((lit[:, :, 1:] - lit[:, :, :-1]) ** 2), 2 * (c0 - lit_cov[N - 1, N]), rtol=0.08)
