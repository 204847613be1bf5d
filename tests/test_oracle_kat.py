"""Known-answer tests that pin the CPU oracle (oracle/) to analytic optics and to SURVEY.md Appendix B.

The reference ships no tests or golden vectors and hcipy cannot be imported here, so these analytic values are
what anchors the restatement ("parity unpinned" with respect to hcipy itself; see oracle/hcipy_restatement.py).
"""
import math

import numpy as np
import pytest
from scipy import ndimage

import oracle.hcipy_restatement as H
from oracle.ao_env_oracle import AOEnvOracle


@pytest.fixture(scope="module")
def env240():
    return AOEnvOracle(act_dim=64, obs_dim=2, num_pupil_pixels=240, screen=np.zeros(240 * 240), timesteps_per_episode=3,
                       verbose=False)


@pytest.fixture(scope="module")
def env240_o5():
    return AOEnvOracle(act_type="zernike", act_dim=6, obs_dim=5, rew_type="smf_ssim", num_pupil_pixels=240,
                       screen=np.zeros(240 * 240), verbose=False)


# ---- SURVEY.md Appendix B numbers (N = 240, flat wavefront) ------------------------------------------------
def test_appendix_b_flat_wavefront_o2(env240):
    env240.reset()
    np.testing.assert_allclose(env240.last_obs_raw, 0.01575277, rtol=2e-7)
    assert int(env240.aperture.sum()) == 45244
    np.testing.assert_allclose(env240.wf_wfs_after_foc.total_power, 0.96312880, rtol=2e-8)
    np.testing.assert_allclose(env240.single_mode_fiber.forward(env240.wf_wfs_after_foc).total_power, 0.75716390, rtol=2e-8)


def test_appendix_b_flat_wavefront_o5(env240_o5):
    env240_o5.reset()
    o = env240_o5.last_obs_raw.reshape(5, 5)
    np.testing.assert_allclose(o[2, 2], 3.01752344, rtol=2e-9)
    np.testing.assert_allclose(o[2, 1], 0.01867363, rtol=3e-7)
    np.testing.assert_allclose(o[1, 1], 0.01163139, rtol=3e-7)
    np.testing.assert_allclose([o[0, 2], o[0, 1], o[0, 0]], [2.9176e-3, 5.5118e-4, 9.5706e-4], rtol=2e-4)
    np.testing.assert_allclose(o, o.T, rtol=1e-9)
    np.testing.assert_allclose(o, o[::-1, ::-1], rtol=1e-9)


def test_aperture_pixel_counts():
    for n, expect in ((128, 12892), (240, 45244), (256, 51468)):
        g = H.make_pupil_grid(n, 0.5)
        assert int(H.make_circular_aperture(0.5)(g).sum()) == expect


def test_lp_mode_roots():
    V = 2 * np.pi / 1.5e-6 * 4.5e-6 * 0.14
    np.testing.assert_allclose(V, 2.63893783, rtol=1e-8)
    u0, w0 = H.find_lp_solutions(0, V)
    u1, w1 = H.find_lp_solutions(1, V)
    u2, _ = H.find_lp_solutions(2, V)
    np.testing.assert_allclose(u0, [1.701114], atol=1e-6)
    np.testing.assert_allclose(w0, [2.017475], atol=1e-6)
    np.testing.assert_allclose(u1, [2.556426], atol=1e-6)
    np.testing.assert_allclose(w1, [0.654738], atol=1e-6)
    assert len(u2) == 0  # LP01 + LP11(cos, sin): three guided modes


# ---- analytic optics ------------------------------------------------------------------------------------------
def test_flat_wavefront_strehl_is_one(env240):
    env240.reset()
    reward, _ = env240.reward_function()
    assert abs(env240.last_strehl - 1.0) < 1e-12
    assert abs(reward) < 1e-10


def test_focal_grid_has_on_axis_sample_and_airy_null(env240):
    fg = env240.focal_grid
    xs, ys = fg.separated_coords
    assert fg.dims.tolist() == [240, 240] and xs[120] == 0.0 and ys[120] == 0.0
    psf = env240.unaberrated_PSF.reshape(240, 240)
    assert np.argmax(psf) == 120 * 240 + 120
    prof = psf[120, 120:140] / psf[120, 120]
    k = int(np.argmin(prof[:8]))
    assert k == 5  # first Airy null at 1.22 lambda/D = 4.88 pixels (q = 4)
    np.testing.assert_allclose(prof[4], (2 * _j1(np.pi * 1.0) / (np.pi * 1.0)) ** 2, rtol=2e-2)  # 1 lambda/D
    # total power is conserved by the propagator up to the finite focal window (30 lambda/D half-width)
    assert 0.98 < psf.sum() < 1.0


def _j1(x):
    from scipy.special import j1
    return j1(x)


def test_marechal_small_aberration():
    N = 128
    env = AOEnvOracle(act_type="zernike", act_dim=6, obs_dim=2, num_pupil_pixels=N, screen=np.zeros(N * N), verbose=False)
    env.reset()
    a = np.array([0, 0, 0, 1, 0.5, -0.3], dtype=np.float32)
    env.step(a)
    surf = env.deformable_mirror.surface
    np.testing.assert_allclose(np.std(surf), 0.1 * 2.2e-6, rtol=1e-12)       # AO_env.py:120 invariant
    phase = 2 * surf[env.aperture > 0] * 2 * np.pi / 2.2e-6
    # exact: S = |<exp(i phase)>|^2 over the aperture
    np.testing.assert_allclose(env.last_strehl, abs(np.mean(np.exp(1j * phase))) ** 2, rtol=1e-12)
    # Marechal regime: shrink the same shape to ~0.25 rad rms and evaluate the reward chain directly
    env.deformable_mirror.actuators = env.deformable_mirror.actuators * 0.18
    env.reward_function()
    phase = 2 * env.deformable_mirror.surface[env.aperture > 0] * 2 * np.pi / 2.2e-6
    sigma2 = np.var(phase)
    assert 0.03 < sigma2 < 0.1
    assert abs(env.last_strehl - math.exp(-sigma2)) < 2e-3


def test_action_scale_invariance_and_fp32_action_promotion():
    N = 48
    scr = ndimage.gaussian_filter(np.random.RandomState(0).randn(N, N), 3) * 1e-5
    kw = dict(act_dim=12, obs_dim=2, num_pupil_pixels=N, screen=scr.ravel(), verbose=False)
    a = np.random.RandomState(1).randn(12).astype(np.float32)
    e1, e2 = AOEnvOracle(**kw), AOEnvOracle(**kw)
    e1.reset(); e2.reset()
    _, r1, _, _, i1 = e1.step(a)
    _, r2, _, _, i2 = e2.step(np.float32(4.0) * a)
    np.testing.assert_allclose(e1.last_obs_raw, e2.last_obs_raw, rtol=1e-12)
    np.testing.assert_allclose(r1, r2, rtol=1e-12)
    assert e1.deformable_mirror.actuators.dtype == np.float64  # float32 action / int64 arange -> float64


def test_zernike_noll_table_and_orthonormality():
    expect = {1: (0, 0), 2: (1, 1), 3: (1, -1), 4: (2, 0), 5: (2, -2), 6: (2, 2), 7: (3, -1), 8: (3, 1), 9: (3, -3),
              10: (3, 3), 11: (4, 0)}
    for j, nm in expect.items():
        assert H.noll_to_zernike(j) == nm
    g = H.make_pupil_grid(256, 0.5)
    modes = np.stack(H.make_zernike_basis(10, 0.5, g))
    ap = H.make_circular_aperture(0.5)(g) > 0
    gram = modes[:, ap] @ modes[:, ap].T / ap.sum()
    np.testing.assert_allclose(gram, np.eye(10), atol=0.02)
    assert np.all(modes[:, ~ap] == 0)


def test_disk_harmonic_order_first_64():
    pm = lambda n, m: [(n, -m), (n, m)]
    expect = [(1, 0)] + pm(1, 1) + pm(1, 2) + pm(1, 3) + pm(1, 4) + pm(2, 1) + pm(1, 5) + pm(2, 2) + [(2, 0)] + pm(1, 6) \
        + pm(2, 3) + pm(3, 1) + pm(1, 7) + pm(2, 4) + pm(1, 8) + pm(3, 2) + [(3, 0)] + pm(2, 5) + pm(1, 9) + pm(3, 3) \
        + pm(4, 1) + pm(2, 6) + pm(1, 10) + pm(3, 4) + pm(1, 11) + pm(2, 7) + pm(4, 2) + [(4, 0)] + pm(1, 12) + pm(3, 5) \
        + pm(2, 8) + pm(4, 3) + pm(5, 1) + pm(1, 13)
    assert H.get_disk_harmonic_orders_sorted(64) == expect[:64]
    # the first mode is J0(3.8317 r): no piston mode in the Neumann basis
    from scipy.special import jnp_zeros
    np.testing.assert_allclose(jnp_zeros(0, 1)[-1], 3.8317059702, rtol=1e-9)


def test_cn_squared_values():
    np.testing.assert_allclose(H.Cn_squared_from_fried_parameter(0.20, 2.2e-6), 4.2374e-12, rtol=2e-5)
    np.testing.assert_allclose(H.Cn_squared_from_fried_parameter(0.15, 2.2e-6), 6.8443e-12, rtol=2e-5)
    r0 = H.fried_parameter_from_Cn_squared(H.Cn_squared_from_fried_parameter(0.17, 1e-6), 1e-6)
    np.testing.assert_allclose(r0, 0.17, rtol=1e-12)


def test_von_karman_covariance_and_psd_agree_on_variance():
    r0, L0 = 0.2, 10.0
    c0 = H.phase_covariance_von_karman(r0, L0)(np.array([0.0]))[0]
    np.testing.assert_allclose(c0, 0.0863 * (L0 / r0) ** (5 / 3), rtol=2e-3)


def test_fft_screen_normalisation_monte_carlo():
    """Variance of the synthesised screen equals the discrete integral of the PSD over the FFT grid, and the
    small-separation structure function follows the von Karman covariance."""
    n, D, L0 = 16, 0.5, 10.0
    g = H.make_pupil_grid(n, D)
    cn2 = H.Cn_squared_from_fried_parameter(0.2, 2.2e-6)
    rng = np.random.RandomState(3)
    scr = np.stack([H.von_karman_screen_fft(g, cn2, L0, 16, rng) for _ in range(300)]).reshape(300, n, n)
    m = 16 * n
    du = 2 * np.pi / (m * D / n)
    u = du * (np.arange(m) - m // 2)
    psd = H.power_spectral_density_von_karman(H.fried_parameter_from_Cn_squared(1, 1), L0)(np.hypot(u[None], u[:, None]).ravel())
    var_expect = psd.sum() * du * du / (2 * np.pi) ** 2 * cn2
    np.testing.assert_allclose(scr.var(), var_expect, rtol=0.12)
    # structure function at a 2-pixel separation vs 2 (C(0) - C(r)) of the infinite von Karman field
    d = g.delta[0] * 2
    cov = H.phase_covariance_von_karman(H.fried_parameter_from_Cn_squared(1, 1), L0)
    D_expect = 2 * (cov(np.array([0.0]))[0] - cov(np.array([d]))[0]) * cn2
    D_mc = np.mean((scr[:, :, 2:] - scr[:, :, :-2]) ** 2)
    np.testing.assert_allclose(D_mc, D_expect, rtol=0.1)


def test_extrusion_matrices_reproduce_conditional_covariance():
    n = 12
    g = H.make_pupil_grid(n, 0.5)
    layer = H.InfiniteAtmosphericLayer(g, 1e-12, 10.0, 3.0, rng=np.random.RandomState(5), initial_screen=np.zeros(n * n))
    nz = layer.num_stencils_vertical
    assert nz == 3 * n or nz < 3 * n  # 2 rows + one extra sample per column (duplicates possible)
    C = layer.cov_matrix_vertical
    A, B = layer.A_vertical, layer.B_vertical
    np.testing.assert_allclose(A @ C[:nz, :nz] @ A.T + B @ B.T, C[nz:, nz:], rtol=1e-6, atol=1e-9 * C.max())
    # rng draw order: rand (direction), geometric x2 (stencils); the velocity vector has the requested modulus
    np.testing.assert_allclose(np.hypot(*layer.velocity), 3.0, rtol=1e-12)


def test_extrusion_shifts_screen_by_whole_pixels():
    n = 10
    g = H.make_pupil_grid(n, 0.5)
    rng = np.random.RandomState(11)
    s0 = rng.randn(n, n) * 1e-6
    layer = H.InfiniteAtmosphericLayer(g, 1e-12, 10.0, np.array([0.0, -60.0]), rng=rng, initial_screen=s0.ravel())
    layer.t = 1e-3  # centre moves by -0.06 m = -1.2 px -> one 'bottom' extrusion: new row 0, rest shifted down
    s1 = layer._achromatic_screen.reshape(n, n)
    np.testing.assert_array_equal(s1[1:], s0[:-1])
    layer2 = H.InfiniteAtmosphericLayer(g, 1e-12, 10.0, np.array([60.0, 0.0]), rng=rng, initial_screen=s0.ravel())
    layer2.t = 1e-3  # 'right': operates on the 180-degree rotated screen -> new last column, rest shifted left
    s2 = layer2._achromatic_screen.reshape(n, n)
    np.testing.assert_array_equal(s2[:, :-1], s0[:, 1:])


def test_ssim_matches_uniform_filter_formulation():
    rng = np.random.RandomState(2)
    x = rng.rand(25)
    y = np.zeros(25); y[12] = 2.8
    # skimage's algorithm restated with scipy.ndimage.uniform_filter + crop
    f = lambda a: ndimage.uniform_filter(a, size=7)
    ux, uy, uxx, uyy, uxy = f(x), f(y), f(x * x), f(y * y), f(x * y)
    cn = 7 / 6
    vx, vy, vxy = cn * (uxx - ux * ux), cn * (uyy - uy * uy), cn * (uxy - ux * uy)
    C1, C2 = (0.01 * 2.8) ** 2, (0.03 * 2.8) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    np.testing.assert_allclose(H.structural_similarity_1d(x, y, 2.8), S[3:-3].mean(), rtol=1e-12)
    with pytest.raises(ValueError):
        H.structural_similarity_1d(np.zeros(4), np.zeros(4), 2.8)  # obs_dim = 2 (AO_env.py:495 raises in the reference)


def test_episode_bookkeeping_and_velocity_messages(capsys):
    N = 32
    env = AOEnvOracle(atm_type="quasi_static", atm_vel=10, act_dim=6, act_type="zernike", obs_dim=2, num_pupil_pixels=N,
                      screen=np.zeros(N * N), timesteps_per_episode=3)
    out = capsys.readouterr().out
    assert "In quasi_static atmospheric condition, the velocity value should be zero." in out
    assert "therefore velocity value is changed to zero" in out
    assert env.velocity == 0
    a = np.ones(6, dtype=np.float32)
    dones = []
    for ep in range(2):
        obs, info = env.reset()
        assert obs.dtype == np.float16 and obs.shape == (4,) and info == {}
        for t in range(3):
            obs, r, d, tr, info = env.step(a)
            dones.append(d)
            assert tr is False and set(info) == {"power"} and isinstance(info["power"], float)
    assert dones == [False, False, True, False, False, True]
    assert env.timestep == 6 and env.episode_no == 2  # timestep is never reset (AO_env.py:123)


def test_smf_ssim_reward_and_threshold():
    N = 48
    scr = ndimage.gaussian_filter(np.random.RandomState(4).randn(N, N), 3) * 1e-5
    kw = dict(act_type="zernike", act_dim=6, obs_dim=5, rew_type="smf_ssim", num_pupil_pixels=N, screen=scr.ravel(), verbose=False)
    env = AOEnvOracle(**kw)
    env.reset()
    a = np.random.RandomState(5).randn(6).astype(np.float32)
    _, r, _, _, info = env.step(a)
    ref = np.zeros(25); ref[12] = 2.8
    expect = 0.8 * info["power"] + 0.2 * H.structural_similarity_1d(env.last_obs_raw, ref, 2.8)
    np.testing.assert_allclose(r, expect, rtol=1e-12)
    env_t = AOEnvOracle(rew_threshold=r + 1.0, **kw)
    env_t.reset()
    assert env_t.step(a)[1] == -1.0
    with pytest.raises(ValueError):
        bad = AOEnvOracle(act_type="zernike", act_dim=6, obs_dim=2, rew_type="smf_ssim", num_pupil_pixels=N, screen=scr.ravel(), verbose=False)
        bad.reset()
        bad.step(a)


def test_shack_hartmann_oracle_known_answers():
    """SH chain of the oracle: flat wavefront -> slopes equal the reference slopes; a pure tilt -> one uniform slope component
    proportional to the tilt; the leaky integrator (gain 0.3, leakage 0.01) raises the Strehl ratio of a static aberration."""
    N, A = 96, 8
    scr = ndimage.gaussian_filter(np.random.RandomState(3).randn(N, N), 5.0)
    scr = scr / scr.std() * 1.5e-6
    env = AOEnvOracle(act_type="zernike", act_dim=A, obs_dim=2, num_pupil_pixels=N, screen=np.zeros(N * N), SH_operation=True,
                      timesteps_per_episode=50, rng=np.random.RandomState(0), verbose=False)
    n_sub = len(env.shwfse.estimation_subapertures)
    assert 60 < n_sub < 12 * 12 and env.reconstruction_matrix.shape == (A, 2 * n_sub)
    assert env.shwfs.micro_lens_array.mla_index.max() < 24 * 24          # lenslet grid spans twice the pupil (arange(-D, D, d))
    # flat: noise-free slopes == reference
    wf = env.magnifier(env.wf_wfs)
    env.camera.integrate(env.shwfs(wf), 1.0)
    s = env.shwfse.estimate([env.camera.read_out()]) - env.slopes_ref
    assert np.abs(s).max() < 1e-15
    # pure x tilt of amplitude eps * (Zernike 2 / ptp): x slopes uniform and non-zero, y slopes ~ 0
    env.deformable_mirror_shack.flatten()
    act = np.zeros(A); act[1] = 2e-8
    env.deformable_mirror_shack.actuators = act
    env.camera.integrate(env.shwfs(env.magnifier(env.deformable_mirror_shack(env.wf_wfs))), 1.0)
    s = env.shwfse.estimate([env.camera.read_out()]) - env.slopes_ref
    inner = np.abs(s[0] - np.median(s[0])) < 0.2 * abs(np.median(s[0]))
    assert inner.mean() > 0.7 and abs(np.median(s[0])) > 20 * np.abs(s[1]).max()
    # the reference's poke formula sum(amp * slopes) / var([-p, p]) = (s+ - s-) / p is TWICE the derivative ds/da
    # (AO_env.py:453-457), so response * a = 2 * slopes(a) in the linear regime
    np.testing.assert_allclose(0.5 * env.response_matrix[:, 1] * 2e-8, s.ravel(), rtol=2e-2, atol=2e-2 * np.abs(s).max())
    # closed loop on a static aberration
    env2 = AOEnvOracle(act_type="zernike", act_dim=A, obs_dim=2, num_pupil_pixels=N, screen=scr.ravel(), SH_operation=True,
                       timesteps_per_episode=50, rng=np.random.RandomState(1), verbose=False)
    env2.reset()
    st = []
    for _ in range(10):
        a, la = env2.SH_step()
        assert la.tolist() == [1]
        env2.step(a)
        st.append(env2.last_strehl)
    assert st[-1] > st[0] + 0.02


def test_large_poisson_and_tikhonov():
    rng = np.random.RandomState(0)
    lam = np.array([0.5, 10.0, 2e6, 5e7])
    out = np.stack([H.large_poisson(lam, rng=rng) for _ in range(4000)])
    np.testing.assert_allclose(out.mean(0), lam, rtol=0.05)
    np.testing.assert_allclose(out.var(0), lam, rtol=0.1)
    assert np.all(out == np.round(out))
    M = np.random.RandomState(1).randn(7, 4)
    np.testing.assert_allclose(H.inverse_tikhonov(M, 1e-12), np.linalg.pinv(M), rtol=1e-8, atol=1e-10)


def test_lp_modes_are_orthonormal_and_fiber_power_ignores_length():
    """StepIndexFiber (AO_env.py:393, 471-474): at V = 2.639 three guided modes (LP01, LP11 cos, LP11 sin), each with sum(mode^2 w) = 1 on the
    128^2 focal grid and mutually orthogonal (by symmetry), so that ``forward(wf).total_power`` = sum |c_k|^2 — the propagation phases
    exp(i beta_k L) are unit-modulus factors on orthogonal modes and cannot change it: the power is the same for any fiber length."""
    grid = H.make_pupil_grid(128, 52.5e-6)
    fiber = H.StepIndexFiber(4.5e-6, 0.14, 10.0)
    M = fiber.modes_for(grid, 1.5e-6)
    assert M.shape == (128 * 128, 3)
    gram = M.T @ (M * grid.weights)
    np.testing.assert_allclose(np.diag(gram), 1.0, rtol=1e-12)
    assert np.abs(gram - np.diag(np.diag(gram))).max() < 1e-12
    rng = np.random.RandomState(0)
    field = rng.randn(128 * 128) + 1j * rng.randn(128 * 128)
    wf = H.Wavefront(field, 1.5e-6, grid)
    c = M.T @ (field * grid.weights)
    p_forward = fiber.forward(wf).total_power
    np.testing.assert_allclose(p_forward, np.sum(np.abs(c) ** 2), rtol=1e-12)
    # with explicit propagation phases of arbitrary lengths the power of sum_k c_k e^{i beta_k L} mode_k is unchanged
    for betas in ([0.0, 0.0, 0.0], [1.234e6, 1.231e6, 1.231e6], [7.0, -3.0, 11.0]):
        for L in (0.0, 10.0, 123.456):
            out = M @ (c * np.exp(1j * np.asarray(betas) * L))
            np.testing.assert_allclose(np.sum(np.abs(out) ** 2 * grid.weights), p_forward, rtol=1e-12)
    assert H.StepIndexFiber(4.5e-6, 0.14, 1e-3).forward(wf).total_power == p_forward


def test_large_poisson_switches_at_the_threshold_and_draws_in_hcipy_order():
    """hcipy.util.large_poisson (AO_env.py:274 through the noiseless detector's read-out): strictly above ``thresh`` the rounded normal
    approximation, at or below it an exact Poisson draw; the normals of ALL large entries are drawn first (one ``normal(size=n_large)`` call),
    then one ``poisson`` call for the small ones — what the single-env drop-in's host-noise path replays."""

    class Spy:
        def __init__(self):
            self.calls = []

        def normal(self, size):
            self.calls.append(("normal", int(size)))
            return np.full(size, 2.0)

        def poisson(self, lam, size):
            self.calls.append(("poisson", int(size)))
            return np.asarray(lam).astype(int) + 7

    lam = np.array([3.0, 1e6, 1e6 + 1, 4e6, 0.0])
    spy = Spy()
    out = H.large_poisson(lam, rng=spy)
    assert spy.calls == [("normal", 2), ("poisson", 3)]
    np.testing.assert_array_equal(out, [10.0, 1e6 + 7, np.round(1e6 + 1 + 2 * np.sqrt(1e6 + 1)), 4e6 + 4000.0, 7.0])
    assert H.large_poisson(np.array([50.0]), thresh=10.0, rng=spy)[0] == np.round(50.0 + 2.0 * np.sqrt(50.0))


def test_inverse_tikhonov_rcond_is_relative_to_the_largest_singular_value():
    """hcipy inverse_tikhonov(M, rcond) (AO_env.py:464-465, rcond = 1e-3): filter s / (s^2 + (rcond s_max)^2) on every singular value — damping,
    not truncation, and relative to s_max (not an absolute threshold)."""
    U, _ = np.linalg.qr(np.random.RandomState(0).randn(6, 6))
    V, _ = np.linalg.qr(np.random.RandomState(1).randn(4, 4))
    s = np.array([10.0, 1.0, 1e-2, 1e-5])
    M = U[:, :4] @ np.diag(s) @ V.T
    for rcond in (1e-3, 0.5):
        inv = H.inverse_tikhonov(M, rcond)
        expect = V @ np.diag(s / (s ** 2 + (rcond * 10.0) ** 2)) @ U[:, :4].T
        np.testing.assert_allclose(inv, expect, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(H.inverse_tikhonov(3.0 * M, 1e-3), H.inverse_tikhonov(M, 1e-3) / 3.0, rtol=1e-9)   # scale-covariant: relative rcond
    # s = rcond s_max is damped to exactly half of its plain inverse; far above it the plain inverse survives
    inv = H.inverse_tikhonov(np.diag([1.0, 1e-3]), 1e-3)
    np.testing.assert_allclose(np.diag(inv), [1.0 / (1.0 + 1e-6), 0.5e3], rtol=1e-12)


def test_shack_hartmann_estimator_measures_from_the_lenslet_centre():
    """ShackHartmannWavefrontSensorEstimator.estimate (AO_env.py:418-425, 277): centre of gravity of each selected lenslet's pixels in the
    IMAGE grid's coordinates, minus that lenslet's position (``mla_points``), x row first; unselected lenslets are not reported."""
    grid = H.make_pupil_grid(8, 8.0)                 # pixel centres at -3.5 .. 3.5
    pts = np.array([[-2.0, -2.0], [2.0, -2.0], [-2.0, 2.0], [2.0, 2.0]])      # four lenslets of 4 x 4 pixels
    ix = (grid.x > 0).astype(int) + 2 * (grid.y > 0).astype(int)
    est = H.ShackHartmannWavefrontSensorEstimator(pts, ix, grid, estimation_subapertures=np.array([1, 0, 1, 1]))
    img = np.zeros(64)
    img[np.flatnonzero((grid.x == -1.5) & (grid.y == -2.5))] = 5.0            # lenslet 0: one spot at (-1.5, -2.5)
    img[np.flatnonzero((grid.x == -3.5) & (grid.y == 0.5))] = 1.0             # lenslet 2: two spots, 1 : 3
    img[np.flatnonzero((grid.x == -0.5) & (grid.y == 2.5))] = 3.0
    img[np.flatnonzero((grid.x == 2.5) & (grid.y == 2.5))] = 2.0              # lenslet 3
    img[np.flatnonzero((grid.x == 1.5) & (grid.y == -1.5))] = 9.0             # lenslet 1: not selected
    c = est.estimate([img])
    assert c.shape == (2, 3)
    np.testing.assert_allclose(c[:, 0], [-1.5 + 2.0, -2.5 + 2.0])
    np.testing.assert_allclose(c[:, 1], [(-3.5 + 3 * -0.5) / 4 + 2.0, (0.5 + 3 * 2.5) / 4 - 2.0])
    np.testing.assert_allclose(c[:, 2], [2.5 - 2.0, 2.5 - 2.0])
