"""CPU tests of the product's host side: table builder vs the oracle, C-ABI export surface, fail-loud behaviour,
env-batch sharding.  No compute call is made without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle.hcipy_restatement as H
from adaptive_optics_gym_amd import _lib, atmosphere_host, optics_host, sharding
from adaptive_optics_gym_amd.params import OpticalParams, coerce_velocity
from adaptive_optics_gym_amd.spaces import make_box
from helpers import smooth_screens
from oracle.ao_env_oracle import AOEnvOracle


@pytest.mark.parametrize("act_type,A", [("num_actuators", 64), ("zernike", 20)])
def test_mode_matrix_and_indexing_match_oracle(act_type, A):
    N = 96
    T = optics_host.build_tables(OpticalParams(num_pupil_pixels=N), act_type, A, 2)
    env = AOEnvOracle(act_type=act_type, act_dim=A, num_pupil_pixels=N, screen=np.zeros(N * N), verbose=False)
    full = env.dm_modes.transformation_matrix
    # actuator indexing bit-exact: identical (n, m) table, identical column order
    if act_type == "zernike":
        assert T.mode_specs == [H.noll_to_zernike(j) for j in range(1, A + 1)]
    else:
        assert T.mode_specs == H.get_disk_harmonic_orders_sorted(A)
    assert np.array_equal(T.ap_index, np.flatnonzero(env.aperture > 0))
    np.testing.assert_allclose(T.modes, full[T.ap_index], rtol=0, atol=2e-15)
    assert np.abs(np.delete(full, T.ap_index, axis=0)).max() == 0
    a = np.random.RandomState(0).randn(A)
    np.testing.assert_allclose(np.sqrt(a @ T.gram @ a), np.std(full @ a), rtol=1e-12)
    # piston-heavy action: the Gram form must not lose the small variance to cancellation
    if act_type == "zernike":
        a = np.zeros(A); a[0] = 1.0; a[3] = 1e-5
        np.testing.assert_allclose(np.sqrt(a @ T.gram @ a), np.std(full @ a), rtol=1e-9)


@pytest.mark.parametrize("act_type,A,o,rew", [("num_actuators", 64, 2, "strehl_ratio"), ("zernike", 6, 5, "smf_ssim"),
                                               ("num_actuators", 10, 3, "strehl_ratio")])
def test_collapsed_tables_reproduce_literal_pipeline(act_type, A, o, rew):
    """The fold-back identity: K fixed pupil-plane dot products == the literal propagate-everything dataflow."""
    N = 64
    T = optics_host.build_tables(OpticalParams(num_pupil_pixels=N), act_type, A, o)
    assert T.wfs_tables.shape[0] == o * o + 3 and T.sci_tables.shape[0] == 1 and T.n_fiber_modes == 3
    assert np.abs(T.wfs_tables).max() == 1.0
    psi = smooth_screens(1, N, 1)[0]
    env = AOEnvOracle(act_type=act_type, act_dim=A, obs_dim=o, rew_type=rew, num_pupil_pixels=N, screen=psi.ravel(), verbose=False)
    env.reset()
    a = np.random.RandomState(2).randn(A).astype(np.float32)
    _, reward, _, _, info = env.step(a)
    ap = a.astype(np.float64) / (np.arange(A) + 10)
    a2 = ap * 0.1 * 2.2e-6 / np.sqrt(ap @ T.gram @ ap)
    np.testing.assert_allclose(a2, env.deformable_mirror.actuators, rtol=1e-12)
    theta = psi.ravel()[T.ap_index] + 4 * np.pi * (T.modes @ a2)
    Z = T.wfs_coef @ (T.wfs_tables @ np.cos(theta / 1.5e-6) + 1j * (T.wfs_tables @ np.sin(theta / 1.5e-6)))
    np.testing.assert_allclose(np.abs(Z[:o * o]) ** 2, env.last_obs_raw, rtol=1e-10)
    np.testing.assert_allclose(np.sum(np.abs(Z[o * o:]) ** 2), info["power"], rtol=1e-10)
    Zs = T.sci_coef @ (T.sci_tables @ np.cos(theta / 2.2e-6) + 1j * (T.sci_tables @ np.sin(theta / 2.2e-6)))
    if rew == "strehl_ratio":
        np.testing.assert_allclose(np.abs(Zs[0]) ** 2, env.last_strehl, rtol=1e-10)
    assert T.strehl_focal_index == 120 * 240 + 120  # on-axis sample of the 240^2 science focal grid


def test_lp_roots_and_realify():
    V = 2 * np.pi / 1.5e-6 * 4.5e-6 * 0.14
    np.testing.assert_allclose(optics_host.lp_roots(0, V), [1.701114], atol=1e-6)
    np.testing.assert_allclose(optics_host.lp_roots(1, V), [2.556426], atol=1e-6)
    assert optics_host.lp_roots(2, V) == []
    rng = np.random.RandomState(0)
    base = rng.randn(3, 50)
    k = np.stack([base[0] + 1j * base[1], base[0] - 1j * base[1], -base[2] + 0j, 1j * base[2]])
    tabs, coef = optics_host._realify(k)
    assert tabs.shape[0] == 3
    np.testing.assert_allclose(coef @ tabs, k, atol=1e-14)


def test_screen_generator_matches_oracle_draw_for_draw():
    n, D, L0 = 12, 0.5, 10.0
    cn2 = atmosphere_host.cn_squared_from_fried_parameter(0.15, 2.2e-6)
    np.testing.assert_allclose(cn2, H.Cn_squared_from_fried_parameter(0.15, 2.2e-6), rtol=1e-15)
    a = atmosphere_host.screen_numpy(n, D / n, cn2, L0, np.random.RandomState(9), 16)
    b = H.von_karman_screen_fft(H.make_pupil_grid(n, D), cn2, L0, 16, np.random.RandomState(9)).reshape(n, n)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12 * np.abs(b).max())


def test_params_spaces_and_velocity_coercion(capsys):
    p = OpticalParams()
    np.testing.assert_allclose(p.fiber_focal_length, 1.7857142857, rtol=1e-10)
    np.testing.assert_allclose(p.fiber_window, 52.5e-6, rtol=1e-12)
    assert coerce_velocity("semi_dynamic", 10) == 0
    assert "In semi_dynamic atmospheric condition, the velocity value should be zero." in capsys.readouterr().out
    assert coerce_velocity("dynamic", 0) == 1
    assert "therefore velocity value is changed to 1 m/s" in capsys.readouterr().out
    assert coerce_velocity("dynamic", 7) == 7 and coerce_velocity("quasi_static", 0) == 0
    box = make_box(-1, 1, (4,), np.float16)
    assert box.shape == (4,) and box.dtype == np.float16 and box.low[0] == -1 and box.high[0] == 1
    assert box.contains(box.sample())


def test_cabi_exports_every_declared_symbol(repo_root):
    header = open(os.path.join(repo_root, "include", "aogym.h")).read()
    declared = set(re.findall(r"\b(aog_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.aog_abi_version() == _lib.ABI_VERSION
    assert re.search(r"#define AOG_ABI_VERSION\s+%d" % _lib.ABI_VERSION, header)
    assert ctypes.sizeof(_lib.AogConfig) == 20 * 4 + 6 * 8
    assert ctypes.sizeof(_lib.AogInfo) == 10 * 4 + 8
    # every struct of the header has the size the library was compiled with (the loader checks the same at import)
    for which, cls in enumerate((_lib.AogConfig, _lib.AogTables, _lib.AogLayerTables, _lib.AogShTables, _lib.AogActor, _lib.AogInfo)):
        assert lib.aog_struct_size(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.aog_struct_size(99) == -1


def test_device_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from adaptive_optics_gym_amd import BatchedAOEnv

    with pytest.raises(RuntimeError, match="no CPU path|HIP device"):
        BatchedAOEnv(2, num_pupil_pixels=32, verbose=False)
    # and the C-ABI itself refuses to create a handle: no silent fallback
    lib = _lib.load()
    cfg = _lib.AogConfig()
    cfg.abi_version = _lib.ABI_VERSION
    cfg.num_envs, cfg.n_pupil, cfg.n_modes, cfg.obs_dim, cfg.n_ap = 1, 8, 4, 2, 10
    cfg.n_wfs_tables, cfg.n_sci_tables, cfg.n_fiber_modes = 7, 1, 3
    h = ctypes.c_void_p()
    rc = lib.aog_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc < 0 and not h.value
    assert len(lib.aog_last_error()) > 0
    cfg.abi_version = 999
    assert lib.aog_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    assert b"abi_version" in lib.aog_last_error()


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 1024, 8192, 8195):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, e = sharding.shard_range(total, r, world)
                seen += list(range(s, e))
            assert seen == list(range(total))
    assert sharding.shard_range(8192, 3, 8) == (3072, 4096)
    with pytest.raises(ValueError):
        sharding.shard_range(8, 8, 8)


def test_layer_tables_match_oracle_layer():
    """Stencils and AR matrices of the product's host builder == the oracle's InfiniteAtmosphericLayer (same seed)."""
    n, D, L0 = 14, 0.5, 10.0
    lay = atmosphere_host.build_layer_tables(n, D / n, L0, np.random.RandomState(21))
    rng = np.random.RandomState(21)
    ref = H.InfiniteAtmosphericLayer(H.make_pupil_grid(n, D), 1e-12, L0, np.array([1.0, 0.0]), rng=rng, initial_screen=np.zeros(n * n))
    assert np.array_equal(lay["stencil_vertical"], np.flatnonzero(ref.stencil_bottom))
    assert np.array_equal(lay["stencil_horizontal"], np.flatnonzero(ref.stencil_left))
    np.testing.assert_allclose(lay["A_vertical"], ref.A_vertical, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(lay["A_horizontal"], ref.A_horizontal, rtol=1e-7, atol=1e-9)
    # B is defined up to the sign of each singular vector: compare B B^T
    np.testing.assert_allclose(lay["B_vertical"] @ lay["B_vertical"].T, ref.B_vertical @ ref.B_vertical.T, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(lay["B_vertical"], ref.B_vertical, rtol=1e-6, atol=1e-9)
    v = np.array([[3.0, -4.0], [0.2, 0.0]])
    s = atmosphere_host.integer_shifts(v, 0.004, 0.005, D / n)
    assert s.tolist() == [[(round(3 * 0.005 / (D / n)) - round(3 * 0.004 / (D / n))), (round(-4 * 0.005 / (D / n)) - round(-4 * 0.004 / (D / n)))], [0, 0]]



@pytest.mark.parametrize("f_number", [50.0, 600.0])
def test_shack_hartmann_host_tables_match_oracle(f_number):
    """Product-side SH construction (sh_host.ShackHartmannHost) == the oracle's shack_hartmann_init (AO_env.py:396-465).  f-number 50
    is the reference's (hcipy's analytic Fresnel transfer function); 600 puts the same 96-pixel pupil past hcipy's sampling test, on the
    impulse-response branch the reference reaches above ~800 pupil pixels (AO_env.py:407)."""
    from adaptive_optics_gym_amd.sh_host import ShackHartmannHost

    N, A = 96, 8
    params = OpticalParams(num_pupil_pixels=N, f_number=f_number)
    T = optics_host.build_tables(params, "zernike", A, 2)
    sh = ShackHartmannHost(params, T)
    ref = AOEnvOracle(act_type="zernike", act_dim=A, obs_dim=2, num_pupil_pixels=N, screen=np.zeros(N * N), SH_operation=True,
                      verbose=False, f_number=f_number)
    assert ref.shwfs.propagator.uses_impulse_response(params.wavelength_wfs) == (f_number > 50)
    tf_ref = ref.shwfs.propagator.transfer_function(params.wavelength_wfs)
    np.testing.assert_allclose(sh.transfer, tf_ref, rtol=0, atol=1e-9 * np.abs(tf_ref).max())
    assert np.array_equal(sh.mla_index, ref.shwfs.micro_lens_array.mla_index)
    assert np.array_equal(sh.subapertures, ref.shwfse.estimation_subapertures)
    np.testing.assert_allclose(sh.slopes_ref, ref.slopes_ref.ravel(), rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(sh.response, ref.response_matrix, rtol=1e-6, atol=1e-9 * np.abs(ref.response_matrix).max())
    # the piston row is the (regularised) inverse of an unobservable direction: rounding-level values, compare absolutely
    np.testing.assert_allclose(sh.reconstruction, ref.reconstruction_matrix, rtol=1e-5,
                               atol=1e-5 * np.abs(ref.reconstruction_matrix).max())
    np.testing.assert_allclose(sh.amp_wfs ** 2 * sh.n_ap * sh.pix_area_pupil, ref.wf_wfs.total_power, rtol=1e-12)


def test_impulse_response_transfer_function_tends_to_the_analytic_one():
    """Known answer for hcipy's impulse-response Fresnel branch: the chirp h(r) = e^{ikz} e^{ik r^2 / 2z} / (i lambda z) has the analytic
    transform e^{ikz} e^{-i z |k|^2 / 2k}; sampled on a grid that resolves it (local frequency k x / z below Nyquist at the grid edge) and
    wide enough to hold it, the discrete transform must reproduce the analytic values in the band the chirp covers, |k| < k L / (2 z).
    Both the product table and the oracle's are checked, on a geometry chosen inside the impulse-response regime."""
    from adaptive_optics_gym_amd.sh_host import _fresnel_transfer
    from oracle import hcipy_restatement as H

    n, lam = 256, 1.5e-6
    pitch = 20e-6
    L = n * pitch
    z = 1.5 * pitch * L / lam            # pitch < lambda z / L: impulse-response branch; edge frequency = 2/1.5 of ... still sampled by q = 2
    tf = _fresnel_transfer(n, pitch, lam, z)
    grid = H.make_pupil_grid(n, L)
    prop = H.FresnelPropagator(grid, z)
    assert prop.uses_impulse_response(lam)
    np.testing.assert_allclose(tf, prop.transfer_function(lam), rtol=0, atol=1e-9)
    k = 2 * np.pi / lam
    kk = 2 * np.pi * np.fft.fftfreq(2 * n, pitch)
    k2 = kk[None, :] ** 2 + kk[:, None] ** 2
    analytic = np.exp(1j * k * z) * np.exp(-0.5j * z * k2 / k)
    band = np.sqrt(k2) < 0.5 * k * L / (2 * z)        # well inside the chirp's band: the stationary point lies inside the padded grid
    assert band.sum() > 1000
    assert np.abs(tf[band] - analytic[band]).max() < 0.08 and np.abs(np.abs(tf[band]) - 1).max() < 0.08   # Fresnel ripple of the truncated chirp
    assert np.abs(np.angle(tf[band] / analytic[band])).mean() < 0.05


def test_gym_registration_through_the_gymnasium_stand_in(repo_root):
    """``import gym_AO`` registers 'AO-v0' with the entry point of the drop-in (gym_AO/__init__.py:9-12 in the reference) when a
    ``gymnasium`` is importable — here the test stand-in, in a child process so that this process's modules are untouched — and the
    spaces become that package's Box (algorithm.py:32-35 asserts the type).  The env itself needs a GPU: tests/test_gpu_gym_make.py."""
    import subprocess
    import sys

    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import gymnasium as gym, gym_AO\n"
            "from adaptive_optics_gym_amd import spaces\n"
            "spec = gym.envs.registration.registry['AO-v0']\n"
            "assert gym_AO.registered and spec.entry_point == 'adaptive_optics_gym_amd.envs:AOEnv'\n"
            "assert type(spaces.make_box(-1, 1, (4,), 'float16')) == gym.spaces.Box\n"
            "from adaptive_optics_gym_amd.envs import AOEnv\n"
            "assert issubclass(AOEnv, gym.Env)\n"
            "print('ok')\n") % (repo_root, os.path.join(repo_root, "tests", "fake_gymnasium"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
