"""Parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs, against the committed golden fixtures, and — at BASELINE.json's full size —
through size-independent properties.

Tolerances (north_star): Strehl ratio and observations within 1e-5 relative *before* the float16 cast
(``obs_raw``); after the cast equality up to 1 float16 ulp; mode ("actuator") indexing bit-exact (CPU test
``test_mode_matrix_and_indexing_match_oracle``); ``done`` exact.
"""
import ast
import glob
import os

import math

import numpy as np
import pytest

from helpers import actions_for, run_oracle, smooth_screens

pytestmark = pytest.mark.gpu

RTOL = 1e-5
KERNELS = [("fast", "mfma"), ("fast", "valu"), ("fp64", "auto")]


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _loaded_native():
    from adaptive_optics_gym_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libaogym.so" in f.read(), "the HIP extension is not the code that ran"


def _ulp16_equal(a, b):
    a = np.asarray(a, dtype=np.float16).view(np.int16).astype(np.int32)
    b = np.asarray(b, dtype=np.float16).view(np.int16).astype(np.int32)
    return np.all(np.abs(a - b) <= 1)


def _drive(env, acts, torch):
    """Mirror of helpers.run_oracle on the device env: [T, B, A] actions, reset after done."""
    out = {k: [] for k in ("obs_raw", "obs", "reward", "done", "power", "strehl")}
    env.reset()
    obs0 = env.last_obs_raw.cpu().numpy().astype(np.float64)
    for t in range(acts.shape[0]):
        obs, r, d, tr, info = env.step(torch.from_numpy(acts[t]).to(env.device))
        out["obs_raw"].append(info["obs_raw"].cpu().numpy().astype(np.float64))
        out["obs"].append(obs.cpu().numpy())
        out["reward"].append(r.cpu().numpy().astype(np.float64))
        out["done"].append(d.cpu().numpy())
        out["power"].append(info["power"].cpu().numpy().astype(np.float64))
        out["strehl"].append(info["strehl"].cpu().numpy().astype(np.float64))
        assert tr.dtype == torch.bool and not bool(tr.any())
        if bool(d.all()):
            env.reset()
    res = {k: np.stack(v) for k, v in out.items()}
    res["obs0"] = obs0
    return res


def _assert_obs_close(got, ref):
    """1e-5 relative per element; elements in deep interference nulls (below 1e-3 of their vector's peak) are held
    to the same ABSOLUTE error instead, 1e-5 * 1e-3 * peak — a relative bound there is conditioning, not accuracy
    (the reference's own float64 result moves by more under a 1e-7 rad phase perturbation)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    peak = ref.max(axis=-1, keepdims=True)
    tol = RTOL * np.maximum(np.abs(ref), 1e-3 * peak)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"max rel err {np.max(np.abs(got - ref) / np.abs(ref)):.3e}, {bad.sum()} elements out of tolerance"


def _compare(got, ref, strehl_reward):
    _assert_obs_close(got["obs0"], ref["obs0"])
    _assert_obs_close(got["obs_raw"], ref["obs_raw"])
    np.testing.assert_allclose(got["power"], ref["power"], rtol=RTOL)
    assert _ulp16_equal(got["obs"], ref["obs"])
    np.testing.assert_array_equal(got["done"], ref["done"].astype(bool))
    if strehl_reward:
        np.testing.assert_allclose(got["strehl"], ref["strehl"], rtol=RTOL)
        np.testing.assert_allclose(got["reward"], ref["reward"], rtol=0, atol=100 * RTOL)  # reward = 100 (S - 1)
    else:
        np.testing.assert_allclose(got["reward"], ref["reward"], rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("precision,kernel", KERNELS)
@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                                        if not os.path.basename(p).startswith("ref_")), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_fixtures(path, precision, kernel):
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    z = np.load(path)
    kw = ast.literal_eval(str(z["kw"]))
    scr, acts = z["screens"], z["actions"]
    env = BatchedAOEnv(scr.shape[0], "cuda:0", num_pupil_pixels=scr.shape[1], screens=scr, precision=precision,
                       kernel=kernel, verbose=False, **kw)
    got = _drive(env, acts, torch)
    ref = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    _compare(got, ref, kw["rew_type"] == "strehl_ratio")
    _loaded_native()
    env.close()


_REF_FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_*.npz")))


@pytest.mark.skipif(not _REF_FIXTURES, reason="no tests/golden/ref_*.npz (hcipy-derived vectors of `tools/make_golden.py --from-reference`): hcipy is not "
                    "importable in the build image")
@pytest.mark.parametrize("path", _REF_FIXTURES or [None], ids=lambda p: os.path.basename(p)[:-4] if p else "absent")
def test_reference_derived_fixtures(path):
    """The HIP path against outputs of the reference's OWN AOEnv (hcipy) on a recorded screen and actions, N = 240: north_star's 1e-5."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    z = np.load(path)
    kw = ast.literal_eval(str(z["kw"]))
    env = BatchedAOEnv(1, "cuda:0", num_pupil_pixels=240, screens=z["screen"].reshape(1, 240, 240), verbose=False, **kw)
    env.reset()
    _assert_obs_close(env.last_obs_raw[0].double().cpu().numpy(), z["obs0_raw"])
    for t, a in enumerate(z["actions"]):
        _, r, d, _, info = env.step(torch.from_numpy(a[None].astype(np.float32)).cuda())
        _assert_obs_close(info["obs_raw"][0].double().cpu().numpy(), z["exp_obs_raw"][t])
        np.testing.assert_allclose(float(r[0]), z["exp_reward"][t], rtol=RTOL, atol=100 * RTOL if kw["rew_type"] == "strehl_ratio" else 1e-7)
        np.testing.assert_allclose(float(info["power"][0]), z["exp_power"][t], rtol=RTOL)
        assert bool(d[0]) == bool(z["exp_done"][t])
        if bool(d[0]):
            env.reset()
    env.close()


@pytest.mark.parametrize("precision,kernel", KERNELS)
@pytest.mark.parametrize("N,B,A,o,act_type,rew", [
    (64, 5, 64, 2, "num_actuators", "strehl_ratio"),      # ragged batch: not a multiple of 32 / 64
    (64, 33, 6, 5, "zernike", "smf_ssim"),
    (128, 2, 6, 2, "zernike", "strehl_ratio"),            # BASELINE configs[0] geometry
    (50, 1, 3, 4, "num_actuators", "strehl_ratio"),       # B = 1, odd sizes, A < 8
    (240, 2, 64, 2, "num_actuators", "strehl_ratio"),     # the reference's hard-coded pupil size
    (64, 3, 100, 2, "num_actuators", "strehl_ratio"),     # 64 < act_dim <= 128: the 128-mode matrix-core variant
    (64, 2, 20, 3, "zernike", "smf_ssim"),                # o = 3: the 12-table variant
])
def test_live_oracle_parity(N, B, A, o, act_type, rew, precision, kernel):
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    T = 3
    scr = smooth_screens(B, N, seed=N + B)
    acts = np.stack([actions_for(B, A, 7 * s + N) for s in range(T)])
    kw = dict(act_type=act_type, act_dim=A, obs_dim=o, rew_type=rew, timesteps_per_episode=2)
    ref = run_oracle(scr[: min(B, 3)], acts[:, : min(B, 3)], **kw)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, precision=precision, kernel=kernel, verbose=False, **kw)
    got = _drive(env, acts, torch)
    got = {k: (v[:, : min(B, 3)] if k != "obs0" else v[: min(B, 3)]) for k, v in got.items()}
    _compare(got, ref, rew == "strehl_ratio")
    env.close()


def test_strong_turbulence_von_karman_screens_parity():
    """Real von Karman screens (tens of radians peak-to-valley at the sensing wavelength): exercises the exact
    range reduction and the piston removal."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screen_numpy

    N, B, A = 96, 2, 64
    cn2 = cn_squared_from_fried_parameter(0.10, 2.2e-6)
    scr = np.stack([screen_numpy(N, 0.5 / N, cn2, 10.0, np.random.RandomState(s), 16) + 3e-5 for s in range(B)])
    assert np.ptp(scr[0]) / 1.5e-6 > 10
    acts = np.stack([actions_for(B, A, s) for s in range(2)])
    kw = dict(act_dim=A, obs_dim=2, timesteps_per_episode=5)
    ref = run_oracle(scr, acts, **kw)
    for precision, kernel in KERNELS:
        env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, precision=precision, kernel=kernel, verbose=False, **kw)
        _compare(_drive(env, acts, torch), ref, True)
        env.close()


@pytest.mark.parametrize("flavour", ["poly", "hw", "hwraw"])
def test_sincos_variant_parity(monkeypatch, flavour):
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    monkeypatch.setenv("AOG_SINCOS", flavour)
    N, B, A = 64, 4, 64
    scr = smooth_screens(B, N, 5)
    acts = np.stack([actions_for(B, A, s) for s in range(2)])
    kw = dict(act_dim=A, obs_dim=2, timesteps_per_episode=5)
    ref = run_oracle(scr, acts, **kw)
    for kernel in ("mfma", "valu"):
        env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, kernel=kernel, verbose=False, **kw)
        _compare(_drive(env, acts, torch), ref, True)
        env.close()


def test_full_size_properties_config2():
    """BASELINE configs[1] (B=1024, N=256, A=64, o=2): fast kernels vs the float64 device kernel on every env,
    action-scale invariance, Strehl in [0, 1], B=1 == env 0 of the batch, lock-step done."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch

    N, B, A = 256, 1024, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(1234)
    scr = screens_torch(B, N, 0.5 / N, cn_squared_from_fried_parameter(0.2, 2.2e-6), 10.0, dev, g, oversampling=4)
    a = torch.randn((B, A), device=dev, generator=g) * 0.7071
    kw = dict(act_dim=A, obs_dim=2, atm_fried=0.2, timesteps_per_episode=2, num_pupil_pixels=N, verbose=False)
    ref = BatchedAOEnv(B, dev, screens=scr, precision="fp64", **kw)
    ref.reset()
    r_obs0 = ref.last_obs_raw.double()
    _, r_rew, r_done, _, r_info = ref.step(a)
    for kernel in ("mfma", "valu"):
        env = BatchedAOEnv(B, dev, screens=scr, kernel=kernel, **kw)
        env.reset()
        _assert_obs_close(env.last_obs_raw.cpu().numpy(), r_obs0.cpu().numpy())
        obs, rew, done, _, info = env.step(a)
        _assert_obs_close(info["obs_raw"].cpu().numpy(), r_info["obs_raw"].cpu().numpy())
        rel = torch.abs(info["obs_raw"].double() / r_info["obs_raw"].double() - 1)
        assert float(rel.median()) < 1e-6 and float((rel > RTOL).double().mean()) < 2e-3
        assert torch.max(torch.abs(info["strehl"].double() / r_info["strehl"].double() - 1)) < RTOL
        assert torch.max(torch.abs(info["power"].double() / r_info["power"].double() - 1)) < RTOL
        assert float(info["strehl"].min()) >= 0 and float(info["strehl"].max()) <= 1
        assert not bool(done.any())
        # second step ends the 2-step episode for every env; scaled action gives the same result (AO_env.py:119-120)
        obs2, rew2, done2, _, info2 = env.step(a * 4.0)  # power of two: exact in fp32
        assert bool(done2.all())
        assert torch.max(torch.abs(info2["obs_raw"].double() / info["obs_raw"].double() - 1)) < 2e-6
        # B = 1 env built on screen 0 equals env 0 of the batch
        one = BatchedAOEnv(1, dev, screens=scr[:1], kernel=kernel, **kw)
        one.reset()
        _, _, _, _, i1 = one.step(a[:1])
        assert torch.max(torch.abs(i1["obs_raw"].double() / info["obs_raw"][:1].double() - 1)) < 2e-6
        assert abs(float(i1["strehl"][0]) - float(info["strehl"][0])) < 1e-6
        one.close()
        env.close()
    ref.close()


@pytest.mark.parametrize("N,A,o,rew", [(256, 64, 5, "smf_ssim"), (512, 20, 5, "smf_ssim"), (256, 100, 4, "strehl_ratio"), (256, 64, 3, "smf_ssim"),
                                      (240, 64, 5, "smf_ssim")])
def test_full_size_many_table_variants_vs_float64_kernel(N, A, o, rew):
    """The 12/20/28-table variants (o = 3, 4, 5: BASELINE configs[2] and [4] shapes) run their table reduction on the matrix
    cores with fp32 accumulation over a whole chunk (<= 13 tiles, 416 terms per accumulator element) and float slabs; at full
    pupil sizes they still meet the observation tolerance against the float64 device kernel, on strong von Karman screens."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch

    B = 48
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(77)
    scr = screens_torch(B, N, 0.5 / N, cn_squared_from_fried_parameter(0.15, 2.2e-6), 10.0, dev, g, oversampling=4)
    a = torch.randn((B, A), device=dev, generator=g) * 0.7071
    kw = dict(act_dim=A, obs_dim=o, rew_type=rew, act_type="num_actuators" if A > 21 else "zernike", atm_fried=0.15,
              timesteps_per_episode=3, num_pupil_pixels=N, verbose=False)
    ref = BatchedAOEnv(B, dev, screens=scr, precision="fp64", **kw)
    env = BatchedAOEnv(B, dev, screens=scr, kernel="mfma", **kw)
    assert env.info.kernel == 2
    ref.reset(); env.reset()
    _assert_obs_close(env.last_obs_raw.cpu().numpy(), ref.last_obs_raw.cpu().numpy())
    for _ in range(2):
        _, r_rew, _, _, r_info = ref.step(a)
        _, rew_, _, _, info = env.step(a)
        _assert_obs_close(info["obs_raw"].cpu().numpy(), r_info["obs_raw"].cpu().numpy())
        rel = torch.abs(info["obs_raw"].double() / r_info["obs_raw"].double() - 1)
        assert float(rel.median()) < 2e-6
        assert torch.max(torch.abs(info["strehl"].double() / r_info["strehl"].double() - 1)) < RTOL
        assert torch.max(torch.abs(info["power"].double() / r_info["power"].double() - 1)) < RTOL
        torch.testing.assert_close(rew_.double(), r_rew.double(), rtol=1e-5, atol=1e-5)
    ref.close(); env.close()


@pytest.mark.parametrize("o", [2, 5])
def test_fast_kernel_is_repeatable_at_full_size(o):
    """Identical launches give identical bits (B = 1024, N = 256: two workgroups per CU, skewed start).  Regression test for a
    race seen with packed fp32 FMAs in the table reduction: wrong low halves in lanes 16-31 once the two waves of a SIMD ran out
    of phase — different envs from run to run."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screens_torch

    N, B, A = 256, 1024, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(4321)
    scr = screens_torch(B, N, 0.5 / N, cn_squared_from_fried_parameter(0.2, 2.2e-6), 10.0, dev, g, oversampling=4)
    env = BatchedAOEnv(B, dev, screens=scr, kernel="mfma", act_dim=A, obs_dim=o, rew_type="strehl_ratio" if o == 2 else "smf_ssim",
                       atm_fried=0.2, timesteps_per_episode=2, num_pupil_pixels=N, verbose=False)
    a = torch.randn((B, A), device=dev, generator=g) * 0.7071
    env.reset()
    first_reset = env.last_obs_raw.clone()
    first_step = env.step(a)[4]["obs_raw"].clone()
    for _ in range(6):
        env.reset()
        assert torch.equal(env.last_obs_raw, first_reset)
        env.set_actuators(torch.zeros((B, A), dtype=torch.float64, device=dev))
        assert torch.equal(env.step(a)[4]["obs_raw"], first_step)
    env.close()


def test_zero_action_is_nan_like_numpy_and_other_envs_unaffected():
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N, B, A = 48, 3, 16
    scr = smooth_screens(B, N, 9)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, act_dim=A, obs_dim=2, screens=scr, verbose=False)
    env.reset()
    a = torch.from_numpy(actions_for(B, A, 3)).cuda()
    a[1] = 0                                     # AO_env.py:120 divides by std(surface) = 0
    obs, rew, done, _, info = env.step(a)
    assert bool(torch.isnan(info["obs_raw"][1]).all()) and bool(torch.isnan(rew[1]))
    assert not bool(torch.isnan(info["obs_raw"][[0, 2]]).any())
    env.close()


def test_masked_reset_and_actuator_state():
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N, B, A = 48, 4, 16
    scr = smooth_screens(B, N, 2)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, act_dim=A, obs_dim=2, screens=scr, timesteps_per_episode=3, verbose=False)
    obs_flat, _ = env.reset()
    a = torch.from_numpy(actions_for(B, A, 1)).cuda()
    env.step(a)
    act = env.get_actuators()
    assert act.dtype == torch.float64 and act.shape == (B, A) and float(act.abs().min()) > 0
    mask = torch.tensor([1, 0, 0, 1], dtype=torch.uint8)
    obs_m, _ = env.reset(mask=mask)
    act2 = env.get_actuators()
    assert float(act2[[0, 3]].abs().max()) == 0 and torch.equal(act2[[1, 2]], act[[1, 2]])
    assert torch.equal(obs_m[[0, 3]], obs_flat[[0, 3]]) and not torch.equal(obs_m[[1, 2]], obs_flat[[1, 2]])
    # per-env episode counters: envs 0,3 restart, envs 1,2 are one step into their episode
    d = [env.step(a)[2].cpu().tolist() for _ in range(3)]
    assert d == [[False] * 4, [False, True, True, False], [True, False, False, True]]
    env.close()


def test_single_env_gym_api_and_seeded_screen_chain():
    """``AOEnv`` (the drop-in) built from the process-global numpy RNG == the oracle built from the same seed:
    checks the hcipy draw order end to end (direction, stencils, 2 x (16 N)^2 normals), incl. a semi_dynamic reset."""
    _torch()
    from adaptive_optics_gym_amd.envs import AOEnv
    from oracle.ao_env_oracle import AOEnvOracle

    kw = dict(atm_type="semi_dynamic", atm_vel=5, atm_fried=0.15, act_type="zernike", act_dim=6, obs_dim=2,
              timesteps_per_episode=2, num_pupil_pixels=24, verbose=False)
    np.random.seed(123)
    env = AOEnv(**kw)
    np.random.seed(123)
    ref = AOEnvOracle(**kw)
    assert env.observation_space.shape == (4,) and env.action_space.shape == (6,)
    assert env.observation_space.dtype == np.float16
    a = np.array([0.3, -1.2, 0.5, 0.9, -0.1, 0.2], dtype=np.float32)
    for ep in range(2):
        # both draw the regenerated screen from the process-global stream: give each the same stream position
        np.random.seed(77 + ep)
        o, info = env.reset()
        np.random.seed(77 + ep)
        ro, _ = ref.reset()
        assert o.dtype == np.float16 and o.shape == (4,) and info == {}
        np.testing.assert_allclose(env.last_obs_raw, ref.last_obs_raw, rtol=RTOL)
        for t in range(2):
            o, r, d, tr, info = env.step(a)
            ro, rr, rd, _, rinfo = ref.step(a)
            assert isinstance(r, float) and isinstance(d, bool) and tr is False and set(info) == {"power"}
            np.testing.assert_allclose(env.last_obs_raw, ref.last_obs_raw, rtol=RTOL)
            np.testing.assert_allclose(r, rr, atol=1e-3)
            np.testing.assert_allclose(info["power"], rinfo["power"], rtol=RTOL)
            assert d == rd == (t == 1)
    assert env.timestep == 4 and env.episode_no == 2
    env.close()


def test_reward_threshold_ssim_guard_and_unsupported_raise():
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N = 32
    scr = smooth_screens(2, N, 4)
    with pytest.raises(ValueError):
        BatchedAOEnv(2, "cuda:0", num_pupil_pixels=N, rew_type="nope", screens=scr, verbose=False)
    env = BatchedAOEnv(2, "cuda:0", num_pupil_pixels=N, act_dim=6, act_type="zernike", obs_dim=2, rew_type="smf_ssim",
                       screens=scr, verbose=False)
    env.reset()                                   # reset works in the reference too; step raises (AO_env.py:495)
    with pytest.raises(ValueError, match="win_size exceeds image extent"):
        env.step(torch.ones(2, 6, device="cuda"))
    env.close()
    env = BatchedAOEnv(2, "cuda:0", num_pupil_pixels=N, act_dim=6, act_type="zernike", obs_dim=2, rew_threshold=-1e-9,
                       screens=scr, verbose=False)
    env.reset()
    _, r, _, _, _ = env.step(torch.ones(2, 6, device="cuda"))
    assert r.cpu().tolist() == [-1.0, -1.0]       # Strehl reward is always < 0 -> clipped to -1.0 (AO_env.py:500-501)
    env.close()


def test_device_sincos_accuracy():
    """The fused kernels' sin/cos (polynomial and hardware forms) against float64, through a tiny 1-mode env:
    obs of a pure-piston-free tilt screen has a closed form, so compare the fast kernel with the fp64 kernel
    over a sweep of screen amplitudes up to +-60 rad."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N, B = 40, 8
    yy, xx = np.mgrid[0:N, 0:N]
    scr = np.stack([(xx - N / 2) * (k + 1) * 2.5e-7 * 1.5 + (yy * yy) * 1e-9 * k for k in range(B)])
    a = torch.ones(B, 4, device="cuda")
    kw = dict(act_dim=4, act_type="zernike", obs_dim=3, num_pupil_pixels=N, screens=scr, verbose=False)
    ref = BatchedAOEnv(B, "cuda:0", precision="fp64", **kw)
    ref.reset()
    _, _, _, _, ri = ref.step(a)
    for kernel in ("mfma", "valu"):
        env = BatchedAOEnv(B, "cuda:0", kernel=kernel, **kw)
        env.reset()
        _, _, _, _, i = env.step(a)
        scale = ri["obs_raw"].double().max(dim=1, keepdim=True).values
        assert float(((i["obs_raw"].double() - ri["obs_raw"].double()).abs() / scale).max()) < 1e-5
        assert float((i["strehl"].double() - ri["strehl"].double()).abs().max()) < 2e-6
        env.close()
    ref.close()


def test_dynamic_atmosphere_matches_oracle_with_shared_numpy_stream():
    """atm_type='dynamic': the wind extrusion on the device (float64 ring-buffer screens, AR matrices, host-supplied
    normals in hcipy's order) reproduces the oracle's InfiniteAtmosphericLayer step by step, observations included."""
    torch = _torch()
    from adaptive_optics_gym_amd.envs import AOEnv
    from oracle.ao_env_oracle import AOEnvOracle

    kw = dict(atm_type="dynamic", atm_vel=45, atm_fried=0.15, act_type="zernike", act_dim=6, obs_dim=2,
              timesteps_per_episode=4, num_pupil_pixels=24, verbose=False)
    np.random.seed(5)
    env = AOEnv(**kw)
    st_env = np.random.get_state()
    np.random.seed(5)
    ref = AOEnvOracle(**kw)
    st_ref = np.random.get_state()
    np.testing.assert_allclose(env._env.velocity_vectors[0], ref.layer.velocity, rtol=1e-14)
    a = np.array([0.3, -1.2, 0.5, 0.9, -0.1, 0.2], dtype=np.float32)
    moved = 0
    for ep in range(2):
        env.reset(); ref.reset()
        np.testing.assert_allclose(env.last_obs_raw, ref.last_obs_raw, rtol=RTOL)
        for t in range(4):
            np.random.set_state(st_env)
            o, r, d, _, info = env.step(a)
            st_env = np.random.get_state()
            np.random.set_state(st_ref)
            before = ref.layer._achromatic_screen.copy()
            ro, rr, rd, _, rinfo = ref.step(a)
            st_ref = np.random.get_state()
            moved += int(not np.array_equal(before, ref.layer._achromatic_screen))
            scr = env._env.get_screens()[0].cpu().numpy().ravel()
            np.testing.assert_allclose(scr, ref.layer._achromatic_screen, rtol=1e-9, atol=1e-12 * np.abs(before).max())
            _assert_obs_close(env.last_obs_raw, ref.last_obs_raw)
            np.testing.assert_allclose(info["power"], rinfo["power"], rtol=RTOL)
            assert d == rd
    assert moved >= 6   # the screen really moved on most steps (45 m/s = 2.2 px per step)
    env.close()


def test_dynamic_atmosphere_device_rng_statistics_and_shift():
    """Batched dynamic mode with the on-device Philox stream: every step moves each screen by the whole-pixel shift hcipy
    would apply (interior pixels are copies of the previous screen), new rows/columns are finite and keep the variance."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import integer_shifts

    B, N = 6, 32
    env = BatchedAOEnv(B, "cuda:0", atm_type="dynamic", atm_vel=30, atm_fried=0.15, act_dim=6, act_type="zernike", obs_dim=2,
                       num_pupil_pixels=N, timesteps_per_episode=100, seed=3, screen_oversampling=4, verbose=False)
    env.reset()
    a = torch.ones(B, 6, device="cuda")
    var0 = float(env.get_screens().var())
    for t in range(12):
        prev = env.get_screens().cpu().numpy()
        sh = integer_shifts(env.velocity_vectors, env.timestep * env.delta_t, (env.timestep + 1) * env.delta_t, env.params.pupil_pixel)
        env.step(a)
        cur = env.get_screens().cpu().numpy()
        assert np.isfinite(cur).all()
        for b in range(B):
            dx, dy = int(sh[b, 0]), int(sh[b, 1])
            # hcipy: dx < 0 -> 'left' = new column 0, content moves to +x by one per extrusion; dx > 0 -> content moves to -x
            mx, my = (-dx if dx < 0 else -dx), (-dy if dy < 0 else -dy)
            shifted = np.roll(prev[b], shift=(-dy if dy > 0 else abs(dy), -dx if dx > 0 else abs(dx)), axis=(0, 1))
            ys = slice(abs(dy), N) if dy < 0 else slice(0, N - abs(dy))
            xs = slice(abs(dx), N) if dx < 0 else slice(0, N - abs(dx))
            np.testing.assert_array_equal(cur[b][ys, xs], shifted[ys, xs])
    var1 = float(env.get_screens().var())
    assert 0.3 * var0 < var1 < 3.0 * var0
    env.close()


def test_batched_rollout_on_device():
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.rollout import make_actor, rollout

    B, T = 8, 5
    env = BatchedAOEnv(B, "cuda:0", act_dim=16, obs_dim=2, num_pupil_pixels=32, timesteps_per_episode=T,
                       screens=smooth_screens(B, 32, 1), verbose=False)
    out = rollout(env, make_actor(4, 16, 32, device="cuda:0"), episodes=2)
    assert out["obs"].shape == (2 * T, B, 4) and out["obs"].dtype == torch.float16
    assert out["rew"].shape == (2 * T, B) and bool(torch.isfinite(out["rew"]).all())
    assert bool(out["done"][T - 1].all()) and not bool(out["done"][T - 2].any())
    assert -100.0 <= out["avg_ep_rew"] <= 0.0     # Strehl reward = 100 (S - 1)
    env.close()


def test_episode_returns_accumulate_inside_the_step():
    """aog_set_return_accumulator: the epilogue adds each step's reward into the caller's [B] float32 buffer with the caller's own
    arithmetic (returns += reward), resets add nothing, detaching stops it; the sharded gatherer uses it through attach()."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.sharding import EpisodeReturnGatherer

    N, B, A = 64, 70, 16      # 70 envs: a ragged last env tile
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=smooth_screens(B, N, 21), act_dim=A, obs_dim=2,
                       timesteps_per_episode=4, verbose=False)
    gather = EpisodeReturnGatherer(B, torch.device("cuda:0"), False)
    gather.attach(env)
    env.reset()
    gather.start_episode()
    manual = torch.zeros(B, dtype=torch.float32, device="cuda")
    for t in range(4):
        _, rew, done, _, _ = env.step(torch.from_numpy(actions_for(B, A, 30 + t)).cuda())
        gather.add(rew)                      # no-op while attached
        manual += rew
    assert bool(done.all())
    ret = gather.finish_episode()
    assert torch.equal(ret, manual) and float(manual.abs().max()) > 0
    env.reset()                              # the reset's observation pass must not touch the sums
    assert torch.equal(gather.returns, manual)
    gather.detach()
    env.step(torch.from_numpy(actions_for(B, A, 40)).cuda())
    torch.cuda.synchronize()
    assert torch.equal(gather.returns, manual)
    with pytest.raises(ValueError):
        env.accumulate_returns(torch.zeros(B + 1, dtype=torch.float32, device="cuda"))
    env.close()


def test_device_actor_matches_torch_module():
    """R1 policy query (network.py:48-69) as one launch: with dropout off the mean equals the torch module's forward (fp32
    matrix cores are exact fp32), the action is mean + sqrt(cov) eps with unit-variance eps, log_prob is the
    MultivariateNormal(mean, cov I) density of the action; with p = 0.5 half of the hidden units are dropped and the
    survivors doubled."""
    torch = _torch()
    from adaptive_optics_gym_amd.rollout import DeviceActor, make_actor

    def module_forward_f64(actor, obs):
        """The module's eval-mode forward in float64 on the host (keeps torch's GPU BLAS / solver libraries, minutes to page
        in on a fresh box, out of the GPU suite: the kernel under test does not use them)."""
        x = obs.detach().cpu().double()
        for layer in actor.hidden:
            x = torch.relu(x @ layer.weight.detach().cpu().double().T + layer.bias.detach().cpu().double())
        return x @ actor.out.weight.detach().cpu().double().T + actor.out.bias.detach().cpu().double()

    torch.manual_seed(3)
    for S, H, A, B in ((4, 150, 64, 1000), (25, 150, 20, 37), (4, 32, 16, 5), (9, 400, 6, 33)):   # last: weight matrix staged in several chunks
        actor = make_actor(S, A, H, device="cuda:0")
        with torch.no_grad():
            actor.out.weight.mul_(100.0)          # make the means O(1) so that errors would show
        obs32 = torch.rand((B, S), device="cuda") * 3
        obs16 = obs32.to(torch.float16)
        dev = DeviceActor(actor, seed=5, dropout_p=0.0)
        ref16 = module_forward_f64(actor, obs16)
        ref32 = module_forward_f64(actor, obs32)
        for obs, ref in ((obs16, ref16), (obs32, ref32)):
            action, log_prob, mean = dev(obs, 0.5)
            torch.testing.assert_close(mean.cpu().double(), ref, rtol=2e-5, atol=2e-6)
            eps = (action - mean) / math.sqrt(0.5)
            if B * A > 10000:
                assert abs(float(eps.mean())) < 0.02 and abs(float(eps.var()) - 1.0) < 0.03
                assert abs(float((eps ** 4).mean()) - 3.0) < 0.15
            # MultivariateNormal(mean, 0.5 I).log_prob(action) in closed form
            d = (action - mean).cpu().double()
            lp_ref = -0.5 * (d * d).sum(-1) / 0.5 - 0.5 * A * math.log(2 * math.pi * 0.5)
            torch.testing.assert_close(log_prob.cpu().double(), lp_ref, rtol=1e-4, atol=1e-3)
        a1, _, _ = dev(obs16, 0.5)
        a2, _, _ = dev(obs16, 0.5)
        assert not torch.equal(a1, a2)              # a new call draws new noise
    # dropout statistics on the first hidden layer: feed an actor whose later layers are identities is overkill; use the
    # fraction of exactly-zero means' change instead: with p = 0.5 the mean differs from the p = 0 mean and varies call to call
    actor = make_actor(4, 64, 150, device="cuda:0")
    obs = torch.rand((256, 4), device="cuda").to(torch.float16)
    d5 = DeviceActor(actor, seed=1, dropout_p=0.5)
    m1 = d5(obs, 0.5)[2]
    m2 = d5(obs, 0.5)[2]
    m0 = DeviceActor(actor, seed=1, dropout_p=0.0)(obs, 0.5)[2]
    assert not torch.equal(m1, m2) and not torch.equal(m1, m0)
    # E[dropout(x)] = x: averaging many masked means approaches the p = 0 mean of a LINEARISED net only; check layer 1 alone
    with torch.no_grad():
        for layer in list(actor.hidden)[1:]:
            layer.weight.copy_(torch.eye(150, device="cuda")); layer.bias.zero_()
        actor.out.weight.zero_(); actor.out.bias.zero_()
        actor.out.weight[:, :64] = torch.eye(64, device="cuda")
    base = DeviceActor(actor, seed=2, dropout_p=0.0)(obs, 0.5)[2]          # relu(W1 x + b1)[:64]
    one = DeviceActor(actor, seed=2, dropout_p=0.5)(obs, 0.5)[2]           # three dropout layers in a row: kept w.p. 1/8, scaled x8
    pos = base > 1e-3
    kept = (one[pos] != 0).float().mean()
    assert abs(float(kept) - 0.125) < 0.02
    torch.testing.assert_close(one[pos][one[pos] != 0], (8.0 * base)[pos][one[pos] != 0], rtol=1e-5, atol=1e-6)


def _assert_power_image_close(power, ref_power):
    """1e-5 relative per pixel; pixels below 1e-3 of the image's peak are held to the same ABSOLUTE error (1e-8 x peak), the rule of
    _assert_obs_close."""
    tol = RTOL * np.maximum(ref_power, 1e-3 * ref_power.max())
    bad = np.abs(power - ref_power) > tol
    assert not bad.any(), f"{bad.sum()} pixels out of tolerance, worst {np.max(np.abs(power - ref_power) / tol):.2f} x tol"


@pytest.mark.parametrize("precision", ["fast", "fp64"])
def test_focal_image_matches_literal_propagation(precision):
    """K4: the materialised 128x128 focal field == the oracle's propagator_fiber output (AO_env.py:138), and projecting it
    on the LP modes (the reference's literal fiber path, AO_env.py:471-474) == the power the fused kernel reports."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from oracle.ao_env_oracle import AOEnvOracle

    N, B, A = 64, 2, 16
    scr = smooth_screens(B, N, 8)
    a = actions_for(B, A, 2)
    kw = dict(act_dim=A, obs_dim=2, timesteps_per_episode=5)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, precision=precision, verbose=False, **kw)
    env.reset()
    _, _, _, _, info = env.step(torch.from_numpy(a).cuda())
    for b in range(B):
        ref = AOEnvOracle(num_pupil_pixels=N, screen=scr[b].ravel(), verbose=False, **kw)
        ref.reset()
        ref.step(a[b])
        F = env.focal_image(b).cpu().numpy().astype(np.complex128)
        power = np.abs(F) ** 2 * env.tables.focal_pixel_area
        _assert_power_image_close(power, ref.wf_wfs_after_foc.power.reshape(128, 128))
        coef = (env.tables.lp_modes * F[None]).sum(axis=(1, 2)) * env.tables.focal_pixel_area
        np.testing.assert_allclose(np.sum(np.abs(coef) ** 2), float(info["power"][b]), rtol=RTOL)
    env.close()


@pytest.mark.parametrize("N,B,A,act_type", [(64, 37, 16, "num_actuators"), (240, 3, 64, "num_actuators"), (256, 70, 64, "num_actuators"),
                                            (128, 5, 6, "zernike")])
def test_batched_focal_images_match_oracle_propagator(N, B, A, act_type):
    """K4 for the whole batch in one call (aog_focal_images: phase contraction + two batched complex GEMMs on the fp32 matrix cores):
    every sampled env's 128 x 128 focal-plane power == the oracle's propagator_fiber (AO_env.py:138) within 1e-5, its fiber
    projection == info["power"], the single-env entry point returns the same field, and ranges / ragged sizes work (N = 240 is not a
    multiple of the 64-wide GEMM tile; 37 and 70 envs leave partial env tiles)."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from oracle.ao_env_oracle import AOEnvOracle

    scr = smooth_screens(B, N, 80 + N)
    a = actions_for(B, A, 5)
    kw = dict(act_type=act_type, act_dim=A, obs_dim=2, timesteps_per_episode=5)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, screens=scr, verbose=False, **kw)
    env.reset()
    _, _, _, _, info = env.step(torch.from_numpy(a).cuda())
    F_all = env.focal_images()
    assert F_all.shape == (B, 128, 128) and F_all.dtype == torch.complex64
    assert torch.equal(env.focal_images(1, 2), F_all[1:3]) and torch.equal(env.focal_image(B - 1), F_all[B - 1])
    area = env.tables.focal_pixel_area
    for b in sorted({0, B // 2, B - 1}):
        ref = AOEnvOracle(num_pupil_pixels=N, screen=scr[b].ravel(), verbose=False, **kw)
        ref.reset()
        ref.step(a[b])
        F = F_all[b].cpu().numpy().astype(np.complex128)
        _assert_power_image_close(np.abs(F) ** 2 * area, ref.wf_wfs_after_foc.power.reshape(128, 128))
        coef = (env.tables.lp_modes * F[None]).sum(axis=(1, 2)) * area
        np.testing.assert_allclose(np.sum(np.abs(coef) ** 2), float(info["power"][b]), rtol=RTOL)
    # the fields of all envs: total power inside the window <= the beam's unit power, and > 0
    tot = (F_all.abs() ** 2).sum(dim=(1, 2)) * area
    assert float(tot.min()) > 0 and float(tot.max()) <= 1.0
    env.close()


@pytest.mark.parametrize("atm,o,rew,sh", [("quasi_static", 2, "strehl_ratio", False), ("semi_dynamic", 5, "smf_ssim", False),
                                          ("dynamic", 2, "strehl_ratio", False)])
def test_pipelined_stepping_is_bit_identical(atm, o, rew, sh):
    """aog_step_pipelined (the epilogue of step t and the action -> actuator prologue of step t + 1 in one launch, for callers that know
    the next action already) against plain aog_step: every output of every step and the final mirror state bit for bit, over two episodes
    with a reset between them; between two calls of a sequence the handle refuses whatever would see the half-advanced mirror."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    B, A, N, T = 70, 16, 64, 6
    kw = dict(atm_type=atm, atm_vel=20.0 if atm == "dynamic" else 0, atm_fried=0.15, act_dim=A, obs_dim=o, rew_type=rew,
              num_pupil_pixels=N, timesteps_per_episode=T, seed=9, screen_oversampling=4, verbose=False)
    acts = torch.from_numpy(np.random.RandomState(3).randn(2 * T, B, A).astype(np.float32)).cuda()

    def run(pipelined):
        env = BatchedAOEnv(B, "cuda:0", **kw)
        outs = []
        for ep in range(2):
            obs0, _ = env.reset()
            outs.append(obs0.clone())
            for t in range(T):
                k = ep * T + t
                if pipelined:
                    r = env.step(acts[k], next_actions=acts[k + 1] if t + 1 < T else None)
                    if t + 1 < T and t == 2:   # mid-sequence: the mirror already belongs to step t + 1
                        for call in (env.reset, env.get_state, lambda: env.focal_images(0, 1), lambda: env.step(acts[k + 1])):
                            with pytest.raises(RuntimeError):
                                call()
                else:
                    r = env.step(acts[k])
                outs.extend([r[0].clone(), r[1].clone(), r[2].clone(), r[4]["obs_raw"].clone(), r[4]["power"].clone(), r[4]["strehl"].clone()])
        mirror = env.get_actuators().clone()
        env.close()
        return outs, mirror

    plain, m0 = run(False)
    piped, m1 = run(True)
    assert len(plain) == len(piped)
    for a, b in zip(plain, piped):
        assert torch.equal(a, b)
    assert torch.equal(m0, m1)


def test_batched_focal_images_in_several_chunks(monkeypatch):
    """aog_focal_images works through the batch in chunks of whole env tiles (work buffers of <= 256 MB: 1024 envs at N = 256, 256 at
    N = 512).  Forced down to chunks of 32 envs here: the fields of a 100-env batch, of sub-ranges that start and end inside tiles and chunks,
    and of single envs equal the one-chunk result bit for bit."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N, B, A = 64, 100, 16
    scr = smooth_screens(B, N, 17)
    a = torch.from_numpy(actions_for(B, A, 3)).cuda()
    kw = dict(act_dim=A, obs_dim=2, num_pupil_pixels=N, timesteps_per_episode=5, verbose=False)

    def fields(chunk):
        if chunk:
            monkeypatch.setenv("AOG_FOCAL_CHUNK", str(chunk))
        else:
            monkeypatch.delenv("AOG_FOCAL_CHUNK", raising=False)
        env = BatchedAOEnv(B, "cuda:0", screens=scr, **kw)
        env.reset()
        env.step(a)
        out = env.focal_images().clone(), env.focal_images(31, 40).clone(), env.focal_images(64, 36).clone(), env.focal_image(97).clone()
        env.close()
        return out

    whole, chunked = fields(0), fields(32)
    for w, c in zip(whole, chunked):
        assert torch.equal(w, c)
    assert torch.equal(chunked[1], chunked[0][31:71]) and torch.equal(chunked[2], chunked[0][64:100]) and torch.equal(chunked[3], chunked[0][97])
    assert float(chunked[0].abs().max()) > 0


@pytest.mark.parametrize("N,vel", [(32, 35.0), (128, 20.0)])
def test_dynamic_extrusion_kernel_variants_agree(monkeypatch, N, vel):
    """The float64 extrusion kernels (``extrusion='f64'``: the validation forms since round 4) — matrix-core form with a group's rows split over
    four workgroups and a group barrier (at N = 128 its clamp-free fast form, at N = 32 the masked one; AOG_EXTRUDE_SAME_XCD: the writer-side
    short form of the group barrier for groups that measured that they share an XCD, instead of agent-scope fences in every round: bit-identical), the same in one
    workgroup per group (AOG_EXTRUDE_NOSPLIT), per-group vector form (AOG_EXTRUDE_SIMPLE) — give the same screens on
    the same Philox stream (only the float64 summation order differs; the two matrix-core forms agree to 1e-12), no inter-workgroup
    wait timed out; and the step kernel reading the fp32 ring copy directly (default) gives the observations of the per-step repack
    form (AOG_DYNAMIC_REPACK)."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    def run(mode):
        for k in ("AOG_EXTRUDE_SIMPLE", "AOG_EXTRUDE_NOSPLIT", "AOG_EXTRUDE_SAME_XCD", "AOG_DYNAMIC_REPACK"):
            monkeypatch.delenv(k, raising=False)
        if mode:
            monkeypatch.setenv(mode, "1")
        env = BatchedAOEnv(70, "cuda:0", atm_type="dynamic", atm_vel=vel, atm_fried=0.15, act_dim=6, act_type="zernike", obs_dim=2,
                           num_pupil_pixels=N, timesteps_per_episode=100, seed=11, screen_oversampling=4, verbose=False, extrusion="f64")
        assert env.info.reserved == (0 if mode == "AOG_DYNAMIC_REPACK" else 1)      # ring-direct unless asked otherwise
        env.reset()
        a = torch.ones(70, 6, device="cuda")
        obs = None
        for _ in range(7):
            obs = env.step(a)[4]["obs_raw"]
        out = env.get_screens().cpu().numpy(), obs.cpu().numpy(), env.phase_screen(3).cpu().numpy(), env.focal_image(69).cpu().numpy()
        assert env.device_status() == 0
        env.close()
        return out

    s_simple, o_simple, ph_simple, f_simple = run("AOG_EXTRUDE_SIMPLE")
    s_split = None
    for mode in (None, "AOG_EXTRUDE_SAME_XCD", "AOG_EXTRUDE_NOSPLIT", "AOG_DYNAMIC_REPACK"):
        s_other, o_other, ph_other, f_other = run(mode)
        np.testing.assert_allclose(s_other, s_simple, rtol=1e-9, atol=1e-12 * np.abs(s_simple).max())
        _assert_obs_close(o_other, o_simple)
        # radians (fp32 screens), up to the piston: the repack form subtracts the aperture mean measured one step earlier
        ap = ph_simple != 0
        np.testing.assert_allclose(ph_other[ap] - ph_other[ap].mean(), ph_simple[ap] - ph_simple[ap].mean(), rtol=0, atol=3e-5)
        assert not ph_other[~ap].any()
        np.testing.assert_allclose(np.abs(f_other), np.abs(f_simple), rtol=0, atol=1e-5 * np.abs(f_simple).max())   # (global phase = piston)
        if mode is None:
            s_split = s_other
        elif mode == "AOG_EXTRUDE_SAME_XCD":
            assert np.array_equal(s_other, s_split)      # the same arithmetic behind a different fence: bit for bit
        elif mode == "AOG_EXTRUDE_NOSPLIT":
            np.testing.assert_allclose(s_other, s_split, rtol=1e-11, atol=1e-13 * np.abs(s_split).max())


@pytest.mark.parametrize("N,vel", [(64, 70.0), (96, 45.0), (240, 12.0)])
def test_dynamic_ring_direct_matches_oracle(N, vel):
    """The step kernel reading the screens straight from the toroidal fp32 ring (per-lane 16-byte loads through the origin offsets, groups
    that straddle an aperture row end patched from the next row) against the oracle's InfiniteAtmosphericLayer fed the same normals:
    several pupil sizes (row lengths not multiples of 4 -> straddling groups; 240 = the reference's size), winds that wrap the ring."""
    torch = _torch()
    from helpers import ScriptedRNG, device_mode_stencil_draws
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import integer_shifts
    from oracle.ao_env_oracle import AOEnvOracle

    B, A, T, seed = 37, 16, 12, 4
    kw = dict(atm_type="dynamic", atm_vel=vel, atm_fried=0.15, act_type="num_actuators", act_dim=A, obs_dim=2, timesteps_per_episode=T)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, seed=seed, screen_source="device", screen_oversampling=4, verbose=False, **kw)
    assert env.info.reserved == 1
    geo = device_mode_stencil_draws(seed, B, N)
    ids = [0, 17, B - 1]
    refs = {}
    for b in ids:
        refs[b] = AOEnvOracle(num_pupil_pixels=N, screen=env.get_screens(b, 1)[0].cpu().numpy().ravel(),
                              rng=ScriptedRNG(env.wind_u[b], [g.copy() for g in geo]), verbose=False, **kw)
    env.reset()
    for b in ids:
        refs[b].reset()
        _assert_obs_close(env.last_obs_raw[b].double().cpu().numpy(), refs[b].last_obs_raw)
    gen = torch.Generator("cuda").manual_seed(5)
    for t in range(2 * T):                       # two episodes: the origins travel more than once around the ring at the fast winds
        a = torch.randn((B, A), device="cuda", generator=gen)
        counts = np.abs(integer_shifts(env.velocity_vectors, env.timestep * env.delta_t, (env.timestep + 1) * env.delta_t,
                                       env.params.pupil_pixel)).sum(axis=1)
        noise = torch.randn((B, max(int(counts.max()), 1), N), device="cuda", dtype=torch.float64, generator=gen)
        env.set_extrusion_noise(noise)
        obs, rew, done, _, info = env.step(a)
        for b in ids:
            refs[b].rng.normals.extend(noise[b, :int(counts[b])].cpu().numpy())
            _, r_rew, r_done, _, r_info = refs[b].step(a[b].cpu().numpy())
            _assert_obs_close(info["obs_raw"][b].double().cpu().numpy(), refs[b].last_obs_raw)
            np.testing.assert_allclose(float(info["strehl"][b]), refs[b].last_strehl, rtol=RTOL)
            np.testing.assert_allclose(float(info["power"][b]), r_info["power"], rtol=RTOL)
            assert bool(done[b]) == r_done
        if bool(done.all()):
            env.reset()
            for b in ids:
                refs[b].reset()
                _assert_obs_close(env.last_obs_raw[b].double().cpu().numpy(), refs[b].last_obs_raw)
    assert env.device_status() == 0
    env.close()


@pytest.mark.parametrize("method", ["twoband", "hcipy16"])
def test_device_screen_synthesis_statistics(method):
    """K8 inside the library (Philox normals; both synthesis methods): same variance / structure function as the literal numpy generator and
    as the discrete integral of the von Karman PSD over the FFT grid; new screens on every call; masked regeneration."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, screen_numpy, spectral_amplitude

    B, N, q = 256, 32, 8
    env = BatchedAOEnv(B, "cuda:0", atm_type="dynamic", atm_vel=1, atm_fried=0.2, act_dim=6, act_type="zernike", obs_dim=2,
                       num_pupil_pixels=N, seed=5, screen_oversampling=q, screen_source="device", screen_method=method, verbose=False)
    dev = env.get_screens().cpu().numpy()
    assert np.isfinite(dev).all()
    cn2 = cn_squared_from_fried_parameter(0.2, 2.2e-6)
    delta = 0.5 / N
    m = N * q
    var_expect = (spectral_amplitude(N, delta, 10.0, q) ** 2).sum() / (float(m) * m * delta ** 4) / (m * m) * cn2 * (m * m) / (m * m)
    # Var = sum a^2 / (M^2 delta^4) * Cn^2  (see atmosphere_host.screen_numpy)
    var_expect = (spectral_amplitude(N, delta, 10.0, q) ** 2).sum() / (float(m) ** 4 * delta ** 4) * cn2
    np.testing.assert_allclose(dev.var(), var_expect, rtol=0.15)
    rng = np.random.RandomState(0)
    lit = np.stack([screen_numpy(N, delta, cn2, 10.0, rng, q) for _ in range(200)])
    d_dev = np.mean((dev[:, :, 2:] - dev[:, :, :-2]) ** 2)
    d_lit = np.mean((lit[:, :, 2:] - lit[:, :, :-2]) ** 2)
    np.testing.assert_allclose(d_dev, d_lit, rtol=0.1)
    d_dev_y = np.mean((dev[:, 5:, :] - dev[:, :-5, :]) ** 2)
    d_lit_y = np.mean((lit[:, 5:, :] - lit[:, :-5, :]) ** 2)
    np.testing.assert_allclose(d_dev_y, d_lit_y, rtol=0.1)
    assert abs(np.corrcoef(dev[0].ravel(), dev[1].ravel())[0, 1]) < 0.9 and not np.array_equal(dev[0], dev[1])
    # masked regeneration: only the selected envs change
    mask = np.zeros(B, dtype=bool); mask[[3, 4, 5, 200]] = True
    env._generate_screens(mask=torch.from_numpy(mask))
    new = env.get_screens().cpu().numpy()
    changed = np.array([not np.array_equal(new[b], dev[b]) for b in range(B)])
    assert np.array_equal(changed, mask)
    env.close()


@pytest.mark.parametrize("method,N,q", [("hcipy16", 64, 8), ("hcipy16", 128, 4), ("hcipy16", 256, 16), ("hcipy16", 512, 2), ("hcipy16", 60, 8),
                                        ("hcipy16", 120, 4), ("hcipy16", 240, 16), ("hcipy16", 480, 2),
                                        ("twoband", 64, 16), ("twoband", 128, 8), ("twoband", 256, 16), ("twoband", 512, 4), ("twoband", 60, 16),
                                        ("twoband", 120, 8), ("twoband", 240, 16), ("twoband", 480, 16)])
def test_pruned_screen_synthesis_matches_full_transform(monkeypatch, method, N, q):
    """K8, pupils of 64 R or 60 R pixels (R = 1, 2, 4, 8; 240 is the reference's size): the pruned two-pass synthesis (Philox lines -> length-N transforms in registers/LDS, never the
    (qN)^2 array) gives the same screens as spectrum fill + hipFFT + centred crop on the same Philox stream; only fp32 rounding
    (and the float amplitude law) differs.  Covers 1, 2, 4 and 8 points per lane, one or two b-groups per line, and both the
    radix-2 64-point and the mixed-radix (2 x 2 x 3 x 5) 60-point in-register transform.  Two-band method: the high band's pruned passes
    (32 / R lines or columns per wave) against hipFFT on the (2N)^2 grid, and the low band's in-kernel direct sums (twiddles by recurrence)
    against the one-thread-per-output kernels (every twiddle from an exactly reduced angle)."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    def run(full):
        if full:
            monkeypatch.setenv("AOG_SCREENS_FULLFFT", "1")
        else:
            monkeypatch.delenv("AOG_SCREENS_FULLFFT", raising=False)
        env = BatchedAOEnv(3, "cuda:0", atm_type="semi_dynamic", atm_fried=0.15, act_dim=6, act_type="zernike", obs_dim=2,
                           num_pupil_pixels=N, timesteps_per_episode=5, seed=7, screen_oversampling=q, screen_method=method, verbose=False)
        env.reset()
        first = np.stack([env.phase_screen(i).cpu().numpy() for i in range(3)])
        env.reset()                                   # semi_dynamic: a new screen per episode
        second = env.phase_screen(0).cpu().numpy()
        env.close()
        return first, second

    f1, f2 = run(True)
    p1, p2 = run(False)
    rms = f1.std()
    assert rms > 0 and not np.allclose(f1[0], f1[1]) and not np.allclose(f1[0], f2)
    assert np.abs(p1 - f1).max() < 3e-5 * rms
    assert np.abs(p2 - f2).max() < 3e-5 * rms


@pytest.mark.parametrize("first,count,B", [(0, 70, 70), (5, 70, 100), (33, 8, 64), (31, 34, 65)])
def test_batched_screen_installation_equals_one_by_one(first, count, B):
    """``set_screens`` converts batches of >= 8 screens with the tiled kernels (k_screen_means + k_pack_tiles: whole-line stores per env
    tile) and fewer with k_pack_screens (one workgroup per env): both must store bit-identical values, for env ranges that start and end
    inside env tiles of 32, and must leave the envs outside the range alone."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    N = 64
    rng = np.random.RandomState(first * 100 + count)
    base = smooth_screens(B, N, 3)
    new = smooth_screens(count, N, 4) * 1.7 + 1e-6 * rng.randn(count, N, N)
    kw = dict(act_type="zernike", act_dim=6, obs_dim=2, num_pupil_pixels=N, verbose=False)
    a = BatchedAOEnv(B, "cuda:0", screens=base, **kw)
    b = BatchedAOEnv(B, "cuda:0", screens=base, **kw)
    a.set_screens(new, first=first)                         # one call: the tiled path
    for i in range(count):
        b.set_screens(new[i:i + 1], first=first + i)        # one env at a time: k_pack_screens
    sa, sb = a.get_screens().cpu(), b.get_screens().cpu()
    assert torch.equal(sa, sb)
    untouched = [i for i in range(B) if not (first <= i < first + count)]
    if untouched:
        ref = BatchedAOEnv(B, "cuda:0", screens=base, **kw)
        assert torch.equal(sa[untouched], ref.get_screens().cpu()[untouched])
        ref.close()
    act = torch.from_numpy(rng.randn(B, 6).astype(np.float32)).cuda()
    assert torch.equal(a.step(act)[4]["obs_raw"], b.step(act)[4]["obs_raw"])
    a.close(); b.close()


def test_switching_the_screen_method_gives_the_workspace_back():
    """``aog_set_screen_method`` between resets: each method draws its own (reproducible) stream, and the workspace of the method that is
    left is released instead of piling up until ``aog_destroy`` (the literal form's is 4 MB per env at N = 64, q = 16)."""
    _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    env = BatchedAOEnv(64, "cuda:0", atm_type="semi_dynamic", act_dim=6, act_type="zernike", obs_dim=2, num_pupil_pixels=64, seed=3, verbose=False)
    env.reset()
    two = env.phase_screen(1).cpu().numpy()
    base = env.device_bytes()
    sizes = []
    for _ in range(3):
        env.set_screen_method("hcipy16")
        env.reset()
        lit = env.phase_screen(1).cpu().numpy()
        sizes.append(env.device_bytes())
        env.set_screen_method("twoband")
        env.reset()
        sizes.append(env.device_bytes())
    assert np.isfinite(lit).all() and lit.std() > 0 and not np.array_equal(lit, two)
    assert sizes[0] == sizes[2] == sizes[4] and sizes[1] == sizes[3] == sizes[5] == base     # nothing accumulates
    assert sizes[0] > base                                                                   # (the literal workspace is the larger one)
    env.close()


@pytest.mark.parametrize("method,N", [("twoband", 64), ("twoband", 60), ("twoband", 96), ("hcipy16", 64)])
def test_device_screens_have_the_literal_covariance(method, N):
    """Monte-Carlo check of the device output against the EXACT covariance of hcipy's literal method (the cosine sum over its (16 N)^2
    spectrum, ``atmosphere_host.literal_covariance``; tests/test_screen_twoband.py shows the two-band model's covariance equals it to 2e-5
    C(0) at every lag): structure function along x, y and the diagonal from 1 pixel to 3/4 of the pupil over 4096 screens (3 sigma of the
    estimator: ~2 % at short lags, ~7 % at the longest, where a screen contributes about one degree of freedom; the seeds are fixed), and independence of the
    screens of different envs.  N = 64 / 60: pruned passes (64- and 60-point forms); N = 96: hipFFT route + direct low band."""
    _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import cn_squared_from_fried_parameter, literal_covariance

    B = 4096
    env = BatchedAOEnv(B, "cuda:0", atm_type="dynamic", atm_vel=1, atm_fried=0.15, act_dim=6, act_type="zernike", obs_dim=2,
                       num_pupil_pixels=N, seed=21, screen_source="device", screen_method=method, verbose=False)
    dev = env.get_screens().cpu().numpy()      # dynamic handle: the full float64 master screens, piston included
    env.close()
    assert np.isfinite(dev).all()
    cn2 = cn_squared_from_fried_parameter(0.15, 2.2e-6)
    cov = literal_covariance(N, 0.5 / N, 10.0, 16) * cn2
    c0 = cov[N - 1, N - 1]
    for r, rtol in ((1, 0.025), (2, 0.025), (4, 0.03), (N // 8, 0.04), (N // 4, 0.06), (N // 2, 0.08), (3 * N // 4, 0.09)):
        np.testing.assert_allclose(np.mean((dev[:, :, r:] - dev[:, :, :-r]) ** 2), 2 * (c0 - cov[N - 1, N - 1 + r]), rtol=rtol)
        np.testing.assert_allclose(np.mean((dev[:, r:, :] - dev[:, :-r, :]) ** 2), 2 * (c0 - cov[N - 1 + r, N - 1]), rtol=rtol)
        np.testing.assert_allclose(np.mean((dev[:, r:, r:] - dev[:, :-r, :-r]) ** 2), 2 * (c0 - cov[N - 1 + r, N - 1 + r]), rtol=rtol)
        np.testing.assert_allclose(np.mean((dev[:, r:, :-r] - dev[:, :-r, r:]) ** 2), 2 * (c0 - cov[N - 1 + r, N - 1 - r]), rtol=rtol)
    # screens of different envs are independent: the mean over envs of a product of two envs' centre-pixel differences vanishes
    tilt = dev[:, N // 2, 3 * N // 4] - dev[:, N // 2, N // 4]
    assert abs(np.mean(tilt[0::2] * tilt[1::2])) < 4 * tilt.var() / np.sqrt(B / 2)
    assert abs(np.mean(tilt)) < 4 * tilt.std() / np.sqrt(B)


def test_shack_hartmann_chain_on_the_impulse_response_fresnel_branch():
    """hcipy's FresnelPropagator switches to its impulse-response transfer function when the pupil pitch falls below lambda z / L — the
    reference's geometry (AO_env.py:407, f-number 50) does above ~800 pupil pixels.  A lenslet f-number of 600 puts a 96-pixel pupil on that
    branch: the device chain (the table is uploaded, it factorises like the analytic one) against the oracle's — noise-free sensor image,
    estimator + reconstructor + integrator on the oracle's noisy image, and the env step that consumes the actuators."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.params import OpticalParams
    from oracle.ao_env_oracle import AOEnvOracle

    N, A, F = 96, 8, 600.0
    scr = smooth_screens(1, N, 3)[0] * 0.5
    kw = dict(act_type="zernike", act_dim=A, obs_dim=2, timesteps_per_episode=50, SH_operation=True, verbose=False)
    env = BatchedAOEnv(1, "cuda:0", screens=scr[None], sh_fft_precision="double", params=OpticalParams(num_pupil_pixels=N, f_number=F), **kw)
    ref = AOEnvOracle(screen=scr.ravel(), rng=np.random.RandomState(42), num_pupil_pixels=N, f_number=F, **kw)
    assert ref.shwfs.propagator.uses_impulse_response(ref.wavelength_wfs)
    env.reset(); ref.reset()
    for t in range(3):
        clean = env.sh_image()[0].cpu().numpy()
        ra, _ = ref.SH_step()
        np.testing.assert_allclose(clean, ref.last_sh_image_noiseless, rtol=1e-5, atol=1e-7 * ref.last_sh_image_noiseless.max())
        noisy = np.round(ref.last_sh_image_noiseless + (ref.last_sh_noisy - ref.last_sh_image_noiseless))
        a = env.sh_update(noisy[None])[0].cpu().numpy()
        np.testing.assert_allclose(a, ra, rtol=1e-6, atol=1e-6 * np.abs(ra).max())
        _, _, _, _, info = env.step(torch.from_numpy(a[None]).cuda())
        ref.step(ra)
        _assert_obs_close(info["obs_raw"].cpu().numpy()[0], ref.last_obs_raw)
        np.testing.assert_allclose(float(info["strehl"][0]), ref.last_strehl, rtol=1e-5)
    env.close()


@pytest.mark.parametrize("N", [96, 240])
def test_shack_hartmann_chain_matches_oracle(N):
    """SH_step (AO_env.py:254-290) on the device vs the oracle (N = 240 is the reference's pupil size), stage by stage: the noise-free sensor image; then the
    estimator + reconstructor + leaky integrator fed with the ORACLE's photon-noisy image (a Poisson stream cannot be replayed
    on images that differ in the last bits); then the env step that consumes the actuators."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from oracle.ao_env_oracle import AOEnvOracle

    A = 8
    scr = smooth_screens(1, N, 3)[0] * 0.5
    kw = dict(act_type="zernike", act_dim=A, obs_dim=2, timesteps_per_episode=50, num_pupil_pixels=N, SH_operation=True, verbose=False)
    env = BatchedAOEnv(1, "cuda:0", screens=scr[None], sh_fft_precision="double", **kw)      # complex128 transforms: image parity to 1e-5
    env32 = BatchedAOEnv(1, "cuda:0", screens=scr[None], **kw)                                # the default complex64 transforms
    ref = AOEnvOracle(screen=scr.ravel(), rng=np.random.RandomState(42), **kw)
    env.reset(); env32.reset(); ref.reset()
    strehl = []
    for t in range(6):
        clean = env.sh_image()[0].cpu().numpy()
        ra, _ = ref.SH_step()
        np.testing.assert_allclose(clean, ref.last_sh_image_noiseless, rtol=1e-5, atol=1e-7 * ref.last_sh_image_noiseless.max())
        # complex64: ~1e-6 of the image peak — three orders below the photon noise large_poisson adds before the estimator reads it
        clean32 = env32.sh_image()[0].cpu().numpy()
        np.testing.assert_allclose(clean32, clean, rtol=0, atol=5e-6 * clean.max())
        # replay the oracle's photon noise exactly: noisy = what its large_poisson produced
        noisy = np.round(ref.last_sh_image_noiseless + (ref.last_sh_noisy - ref.last_sh_image_noiseless))
        a = env.sh_update(noisy[None])[0].cpu().numpy()
        np.testing.assert_allclose(a, ra, rtol=1e-6, atol=1e-6 * np.abs(ra).max())
        # the estimator does not depend on the transform precision (float64 LDS atomics: equal to rounding, not bit for bit)
        np.testing.assert_allclose(env32.sh_update(noisy[None])[0].cpu().numpy(), a, rtol=1e-10, atol=0)
        env32.step(torch.from_numpy(a[None]).cuda())
        _, _, _, _, info = env.step(torch.from_numpy(a[None]).cuda())
        ref.step(ra)
        _assert_obs_close(info["obs_raw"].cpu().numpy()[0], ref.last_obs_raw)
        np.testing.assert_allclose(float(info["strehl"][0]), ref.last_strehl, rtol=1e-5)
        strehl.append(ref.last_strehl)
    assert strehl[-1] > strehl[0]   # the leaky integrator closes the loop
    # the single-env drop-in exposes the reference signature: (actuators [A] float64, torch.tensor([1]))
    from adaptive_optics_gym_amd.envs import AOEnv
    one = AOEnv(screens=scr[None], rng=np.random.RandomState(1), **kw)
    one.reset()
    act, la = one.SH_step()
    assert act.dtype == np.float64 and act.shape == (A,) and la.tolist() == [1]
    one.close()
    env.close()
    env32.close()


@pytest.mark.parametrize("N,B", [(128, 5), (240, 3), (256, 3), (480, 2), (512, 2)])
def test_shack_hartmann_pruned_propagation_matches_2d_transforms(N, B):
    """Pupils of 128 / 256 / 512 pixels (lines of 64 R) and of 240 pixels (the reference's size: lines of 60 R) run the Fresnel propagation as three pruned passes of in-register length-2N transforms (complex64:
    k_sh_rows_fwd, k_sh_cols, k_sh_rows_inv) instead of zero-padded 2-D FFTs; the detector image must equal the complex128 2-D route
    (hipFFT Z2Z, the form the oracle test pins at N = 96 / 240) to complex64 rounding, for every env of the batch, also after the mirror
    has moved."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    scr = smooth_screens(B, N, 5) * 0.5
    kw = dict(act_type="zernike", act_dim=8, obs_dim=2, timesteps_per_episode=50, num_pupil_pixels=N, SH_operation=True, verbose=False)
    env64 = BatchedAOEnv(B, "cuda:0", screens=scr, sh_fft_precision="double", **kw)
    env32 = BatchedAOEnv(B, "cuda:0", screens=scr, **kw)
    env64.reset(); env32.reset()
    for it in range(2):
        ref = env64.sh_image().cpu().numpy()
        got = env32.sh_image().cpu().numpy()
        assert ref.shape == (B, N * N) and ref.max() > 0
        for b in range(B):
            np.testing.assert_allclose(got[b], ref[b], rtol=0, atol=5e-6 * ref[b].max())
        assert not np.allclose(ref[0], ref[1], rtol=1e-3, atol=1e-6 * ref.max())
        # the same (deterministically rounded) camera frame into both estimators: both Shack-Hartmann mirrors move identically
        noisy = np.round(ref)
        a = env64.sh_update(noisy)
        torch.testing.assert_close(env32.sh_update(noisy), a, rtol=1e-6, atol=1e-8 * float(a.abs().max()))   # (order of the float64 atomics)
        env64.step(a)
        env32.step(a)
    env64.close()
    env32.close()
    # SH_step never asks for the image: photon noise and the lenslet sums are then taken inside the last propagation pass.  Same Philox
    # stream, same pixels -> the actuators of the image -> k_sh_noise -> k_sh_estimate route (float64 sums in another order)
    fused = BatchedAOEnv(B, "cuda:0", screens=scr, seed=21, **kw)
    plain = BatchedAOEnv(B, "cuda:0", screens=scr, seed=21, **kw)
    fused.reset(); plain.reset()
    for it in range(3):
        a_f, _ = fused.SH_step()
        plain.sh_image()
        a_p = plain.sh_update(None)
        # (the two row kernels are separate instantiations: their images agree to complex64 rounding, and a pixel whose expectation sits
        # within that of a rounding boundary of the sampler draws one count more or less in one of them — a few 1e-7 of an actuator;
        # a wrong stream or pixel mapping would show at order 1)
        torch.testing.assert_close(a_f, a_p, rtol=1e-4, atol=1e-5 * float(a_p.abs().max()))
        assert float(a_f.abs().max()) > 0
        fused.step(a_f)
        plain.step(a_p)
    fused.close()
    plain.close()


def test_shack_hartmann_three_pass_form_matches_the_separable_form(monkeypatch):
    """The separable two-pass propagation (default: hcipy's Fresnel transfer function factorises) and the three-pass form kept for transfer
    functions that do not (forced here with AOG_SH_THREE_PASS=1) give the same camera image to complex64 rounding, and each form's fused
    noise + lenslet-sum path closes the loop (the two forms key the photon-noise stream differently, so only statistics are shared)."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    B, N = 3, 256
    scr = smooth_screens(B, N, 9) * 0.5
    kw = dict(act_type="zernike", act_dim=8, obs_dim=2, timesteps_per_episode=50, num_pupil_pixels=N, SH_operation=True, verbose=False, seed=4)
    sep = BatchedAOEnv(B, "cuda:0", screens=scr, **kw)
    monkeypatch.setenv("AOG_SH_THREE_PASS", "1")
    three = BatchedAOEnv(B, "cuda:0", screens=scr, **kw)
    monkeypatch.delenv("AOG_SH_THREE_PASS")
    sep.reset(); three.reset()
    a = sep.sh_image().cpu().numpy()
    b = three.sh_image().cpu().numpy()
    for e in range(B):
        np.testing.assert_allclose(a[e], b[e], rtol=0, atol=5e-6 * b[e].max())
    s0 = sep.step(torch.zeros((B, 8), device="cuda"))[4]["strehl"].cpu().numpy()
    for env in (sep, three):
        for _ in range(10):
            act, _ = env.SH_step()
            s1 = env.step(act)[4]["strehl"].cpu().numpy()
        assert np.all(s1 > s0)          # both controllers flatten the wavefront
        env.close()


def test_shack_hartmann_device_noise_closed_loop():
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    B, N, A = 4, 96, 8
    scr = smooth_screens(B, N, 11) * 0.5
    env = BatchedAOEnv(B, "cuda:0", act_type="zernike", act_dim=A, obs_dim=2, timesteps_per_episode=50, num_pupil_pixels=N,
                       SH_operation=True, screens=scr, verbose=False)
    env.reset()
    s = []
    for t in range(12):
        a, _ = env.SH_step()
        assert a.shape == (B, A) and a.dtype == torch.float64 and bool(torch.isfinite(a).all())
        s.append(env.step(a)[4]["strehl"].cpu().numpy())
    assert np.all(s[-1] > s[0]) and np.all(s[-1] <= 1.0)
    env.close()


@pytest.mark.parametrize("lam", [0.02, 0.5, 3.0, 11.0, 11.99, 40.0, 2000.0])
def test_device_poisson_sampler_matches_scipy(lam):
    """The camera's photon-noise sampler (``large_poisson``, AO_env.py:272-275; device: exact inversion below 12 counts walked four terms
    per wave vote in fp32, skew-corrected rounded normal above) against ``scipy.stats.poisson``: chi-square of 1.05 M draws over the bins
    with expectation >= 20, plus mean / variance / third central moment.  Below the switch the law is Poisson to ~1e-6 in total
    variation, so the statistic is chi-square distributed; above it the rounded normal matches three moments but not every bin, so there
    only the moments are held (the sensor reads flux-weighted centroids)."""
    import ctypes as C

    from scipy import stats

    torch = _torch()
    from adaptive_optics_gym_amd import _lib

    lib = _lib.load()
    n_env, n = 16, 256
    x = torch.full((n_env, n, n), float(lam), dtype=torch.float64, device="cuda")
    out = torch.empty_like(x)
    _lib.check(lib.aog_selftest_poisson(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), n_env, n, 77, 3, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    k = out.cpu().numpy().ravel()
    assert np.all(k >= 0) and np.all(k == np.rint(k))
    m = k.size
    se_mean = np.sqrt(lam / m)
    assert abs(k.mean() - lam) < 5 * se_mean
    assert abs(k.var() - lam) < 5 * lam * np.sqrt(2.0 / m + 1.0 / (lam * m))           # var of the sample variance of a Poisson law
    mu3 = np.mean((k - k.mean()) ** 3)
    assert abs(mu3 - lam) < 6 * np.sqrt((15 * lam ** 3 + 25 * lam ** 2 + lam) / m) + 0.02 * lam   # third central moment of Poisson = lam
    if lam < 12:
        kmax = int(k.max())
        obs = np.bincount(k.astype(np.int64), minlength=kmax + 1).astype(np.float64)
        exp = stats.poisson.pmf(np.arange(kmax + 1), lam) * m
        keep = exp >= 20
        obs_k, exp_k = obs[keep], exp[keep]
        obs_rest, exp_rest = obs[~keep].sum(), m - exp_k.sum()
        chi2 = ((obs_k - exp_k) ** 2 / exp_k).sum() + ((obs_rest - exp_rest) ** 2 / exp_rest if exp_rest >= 5 else 0.0)
        dof = keep.sum() - 1 + (1 if exp_rest >= 5 else 0)
        assert chi2 < stats.chi2.ppf(1 - 1e-4, dof), (chi2, dof)
    # different call index / env: different draws
    _lib.check(lib.aog_selftest_poisson(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), n_env, n, 77, 4, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    k2 = out.cpu().numpy().ravel()
    assert not np.array_equal(k, k2) and not np.array_equal(k[:n * n], k[n * n:2 * n * n])


def test_state_save_restore_resumes_bit_identically():
    """get_state / set_state (checkpointing; the reference has none for the env): a dynamic-atmosphere batch with Shack-Hartmann
    state resumes from a snapshot with bit-identical observations, rewards and screens."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    def make():
        return BatchedAOEnv(5, "cuda:0", atm_type="dynamic", atm_vel=30, atm_fried=0.15, act_dim=8, act_type="zernike", obs_dim=2,
                            num_pupil_pixels=48, timesteps_per_episode=4, seed=9, screen_oversampling=4, SH_operation=True, verbose=False)

    env = make()
    env.reset()
    def run(e, n):
        out = []
        for _ in range(n):
            a, _ = e.SH_step()
            o, r, d, _, info = e.step(a)
            out.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), info["obs_raw"].cpu().numpy().copy()))
            if bool(d.all()):
                e.reset()
        return out

    run(env, 3)
    snap = env.get_state()
    first = run(env, 5)
    scr_first = env.get_screens().cpu().numpy()
    other = make()          # a different object: fresh handle, same configuration
    other.set_state(snap)
    second = run(other, 5)
    for x, y in zip(first, second):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(u, v)
    np.testing.assert_array_equal(other.get_screens().cpu().numpy(), scr_first)
    assert other.timestep == env.timestep
    with pytest.raises(ValueError):
        BatchedAOEnv(2, "cuda:0", act_dim=8, act_type="zernike", num_pupil_pixels=48, screens=smooth_screens(2, 48, 1), verbose=False).set_state(snap)
    env.close(); other.close()


def test_render_data_and_phase_screen():
    _torch()
    from adaptive_optics_gym_amd.envs import AOEnv

    N = 64
    scr = smooth_screens(1, N, 4)
    env = AOEnv(act_dim=8, act_type="zernike", obs_dim=5, num_pupil_pixels=N, screens=scr, rng=np.random.RandomState(0), verbose=False)
    env.reset()
    env.step(np.ones(8, dtype=np.float32))
    d = env.render_data()
    assert d["phase_screen_opd"].shape == (N, N) and d["focal_power"].shape == (128, 128) and d["obs_power"].shape == (5, 5)
    ap = env._env.tables.ap_index
    expect = scr[0].ravel()[ap]
    expect = (expect - expect.mean()) * 1e6 / (2 * np.pi)            # achromatic screen (phase x lambda) -> OPD in micrometres
    np.testing.assert_allclose(d["phase_screen_opd"].ravel()[ap], expect, rtol=1e-5, atol=1e-6)
    assert np.count_nonzero(np.delete(d["phase_screen_opd"].ravel(), ap)) == 0
    assert 0.05 < d["focal_power"].sum() <= 1.0 and d["focal_power"].min() >= 0      # power-1 beam; the aberrated part misses the window
    env.close()
