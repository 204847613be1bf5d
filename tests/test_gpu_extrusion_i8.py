"""Dynamic atmosphere on the int8 matrix cores (csrc/k_extrude_i8.h; the default whenever the composite operators cover a step's shifts):
against the float64 round kernels on the same device random stream, against the oracle's InfiniteAtmosphericLayer over a 30-step episode with
replayed normals (observations, Strehl, fiber power at north_star's 1e-5; the screen error is reported), under the group-barrier stress the
float64 kernels need, and for the bookkeeping both forms share (origins, stream positions, state save / restore)."""
import numpy as np
import pytest

from helpers import ScriptedRNG, device_mode_stencil_draws

pytestmark = pytest.mark.gpu
RTOL = 1e-5
LAM_WFS = 1.5e-6


def _torch():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _obs_close(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    peak = ref.max(axis=-1, keepdims=True)
    bad = np.abs(got - ref) > RTOL * np.maximum(np.abs(ref), 1e-3 * peak)
    assert not bad.any(), f"max rel err {np.max(np.abs(got - ref) / np.abs(ref)):.3e}, {bad.sum()} elements out of tolerance"


@pytest.mark.parametrize("N,vel,B", [(64, 10, 96), (96, 31, 40), (240, 10, 70), (128, 3, 33)])
def test_int8_extrusion_matches_the_float64_kernels(N, vel, B, record_property):
    """Same seeds, same Philox normals: the composite int8 form and the chain of one-pixel float64 rounds leave the same screens (origins and
    stream positions exactly; samples to a few 1e-8 rad of ~5 rad rms after a dozen steps — the operands' quantisation, amplified by the AR
    recursion) and the same step outputs.  Winds in every direction (all four of hcipy's 'left' / 'right' / 'bottom' / 'top' forms), shift counts
    from 0 to 7 per axis and step, pupil sizes that are not multiples of 64 (240 = the reference's)."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    T = 12
    kw = dict(atm_type="dynamic", atm_vel=vel, atm_fried=0.15, act_type="num_actuators", act_dim=16, obs_dim=2, timesteps_per_episode=T,
              num_pupil_pixels=N, seed=4, screen_source="device", screen_oversampling=4, verbose=False)
    e8 = BatchedAOEnv(B, "cuda:0", **kw)
    e64 = BatchedAOEnv(B, "cuda:0", extrusion="f64", **kw)
    assert e8.extrusion_kmax >= 1 and e64.extrusion_kmax == 0
    assert torch.equal(e8.get_screens(), e64.get_screens())
    e8.reset()
    e64.reset()
    gen = torch.Generator("cuda").manual_seed(5)
    worst = 0.0
    for t in range(T):
        a = torch.randn((B, 16), device="cuda", generator=gen)
        o8, o64 = e8.step(a), e64.step(a)
        s8, s64 = e8.get_screens(), e64.get_screens()
        err = float((s8 - s64).abs().max()) / LAM_WFS
        worst = max(worst, err)
        assert err < 1e-6, f"step {t}: screens differ by {err:.2e} rad"
        np.testing.assert_allclose(o8[4]["strehl"].cpu().numpy(), o64[4]["strehl"].cpu().numpy(), rtol=RTOL)
        np.testing.assert_allclose(o8[4]["power"].cpu().numpy(), o64[4]["power"].cpu().numpy(), rtol=RTOL)
        _obs_close(o8[4]["obs_raw"].cpu().numpy(), o64[4]["obs_raw"].cpu().numpy())
    # the bookkeeping is exact: origins, stream positions and step counters travel in the saved state
    st8, st64 = e8.get_state(), e64.get_state()
    assert e8.device_status() == 0 and e64.device_status() == 0
    record_property("max_screen_error_rad", worst)
    print(f"int8 vs float64 extrusion, N = {N}, v = {vel}: worst screen difference {worst:.2e} rad over {T} steps")
    # a state saved by one form resumes under the other (same layout, same counters)
    e64.set_state(st8)
    a = torch.randn((B, 16), device="cuda", generator=gen)
    o8, o64 = e8.step(a), e64.step(a)
    assert float((e8.get_screens() - e64.get_screens()).abs().max()) / LAM_WFS < 1e-7
    del st64
    e8.close()
    e64.close()


def test_int8_extrusion_episode_against_the_oracle(record_property):
    """A 30-step episode at v = 10 m/s with host-supplied normals replayed through the oracle's InfiniteAtmosphericLayer for three envs:
    observations (before the float16 cast), Strehl ratio and fiber power at 1e-5 on every step, `done` exact; the screen error against the
    oracle's float64 recursion is measured at the end of the episode and reported."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv
    from adaptive_optics_gym_amd.atmosphere_host import integer_shifts
    from oracle.ao_env_oracle import AOEnvOracle

    B, N, A, T, seed = 48, 128, 16, 30, 9
    kw = dict(atm_type="dynamic", atm_vel=10, atm_fried=0.15, act_type="num_actuators", act_dim=A, obs_dim=2, timesteps_per_episode=T)
    env = BatchedAOEnv(B, "cuda:0", num_pupil_pixels=N, seed=seed, screen_source="device", screen_oversampling=4, verbose=False, **kw)
    assert env.extrusion_kmax >= 3
    geo = device_mode_stencil_draws(seed, B, N)
    ids = [0, 23, B - 1]
    refs = {b: AOEnvOracle(num_pupil_pixels=N, screen=env.get_screens(b, 1)[0].cpu().numpy().ravel(),
                           rng=ScriptedRNG(env.wind_u[b], [g.copy() for g in geo]), verbose=False, **kw) for b in ids}
    env.reset()
    for b in ids:
        refs[b].reset()
    gen = torch.Generator("cuda").manual_seed(5)
    for t in range(T):
        a = torch.randn((B, A), device="cuda", generator=gen)
        counts = np.abs(integer_shifts(env.velocity_vectors, env.timestep * env.delta_t, (env.timestep + 1) * env.delta_t,
                                       env.params.pupil_pixel)).sum(axis=1)
        noise = torch.randn((B, max(int(counts.max()), 1), N), device="cuda", dtype=torch.float64, generator=gen)
        env.set_extrusion_noise(noise)
        obs, rew, done, _, info = env.step(a)
        for b in ids:
            refs[b].rng.normals.extend(noise[b, :int(counts[b])].cpu().numpy())
            _, _, r_done, _, r_info = refs[b].step(a[b].cpu().numpy())
            _obs_close(info["obs_raw"][b].double().cpu().numpy(), refs[b].last_obs_raw)
            np.testing.assert_allclose(float(info["strehl"][b]), refs[b].last_strehl, rtol=RTOL)
            np.testing.assert_allclose(float(info["power"][b]), r_info["power"], rtol=RTOL)
            assert bool(done[b]) == r_done
    worst = 0.0
    for b in ids:
        dev = env.get_screens(b, 1)[0].cpu().numpy()
        ref = refs[b].layer._achromatic_screen.reshape(N, N)
        worst = max(worst, float(np.abs(dev - ref).max()) / LAM_WFS)
    record_property("max_screen_error_rad", worst)
    print(f"int8 extrusion vs oracle, N = {N}, 30 steps at 10 m/s: worst screen error {worst:.2e} rad (screen rms {ref.std() / LAM_WFS:.1f} rad)")
    assert worst < 2e-6
    assert env.device_status() == 0
    env.close()


def test_extrusion_mode_switch_and_operator_coverage():
    """'f64' keeps the float64 kernels for a handle that has the operators; a wind the operators do not cover (more than 8 shifts per axis and
    step) never uploads them; both still step."""
    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv

    kw = dict(atm_type="dynamic", atm_fried=0.15, act_dim=8, act_type="zernike", obs_dim=2, timesteps_per_episode=4, num_pupil_pixels=32, seed=2,
              screen_oversampling=4, verbose=False)
    fast = BatchedAOEnv(5, "cuda:0", atm_vel=200, **kw)          # 200 m/s: 12.8 pixels per step
    assert fast.extrusion_kmax == 0
    slow = BatchedAOEnv(5, "cuda:0", atm_vel=20, **kw)
    twin = BatchedAOEnv(5, "cuda:0", atm_vel=20, **kw)
    assert slow.extrusion_kmax == 2
    twin.set_extrusion_mode("f64")
    a = torch.randn((5, 8), device="cuda")
    for env in (fast, slow, twin):
        env.reset()
    for _ in range(4):
        fast.step(a)
        slow.step(a)
        twin.step(a)
    assert float((slow.get_screens() - twin.get_screens()).abs().max()) / LAM_WFS < 1e-7
    assert not torch.equal(fast.get_screens(), slow.get_screens())
    for env in (fast, slow, twin):
        assert env.device_status() == 0
        env.close()


def test_plan_made_ahead_is_the_plan_made_in_line(monkeypatch):
    """The shifts / slots / workgroup list of step t + 1 AND its whole x phase (operands, product, staged columns) run on a side stream beside
    step t's fused kernel (x8_evolve).  Nothing may depend on that: a handle that works ahead and one that does everything in line
    (AOG_X8_NO_PLAN_AHEAD) agree bit for bit through episode resets, a restored state (the clock jumps back), a changed wind (aog_set_wind
    drops what was made for the old one) and steps whose normals the caller supplies after the x phase already drew its own."""
    import ctypes as C

    torch = _torch()
    from adaptive_optics_gym_amd import BatchedAOEnv, _lib

    B, N, T = 160, 64, 5
    kw = dict(atm_type="dynamic", atm_vel=12, atm_fried=0.15, act_type="num_actuators", act_dim=16, obs_dim=2, timesteps_per_episode=T,
              num_pupil_pixels=N, seed=9, screen_source="device", screen_oversampling=4, verbose=False)
    monkeypatch.delenv("AOG_X8_NO_PLAN_AHEAD", raising=False)
    ahead = BatchedAOEnv(B, "cuda:0", **kw)
    inline = BatchedAOEnv(B, "cuda:0", **kw)
    assert ahead.extrusion_kmax >= 1

    def both(f):
        monkeypatch.delenv("AOG_X8_NO_PLAN_AHEAD", raising=False)
        ra = f(ahead)
        monkeypatch.setenv("AOG_X8_NO_PLAN_AHEAD", "1")
        rb = f(inline)
        return ra, rb

    gen = torch.Generator("cuda").manual_seed(2)
    both(lambda e: e.reset())
    saved = None
    for t in range(3 * T + 2):
        a = torch.randn((B, 16), device="cuda", generator=gen)
        (oa, ra, da, _, ia), (ob, rb, db, _, ib) = both(lambda e: e.step(a))
        assert torch.equal(ia["obs_raw"], ib["obs_raw"]) and torch.equal(ra, rb) and torch.equal(da, db), f"step {t}"
        sa, sb = both(lambda e: e.get_screens())
        assert torch.equal(sa, sb), f"step {t}: screens differ"
        if t in (4, 5, 13):   # caller-supplied normals for the NEXT step: an x phase already run ahead on the device stream is redone with them
            noise = torch.randn((B, 24, N), device="cuda", dtype=torch.float64, generator=gen)
            both(lambda e: e.set_extrusion_noise(noise))
        if t == 2:
            saved = both(lambda e: e.get_state())
        if t == 7:   # the clock jumps back to step 3: the plan made for step 9 is not the one step 4 needs
            ahead.set_state(saved[0])
            inline.set_state(saved[1])
        if t == 11:  # new winds: the plan made ahead read the old ones
            v = np.ascontiguousarray(ahead.velocity_vectors[::-1] * 0.7)
            for e in (ahead, inline):
                e.velocity_vectors = v
                vt = torch.from_numpy(v).cuda()
                _lib.check(e.lib.aog_set_wind(e._handle, C.c_void_p(vt.data_ptr()), float(np.abs(v).max()), e._stream()))
                torch.cuda.synchronize()
        if bool(da.all()):
            both(lambda e: e.reset())
    assert ahead.device_status() == 0 and inline.device_status() == 0
