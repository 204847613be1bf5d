"""Literal CPU (numpy float64) restatement of ``AOEnv`` from the reference
(``/root/reference/gym_AO/envs/AO_env.py``), built on ``oracle.hcipy_restatement``.

TEST INFRASTRUCTURE ONLY — see the header of ``hcipy_restatement.py``.  PARITY UNPINNED
(no HCIPy in this image, no tests/fixtures in the reference).

It keeps the reference's *dataflow* — two surface GEMVs per step, materialised complex
fields at both wavelengths, the N^2 -> 128^2 and N^2 -> o^2 Fraunhofer transforms, the full
N^2 -> 240^2 science PSF of which one pixel is read, the 3-mode fiber projection — so that it
doubles as the "reference-equivalent" CPU baseline timed by ``bench.py`` (``cpu_baseline.kind
= "port"``).  The only additions are the ``num_pupil_pixels`` override (the reference hard-codes
240, AO_env.py:216) and ``screen=`` for injecting a given achromatic screen.
"""
from __future__ import annotations

import numpy as np

from . import hcipy_restatement as H


class AOEnvOracle:
    def __init__(self, atm_type="quasi_static", atm_vel=0, atm_fried=0.15, act_type="num_actuators",
                 act_dim=64, obs_dim=2, rew_type="strehl_ratio", rew_threshold=None,
                 timesteps_per_episode=20, flat_mirror_start_per_episode=True, SH_operation=False,
                 num_pupil_pixels=240, rng=np.random, screen=None, verbose=True, f_number=None):
        # AO_env.py:33-39
        self.atm_type = atm_type
        self.rew_type = rew_type
        self.act_type = act_type
        self.flat_mirror_start_per_episode = flat_mirror_start_per_episode
        self.rew_threshold = rew_threshold
        self.SH_operation = SH_operation
        self.rng = rng
        self._verbose = verbose
        self._parameters_init(act_dim, atm_vel, obs_dim, timesteps_per_episode, atm_fried, num_pupil_pixels)
        if f_number is not None:   # test knob (the reference fixes 50, AO_env.py:243): long lenslet focal lengths reach hcipy's impulse-response Fresnel branch at small pupils
            self.f_number = f_number

        # AO_env.py:293-303 pupil_simulation
        self.pupil_grid = H.make_pupil_grid(self.num_pupil_pixels, self.telescope_diameter)
        self.aperture = H.make_circular_aperture(self.telescope_diameter)(self.pupil_grid)

        # AO_env.py:306-336 incoming_wavefront
        spatial_resolution = self.wavelength_sci / self.telescope_diameter
        self.focal_grid = H.make_focal_grid(4, 30, spatial_resolution)
        self.propagator = H.FraunhoferPropagator(self.pupil_grid, self.focal_grid)
        wf = H.Wavefront(self.aperture, self.wavelength_sci, self.pupil_grid)
        wf.total_power = 1
        self.unaberrated_PSF = self.propagator.forward(wf).power
        zero_magnitude_flux = 3.9e10
        self.wf_wfs = H.Wavefront(self.aperture, self.wavelength_wfs, self.pupil_grid)
        self.wf_wfs.total_power = zero_magnitude_flux * 10 ** (-self.stellar_magnitude / 2.5)
        self.wf_wfs_fiber = H.Wavefront(self.aperture, self.wavelength_wfs, self.pupil_grid)
        self.wf_wfs_fiber.total_power = 1
        self.wf_sci = H.Wavefront(self.aperture, self.wavelength_sci, self.pupil_grid)
        self.wf_sci.total_power = zero_magnitude_flux * 10 ** (-self.stellar_magnitude / 2.5)

        # AO_env.py:339-358 DM_function
        if act_type == "zernike":
            modes = H.make_zernike_basis(self.num_modes, self.telescope_diameter, self.pupil_grid)
        else:
            modes = H.make_disk_harmonic_basis(self.pupil_grid, self.num_modes, self.telescope_diameter, "neumann")
        self.dm_modes = H.ModeBasis([m / np.ptp(m) for m in modes])
        self.deformable_mirror = H.DeformableMirror(self.dm_modes)
        self.deformable_mirror.flatten()

        # AO_env.py:361-370 atmospheric_turbulence
        Cn_squared = H.Cn_squared_from_fried_parameter(self.fried_parameter, self.wavelength_sci)
        self.layer = H.InfiniteAtmosphericLayer(self.pupil_grid, Cn_squared, self.outer_scale, self.velocity,
                                                rng=rng, initial_screen=screen)

        # AO_env.py:373-393 fiber_coupling
        D_focus_fiber = 2.1 * self.multimode_fiber_core_radius
        self.focal_grid_fiber = H.make_pupil_grid(self.num_focal_pixels_fiber, D_focus_fiber)
        self.focal_grid_fiber_subsample = H.make_pupil_grid(self.num_focal_pixels_fiber_subsample, D_focus_fiber)
        focal_length = self.D_pupil_fiber / (2 * self.fiber_NA)
        pupil_grid_fiber = H.make_pupil_grid(self.num_pupil_pixels_fiber, self.D_pupil_fiber)
        self.propagator_fiber = H.FraunhoferPropagator(pupil_grid_fiber, self.focal_grid_fiber, focal_length)
        self.propagator_fiber_subsample = H.FraunhoferPropagator(pupil_grid_fiber, self.focal_grid_fiber_subsample,
                                                                 focal_length)
        self.single_mode_fiber = H.StepIndexFiber(self.singlemode_fiber_core_radius, self.fiber_NA, self.fiber_length)

        if self.SH_operation:
            self.shack_hartmann_init()

        self.timestep = 0
        self.episode_no = 0
        self.timestep_render = 0

    # AO_env.py:197-251
    def _parameters_init(self, act_dim, velocity_value, obs_dim, timesteps_per_episode, fried_parameter, npix):
        if self.atm_type in ("quasi_static", "semi_dynamic") and velocity_value != 0:
            if self._verbose:
                print("In " + self.atm_type + " atmospheric condition, the velocity value should be zero.")
                print("therefore velocity value is changed to zero")
            velocity_value = 0
        elif self.atm_type == "dynamic" and velocity_value == 0:
            if self._verbose:
                print("In " + self.atm_type + " atmospheric condition, the velocity value cannot be zero.")
                print("therefore velocity value is changed to 1 m/s")
            velocity_value = 1
        self.telescope_diameter = 0.5
        self.num_pupil_pixels = int(npix)
        self.wavelength_wfs = 1.5e-6
        self.wavelength_sci = 2.2e-6
        self.num_modes = int(act_dim)
        self.delta_t = 1e-3
        self.max_steps = timesteps_per_episode
        self.velocity = velocity_value
        self.fried_parameter = fried_parameter
        self.outer_scale = 10
        self.D_pupil_fiber = 0.5
        self.num_pupil_pixels_fiber = 128
        self.num_focal_pixels_fiber = 128
        self.num_focal_pixels_fiber_subsample = int(obs_dim)
        self.multimode_fiber_core_radius = 25 * 1e-6
        self.singlemode_fiber_core_radius = 4.5 * 1e-6
        self.fiber_NA = 0.14
        self.fiber_length = 10
        self.f_number = 50
        self.num_lenslets = 12
        self.sh_diameter = 5e-3
        self.stellar_magnitude = -5

    # AO_env.py:396-465
    def shack_hartmann_init(self):
        magnification = self.sh_diameter / self.telescope_diameter
        self.magnifier = H.Magnifier(magnification)
        sh_grid = self.pupil_grid.scaled(magnification)
        self.shwfs = H.SquareShackHartmannWavefrontSensorOptics(sh_grid, self.f_number, self.num_lenslets, self.sh_diameter)
        # NoiselessDetector(focal_grid): the reference hands it the 240^2 science focal grid (it only works because that has as
        # many samples as the 240^2 pupil); for other pupil sizes use a centred grid of N samples with the same pitch
        if self.num_pupil_pixels == int(self.focal_grid.dims[0]):
            det_grid = self.focal_grid
        else:
            n = self.num_pupil_pixels
            d = self.focal_grid.delta
            det_grid = H.Grid(d, [n, n], d * (-n / 2 + (n % 2) * 0.5))
        self.detector_grid = det_grid
        self.camera = H.NoiselessDetector(det_grid)
        mla = self.shwfs.micro_lens_array
        self.shwfse = H.ShackHartmannWavefrontSensorEstimator(self.shwfs.mla_points, mla.mla_index, det_grid)
        wf_camera = H.Wavefront(self.aperture, self.wavelength_wfs, self.pupil_grid)
        self.camera.integrate(self.shwfs(self.magnifier(wf_camera)), 1)
        image_ref = self.camera.read_out()
        import scipy.ndimage as ndimage

        fluxes = ndimage.sum(image_ref, mla.mla_index, self.shwfse.estimation_subapertures)
        flux_limit = fluxes.max() * 0.5
        sel = np.zeros(len(self.shwfs.mla_points), dtype=bool)
        sel[self.shwfse.estimation_subapertures[fluxes > flux_limit]] = True
        self.shwfse = H.ShackHartmannWavefrontSensorEstimator(self.shwfs.mla_points, mla.mla_index, det_grid, sel)
        self.slopes_ref = self.shwfse.estimate([image_ref])
        self.deformable_mirror_shack = H.DeformableMirror(self.dm_modes)
        probe_amp = 0.01 * self.wavelength_wfs
        response = []
        wf_cal = H.Wavefront(self.aperture, self.wavelength_wfs, self.pupil_grid)
        wf_cal.total_power = 1
        for i in range(self.num_modes):
            slope = 0
            amps = [-probe_amp, probe_amp]
            for amp in amps:
                self.deformable_mirror_shack.flatten()
                act = self.deformable_mirror_shack.actuators
                act[i] = amp
                self.deformable_mirror_shack.actuators = act
                self.camera.integrate(self.shwfs(self.magnifier(self.deformable_mirror_shack.forward(wf_cal))), 1)
                slopes = self.shwfse.estimate([self.camera.read_out()])
                slope = slope + amp * slopes / np.var(amps)
            response.append(np.asarray(slope).ravel())
        self.response_matrix = np.stack(response, axis=-1)          # ModeBasis(...).transformation_matrix: [2 n_sub, A]
        self.reconstruction_matrix = H.inverse_tikhonov(self.response_matrix, rcond=1e-3)
        self.deformable_mirror_shack.flatten()

    # AO_env.py:254-290
    def SH_step(self):
        wf = self.deformable_mirror_shack(self.layer(self.wf_wfs))
        wf_on_sh = self.shwfs(self.magnifier(wf))
        self.camera.integrate(wf_on_sh, self.delta_t)
        wfs_image = self.camera.read_out()
        self.last_sh_image_noiseless = wfs_image.copy()
        wfs_image = H.large_poisson(wfs_image, rng=self.rng).astype("float")
        self.last_sh_noisy = wfs_image.copy()
        slopes = self.shwfse.estimate([wfs_image + 1e-10])
        slopes = slopes - self.slopes_ref
        slopes = slopes.ravel()
        self.last_sh_slopes = slopes
        gain, leakage = 0.3, 0.01
        self.deformable_mirror_shack.actuators = ((1 - leakage) * self.deformable_mirror_shack.actuators
                                                  - gain * self.reconstruction_matrix.dot(slopes))
        return self.deformable_mirror_shack.actuators, np.array([1])

    # AO_env.py:74-103
    def reset(self, seed=None, options=None):
        if self.atm_type == "semi_dynamic":
            self.layer.reset()
        if self.flat_mirror_start_per_episode:
            self.deformable_mirror.flatten()
        self.timestep_render = 0
        self.layer.t = self.timestep * self.delta_t
        wf_after_atmos = self.layer(self.wf_wfs_fiber)
        wf_after_dm = self.deformable_mirror(wf_after_atmos)
        self.wf_wfs_after_foc = self.propagator_fiber(wf_after_dm)
        self.wf_wfs_after_foc_subsample = self.propagator_fiber_subsample(wf_after_dm)
        state = self.wf_wfs_after_foc_subsample.power
        self.last_obs_raw = np.array(state, dtype=np.float64)
        return np.array(state, dtype=np.float16), {}

    # AO_env.py:106-153
    def step(self, action):
        trunc = False
        if self.SH_operation:
            self.deformable_mirror.actuators = action
        else:
            self.deformable_mirror.actuators = action / (np.arange(self.num_modes) + 10)
            self.deformable_mirror.actuators *= 0.1 * self.wavelength_sci / (np.std(self.deformable_mirror.surface))
        self.timestep += 1
        self.timestep_render += 1
        self.layer.t = self.timestep * self.delta_t
        wf_after_atmos = self.layer(self.wf_wfs_fiber)
        wf_after_dm = self.deformable_mirror(wf_after_atmos)
        self.wf_wfs_after_foc = self.propagator_fiber(wf_after_dm)
        self.wf_wfs_after_foc_subsample = self.propagator_fiber_subsample(wf_after_dm)
        next_state = self.wf_wfs_after_foc_subsample.power
        self.last_obs_raw = np.array(next_state, dtype=np.float64)
        reward, rew_fiber = self.reward_function()
        if self.timestep_render == self.max_steps:
            done = True
            self.episode_no += 1
        else:
            done = False
        return np.array(next_state, dtype=np.float16), reward, done, trunc, {"power": float(rew_fiber)}

    # AO_env.py:468-503
    def reward_function(self):
        wf_smf = self.single_mode_fiber.forward(self.wf_wfs_after_foc)
        rew_fiber = wf_smf.total_power
        if self.rew_type == "strehl_ratio":
            self.wf_sci_focal_plane = self.propagator(self.deformable_mirror(self.layer(self.wf_sci)))
            strehl_ratio = H.get_strehl_from_focal(self.wf_sci_focal_plane.power,
                                                   self.unaberrated_PSF * self.wf_wfs.total_power) * 100
            self.last_strehl = strehl_ratio / 100
            reward = -(100 - strehl_ratio)
        elif self.rew_type == "smf_ssim":
            focal_power = self.wf_wfs_after_foc_subsample.power
            ref_power = np.zeros(self.num_focal_pixels_fiber_subsample ** 2)
            ref_power[int(self.num_focal_pixels_fiber_subsample ** 2 / 2)] = 2.8
            data_range = ref_power.max() - ref_power.min()
            ssim_score = H.structural_similarity_1d(focal_power, ref_power, data_range)
            alpha = 0.8
            reward = alpha * rew_fiber + (1 - alpha) * ssim_score
        else:
            raise ValueError("rew_type must be 'strehl_ratio' or 'smf_ssim' (reference leaves reward undefined)")
        if self.rew_threshold is not None and reward < self.rew_threshold:
            reward = -1.0
        return reward, rew_fiber
