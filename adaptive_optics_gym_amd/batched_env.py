"""``BatchedAOEnv`` — B independent adaptive-optics environments stepped in lock-step on one MI355X.

Host-side mirror of the reference's ``AOEnv`` (``/root/reference/gym_AO/envs/AO_env.py``): same constructor
keywords, same ``reset``/``step`` semantics, but tensors carry a leading env dimension and live on the GPU.
All arithmetic of ``reset``/``step`` runs in ``libaogym.so`` (hand-written HIP, called through the C-ABI of
``include/aogym.h``); PyTorch only owns the buffers and the stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .atmosphere_host import (build_layer_tables, cn_squared_from_fried_parameter, integer_shifts, screen_numpy,
                              screens_torch)
from .optics_host import HostTables, build_tables
from .params import OpticalParams, coerce_velocity
from .spaces import make_box


def _dptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class BatchedAOEnv:
    """Constructor keywords = AOEnv's (AO_env.py:17-29) plus:

    num_envs            B
    device              torch device (default ``cuda:0``)
    num_pupil_pixels    pupil grid side N (reference: 240)
    seed                seed of the screen generator (global env g uses seed + g for ``screen_source='numpy'``)
    screen_source       'device' (default: hipFFT synthesis inside libaogym, Philox normals) | 'torch' (same algorithm through
                        torch.fft) | 'numpy' (hcipy's draw order on a numpy legacy stream, float64, host)
    screen_method       'twoband' (default: device synthesis splits the von Karman spectrum into a low band on hcipy's (16 N)^2 grid and a
                        high band on the (2 N)^2 grid — the same stationary Gaussian field on the pupil to < 1e-4 of its variance at every
                        lag, 60x fewer spectrum samples) | 'hcipy16' (the literal (16 N)^2 draw); only ``screen_source='device'`` reads it
    screens             optional [B, N, N] achromatic screens to use instead of generating them
    global_env_offset   global id of env 0 of this instance (multi-GPU: ``sharding.shard_range(total, rank, world)[0]``)
    total_envs          size of the global batch this instance is a contiguous slice of (default: offset + num_envs).
                        Every per-env random stream (wind direction, device screen synthesis, extrusion normals, photon noise,
                        numpy per-env seeds) is keyed by the GLOBAL env id, so a batch split over several instances / GPUs
                        gives bit-identical screens to one instance holding all of it
    sh_fft_precision    'single' (default: complex64 Fresnel transforms in the Shack-Hartmann chain; error ~1e-6 of the image peak, far
                        below the photon noise added before the image is read) | 'double' (complex128, for comparing the noise-free
                        sensor image with a float64 oracle)
    precision           'fast' (fp32 data, float64 accumulators) | 'fp64' (validation kernel)
    extrusion           'auto' (dynamic atmosphere: the int8 matrix-core composite form, new samples good to ~1e-9 rad) | 'f64' (float64 round
                        kernels only: the validation form, bit-comparable with the oracle's recursion to 1e-12)
    kernel              'auto' | 'mfma' | 'valu'
    """

    def __init__(self, num_envs=1, device=None, atm_type="quasi_static", atm_vel=0, atm_fried=0.15,
                 act_type="num_actuators", act_dim=64, obs_dim=2, rew_type="strehl_ratio", rew_threshold=None,
                 timesteps_per_episode=20, flat_mirror_start_per_episode=True, SH_operation=False, *,
                 num_pupil_pixels=240, seed=None, screen_source="device", screen_oversampling=16, screens=None,
                 precision="fast", kernel="auto", pixel_chunks=0, rng=None, verbose=True, params=None,
                 global_env_offset=0, total_envs=None, sh_fft_precision="single", screen_method="twoband", tables=None, extrusion="auto"):
        import torch

        self._torch = torch
        self.lib = _lib.load()  # raises if the HIP extension is missing — no CPU fallback
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedAOEnv needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError("BatchedAOEnv runs on a HIP device only")
        if rew_type not in _lib.AOG_REWARD:
            # the reference leaves `reward` undefined in this case (AO_env.py:476,487); raise something clear instead
            raise ValueError("rew_type must be 'strehl_ratio' or 'smf_ssim'")
        if atm_type not in ("quasi_static", "semi_dynamic", "dynamic"):
            raise ValueError("atm_type must be 'quasi_static', 'semi_dynamic' or 'dynamic'")

        self.num_envs = int(num_envs)
        self.global_env_offset = int(global_env_offset)
        self.total_envs = int(total_envs) if total_envs is not None else self.global_env_offset + self.num_envs
        if self.global_env_offset < 0 or self.global_env_offset + self.num_envs > self.total_envs:
            raise ValueError("global_env_offset / total_envs: this instance's envs must lie inside range(total_envs)")
        self.atm_type = atm_type
        self.rew_type = rew_type
        self.act_type = act_type
        self.flat_mirror_start_per_episode = bool(flat_mirror_start_per_episode)
        self.rew_threshold = rew_threshold
        self.SH_operation = bool(SH_operation)
        if sh_fft_precision not in ("single", "double"):
            raise ValueError("sh_fft_precision must be 'single' or 'double'")
        self.sh_fft_precision = sh_fft_precision
        self.velocity = coerce_velocity(atm_type, atm_vel, verbose)
        self.fried_parameter = atm_fried
        self.params = params if params is not None else OpticalParams(num_pupil_pixels=int(num_pupil_pixels))
        self.num_pupil_pixels = self.params.num_pupil_pixels
        self.num_modes = int(act_dim)
        self.obs_dim = int(obs_dim)
        self.num_focal_pixels_fiber_subsample = int(obs_dim)
        self.max_steps = timesteps_per_episode
        self.delta_t = self.params.delta_t
        self.wavelength_wfs = self.params.wavelength_wfs
        self.wavelength_sci = self.params.wavelength_sci
        self.timestep = 0
        self.episode_no = 0
        self.seed = seed
        self.screen_source = screen_source
        self.screen_oversampling = int(screen_oversampling)
        if screen_method not in _lib.AOG_SCREENS:
            raise ValueError("screen_method must be 'twoband' or 'hcipy16'")
        self.screen_method = screen_method
        if extrusion not in _lib.AOG_EXTRUDE:
            raise ValueError("extrusion must be 'auto' or 'f64'")
        self._extrusion = extrusion
        self._rng = rng
        self._episode_returns = None
        self._trunc = None
        self._persistent_out = False   # see persistent_outputs()
        self._step_cache = None        # (views of the persistent block + their addresses)
        self._pack = None


        self.observation_space = make_box(-1, 1, (self.obs_dim ** 2,), np.float16)  # AO_env.py:45
        self.action_space = make_box(-1, 1, (self.num_modes,), np.float16)          # AO_env.py:46

        self.Cn_squared = cn_squared_from_fried_parameter(self.fried_parameter, self.params.wavelength_sci)
        # (tables=: the HostTables of another instance with the same params / act_type / act_dim / obs_dim, to skip the host precompute)
        self.tables: HostTables = tables if tables is not None else build_tables(self.params, act_type, self.num_modes, self.obs_dim)
        t = self.tables
        cfg = _lib.AogConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.num_envs = self.num_envs
        cfg.n_pupil = self.num_pupil_pixels
        cfg.n_modes = self.num_modes
        cfg.obs_dim = self.obs_dim
        cfg.n_ap = t.n_ap
        cfg.n_wfs_tables = t.wfs_tables.shape[0]
        cfg.n_sci_tables = t.sci_tables.shape[0]
        cfg.n_fiber_modes = t.n_fiber_modes
        cfg.reward_type = _lib.AOG_REWARD[rew_type]
        cfg.sh_operation = int(self.SH_operation)
        cfg.max_steps = int(timesteps_per_episode)
        cfg.flat_mirror_start = int(self.flat_mirror_start_per_episode)
        cfg.has_rew_threshold = int(rew_threshold is not None)
        cfg.precision = _lib.AOG_PRECISION[precision]
        cfg.kernel = _lib.AOG_KERNEL[kernel]
        cfg.pixel_chunks = int(pixel_chunks)
        cfg.atm_dynamic = int(atm_type == "dynamic")
        cfg.env_id_base = self.global_env_offset
        cfg.wavelength_wfs = self.params.wavelength_wfs
        cfg.wavelength_sci = self.params.wavelength_sci
        cfg.surface_rms_target = self.params.action_rms_fraction * self.params.wavelength_sci
        cfg.rew_threshold = float(rew_threshold) if rew_threshold is not None else 0.0
        cfg.ssim_ref_peak = self.params.ssim_ref_peak
        cfg.ssim_alpha = self.params.ssim_alpha
        self._handle = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.aog_create(C.byref(cfg), dev_index, C.byref(self._handle)))

        keep = dict(
            ap=np.ascontiguousarray(t.ap_index, dtype=np.int32),
            modes=np.ascontiguousarray(t.modes, dtype=np.float64),
            gram=np.ascontiguousarray(t.gram, dtype=np.float64),
            wt=np.ascontiguousarray(t.wfs_tables, dtype=np.float64),
            st=np.ascontiguousarray(t.sci_tables, dtype=np.float64),
            wc=np.ascontiguousarray(np.stack([t.wfs_coef.real, t.wfs_coef.imag], axis=-1), dtype=np.float64),
            sc=np.ascontiguousarray(np.stack([t.sci_coef.real, t.sci_coef.imag], axis=-1), dtype=np.float64),
            m1=np.ascontiguousarray(np.stack([t.focal_m1.real, t.focal_m1.imag], axis=-1), dtype=np.float64),
            m2=np.ascontiguousarray(np.stack([t.focal_m2.real, t.focal_m2.imag], axis=-1), dtype=np.float64),
        )
        tabs = _lib.AogTables(_dptr(keep["ap"], C.c_int32), _dptr(keep["modes"], C.c_double), _dptr(keep["gram"], C.c_double),
                              _dptr(keep["wt"], C.c_double), _dptr(keep["st"], C.c_double), _dptr(keep["wc"], C.c_double),
                              _dptr(keep["sc"], C.c_double), _dptr(keep["m1"], C.c_double), _dptr(keep["m2"], C.c_double),
                              int(t.focal_m1.shape[0]))
        _lib.check(self.lib.aog_upload_tables(self._handle, C.byref(tabs)))
        _lib.check(self.lib.aog_set_screen_method(self._handle, _lib.AOG_SCREENS[screen_method]))
        self.info = _lib.AogInfo()
        _lib.check(self.lib.aog_get_info(self._handle, C.byref(self.info)))

        # atmosphere (AO_env.py:361-370)
        # hcipy's construction order: wind direction (rand), the two stencil draws (geometric x2), then
        # the screen normals.  Host-RNG mode consumes the numpy stream in that order; env 0's draws define the stencils /
        # AR matrices shared by the whole batch (for B = 1 this is exactly the reference's layer).
        self._host_rng = self._rng is not None or screen_source == "numpy"
        N = self.num_pupil_pixels
        wind_u = np.zeros(self.num_envs)   # hcipy draws theta = rand() * 2 pi per layer
        layer = None
        # GLOBAL env 0's draws define the stencils whatever slice of the batch this instance holds.
        if self._host_rng:
            if self.atm_type == "dynamic" and self.global_env_offset > 0 and self._rng is None and self.seed is not None:
                r0 = np.random.RandomState(self.seed)      # global env 0's stream, replayed: rand() then the two stencil draws
                r0.rand()
                layer = build_layer_tables(N, self.params.pupil_pixel, self.params.outer_scale, r0)
            for e in range(self.num_envs):
                r = self._env_rng(e)
                wind_u[e] = r.rand()
                if e == 0 and self.atm_type == "dynamic" and layer is None:
                    layer = build_layer_tables(N, self.params.pupil_pixel, self.params.outer_scale, r)
                else:
                    r.geometric(0.5, N)
                    r.geometric(0.5, N)
        else:
            base_seed = 1234 if seed is None else int(seed)
            trng = np.random.RandomState(base_seed)
            wind_u = trng.rand(self.total_envs)[self.global_env_offset:self.global_env_offset + self.num_envs]
            if self.atm_type == "dynamic":
                # the stencil draws have a stream of their own: drawn after the wind directions they would depend on total_envs,
                # and instances holding different slices of one batch must share the AR matrices whatever total_envs they were given
                layer = build_layer_tables(N, self.params.pupil_pixel, self.params.outer_scale, np.random.RandomState([base_seed & 0xFFFFFFFF, 0x57E9C11]))
        self.wind_u = np.array(wind_u, dtype=np.float64)
        theta = self.wind_u * 2 * np.pi
        self.velocity_vectors = float(self.velocity) * np.stack([np.cos(theta), np.sin(theta)], axis=1)  # [B, 2] m/s
        if self.atm_type == "dynamic":
            self._upload_layer(layer)
        if screens is not None:
            self.set_screens(screens)
        else:
            self._generate_screens(first_call=True)
        if self.SH_operation:
            self._upload_shack_hartmann()

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _env_rng(self, e):
        if self._rng is not None:
            return self._rng
        if self.seed is None:
            return np.random
        if not hasattr(self, "_rngs"):
            self._rngs = {}
        if e not in self._rngs:
            self._rngs[e] = np.random.RandomState(self.seed + self.global_env_offset + e)   # seed + GLOBAL env id
        return self._rngs[e]

    def _generate_screens(self, first_call=False, mask=None):
        torch = self._torch
        p = self.params
        if self._rng is not None or self.screen_source == "numpy":
            for e in range(self.num_envs):
                if mask is not None and not bool(mask[e]):
                    continue
                psi = screen_numpy(p.num_pupil_pixels, p.pupil_pixel, self.Cn_squared, p.outer_scale, self._env_rng(e),
                                   self.screen_oversampling)
                self.set_screens(psi[None], first=e)
        elif self.screen_source == "device":
            _lib.check(self.lib.aog_set_rng_seed(self._handle, C.c_uint64(1234 if self.seed is None else int(self.seed))))
            args = (int(self.screen_oversampling), float(self.Cn_squared), float(p.outer_scale), float(p.pupil_pixel), self._stream())
            if mask is None:
                _lib.check(self.lib.aog_generate_screens(self._handle, 0, self.num_envs, *args))
            else:
                idx = np.flatnonzero(np.asarray(mask.cpu() if hasattr(mask, "cpu") else mask))
                runs = np.split(idx, np.flatnonzero(np.diff(idx) != 1) + 1) if idx.size else []
                for r in runs:
                    _lib.check(self.lib.aog_generate_screens(self._handle, int(r[0]), int(r.size), *args))
        else:
            if not hasattr(self, "_gen"):
                self._gen = torch.Generator(device=self.device)
                # (one torch stream per instance, offset by the instance's first global env id: distinct atmospheres on every rank,
                # but — unlike 'device' and 'numpy' — not invariant to how the batch is split)
                self._gen.manual_seed((1234 if self.seed is None else int(self.seed)) + self.global_env_offset)
            psi = screens_torch(self.num_envs, p.num_pupil_pixels, p.pupil_pixel, self.Cn_squared, p.outer_scale,
                                self.device, self._gen, self.screen_oversampling)
            if mask is None:
                self.set_screens(psi)
            else:
                for e in np.flatnonzero(np.asarray(mask.cpu() if hasattr(mask, "cpu") else mask)):
                    self.set_screens(psi[e:e + 1], first=int(e))

    def _upload_layer(self, layer):
        torch = self._torch
        self._layer = layer  # keep the host arrays alive during the call
        lt = _lib.AogLayerTables(
            int(layer["stencil_vertical"].size), int(layer["stencil_horizontal"].size),
            _dptr(layer["stencil_vertical"], C.c_int32), _dptr(layer["stencil_horizontal"], C.c_int32),
            _dptr(layer["A_vertical"], C.c_double), _dptr(layer["B_vertical"], C.c_double),
            _dptr(layer["A_horizontal"], C.c_double), _dptr(layer["B_horizontal"], C.c_double),
            float(np.sqrt(self.Cn_squared)), float(self.params.pupil_pixel), float(self.params.delta_t))
        _lib.check(self.lib.aog_upload_layer(self._handle, C.byref(lt)))
        v = torch.from_numpy(np.ascontiguousarray(self.velocity_vectors)).to(self.device)
        _lib.check(self.lib.aog_set_wind(self._handle, C.c_void_p(v.data_ptr()), float(np.abs(self.velocity_vectors).max()), self._stream()))
        _lib.check(self.lib.aog_set_rng_seed(self._handle, C.c_uint64(1234 if self.seed is None else int(self.seed))))
        torch.cuda.current_stream(self.device).synchronize()
        self._upload_composite(layer)

    def _upload_composite(self, layer):
        """The step's shifts along each axis as ONE operator (``extrusion_host.compose_extrusions``) for the int8 matrix-core extrusion
        (``aog_upload_layer_composite``).  k_max = the largest whole-pixel shift any env makes per step; winds that would need more than 8
        shifts per axis and step (or ``extrusion='f64'``) keep the float64 round kernels."""
        from .extrusion_host import compose_extrusions

        self.extrusion_kmax = 0
        vmax = float(np.abs(self.velocity_vectors).max()) if self.velocity_vectors.size else 0.0
        k_need = int(np.floor(vmax * self.params.delta_t / self.params.pupil_pixel)) + 1
        if self._extrusion == "f64" or k_need > 8 or os.environ.get("AOG_EXTRUDE_F64"):
            return
        N = self.num_pupil_pixels
        keep = []
        self.extrusion_union = {}     # axis -> [U_1 .. U_kmax]: union stencil sizes of the operators the library cuts out of the uploaded one
        for axis, (st, A, B) in enumerate(((layer["stencil_vertical"], layer["A_vertical"], layer["B_vertical"]),
                                            (layer["stencil_horizontal"], layer["A_horizontal"], layer["B_horizontal"]))):
            yx, Ak, Bk = compose_extrusions(st, A, B, N, k_need, vertical=axis == 0)
            yx, Ak, Bk = np.ascontiguousarray(yx, dtype=np.int32), np.ascontiguousarray(Ak), np.ascontiguousarray(Bk)
            keep.append((yx, Ak, Bk))
            first_use = np.full(yx.size, k_need + 1)
            for j in range(k_need, 0, -1):
                first_use[np.abs(Ak[(j - 1) * N:j * N]).sum(axis=0) > 0] = j
            self.extrusion_union[axis] = [int((first_use <= k).sum()) for k in range(1, k_need + 1)]
            op = _lib.AogLayerComposite(axis, k_need, int(yx.size), 0, _dptr(yx, C.c_int32), _dptr(Ak, C.c_double), _dptr(Bk, C.c_double))
            _lib.check(self.lib.aog_upload_layer_composite(self._handle, C.byref(op)))
        self.extrusion_kmax = k_need

    def set_extrusion_mode(self, mode):
        """'auto' (default: the int8 composite form when its operators were uploaded) | 'f64' (the float64 round kernels: validation form)."""
        _lib.check(self.lib.aog_set_extrusion_mode(self._handle, _lib.AOG_EXTRUDE[mode]))

    def _upload_shack_hartmann(self):
        """shack_hartmann_init (AO_env.py:396-465): host calibration, then the tables of the device chain."""
        from .sh_host import ShackHartmannHost

        sh = ShackHartmannHost(self.params, self.tables)
        self.sh = sh
        c = np.ascontiguousarray
        keep = dict(slot=c(sh.sub_slot, dtype=np.int32), cen=c(sh.centres, dtype=np.float64), ref=c(sh.slopes_ref, dtype=np.float64),
                    rec=c(sh.reconstruction, dtype=np.float64),
                    mla=c(np.stack([sh.mla_phase.real, sh.mla_phase.imag], axis=-1), dtype=np.float64),
                    tf=c(np.stack([sh.transfer.real, sh.transfer.imag], axis=-1), dtype=np.float64), xd=c(sh.x_det, dtype=np.float64))
        t = _lib.AogShTables(int(sh.n_sub), _dptr(keep["slot"], C.c_int32), _dptr(keep["cen"], C.c_double), _dptr(keep["ref"], C.c_double),
                             _dptr(keep["rec"], C.c_double), _dptr(keep["mla"], C.c_double), _dptr(keep["tf"], C.c_double),
                             _dptr(keep["xd"], C.c_double), float(sh.amp_wfs / sh.mag), float(sh.pitch ** 2 * self.params.delta_t), 0.3, 0.01,
                             int(self.sh_fft_precision == "double"), 0)
        _lib.check(self.lib.aog_upload_sh(self._handle, C.byref(t)))

    def SH_step(self):
        """AOEnv.SH_step (AO_env.py:254-290) for every env: returns (actions [B, A] float64 = deformable_mirror_shack.actuators,
        torch.tensor([1])).  Photon noise comes from the handle's Philox stream, or — in host-RNG (parity) mode — from each env's
        numpy stream through hcipy's large_poisson draw order."""
        torch = self._torch
        if not self.SH_operation:
            raise RuntimeError("SH_step needs SH_operation=True (AO_env.py:67-68 only initialises the sensor then)")
        B, N = self.num_envs, self.num_pupil_pixels
        action = torch.empty((B, self.num_modes), dtype=torch.float64, device=self.device)
        if self._host_rng:
            img = torch.empty((B, N * N), dtype=torch.float64, device=self.device)
            _lib.check(self.lib.aog_sh_image(self._handle, C.c_void_p(img.data_ptr()), self._stream()))
            lam = img.cpu().numpy()
            noisy = np.empty_like(lam)
            for e in range(B):
                r = self._env_rng(e)
                large = lam[e] > 1e6
                out = np.zeros(N * N)
                out[large] = np.round(lam[e][large] + r.normal(size=int(large.sum())) * np.sqrt(lam[e][large]))
                out[~large] = r.poisson(lam[e][~large], size=int((~large).sum()))
                noisy[e] = out
            nd = torch.from_numpy(noisy).to(self.device)
            _lib.check(self.lib.aog_sh_update(self._handle, C.c_void_p(nd.data_ptr()), C.c_void_p(action.data_ptr()), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        else:
            _lib.check(self.lib.aog_sh_image(self._handle, None, self._stream()))
            _lib.check(self.lib.aog_sh_update(self._handle, None, C.c_void_p(action.data_ptr()), self._stream()))
        return action, torch.tensor([1])

    def sh_image(self):
        """Noise-free Shack-Hartmann camera image of every env, [B, N*N] float64 (camera.read_out(), AO_env.py:274)."""
        torch = self._torch
        img = torch.empty((self.num_envs, self.num_pupil_pixels ** 2), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.aog_sh_image(self._handle, C.c_void_p(img.data_ptr()), self._stream()))
        return img

    def sh_update(self, noisy_image):
        """Estimator + leaky integrator on a caller-supplied (already noisy) image [B, N*N] float64 -> actions [B, A] float64.
        ``None``: photon noise from the handle's Philox stream on the image of the last ``sh_image()`` call (what ``SH_step`` does)."""
        torch = self._torch
        action = torch.empty((self.num_envs, self.num_modes), dtype=torch.float64, device=self.device)
        if noisy_image is None:
            nd_ptr = None
        else:
            nd = torch.as_tensor(noisy_image, device=self.device).to(torch.float64).reshape(self.num_envs, -1).contiguous()
            nd_ptr = C.c_void_p(nd.data_ptr())
        _lib.check(self.lib.aog_sh_update(self._handle, nd_ptr, C.c_void_p(action.data_ptr()), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        return action

    def _host_extrusion_noise(self):
        """Host-RNG (parity) mode: draw the normals of the coming step's extrusions from each env's numpy stream in hcipy's
        order (all x shifts, then all y shifts; ``normal(0, 1, N)`` per extrusion) and hand them to the library."""
        torch = self._torch
        N = self.num_pupil_pixels
        t_prev, t_new = self.timestep * self.delta_t, (self.timestep + 1) * self.delta_t
        shifts = integer_shifts(self.velocity_vectors, t_prev, t_new, self.params.pupil_pixel)  # [B, 2]
        counts = np.abs(shifts).sum(axis=1)
        max_ext = int(counts.max()) if counts.size else 0
        if max_ext == 0:
            return
        noise = np.zeros((self.num_envs, max_ext, N))
        for e in range(self.num_envs):
            r = self._env_rng(e)
            for k in range(int(counts[e])):
                noise[e, k] = r.normal(0, 1, size=N)
        self._noise_dev = torch.from_numpy(noise).to(self.device)  # kept alive until the step has run
        _lib.check(self.lib.aog_set_extrusion_noise(self._handle, C.c_void_p(self._noise_dev.data_ptr()), max_ext, self._stream()))

    def set_extrusion_noise(self, noise):
        """Standard normals for the extrusions of the NEXT ``step`` (dynamic atmosphere, parity runs): ``[B, max_ext, N]`` float64
        device tensor, row k of env b = the ``normal(0, 1, N)`` draw of its k-th extrusion in hcipy's order (x shifts first, then
        y).  Without this call the step draws from the handle's Philox stream."""
        torch = self._torch
        n = torch.as_tensor(noise, device=self.device).to(torch.float64).contiguous()
        if n.dim() != 3 or n.shape[0] != self.num_envs or n.shape[2] != self.num_pupil_pixels:
            raise ValueError("set_extrusion_noise: expected [num_envs, max_ext, N]")
        self._noise_dev = n   # kept alive until the step has run
        _lib.check(self.lib.aog_set_extrusion_noise(self._handle, C.c_void_p(n.data_ptr()), int(n.shape[1]), self._stream()))

    def lookahead(self, enable=True):
        """Dynamic atmosphere, device random stream: let every ``step`` launch the NEXT step's wind extrusion on a stream of the library's
        own, beside its epilogue and the caller's policy query (``aog_set_lookahead``).  Same results bit for bit; between two steps of an
        episode ``reset`` / ``get_screens`` / ``get_state`` / ``phase_screen`` / ``focal_image(s)`` / ``sh_image`` raise (the screens already
        stand at the next step) — at episode boundaries they work as ever.  Opt-in (``rollout(lookahead=True)``, ``bench.py --config 4
        --lookahead``): measured on ROCm 7.2 the two cross-stream event hand-offs cost ~20 us each, which is what the overlap saves."""
        if self.atm_type != "dynamic" or self._host_rng:
            return False
        _lib.check(self.lib.aog_set_lookahead(self._handle, int(bool(enable))))
        return bool(enable)

    def set_screen_method(self, screen_method):
        """Switch the device screen synthesis between 'twoband' and 'hcipy16' (``aog_set_screen_method``); takes effect at the next
        regeneration (semi_dynamic ``reset``).  The workspace of the method that is no longer used is given back then."""
        if screen_method not in _lib.AOG_SCREENS:
            raise ValueError("screen_method must be 'twoband' or 'hcipy16'")
        _lib.check(self.lib.aog_set_screen_method(self._handle, _lib.AOG_SCREENS[screen_method]))
        self.screen_method = screen_method

    def device_bytes(self):
        """Bytes of HBM the library handle owns right now (``aog_info.device_bytes``)."""
        _lib.check(self.lib.aog_get_info(self._handle, C.byref(self.info)))
        return int(self.info.device_bytes)

    def get_screens(self, first=0, count=None):
        """Current achromatic screens (hcipy's ``layer._achromatic_screen``: phase * lambda) of envs [first, first + count), default
        all: [count, N, N] float64.  Dynamic atmosphere: the float64 master screens.  Otherwise the stored screen exactly as the step
        kernels read it (aperture pixels only, aperture mean removed)."""
        torch = self._torch
        N = self.num_pupil_pixels
        count = self.num_envs - first if count is None else int(count)
        out = torch.empty((count, N, N), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.aog_get_screens_f64(self._handle, C.c_void_p(out.data_ptr()), int(first), count, self._stream()))
        return out

    def set_screens(self, screens, first=0):
        """Install achromatic screens (hcipy's ``layer._achromatic_screen``: phase * lambda) for envs
        [first, first + len(screens)).  numpy or torch, float32 or float64, shape [k, N, N] or [k, N*N]."""
        torch = self._torch
        if isinstance(screens, np.ndarray):
            screens = torch.from_numpy(np.ascontiguousarray(screens))
        N = self.num_pupil_pixels
        screens = screens.reshape(-1, N, N)
        if screens.dtype not in (torch.float32, torch.float64):
            screens = screens.to(torch.float64)
        screens = screens.to(self.device).contiguous()
        fn = self.lib.aog_set_screens_f64 if screens.dtype == torch.float64 else self.lib.aog_set_screens_f32
        _lib.check(fn(self._handle, C.c_void_p(screens.data_ptr()), int(first), int(screens.shape[0]), self._stream()))
        # the kernel is stream-ordered; keep the source alive until it has run
        torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------------------------------------
    def reset(self, mask=None, seed=None, options=None):
        """AOEnv.reset (AO_env.py:74-103) for every env (or those selected by ``mask``).  ``seed``/``options`` are
        accepted and ignored exactly like the reference.  Returns (obs [B, o^2] float16, {})."""
        torch = self._torch
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        if self.atm_type == "semi_dynamic":
            self._generate_screens(mask=m)  # layer.reset() (AO_env.py:76-77)
        n = self.obs_dim ** 2
        obs = torch.empty((self.num_envs, n), dtype=torch.float16, device=self.device)
        obs_raw = torch.empty((self.num_envs, n), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aog_reset(self._handle, C.c_void_p(m.data_ptr()) if m is not None else None,
                                      C.c_void_p(obs_raw.data_ptr()), C.c_void_p(obs.data_ptr()), self._stream()))
        self.last_obs_raw = obs_raw
        return obs, {}

    _PIPELINE_END = object()

    def step(self, actions, out=None, next_actions=_PIPELINE_END):
        """AOEnv.step (AO_env.py:106-153).  ``actions``: [B, A] float32 tensor on the device.
        ``next_actions`` (optional; callers that know the next action already — open-loop sequences, replays, synthetic benchmarks):
        a [B, A] float32 device tensor = the actions of the NEXT call, or None on the last step of such a sequence.  The step then goes
        through ``aog_step_pipelined``: identical results, one kernel launch less per step; between two calls of a sequence the mirror
        already holds the next action, so ``reset`` / ``get_state`` / ``focal_images`` ... raise until the sequence is ended with
        ``next_actions=None``.
        Returns (obs float16 [B, o^2], reward float32 [B], done bool [B], trunc bool [B] (all False),
        {"power": [B] float32, "obs_raw": [B, o^2] float32, "strehl": [B] float32}).
        ``out`` (optional): (obs float16 [B, o^2], reward float32 [B], done bool/uint8 [B]) contiguous device tensors to write
        into — a rollout hands in slices of its transition buffers, so nothing is copied afterwards."""
        torch = self._torch
        a = self._as_actions(actions)
        if self.atm_type == "dynamic" and self._host_rng:
            self._host_extrusion_noise()
        n = self.obs_dim ** 2
        B = self.num_envs
        if self._persistent_out and out is None and self._step_cache is not None:
            # persistent outputs: the views of the block and their addresses are made once (a step's host cost drops from ~22 to ~10 us, which
            # matters right after a synchronisation, when the first launches of a burst cost the host twice their steady-state time)
            ret, ptrs = self._step_cache
            self._launch_step(a, next_actions, ptrs)
            self.timestep += 1
            self.last_obs_raw = ret[4]["obs_raw"]
            return ret
        # ONE allocation per step: fp32 block (obs_raw | reward | power | strehl), fp16 obs, uint8 done — or none at all when the
        # caller asked for a persistent block (``persistent_outputs``: the single-env wrapper copies it to the host in one transfer)
        nb32, nb16 = 4 * B * (n + 3), 2 * B * n
        if self._persistent_out:
            if self._pack is None:
                self._pack = torch.empty((nb32 + nb16 + B,), dtype=torch.uint8, device=self.device)
            pack = self._pack
        else:
            pack = torch.empty((nb32 + nb16 + B,), dtype=torch.uint8, device=self.device)
        f32 = pack[:nb32].view(torch.float32)
        obs_raw = f32[: B * n].view(B, n)
        power = f32[B * (n + 1): B * (n + 2)]
        strehl = f32[B * (n + 2):]
        if out is None:
            reward = f32[B * n: B * (n + 1)]
            obs = pack[nb32:nb32 + nb16].view(torch.float16).view(B, n)
            done = pack[nb32 + nb16:]
        else:
            obs, reward, done = out
            ok = (obs.dtype == torch.float16 and tuple(obs.shape) == (B, n) and reward.dtype == torch.float32 and tuple(reward.shape) == (B,)
                  and done.dtype in (torch.bool, torch.uint8) and tuple(done.shape) == (B,)
                  and obs.is_contiguous() and reward.is_contiguous() and done.is_contiguous())
            if not ok:
                raise ValueError("step(out=...): expected contiguous (float16 [B, o^2], float32 [B], bool/uint8 [B]) device tensors")
        base = f32.data_ptr()
        ptrs = (base, obs.data_ptr(), reward.data_ptr(), done.data_ptr(), base + 4 * B * (n + 1), base + 4 * B * (n + 2))
        self._launch_step(a, next_actions, ptrs)
        self.timestep += 1
        self.last_obs_raw = obs_raw
        if self._trunc is None:
            self._trunc = torch.zeros((B,), dtype=torch.bool, device=self.device)
        ret = (obs, reward, done if done.dtype == torch.bool else done.view(torch.bool), self._trunc, {"power": power, "obs_raw": obs_raw, "strehl": strehl})
        if self._persistent_out and out is None:
            self._step_cache = (ret, ptrs)
        return ret

    def _as_actions(self, actions):
        """[B, A] float32 contiguous device tensor (as is when it already is one)."""
        torch = self._torch
        if (isinstance(actions, torch.Tensor) and actions.dtype == torch.float32 and actions.device == self.device and actions.is_contiguous()
                and actions.dim() == 2 and actions.shape[0] == self.num_envs and actions.shape[1] == self.num_modes):
            return actions
        a = torch.as_tensor(actions, device=self.device)
        if a.dtype != torch.float32:
            a = a.to(torch.float32)
        return a.reshape(self.num_envs, self.num_modes).contiguous()

    def _launch_step(self, a, next_actions, ptrs):
        p = C.c_void_p
        pending = getattr(self, "_next_actions_keepalive", None)
        if next_actions is BatchedAOEnv._PIPELINE_END:
            self._next_actions_keepalive = None
            _lib.check(self.lib.aog_step(self._handle, p(a.data_ptr()), p(ptrs[0]), p(ptrs[1]), p(ptrs[2]), p(ptrs[3]), p(ptrs[4]), p(ptrs[5]),
                                         self._stream()))
        else:
            # continuation of a pipelined sequence: the mirror already holds what the PREVIOUS call announced as next_actions and the library
            # ignores `actions` — a caller that passes something else (a corrected action, say) would silently get the old one's results
            if pending is not None and a is not pending and not (a.data_ptr() == pending.data_ptr() and a.shape == pending.shape):
                # (other storage: compared by value, which synchronises — AOG_CHECK_PIPELINE=0 skips it for callers that copy their actions around)
                if os.environ.get("AOG_CHECK_PIPELINE") != "0" and not bool(self._torch.equal(a, pending)):
                    raise ValueError("step(next_actions=...): `actions` differs from the next_actions announced by the previous call of this "
                                     "pipelined sequence (the mirror already holds those); end the sequence with next_actions=None first")
            nxt = None
            if next_actions is not None:
                nxt = self._as_actions(next_actions)
            self._next_actions_keepalive = nxt   # (read by the launch enqueued here; checked against the next call's actions)
            _lib.check(self.lib.aog_step_pipelined(self._handle, p(a.data_ptr()), p(nxt.data_ptr() if nxt is not None else None), p(ptrs[0]),
                                                   p(ptrs[1]), p(ptrs[2]), p(ptrs[3]), p(ptrs[4]), p(ptrs[5]), self._stream()))

    def persistent_outputs(self, enable=True):
        """Write every ``step``'s outputs into ONE block that lives as long as the env instead of fresh tensors: the tensors a step
        returns are then views that the NEXT step overwrites.  For callers that consume a step's outputs before stepping again (the
        single-env wrapper: one device-to-host copy of the block per step); returns the block layout (n = obs_dim^2):
        float32 [B n] obs_raw | [B] reward | [B] power | [B] strehl, then float16 [B n] obs, then uint8 [B] done."""
        self._persistent_out = bool(enable)
        self._step_cache = None
        if not enable:
            self._pack = None
        n, B = self.obs_dim ** 2, self.num_envs
        return {"float32_bytes": 4 * B * (n + 3), "float16_bytes": 2 * B * n, "uint8_bytes": B}

    # ------------------------------------------------------------------------------------------------
    def get_actuators(self):
        """deformable_mirror.actuators of every env, [B, A] float64 (metres)."""
        torch = self._torch
        out = torch.empty((self.num_envs, self.num_modes), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.aog_get_actuators(self._handle, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_actuators(self, act):
        torch = self._torch
        a = torch.as_tensor(act, device=self.device).to(torch.float64).reshape(self.num_envs, self.num_modes).contiguous()
        _lib.check(self.lib.aog_set_actuators(self._handle, C.c_void_p(a.data_ptr()), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def focal_image(self, env_index=0):
        """``wf_wfs_after_foc.electric_field`` of one env (AO_env.py:138): complex64 [n_focal, n_focal] (y, x), up to a global
        phase.  ``abs()**2 * tables.focal_pixel_area`` is the ``.power`` image the reference's render() shows."""
        torch = self._torch
        nf = int(self.tables.focal_m1.shape[0])
        out = torch.empty((nf, nf, 2), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aog_focal_image(self._handle, int(env_index), C.c_void_p(out.data_ptr()), self._stream()))
        return torch.view_as_complex(out)

    def focal_images(self, first=0, count=None):
        """``wf_wfs_after_foc.electric_field`` (AO_env.py:138) of envs [first, first + count), default all: complex64
        [count, n_focal, n_focal] — the batched form of ``focal_image`` (matrix-core complex GEMM pair, fast precision only)."""
        torch = self._torch
        nf = int(self.tables.focal_m1.shape[0])
        count = self.num_envs - first if count is None else int(count)
        out = torch.empty((count, nf, nf, 2), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aog_focal_images(self._handle, int(first), count, C.c_void_p(out.data_ptr()), self._stream()))
        return torch.view_as_complex(out)

    def phase_screen(self, env_index=0):
        """Atmospheric phase at the sensing wavelength [N, N] float32 radians (0 outside the aperture, aperture mean removed) — the
        quantity render() displays as ``phase_screen_opd`` after scaling by lambda_wfs / (2 pi) * 1e6 (AO_env.py:87-88)."""
        torch = self._torch
        N = self.num_pupil_pixels
        out = torch.empty((N, N), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.aog_get_phase_screen(self._handle, int(env_index), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def get_state(self):
        """Snapshot of everything that evolves: library state blob (screens, actuators, counters) + Python-side counters and the
        host RNG streams.  ``set_state`` on an env built with the same arguments resumes bit-identically."""
        torch = self._torch
        n = int(self.lib.aog_state_bytes(self._handle))
        blob = torch.empty((n,), dtype=torch.uint8, device=self.device)
        ts = C.c_int64()
        _lib.check(self.lib.aog_get_state(self._handle, C.c_void_p(blob.data_ptr()), C.byref(ts), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        rng = [self._env_rng(e).get_state() for e in range(self.num_envs)] if self._host_rng else None
        return {"blob": blob, "lib_timestep": int(ts.value), "timestep": self.timestep, "episode_no": self.episode_no, "rng": rng}

    def set_state(self, state):
        torch = self._torch
        blob = state["blob"].to(self.device).contiguous()
        if blob.numel() != int(self.lib.aog_state_bytes(self._handle)):
            raise ValueError("state blob does not match this environment's configuration")
        _lib.check(self.lib.aog_set_state(self._handle, C.c_void_p(blob.data_ptr()), int(state["lib_timestep"]), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        self.timestep = int(state["timestep"])
        self.episode_no = int(state["episode_no"])
        if state.get("rng") is not None:
            for e, st in enumerate(state["rng"]):
                self._env_rng(e).set_state(st)

    def accumulate_returns(self, returns=None):
        """Have every ``step`` add its rewards into ``returns`` ([B] float32 contiguous device tensor; the caller zeroes it at
        episode start) inside the step's last kernel — the episode-return sum of the rollout without a launch of its own.
        ``None`` detaches.  The tensor must stay alive while attached (a reference is kept here)."""
        torch = self._torch
        if getattr(self, "_handle", None) is None:   # closed env: nothing to attach to or detach from
            self._returns_ref = None
            return
        if returns is not None:
            ok = returns.dtype == torch.float32 and tuple(returns.shape) == (self.num_envs,) and returns.is_contiguous() and returns.is_cuda
            if not ok:
                raise ValueError("accumulate_returns: expected a contiguous float32 [num_envs] tensor on the env's device")
        self._returns_ref = returns
        _lib.check(self.lib.aog_set_return_accumulator(self._handle, C.c_void_p(returns.data_ptr() if returns is not None else None)))

    def device_status(self):
        """Synchronise and return the library's sticky device status word (0 = fine)."""
        v = C.c_int32()
        _lib.check(self.lib.aog_device_status(self._handle, C.byref(v)))
        return int(v.value)

    def profile(self, enable=True, every=1, block=8):
        """HIP-event timing of the fused kernel; ``every`` = n times one block of ``block`` consecutive launches in n (the records hold the
        stream ~6 us per timed launch)."""
        _lib.check(self.lib.aog_profile_block(self._handle, int(block)))
        _lib.check(self.lib.aog_profile_enable(self._handle, max(1, int(every)) if enable else 0))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int()
        _lib.check(self.lib.aog_profile_read(self._handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_kernels(self):
        """{name: (mean ms, launches)} of every kernel id timed since profiling was switched on, as collected by the LAST ``profile_read()``
        (``_lib.AOG_PROF``: screen synthesis passes, screen packing, extrusion, Shack-Hartmann passes; HIP events on the launch stream)."""
        out = {}
        for name, kid in _lib.AOG_PROF.items():
            ms, n = C.c_double(), C.c_int()
            _lib.check(self.lib.aog_profile_read_kernel(self._handle, kid, C.byref(ms), C.byref(n)))
            if n.value:
                out[name] = (ms.value, n.value)
        return out

    def close(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self.lib.aog_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
