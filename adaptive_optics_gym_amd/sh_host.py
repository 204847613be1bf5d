"""Host-side (numpy float64, one-time) construction of the Shack-Hartmann baseline controller — the product counterpart of
``AOEnv.shack_hartmann_init`` (AO_env.py:396-465): magnifier, square micro-lens array + Fresnel propagation over one lenslet
focal length, flux-selected sub-apertures, reference slopes, poke-calibrated interaction matrix and its Tikhonov inverse.

Everything the per-step device chain (``aog_sh_image`` / ``aog_sh_update``) needs is returned as plain arrays; the
calibration itself (2 x act_dim propagations of 2N x 2N FFTs) runs here once, like it does in the reference.
"""
from __future__ import annotations

import numpy as np

from .optics_host import HostTables, centred_axis, focal_axis
from .params import OpticalParams


def _fresnel_transfer(n: int, pitch: float, wavelength: float, distance: float, q: int = 2) -> np.ndarray:
    """hcipy FresnelPropagator's transfer function on the UNSHIFTED (q n)^2 FFT grid.  hcipy chooses by a sampling test: the analytic
    paraxial transfer function while pitch >= lambda z / L (every configured geometry: the reference's f-number 50 reaches the other
    branch only above ~800 pupil pixels), otherwise the impulse-response method — the Fourier transform, over the padded grid with a
    sample at r = 0, of h(r) = e^{ikz} e^{ik r^2 / 2z} / (i lambda z).  h factorises, so its transform is the outer product of two
    1-D transforms: both branches give a table the separable Shack-Hartmann passes accept."""
    k = 2 * np.pi / wavelength
    m = n * q
    kk = 2 * np.pi * np.fft.fftfreq(m, pitch)
    if pitch < wavelength * distance / (n * pitch):
        x = (np.arange(m) - m / 2 + (m % 2) * 0.5) * pitch                 # make_fft_grid of the Fourier grid
        chirp = np.exp(1j * k * x ** 2 / (2 * distance))
        line = pitch * (np.exp(-1j * np.outer(kk, x)) @ chirp)              # sum_x e^{ik x^2 / 2z} e^{-i kx x} dx, true coordinates
        return np.exp(1j * k * distance) / (1j * wavelength * distance) * np.outer(line, line)
    k2 = kk[None, :] ** 2 + kk[:, None] ** 2
    return np.exp(-0.5j * distance * k2 / k) * np.exp(1j * k * distance)


class ShackHartmannHost:
    def __init__(self, params: OpticalParams, tables: HostTables):
        p = params
        N = p.num_pupil_pixels
        self.N = N
        self.mag = p.sh_diameter / p.telescope_diameter
        self.pitch = p.pupil_pixel * self.mag                      # pupil_grid.scaled(magnification)
        ax = centred_axis(N, p.telescope_diameter) * self.mag
        lenslet_d = p.sh_diameter / p.num_lenslets
        centres = np.arange(-p.sh_diameter, p.sh_diameter, lenslet_d)
        self.n_lenslets_axis = len(centres)
        focal = p.f_number * lenslet_d
        near = np.argmin(np.abs(ax[:, None] - centres[None, :]), axis=1)
        self.mla_index = (near[:, None] * len(centres) + near[None, :]).ravel()          # iy * nl + ix, x fastest
        d2 = ((ax - centres[near]) ** 2)
        opd = (-1.0 / (2 * focal)) * (d2[:, None] + d2[None, :]).ravel()
        self.k_wfs = 2 * np.pi / p.wavelength_wfs
        self.mla_phase = np.exp(1j * opd * self.k_wfs)                                      # SurfaceApodizer(n = 2)
        self.transfer = _fresnel_transfer(N, self.pitch, p.wavelength_wfs, focal)           # [2N, 2N] unshifted
        # detector coordinates: NoiselessDetector(focal_grid) — the 240^2 science focal grid in the reference (AO_env.py:412)
        self.x_det = focal_axis(p.focal_q, p.focal_num_airy, p.wavelength_sci / p.telescope_diameter) if N == 240 else \
            (p.wavelength_sci / p.telescope_diameter / p.focal_q) * (np.arange(N) - N / 2 + (N % 2) * 0.5)
        gx, gy = np.meshgrid(centres, centres)
        self.mla_points = np.stack([gx.ravel(), gy.ravel()], axis=1)
        self.modes = tables.modes
        self.ap_index = tables.ap_index
        self.n_ap = tables.n_ap
        self.pix_area_pupil = p.pupil_pixel ** 2
        aperture = np.zeros(N * N)
        aperture[self.ap_index] = 1.0
        self.aperture = aperture

        # sub-aperture selection from the reference image (AO_env.py:413-425)
        image_ref = self.image(aperture.astype(complex), 1.0)
        present = np.unique(self.mla_index)
        flux = np.bincount(self.mla_index, weights=image_ref, minlength=len(self.mla_points))[present]
        self.subapertures = present[flux > 0.5 * flux.max()]
        self.n_sub = len(self.subapertures)
        slot = -np.ones(len(self.mla_points), dtype=np.int32)
        slot[self.subapertures] = np.arange(self.n_sub, dtype=np.int32)
        self.sub_slot = slot[self.mla_index].astype(np.int32)                               # per pixel: slot or -1
        self.centres = self.mla_points[self.subapertures]                                   # [n_sub, 2] (x, y)
        self.slopes_ref = self.slopes(image_ref)

        # interaction matrix by +-0.01 lambda pokes of a power-1 wavefront (AO_env.py:433-461), Tikhonov inverse (:464-465)
        amp_cal = 1.0 / np.sqrt(self.n_ap * self.pix_area_pupil)
        probe = 0.01 * p.wavelength_wfs
        A = self.modes.shape[1]
        response = np.empty((2 * self.n_sub, A))
        for i in range(A):
            acc = 0
            for amp in (-probe, probe):
                field = np.zeros(N * N, dtype=complex)
                field[self.ap_index] = amp_cal * np.exp(2j * self.k_wfs * amp * self.modes[:, i])
                acc = acc + amp * self.slopes(self.image(field, 1.0)) / probe ** 2          # np.var([-p, p]) = p^2
            response[:, i] = acc
        self.response = response
        U, S, Vt = np.linalg.svd(response, full_matrices=False)
        self.reconstruction = (Vt.T * (S / (S ** 2 + (1e-3 * S.max()) ** 2))) @ U.T        # [A, 2 n_sub]
        # source used by SH_step: Wavefront(aperture, lambda_wfs).total_power = 3.9e10 * 10^(-m/2.5) (AO_env.py:324-326)
        self.amp_wfs = np.sqrt(3.9e10 * 10 ** (-p.stellar_magnitude / 2.5) / (self.n_ap * self.pix_area_pupil))

    def image(self, pupil_field: np.ndarray, dt: float) -> np.ndarray:
        """camera.integrate(shwfs(magnifier(wf)), dt) for a pupil-plane field given on the full N x N grid."""
        N = self.N
        e = (pupil_field / self.mag * self.mla_phase).reshape(N, N)
        pad = np.zeros((2 * N, 2 * N), dtype=complex)
        pad[:N, :N] = e     # placement inside the padded array only cyclically shifts the (shift-equivariant) result
        out = np.fft.ifft2(np.fft.fft2(pad) * self.transfer)[:N, :N]
        return (np.abs(out) ** 2).ravel() * self.pitch ** 2 * dt

    def slopes(self, image: np.ndarray) -> np.ndarray:
        """estimate([image]).ravel(): all x centroids then all y centroids, relative to the lenslet positions."""
        N = self.N
        sel = self.sub_slot >= 0
        w = image[sel]
        s = self.sub_slot[sel]
        flux = np.bincount(s, weights=w, minlength=self.n_sub)
        xs = np.tile(self.x_det, N)[sel]
        ys = np.repeat(self.x_det, N)[sel]
        cx = np.bincount(s, weights=w * xs, minlength=self.n_sub) / flux - self.centres[:, 0]
        cy = np.bincount(s, weights=w * ys, minlength=self.n_sub) / flux - self.centres[:, 1]
        return np.concatenate([cx, cy])
