"""CPU float64 restatement of the hcipy==0.5.1 functions used by the AOEnv hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``adaptive_optics_gym_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / the CPU number printed beside the
GPU number.

PARITY UNPINNED.  The reference (``/root/reference/gym_AO/envs/AO_env.py``) delegates all
arithmetic to the third-party package ``hcipy==0.5.1`` (``requirements.txt:1``), which is
neither vendored in the reference nor installed in this image, and the reference ships no
tests, fixtures or golden vectors.  Every function below therefore restates the *published*
HCIPy algorithm (names of the HCIPy symbols are given per function, together with the
``AO_env.py`` call site that uses it) and is pinned only by analytic known-answer tests
(``tests/test_oracle_kat.py``) and by SURVEY.md Appendix B's consistency numbers.

Conventions (HCIPy's): a field is a flat 1-D array over a separable regular grid with x
fastest (``index = iy*nx + ix``); ``.shaped`` is ``[ny, nx]``; ``weights`` is the scalar
pixel area.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import scipy.optimize
from scipy.special import gamma, jnp_zeros, jv, kn, kv


# --------------------------------------------------------------------------------------
# Grids  (hcipy.field: CartesianGrid(RegularCoords), make_pupil_grid, make_focal_grid)
# --------------------------------------------------------------------------------------
class Grid:
    """Regular separable Cartesian grid; ``dims = (nx, ny)``, ``delta``, ``zero`` per axis."""

    def __init__(self, delta, dims, zero):
        self.delta = np.asarray(delta, dtype=float) * np.ones(2)
        self.dims = (np.asarray(dims) * np.ones(2)).astype(int)
        self.zero = np.asarray(zero, dtype=float) * np.ones(2)

    @property
    def shape(self):  # numpy shape of a ``.shaped`` field: (ny, nx)
        return (int(self.dims[1]), int(self.dims[0]))

    @property
    def size(self):
        return int(self.dims[0] * self.dims[1])

    @property
    def weights(self):
        return float(self.delta[0] * self.delta[1])

    @property
    def separated_coords(self):
        xs = self.zero[0] + self.delta[0] * np.arange(self.dims[0])
        ys = self.zero[1] + self.delta[1] * np.arange(self.dims[1])
        return xs, ys

    @property
    def x(self):
        xs, ys = self.separated_coords
        return np.tile(xs, len(ys))

    @property
    def y(self):
        xs, ys = self.separated_coords
        return np.repeat(ys, len(xs))

    def scaled(self, s):
        return Grid(self.delta * s, self.dims, self.zero * s)


def make_pupil_grid(dims, diameter=1.0):
    """hcipy.field.make_pupil_grid (AO_env.py:300,378,384,385): delta = D/n, symmetric,
    x_i = -D/2 + delta/2 + i*delta."""
    diameter = np.ones(2) * float(diameter)
    dims = (np.ones(2) * dims).astype(int)
    delta = diameter / dims
    zero = -diameter / 2 + delta / 2
    return Grid(delta, dims, zero)


def make_focal_grid(q, num_airy, spatial_resolution):
    """hcipy.field.make_focal_grid (AO_env.py:314): delta = res/q, dims = 2*q*num_airy,
    zero = delta*(-dims/2 + (dims mod 2)/2) -> a sample exactly on the axis for even dims."""
    q = np.ones(2) * float(q)
    num_airy = np.ones(2) * float(num_airy)
    delta = spatial_resolution / q * np.ones(2)
    dims = (2 * num_airy * q).astype(int)
    zero = delta * (-dims / 2 + np.mod(dims, 2) * 0.5)
    return Grid(delta, dims, zero)


def make_circular_aperture(diameter):
    """hcipy.aperture.make_circular_aperture (AO_env.py:301): 1.0 where x^2+y^2 <= (D/2)^2."""

    def func(grid):
        return ((grid.x ** 2 + grid.y ** 2) <= (diameter / 2) ** 2).astype(float)

    return func


# --------------------------------------------------------------------------------------
# Wavefront  (hcipy.optics.Wavefront)
# --------------------------------------------------------------------------------------
class Wavefront:
    def __init__(self, electric_field, wavelength, grid):
        self.electric_field = np.array(electric_field, dtype=complex)  # independent copy
        self.wavelength = float(wavelength)
        self.grid = grid

    def copy(self):
        return Wavefront(self.electric_field, self.wavelength, self.grid)

    @property
    def wavenumber(self):
        return 2 * np.pi / self.wavelength

    @property
    def intensity(self):
        return np.abs(self.electric_field) ** 2

    @property
    def power(self):
        return self.intensity * self.grid.weights

    @property
    def total_power(self):
        return float(np.sum(self.power))

    @total_power.setter
    def total_power(self, p):
        self.electric_field *= np.sqrt(p / self.total_power)


# --------------------------------------------------------------------------------------
# Fraunhofer propagation (hcipy.propagation.FraunhoferPropagator + MatrixFourierTransform)
# --------------------------------------------------------------------------------------
class FraunhoferPropagator:
    """E_out(X) = 1/(i lambda f) * sum_x E(x) w_in exp(-i (2 pi/(lambda f)) X.x)
    (AO_env.py:316,390,391).  The transform is instantiated for the *actual* grid of the
    wavefront at call time (HCIPy's agnostic optical element), which is why the reference can
    hand 240^2 wavefronts to a propagator constructed on a 128^2 grid (AO_env.py:390 vs 329)."""

    def __init__(self, input_grid, output_grid, focal_length=1.0):
        self.output_grid = output_grid
        self.focal_length = float(focal_length)
        self._cache = {}

    def _matrices(self, grid, wavelength):
        key = (tuple(grid.delta), tuple(grid.dims), tuple(grid.zero), wavelength)
        if key not in self._cache:
            uv = self.output_grid.scaled(2 * np.pi / (self.focal_length * wavelength))
            xin, yin = grid.separated_coords
            u, v = uv.separated_coords
            m1 = np.exp(-1j * np.outer(v, yin))  # [ny_out, ny_in]
            m2 = np.exp(-1j * np.outer(xin, u))  # [nx_in, nx_out]
            self._cache[key] = (m1, m2)
        return self._cache[key]

    def forward(self, wf):
        m1, m2 = self._matrices(wf.grid, wf.wavelength)
        f = wf.electric_field.reshape(wf.grid.shape) * wf.grid.weights
        out = (m1 @ f @ m2).ravel() / (1j * self.focal_length * wf.wavelength)
        return Wavefront(out, wf.wavelength, self.output_grid)

    __call__ = forward


# --------------------------------------------------------------------------------------
# Mode bases (hcipy.mode_basis: make_zernike_basis, make_disk_harmonic_basis, ModeBasis)
# --------------------------------------------------------------------------------------
def noll_to_zernike(j):
    """hcipy.mode_basis.noll_to_zernike: Noll index (1-based) -> (n, m); even j -> +m (cos)."""
    n = int(math.sqrt(2 * j - 1) + 0.5) - 1
    if n % 2:
        m = 2 * int((2 * (j + 1) - n * (n + 1)) // 4) - 1
    else:
        m = 2 * int((2 * j + 1 - n * (n + 1)) // 4)
    return n, m * (-1) ** (j % 2)


def zernike_radial(n, m, r):
    """R_n^|m|(r) by the explicit factorial sum (HCIPy uses a recursion; same polynomial)."""
    m = abs(m)
    out = np.zeros_like(r)
    for k in range((n - m) // 2 + 1):
        c = ((-1) ** k * math.factorial(n - k)
             / (math.factorial(k) * math.factorial((n + m) // 2 - k) * math.factorial((n - m) // 2 - k)))
        out = out + c * r ** (n - 2 * k)
    return out


def zernike(n, m, D, grid):
    """hcipy.mode_basis.zernike with radial_cutoff=True: sqrt(n+1) R_n^|m|(2r/D) * azimuthal,
    azimuthal = sqrt2 cos(m t) (m>0) | sqrt2 sin(|m| t) (m<0) | 1; zero where 2r >= D."""
    x, y = grid.x, grid.y
    r = np.hypot(x, y)
    theta = np.arctan2(y, x)
    if m < 0:
        az = math.sqrt(2) * np.sin(-m * theta)
    elif m == 0:
        az = np.ones_like(theta)
    else:
        az = math.sqrt(2) * np.cos(m * theta)
    z = math.sqrt(n + 1) * az * zernike_radial(n, m, 2 * r / D)
    return z * ((2 * r) < D)


def make_zernike_basis(num_modes, D, grid, starting_mode=1):
    """AO_env.py:346 — Noll-ordered modes starting at piston.  Returns a list of flat fields."""
    return [zernike(*noll_to_zernike(j), D, grid) for j in range(starting_mode, starting_mode + num_modes)]


def disk_harmonic_energy(n, m, bc="neumann"):
    m = abs(m)
    if bc != "neumann":
        raise NotImplementedError(bc)
    return float(jnp_zeros(m, n)[-1]) ** 2


def get_disk_harmonic_orders_sorted(num_modes, bc="neumann"):
    """hcipy.mode_basis.get_disk_harmonic_orders_sorted — frontier search seeded with (1,0);
    pop lowest-energy frontier entry, emit (n,-m) then (n,m) (only (n,0) if m==0), push
    (n,m+1),(n+1,m) when unseen.  (ordering is the top parity risk.)"""
    orders = [(1, 0)]
    energies = [disk_harmonic_energy(1, 0, bc)]
    results = []
    while len(results) < num_modes:
        k = int(np.argmin(energies))
        order = orders[k]
        if order[1] != 0:
            results.append((order[0], -order[1]))
        results.append(order)
        del orders[k]
        del energies[k]
        for new in ((order[0], order[1] + 1), (order[0] + 1, order[1])):
            if new not in results and new not in orders:
                orders.append(new)
                energies.append(disk_harmonic_energy(new[0], new[1], bc))
    return results[:num_modes]


def disk_harmonic(n, m, D, grid, bc="neumann"):
    """hcipy.mode_basis.disk_harmonic: J_|m|(lambda_mn * 2r/D) * {cos m t | sin |m| t}, masked to
    the aperture and L2-normalised over it (the normalisation cancels against the ptp division
    at AO_env.py:353)."""
    x, y = grid.x, grid.y
    r = 2 * np.hypot(x, y) / D
    theta = np.arctan2(y, x)
    m_neg = m < 0
    m = abs(m)
    lam = float(jnp_zeros(m, n)[-1])
    z = jv(m, lam * r) * (np.sin(m * theta) if m_neg else np.cos(m * theta))
    mask = make_circular_aperture(D)(grid) > 0.5
    z = z * mask
    z = z / np.sqrt(np.sum(z[mask] ** 2 * grid.weights))
    return z


def make_disk_harmonic_basis(grid, num_modes, D, bc="neumann"):
    """AO_env.py:352."""
    return [disk_harmonic(n, m, D, grid, bc) for (n, m) in get_disk_harmonic_orders_sorted(num_modes, bc)]


class ModeBasis:
    """hcipy.mode_basis.ModeBasis: modes stacked as columns of ``transformation_matrix``."""

    def __init__(self, modes):
        self.transformation_matrix = np.stack([np.asarray(m, dtype=float) for m in modes], axis=-1)

    def __len__(self):
        return self.transformation_matrix.shape[1]

    def linear_combination(self, coefficients):
        return self.transformation_matrix.dot(coefficients)


class DeformableMirror:
    """hcipy.optics.DeformableMirror (AO_env.py:348,354,119-120,135): surface = T.actuators,
    forward multiplies by exp(2i * surface * k)."""

    def __init__(self, influence_functions):
        self.influence_functions = influence_functions
        self._actuators = np.zeros(len(influence_functions))
        self._surface = None

    @property
    def actuators(self):
        return self._actuators

    @actuators.setter
    def actuators(self, a):
        self._actuators = a
        self._surface = None

    @property
    def surface(self):
        if self._surface is None:
            self._surface = self.influence_functions.linear_combination(self._actuators)
        return self._surface

    def flatten(self):
        self._actuators = np.zeros(len(self._actuators))
        self._surface = None

    def forward(self, wf):
        out = wf.copy()
        out.electric_field *= np.exp(2j * self.surface * wf.wavenumber)
        return out

    __call__ = forward


# --------------------------------------------------------------------------------------
# Atmosphere (hcipy.atmosphere: Cn_squared_from_fried_parameter, InfiniteAtmosphericLayer,
# FiniteAtmosphericLayer, SpectralNoiseFactoryFFT, phase_covariance_von_karman, ...)
# --------------------------------------------------------------------------------------
def Cn_squared_from_fried_parameter(r0, wavelength):
    """AO_env.py:367."""
    k = 2 * np.pi / wavelength
    return r0 ** (-5.0 / 3) / (0.423 * k ** 2)


def fried_parameter_from_Cn_squared(Cn_squared, wavelength):
    k = 2 * np.pi / wavelength
    return (0.423 * Cn_squared * k ** 2) ** (-3.0 / 5)


def phase_covariance_von_karman(r0, L0):
    def func(r):
        r = r + 1e-10
        a = (L0 / r0) ** (5 / 3)
        b = gamma(11 / 6) / (2 ** (5 / 6) * np.pi ** (8 / 3))
        c = (24 / 5 * gamma(6 / 5)) ** (5 / 6)
        d = (2 * np.pi * r / L0) ** (5 / 6)
        e = kv(5 / 6, 2 * np.pi * r / L0)
        return a * b * c * d * e

    return func


def power_spectral_density_von_karman(r0, L0):
    def func(u):
        u = u + 1e-10
        u0 = 2 * np.pi / L0
        res = 0.0229 * ((u ** 2 + u0 ** 2) / (2 * np.pi) ** 2) ** (-11 / 6.0) * r0 ** (-5.0 / 3)
        res[u < 1e-9] = 0
        return res

    return func


def von_karman_screen_fft(grid, Cn_squared, L0, oversampling=16, rng=np.random):
    """FiniteAtmosphericLayer(...).phase_for(1) (hcipy): SpectralNoiseFactoryFFT on a
    (oversampling*N)^2 FFT grid, du = 2 pi/(oversampling*N*delta);
    C = sqrt(PSD(u) (2 pi)^2 / du^2); draws randn(M) (real parts) THEN randn(M) (imag parts);
    screen = Re[centred ifft2(C*g)] / delta^2 * sqrt(Cn^2), cropped to the central N^2."""
    nx, ny = int(grid.dims[0]), int(grid.dims[1])
    q = int(oversampling)
    mx, my = nx * q, ny * q
    du = (2 * np.pi / (grid.delta * grid.dims)) / q
    ux = du[0] * (np.arange(mx) - mx // 2)
    uy = du[1] * (np.arange(my) - my // 2)
    r0 = fried_parameter_from_Cn_squared(1, 1)
    psd = power_spectral_density_von_karman(r0, L0)
    ur = np.hypot(ux[np.newaxis, :], uy[:, np.newaxis]).ravel()
    C = np.sqrt(psd(ur) / (du[0] * du[1]) * (2 * np.pi) ** 2)
    del ur
    M = mx * my
    g_re = rng.randn(M)
    g_im = rng.randn(M)
    C = (C * (g_re + 1j * g_im)).reshape(my, mx)
    del g_re, g_im
    f = np.fft.fftshift(np.fft.ifftn(np.fft.ifftshift(C)))
    y0 = my // 2 - ny // 2
    x0 = mx // 2 - nx // 2
    res = f[y0:y0 + ny, x0:x0 + nx].ravel() / grid.weights
    return res.real * np.sqrt(Cn_squared)


class InfiniteAtmosphericLayer:
    """hcipy.atmosphere.InfiniteAtmosphericLayer(pupil_grid, Cn^2, L0, velocity) (AO_env.py:370).

    RNG draw order on the module-level legacy generator:
      1. ``rand()``                     wind direction (scalar velocity, even when it is 0)
      2. ``geometric(0.5, nx)``         extra stencil sample per column (bottom stencil)
      3. ``geometric(0.5, ny)``         extra stencil sample per row (left stencil)
      4. ``randn(M)``, ``randn(M)``     initial screen, M = (16 N)^2
      per extrusion: ``normal(0, 1, N)``.
    """

    def __init__(self, input_grid, Cn_squared, L0, velocity, stencil_length=2, rng=np.random,
                 initial_screen=None):
        self.rng = rng
        self.input_grid = input_grid
        self.Cn_squared = float(Cn_squared)
        self.L0 = float(L0)
        if np.isscalar(velocity):
            theta = rng.rand() * 2 * np.pi
            self.velocity = velocity * np.array([np.cos(theta), np.sin(theta)])
        else:
            self.velocity = np.array(velocity, dtype=float)
        self.stencil_length = stencil_length
        self._make_stencils()
        self._make_covariance_matrices()
        self._make_AB_matrices()
        if initial_screen is None:
            self._make_initial_phase_screen()
        else:
            self._achromatic_screen = np.array(initial_screen, dtype=float).ravel()
        self.center = np.zeros(2)
        self._t = 0.0

    # -- construction ---------------------------------------------------------------
    def _make_stencils(self):
        g = self.input_grid
        nx, ny = int(g.dims[0]), int(g.dims[1])
        sb = np.zeros((ny, nx), dtype=bool)
        sb[: self.stencil_length, :] = True
        for i, n in enumerate(self.rng.geometric(0.5, nx)):
            sb[(n + self.stencil_length - 1) % ny, i] = True
        self.stencil_bottom = sb.ravel()
        self.num_stencils_vertical = int(np.sum(self.stencil_bottom))

        sl = np.zeros((ny, nx), dtype=bool)
        sl[:, : self.stencil_length] = True
        for i, n in enumerate(self.rng.geometric(0.5, ny)):
            sl[i, (n + self.stencil_length - 1) % nx] = True
        self.stencil_left = sl.ravel()
        self.num_stencils_horizontal = int(np.sum(self.stencil_left))

    def _make_covariance_matrices(self):
        g = self.input_grid
        xs, ys = g.separated_coords
        cov = phase_covariance_von_karman(fried_parameter_from_Cn_squared(1, 1), self.L0)
        gx, gy = g.x, g.y
        # vertical: new row one pixel below the first row
        new_x, new_y = xs, np.full(len(xs), g.zero[1] - g.delta[1])
        x = np.concatenate((gx[self.stencil_bottom], new_x))
        y = np.concatenate((gy[self.stencil_bottom], new_y))
        sep = np.hypot(x[:, None] - x[None, :], y[:, None] - y[None, :])
        self.cov_matrix_vertical = cov(sep)
        # horizontal: new column one pixel left of the first column
        new_x, new_y = np.full(len(ys), g.zero[0] - g.delta[0]), ys
        x = np.concatenate((gx[self.stencil_left], new_x))
        y = np.concatenate((gy[self.stencil_left], new_y))
        sep = np.hypot(x[:, None] - x[None, :], y[:, None] - y[None, :])
        self.cov_matrix_horizontal = cov(sep)

    @staticmethod
    def _ab(cov, n, n_new):
        cov_zz = cov[:n, :n]
        cov_xz = cov[n:, :n]
        cov_zx = cov[:n, n:]
        cov_xx = cov[n:, n:]
        cf = scipy.linalg.cho_factor(cov_zz)
        inv_cov_zz = scipy.linalg.cho_solve(cf, np.eye(n))
        A = cov_xz.dot(inv_cov_zz)
        BBt = cov_xx - A.dot(cov_zx)
        U, S, _ = np.linalg.svd(BBt)
        L = np.sqrt(S[:n_new])
        return A, U * L

    def _make_AB_matrices(self):
        nx, ny = int(self.input_grid.dims[0]), int(self.input_grid.dims[1])
        self.A_vertical, self.B_vertical = self._ab(self.cov_matrix_vertical, self.num_stencils_vertical, nx)
        self.A_horizontal, self.B_horizontal = self._ab(self.cov_matrix_horizontal,
                                                        self.num_stencils_horizontal, ny)

    def _make_initial_phase_screen(self):
        self._achromatic_screen = von_karman_screen_fft(self.input_grid, self.Cn_squared, self.L0, 16, self.rng)

    # -- evolution ------------------------------------------------------------------
    def _extrude(self, where):
        flipped = where in ("top", "right")
        horizontal = where in ("left", "right")
        screen = self._achromatic_screen[::-1] if flipped else self._achromatic_screen
        if horizontal:
            stencil, A, B = self.stencil_left, self.A_horizontal, self.B_horizontal
        else:
            stencil, A, B = self.stencil_bottom, self.A_vertical, self.B_vertical
        stencil_data = screen[stencil]
        random_data = self.rng.normal(0, 1, size=B.shape[1])
        new_slice = A.dot(stencil_data) + B.dot(random_data) * np.sqrt(self.Cn_squared)
        screen = screen.reshape(self.input_grid.shape)
        if horizontal:
            screen = np.hstack((new_slice[:, np.newaxis], screen[:, :-1]))
        else:
            screen = np.vstack((new_slice[np.newaxis, :], screen[:-1, :]))
        if flipped:
            self._achromatic_screen = screen[::-1, ::-1].ravel()
        else:
            self._achromatic_screen = screen.ravel()

    def reset(self):
        self._make_initial_phase_screen()
        self.center = np.zeros(2)
        self._t = 0.0

    def evolve_until(self, t):
        old_center = np.round(self.center / self.input_grid.delta).astype(int)
        self.center = self.velocity * t
        new_center = np.round(self.center / self.input_grid.delta).astype(int)
        delta = new_center - old_center
        for _ in range(abs(int(delta[0]))):
            self._extrude("left" if delta[0] < 0 else "right")
        for _ in range(abs(int(delta[1]))):
            self._extrude("bottom" if delta[1] < 0 else "top")

    @property
    def t(self):
        return self._t

    @t.setter
    def t(self, t):
        self.evolve_until(t)
        self._t = t

    def phase_for(self, wavelength):
        return self._achromatic_screen / wavelength

    def forward(self, wf):
        out = wf.copy()
        out.electric_field *= np.exp(1j * self.phase_for(wf.wavelength))
        return out

    __call__ = forward


# --------------------------------------------------------------------------------------
# Step-index fiber (hcipy.fiber: StepIndexFiber, make_LP_modes)
# --------------------------------------------------------------------------------------
def _lp_eigenvalue_equation(u, m, V):
    w = np.sqrt(V ** 2 - u ** 2)
    return jv(m, u) / (u * jv(m + 1, u)) - kn(m, w) / (w * kn(m + 1, w))


def find_lp_solutions(m, V):
    """Roots u in (0, V) of J_m(u)/(u J_{m+1}(u)) = K_m(w)/(w K_{m+1}(w)), u^2+w^2=V^2."""
    u = np.linspace(0, V, 20001)[1:-1]
    f = _lp_eigenvalue_equation(u, m, V)
    roots = []
    for i in np.flatnonzero(np.sign(f[:-1]) != np.sign(f[1:])):
        if not (np.isfinite(f[i]) and np.isfinite(f[i + 1])):
            continue
        if abs(f[i]) > 1e3 or abs(f[i + 1]) > 1e3:  # sign flip through a pole of 1/J_{m+1}
            continue
        r = scipy.optimize.brentq(_lp_eigenvalue_equation, u[i], u[i + 1], args=(m, V), xtol=1e-15, rtol=1e-15)
        if abs(_lp_eigenvalue_equation(r, m, V)) < 1e-6:
            roots.append(r)
    us = np.array(roots)
    return us, np.sqrt(V ** 2 - us ** 2)


def make_LP_modes(grid, V, core_radius):
    """hcipy.fiber.make_LP_modes: for m = 0,1,... while solutions exist: radial J_m(u r) inside the
    core, J_m(u)/K_m(w) K_m(w r) outside; azimuthal cos(m t) then (m>0) sin; each mode normalised
    numerically so sum(mode^2 w) = 1."""
    x, y = grid.x / core_radius, grid.y / core_radius
    R = np.hypot(x, y)
    T = np.arctan2(y, x)
    modes = []
    m = 0
    while True:
        us, ws = find_lp_solutions(m, V)
        if len(us) == 0:
            break
        for u, w in zip(us, ws):
            inside = R < 1
            radial = np.zeros_like(R)
            radial[inside] = jv(m, u * R[inside])
            radial[~inside] = jv(m, u) / kn(m, w) * kn(m, w * R[~inside])
            for mi in ([m, -m] if m > 0 else [m]):
                az = np.cos(mi * T) if mi >= 0 else np.sin(mi * T)
                prof = radial * az
                prof = prof / np.sqrt(np.sum(prof ** 2 * grid.weights))
                modes.append(prof)
        m += 1
    return modes


class StepIndexFiber:
    """hcipy.fiber.StepIndexFiber(core_radius, NA, length) (AO_env.py:393,471).  forward():
    c_k = sum(mode_k E w); output = sum_k c_k exp(i beta_k L) mode_k."""

    def __init__(self, core_radius, NA, fiber_length):
        self.core_radius = core_radius
        self.NA = NA
        self.fiber_length = fiber_length
        self._cache = {}

    def V(self, wavelength):
        return 2 * np.pi / wavelength * self.core_radius * self.NA

    def modes_for(self, grid, wavelength):
        key = (tuple(grid.delta), tuple(grid.dims), wavelength)
        if key not in self._cache:
            modes = make_LP_modes(grid, self.V(wavelength), self.core_radius)
            modes = [m / np.sqrt(np.sum(np.abs(m) ** 2 * grid.weights)) for m in modes]
            self._cache[key] = np.stack(modes, axis=-1)
        return self._cache[key]

    def forward(self, wf):
        M = self.modes_for(wf.grid, wf.wavelength)
        coeffs = M.T.dot(wf.electric_field * wf.grid.weights)
        # the propagation phases exp(i beta_k L) are unit-modulus; with mutually orthogonal modes
        # they drop out of total_power, which is all the reference consumes (AO_env.py:474).
        return Wavefront(M.dot(coeffs), wf.wavelength, wf.grid)


# --------------------------------------------------------------------------------------
# Metrics
# --------------------------------------------------------------------------------------
def get_strehl_from_focal(img, ref_img):
    """hcipy.metrics.get_strehl_from_focal (AO_env.py:482)."""
    return img[np.argmax(ref_img)] / ref_img.max()


def structural_similarity_1d(im1, im2, data_range, win_size=7, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity (0.22) on 1-D float input with default arguments
    (AO_env.py:495): uniform 7-window means, sample covariance (norm 7/6), mean of S over the
    interior [3:-3] (so the filter's boundary mode never matters)."""
    im1 = np.asarray(im1, dtype=float)
    im2 = np.asarray(im2, dtype=float)
    if im1.shape[0] < win_size:
        raise ValueError("win_size exceeds image extent.")
    n = im1.shape[0]
    pad = (win_size - 1) // 2
    cov_norm = win_size / (win_size - 1)
    C1 = (K1 * data_range) ** 2
    C2 = (K2 * data_range) ** 2
    S = []
    for i in range(pad, n - pad):
        a = im1[i - pad:i + pad + 1]
        b = im2[i - pad:i + pad + 1]
        ux, uy = a.mean(), b.mean()
        uxx, uyy, uxy = (a * a).mean(), (b * b).mean(), (a * b).mean()
        vx = cov_norm * (uxx - ux * ux)
        vy = cov_norm * (uyy - uy * uy)
        vxy = cov_norm * (uxy - ux * uy)
        S.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2)))
    return float(np.mean(S))


# --------------------------------------------------------------------------------------
# Shack-Hartmann chain (hcipy.optics.Magnifier, hcipy.wavefront_sensing.shack_hartmann, hcipy.propagation.FresnelPropagator,
# hcipy.optics.NoiselessDetector, hcipy.util.large_poisson, hcipy.math_util.inverse_tikhonov) — AO_env.py:254-290, 396-465.
# Confidence in these restatements is lower than for the Fraunhofer path (tags L-M).
# --------------------------------------------------------------------------------------
class Magnifier:
    """Rescales the grid by ``magnification`` and the field by 1/magnification, conserving total power (AO_env.py:404)."""

    def __init__(self, magnification):
        self.magnification = float(magnification)

    def forward(self, wf):
        out = Wavefront(wf.electric_field / self.magnification, wf.wavelength, wf.grid.scaled(self.magnification))
        return out

    __call__ = forward


class MicroLensArray:
    """hcipy MicroLensArray with lenslet_shape=None: each pixel belongs to the closest lenslet centre; sag
    -(d^2)/(2 f) applied as a SurfaceApodizer of refractive index 2 (opd = sag)."""

    def __init__(self, input_grid, lenslet_centres, focal_length):
        xs, ys = input_grid.x, input_grid.y
        cx, cy = lenslet_centres
        # separable nearest centre (the lenslet grid is a regular separable grid)
        ix = np.argmin(np.abs(xs[:, None] - cx[None, :]), axis=1)
        iy = np.argmin(np.abs(ys[:, None] - cy[None, :]), axis=1)
        self.mla_index = iy * len(cx) + ix
        d2 = (xs - cx[ix]) ** 2 + (ys - cy[iy]) ** 2
        self.mla_opd = (-1.0 / (2 * focal_length)) * d2

    def forward(self, wf):
        out = wf.copy()
        out.electric_field *= np.exp(1j * self.mla_opd * wf.wavenumber)
        return out


class FresnelPropagator:
    """hcipy FresnelPropagator(grid, distance, num_oversampling=2) = FourierFilter(grid, transfer_function, q): FFT with 2x zero
    padding, multiply by the transfer function, inverse FFT, crop.  hcipy picks the transfer function by a sampling test
    (propagation/fresnel.py, get_instance_data):

    * ``delta >= lambda z / L_max`` (the reference's geometry: 20.8 um >= 6.25 um): the analytic paraxial transfer function
      exp(-i z |k|^2 / (2 k)) exp(i k z) on the Fourier grid;
    * otherwise (pupils of more than ~800 pixels at the reference's f-number, or longer lenslet focal lengths) the IMPULSE-RESPONSE
      method: h(r) = exp(i k z) exp(i k r^2 / (2 z)) / (i lambda z) sampled on the enlarged spatial grid that is dual to the Fourier
      grid (make_fft_grid(fourier_grid): the q-times padded grid, pitch delta, a sample exactly at r = 0) and carried to the
      Fourier grid by FastFourierTransform.forward, i.e. sum_r h(r) exp(-i k.r) delta^2 with the grid's true coordinates.
    """

    def __init__(self, input_grid, distance, num_oversampling=2):
        self.grid = input_grid
        self.distance = float(distance)
        self.q = int(num_oversampling)
        self._tf = {}

    def uses_impulse_response(self, wavelength):
        g = self.grid
        return bool(np.any(g.delta < wavelength * self.distance / np.max(g.dims * g.delta)))

    def transfer_function(self, wavelength):
        """On the UNSHIFTED (q ny, q nx) FFT grid (numpy.fft frequency order)."""
        g = self.grid
        ny, nx = g.shape
        my, mx = ny * self.q, nx * self.q
        k = 2 * np.pi / wavelength
        if self.uses_impulse_response(wavelength):
            # enlarged grid: x_i = (i - m/2) delta for even m (hcipy make_fft_grid: zero = delta (-dims/2 + (dims mod 2)/2))
            xs = (np.arange(mx) - mx / 2 + (mx % 2) * 0.5) * g.delta[0]
            ys = (np.arange(my) - my / 2 + (my % 2) * 0.5) * g.delta[1]
            r2 = xs[None, :] ** 2 + ys[:, None] ** 2
            h = np.exp(1j * k * self.distance) * np.exp(1j * k * r2 / (2 * self.distance)) / (1j * self.distance * wavelength)
            kx = 2 * np.pi * np.fft.fftfreq(mx, g.delta[0])
            ky = 2 * np.pi * np.fft.fftfreq(my, g.delta[1])
            # sum_r h(r) exp(-i k.r) delta^2 with r the true coordinates: an FFT of h plus the phase ramp of the grid origin
            ramp = np.exp(-1j * (kx[None, :] * xs[0] + ky[:, None] * ys[0]))
            return np.fft.fft2(h) * ramp * (g.delta[0] * g.delta[1])
        kx = 2 * np.pi * np.fft.fftfreq(mx, g.delta[0])
        ky = 2 * np.pi * np.fft.fftfreq(my, g.delta[1])
        k2 = kx[None, :] ** 2 + ky[:, None] ** 2
        return np.exp(-0.5j * self.distance * k2 / k) * np.exp(1j * k * self.distance)

    def forward(self, wf):
        g = wf.grid
        ny, nx = g.shape
        my, mx = ny * self.q, nx * self.q
        lam = wf.wavelength
        if lam not in self._tf:
            self._tf[lam] = self.transfer_function(lam)
        pad = np.zeros((my, mx), dtype=complex)
        y0, x0 = my // 2 - ny // 2, mx // 2 - nx // 2
        pad[y0:y0 + ny, x0:x0 + nx] = wf.electric_field.reshape(ny, nx)
        out = np.fft.ifft2(np.fft.fft2(pad) * self._tf[lam])[y0:y0 + ny, x0:x0 + nx]
        return Wavefront(out.ravel(), lam, g)

    __call__ = forward


class SquareShackHartmannWavefrontSensorOptics:
    """hcipy SquareShackHartmannWavefrontSensorOptics(input_grid, f_number, num_lenslets, pupil_diameter) (AO_env.py:407)."""

    def __init__(self, input_grid, f_number, num_lenslets, pupil_diameter):
        lenslet_diameter = float(pupil_diameter) / num_lenslets
        x = np.arange(-pupil_diameter, pupil_diameter, lenslet_diameter)
        self.mla_centres = (x, x)
        self.num_lenslet_axis = len(x)
        gx, gy = np.meshgrid(x, x)
        self.mla_points = np.stack([gx.ravel(), gy.ravel()], axis=1)  # [n_lenslets, 2] (x, y), x fastest
        focal_length = f_number * lenslet_diameter
        self.micro_lens_array = MicroLensArray(input_grid, self.mla_centres, focal_length)
        self.propagator = FresnelPropagator(input_grid, focal_length)

    def forward(self, wf):
        return self.propagator(self.micro_lens_array.forward(wf))

    __call__ = forward


class NoiselessDetector:
    def __init__(self, detector_grid):
        self.detector_grid = detector_grid
        self.accumulated = 0.0

    def integrate(self, wf, dt, weight=1):
        self.accumulated = self.accumulated + wf.power * dt * weight

    def read_out(self):
        out = np.array(self.accumulated, dtype=float)
        self.accumulated = 0.0
        return out


def large_poisson(lam, thresh=1e6, rng=np.random):
    """hcipy.util.large_poisson: normal approximation above ``thresh`` (draws normal(size=n_large) first), exact Poisson below."""
    lam = np.asarray(lam, dtype=float)
    large = lam > thresh
    small = ~large
    n = np.zeros(lam.shape)
    n[large] = np.round(lam[large] + rng.normal(size=np.sum(large)) * np.sqrt(lam[large]))
    n[small] = rng.poisson(lam[small], size=np.sum(small))
    return n


class ShackHartmannWavefrontSensorEstimator:
    """Centre of gravity per lenslet (scipy.ndimage sums over the lenslet labels) minus the lenslet position
    (hcipy ShackHartmannWavefrontSensorEstimator.estimate).  Coordinates are those of the IMAGE's grid."""

    def __init__(self, mla_points, mla_index, image_grid, estimation_subapertures=None):
        self.mla_points = mla_points
        self.mla_index = mla_index
        self.image_grid = image_grid
        if estimation_subapertures is None:
            self.estimation_subapertures = np.unique(mla_index)
        else:
            self.estimation_subapertures = np.flatnonzero(np.asarray(estimation_subapertures))

    def estimate(self, images):
        import scipy.ndimage as ndimage

        image = np.asarray(images[0], dtype=float)
        sub = self.estimation_subapertures
        fluxes = ndimage.sum(image, self.mla_index, sub)
        sum_x = ndimage.sum(image * self.image_grid.x, self.mla_index, sub)
        sum_y = ndimage.sum(image * self.image_grid.y, self.mla_index, sub)
        centroids = np.array((sum_x / fluxes, sum_y / fluxes)) - self.mla_points[sub, :].T
        return centroids  # [2, n_sub]


def inverse_tikhonov(M, rcond=1e-15):
    """hcipy.math_util.inverse_tikhonov: V diag(s / (s^2 + (rcond s_max)^2)) U^T."""
    U, S, Vt = np.linalg.svd(M, full_matrices=False)
    S_inv = S / (S ** 2 + (rcond * S.max()) ** 2)
    return (Vt.T * S_inv).dot(U.T)
