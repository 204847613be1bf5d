"""Host-side (numpy float64) one-time precomputation of the constant tables the device path consumes —
the product counterpart of what ``AOEnv.__init__`` obtains through hcipy (AO_env.py:42-68, helpers
AO_env.py:293-393).  Runs once per environment construction; nothing here is on the step() path.

The whole non-Shack-Hartmann ``step()`` collapses algebraically to: per aperture pixel p
and env, a phase  phi_p = (psi_p + 4 pi (M a)_p) / lambda, and K fixed complex pupil-plane dot products
``Z_j = sum_p exp(i phi_p) G_j(p)``.  This module builds M (modes), its centred Gram matrix, and the
G_j, expressed over *real* tables g_m with a small complex coefficient matrix (``Z_j = sum_m C_jm (U_m + i
V_m)``, ``U_m = sum_p cos(phi_p) g_m(p)``, ``V_m = sum_p sin(phi_p) g_m(p)``) so the device only does real
FMAs.

Field layout follows hcipy: flat index = iy*N + ix, x fastest.
"""
from __future__ import annotations

import heapq
from dataclasses import dataclass, field

import numpy as np
from scipy import optimize, special

from .params import OpticalParams


# ------------------------------------------------------------------------------------------------
# grids / aperture (hcipy make_pupil_grid, make_focal_grid, make_circular_aperture; AO_env.py:300-301,314)
# ------------------------------------------------------------------------------------------------
def centred_axis(n: int, extent: float) -> np.ndarray:
    """make_pupil_grid axis: n samples of pitch extent/n, symmetric about 0 (no sample at 0 for even n)."""
    d = extent / n
    return -extent / 2 + d / 2 + d * np.arange(n)


def focal_axis(q: int, num_airy: int, resolution: float) -> np.ndarray:
    """make_focal_grid axis: pitch resolution/q, 2*q*num_airy samples, one exactly on the axis."""
    n = int(2 * q * num_airy)
    d = resolution / q
    return d * (np.arange(n) - n / 2 + (n % 2) * 0.5)


def aperture_mask(n: int, diameter: float) -> np.ndarray:
    x = centred_axis(n, diameter)
    return (x[None, :] ** 2 + x[:, None] ** 2) <= (diameter / 2) ** 2


# ------------------------------------------------------------------------------------------------
# deformable-mirror mode bases (AO_env.py:346-347, 352-353)
# ------------------------------------------------------------------------------------------------
def noll_indices(j: int):
    """Noll index j (1 = piston) -> (n, signed m); even j carries cos (m > 0), odd j sin (m < 0)."""
    n = int(np.sqrt(2 * j - 1) + 0.5) - 1
    if n % 2:
        m = 2 * ((2 * (j + 1) - n * (n + 1)) // 4) - 1
    else:
        m = 2 * ((2 * j + 1 - n * (n + 1)) // 4)
    return n, (m if j % 2 == 0 else -m)


def zernike_on(rho: np.ndarray, theta: np.ndarray, n: int, m: int) -> np.ndarray:
    """sqrt(n+1) R_n^|m|(rho) {sqrt2 cos | sqrt2 sin | 1}; radial part through the Jacobi polynomial identity
    R_n^m(rho) = (-1)^((n-m)/2) rho^m P^{(m,0)}_{(n-m)/2}(1 - 2 rho^2)."""
    am = abs(m)
    k = (n - am) // 2
    radial = (-1) ** k * rho ** am * special.eval_jacobi(k, am, 0, 1 - 2 * rho ** 2)
    if m > 0:
        az = np.sqrt(2.0) * np.cos(am * theta)
    elif m < 0:
        az = np.sqrt(2.0) * np.sin(am * theta)
    else:
        az = 1.0
    return np.sqrt(n + 1.0) * radial * az


def disk_harmonic_orders(num_modes: int):
    """Order of hcipy's ``make_disk_harmonic_basis(..., 'neumann')``: a frontier search seeded with (n, m) =
    (1, 0); the frontier entry with the smallest Bessel-derivative zero is emitted — as (n, -m) then (n, m)
    when m != 0 — and its two successors (n, m+1), (n+1, m) join the frontier unless already seen.  The seed
    (1, 0) (zero 3.83) therefore precedes (1, +-1) (zero 1.84): the list is not globally sorted.
    "Actuator indexing bit-exact" hinges on this table."""
    def energy(n, m):
        return float(special.jnp_zeros(m, n)[-1]) ** 2

    emitted, seen, count = [], {(1, 0)}, 0
    frontier = [(energy(1, 0), count, (1, 0))]  # the insertion counter reproduces argmin's first-wins tie rule
    while len(emitted) < num_modes:
        _, _, (n, m) = heapq.heappop(frontier)
        if m != 0:
            emitted.append((n, -m))
        emitted.append((n, m))
        for succ in ((n, m + 1), (n + 1, m)):
            if succ not in seen:
                seen.add(succ)
                count += 1
                heapq.heappush(frontier, (energy(*succ), count, succ))
    return emitted[:num_modes]


def disk_harmonic_on(rho: np.ndarray, theta: np.ndarray, n: int, m: int) -> np.ndarray:
    am = abs(m)
    lam = float(special.jnp_zeros(am, n)[-1])
    return special.jv(am, lam * rho) * (np.sin(am * theta) if m < 0 else np.cos(am * theta))


def mode_matrix(act_type: str, num_modes: int, x_ap: np.ndarray, y_ap: np.ndarray, diameter: float, has_outside: bool):
    """[n_ap, A] influence matrix restricted to the aperture, each mode divided by its peak-to-peak over the
    WHOLE grid (np.ptp includes the zeros outside the aperture; AO_env.py:347,353)."""
    rho = 2 * np.hypot(x_ap, y_ap) / diameter
    theta = np.arctan2(y_ap, x_ap)
    cols = []
    if act_type == "zernike":
        specs = [noll_indices(j) for j in range(1, num_modes + 1)]
        cols = [zernike_on(rho, theta, n, m) for n, m in specs]
    else:
        specs = disk_harmonic_orders(num_modes)
        cols = [disk_harmonic_on(rho, theta, n, m) for n, m in specs]
    out = np.empty((len(x_ap), num_modes))
    for i, c in enumerate(cols):
        hi, lo = c.max(), c.min()
        if has_outside:
            hi, lo = max(hi, 0.0), min(lo, 0.0)
        out[:, i] = c / (hi - lo)
    return out, specs


def centred_gram(modes: np.ndarray, n_total: int) -> np.ndarray:
    """G with  np.std(M a over all n_total grid pixels)^2 == a^T G a  (modes vanish outside the aperture)."""
    mean = modes.sum(axis=0) / n_total
    c = modes - mean
    return (c.T @ c + (n_total - modes.shape[0]) * np.outer(mean, mean)) / n_total


# ------------------------------------------------------------------------------------------------
# step-index fiber LP modes (hcipy StepIndexFiber / make_LP_modes; AO_env.py:393)
# ------------------------------------------------------------------------------------------------
def lp_roots(m: int, V: float):
    """Guided-mode roots u in (0, V) of the LP characteristic equation, written pole-free:
    J_m(u) w K_{m+1}(w) - K_m(w) u J_{m+1}(u) = 0,  w = sqrt(V^2 - u^2)."""
    def g(u):
        w = np.sqrt(V * V - u * u)
        return special.jv(m, u) * w * special.kn(m + 1, w) - special.kn(m, w) * u * special.jv(m + 1, u)

    us = np.linspace(0, V, 4001)[1:-1]
    vals = g(us)
    roots = []
    for i in np.flatnonzero(np.sign(vals[:-1]) * np.sign(vals[1:]) < 0):
        roots.append(optimize.brentq(g, us[i], us[i + 1], xtol=1e-15, rtol=1e-15))
    return roots


def lp_modes(axis: np.ndarray, core_radius: float, V: float):
    """All guided LP modes on the separable focal grid ``axis x axis``, each normalised to sum(mode^2 dA) = 1.
    Order: m = 0, 1, ...; per radial root; cos then sin azimuth.  Returns [n_modes, n, n] (y, x)."""
    X, Y = np.meshgrid(axis / core_radius, axis / core_radius)
    R = np.hypot(X, Y)
    T = np.arctan2(Y, X)
    dA = (axis[1] - axis[0]) ** 2
    out = []
    m = 0
    while True:
        roots = lp_roots(m, V)
        if not roots:
            break
        for u in roots:
            w = np.sqrt(V * V - u * u)
            core = R < 1
            radial = np.where(core, special.jv(m, u * R), special.jv(m, u) / special.kn(m, w) * special.kn(m, w * np.where(core, 1.0, R)))
            for az in ([np.cos(m * T), np.sin(-m * T)] if m > 0 else [np.ones_like(T)]):
                prof = radial * az
                out.append(prof / np.sqrt(np.sum(prof ** 2) * dA))
        m += 1
    return np.stack(out)


# ------------------------------------------------------------------------------------------------
# the table set
# ------------------------------------------------------------------------------------------------
@dataclass
class HostTables:
    params: OpticalParams
    act_type: str
    act_dim: int
    obs_dim: int
    n_ap: int
    ap_index: np.ndarray            # [n_ap] int32
    x_ap: np.ndarray                # [n_ap] pupil coordinates of the packed pixels
    y_ap: np.ndarray
    modes: np.ndarray               # [n_ap, A]
    mode_specs: list                # (n, m) per mode — the "actuator indexing" table
    gram: np.ndarray                # [A, A]
    wfs_tables: np.ndarray          # [MRW, n_ap]
    wfs_coef: np.ndarray            # [o^2 + n_fiber, MRW] complex
    sci_tables: np.ndarray          # [MRS, n_ap]
    sci_coef: np.ndarray            # [1, MRS] complex
    n_fiber_modes: int
    lp_u: list = field(default_factory=list)
    focal_m1: np.ndarray = None      # [n_focal, N] complex (scale factors folded in)
    focal_m2: np.ndarray = None      # [N, n_focal] complex
    lp_modes: np.ndarray = None      # [n_fiber, n_focal, n_focal]
    focal_pixel_area: float = 0.0
    strehl_focal_index: int = 0


def _realify(kernels: np.ndarray, tol: float = 1e-12):
    """Complex kernels [K, n] -> (real tables [M, n], coef [K, M] complex) with kernels == coef @ tables.
    Real/imaginary parts that vanish are dropped; parts equal up to sign are shared."""
    tables, coef_rows = [], []
    scale = np.abs(kernels).max()
    for k in range(kernels.shape[0]):
        row = {}
        for part, unit in ((kernels[k].real, 1.0 + 0j), (kernels[k].imag, 1j)):
            if np.abs(part).max() <= tol * scale:
                continue
            for idx, t in enumerate(tables):
                if np.abs(part - t).max() <= tol * scale:
                    row[idx] = row.get(idx, 0) + unit
                    break
                if np.abs(part + t).max() <= tol * scale:
                    row[idx] = row.get(idx, 0) - unit
                    break
            else:
                tables.append(part.copy())
                row[len(tables) - 1] = unit
        coef_rows.append(row)
    coef = np.zeros((kernels.shape[0], len(tables)), dtype=complex)
    for k, row in enumerate(coef_rows):
        for idx, c in row.items():
            coef[k, idx] = c
    # normalise every table to max |g| = 1 (exactly representable peaks in fp32); the scale moves into coef
    tables = np.stack(tables)
    peak = np.abs(tables).max(axis=1)
    return tables / peak[:, None], coef * peak[None, :]


def build_tables(params: OpticalParams, act_type: str, act_dim: int, obs_dim: int) -> HostTables:
    N = params.num_pupil_pixels
    D = params.telescope_diameter
    ax = centred_axis(N, D)
    mask = aperture_mask(N, D)
    ap_index = np.flatnonzero(mask.ravel()).astype(np.int32)
    n_ap = int(ap_index.size)
    iy, ix = np.divmod(ap_index, N)
    x_ap, y_ap = ax[ix], ax[iy]
    pix_area = (D / N) ** 2

    modes, specs = mode_matrix(act_type, act_dim, x_ap, y_ap, D, has_outside=n_ap < N * N)
    gram = centred_gram(modes, N * N)

    # --- wavefront-sensing arm: Wavefront(aperture, lambda_wfs).total_power = 1 (AO_env.py:329-330) ---------
    amp = 1.0 / np.sqrt(n_ap * pix_area)
    lam, f = params.wavelength_wfs, params.fiber_focal_length
    kappa = 2 * np.pi / (lam * f)
    o = obs_dim
    # observation: FraunhoferPropagator onto make_pupil_grid(o, 52.5 um), then .power (AO_env.py:385,391,139,142)
    Xo = centred_axis(o, params.fiber_window)
    dXo = params.fiber_window / o
    obs_scale = amp * pix_area * dXo / (lam * f)
    obs_k = np.empty((o * o, n_ap), dtype=complex)
    for b in range(o):
        for a in range(o):
            obs_k[b * o + a] = obs_scale * np.exp(-1j * kappa * (Xo[a] * x_ap + Xo[b] * y_ap))
    # fiber: LP modes on make_pupil_grid(128, 52.5 um), folded back to the pupil ("receive modes"):
    #   c_k = sum_X m_k(X) E_f(X) dA_f,  E_f(X) = 1/(i lam f) sum_x E(x) dA exp(-i kappa X.x)
    nf = params.num_focal_pixels_fiber
    Xf = centred_axis(nf, params.fiber_window)
    dAf = (params.fiber_window / nf) ** 2
    V = 2 * np.pi / lam * params.singlemode_fiber_core_radius * params.fiber_NA
    lps = lp_modes(Xf, params.singlemode_fiber_core_radius, V)
    Ey = np.exp(-1j * kappa * np.outer(ax, Xf))      # [N(y), nf]
    Ex = np.exp(-1j * kappa * np.outer(Xf, ax))      # [nf, N(x)]
    fib_k = np.empty((lps.shape[0], n_ap), dtype=complex)
    for k in range(lps.shape[0]):
        full = Ey @ lps[k] @ Ex                      # [N, N] (y, x)
        fib_k[k] = amp * pix_area * dAf / (1j * lam * f) * full.ravel()[ap_index]
    wfs_tables, wfs_coef = _realify(np.concatenate([obs_k, fib_k], axis=0))

    # --- science arm: Strehl = img[argmax(ref)] / ref.max() (AO_env.py:479-483) -------------------------------
    lam_s = params.wavelength_sci
    Xs = focal_axis(params.focal_q, params.focal_num_airy, lam_s / D)
    My = np.exp(-1j * (2 * np.pi / lam_s) * np.outer(Xs, ax))
    ref = np.abs(My @ mask.astype(float) @ My.T) ** 2          # unaberrated PSF up to a constant
    kstar = int(np.argmax(ref.ravel()))
    ky, kx = divmod(kstar, Xs.size)
    sci_k = np.exp(-1j * (2 * np.pi / lam_s) * (Xs[kx] * x_ap + Xs[ky] * y_ap))[None, :]
    sci_k = sci_k / np.abs(sci_k.sum())                         # |sum_p exp(i phi_p) K_p|^2 is then the Strehl ratio
    sci_tables, sci_coef = _realify(sci_k)

    return HostTables(params=params, act_type=act_type, act_dim=act_dim, obs_dim=obs_dim, n_ap=n_ap, ap_index=ap_index,
                      x_ap=x_ap, y_ap=y_ap, modes=modes, mode_specs=specs, gram=gram, wfs_tables=wfs_tables,
                      wfs_coef=wfs_coef, sci_tables=sci_tables, sci_coef=sci_coef, n_fiber_modes=int(lps.shape[0]),
                      lp_u=[lp_roots(0, V), lp_roots(1, V)], strehl_focal_index=kstar,
                      focal_m1=amp * pix_area / (1j * lam * f) * np.exp(-1j * kappa * np.outer(Xf, ax)),
                      focal_m2=np.exp(-1j * kappa * np.outer(ax, Xf)), lp_modes=lps, focal_pixel_area=dAf)
