"""Every constant the reference environment fixes (``parameters_init``, AO_env.py:197-251), as a typed
dataclass instead of the reference's ``exec``-injected dict.  ``num_pupil_pixels`` is the one new knob:
the reference hard-codes 240 (AO_env.py:216); BASELINE.json's configs use 128/256/512."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class OpticalParams:
    telescope_diameter: float = 0.5                 # AO_env.py:213
    num_pupil_pixels: int = 240                     # AO_env.py:216
    wavelength_wfs: float = 1.5e-6                  # AO_env.py:219
    wavelength_sci: float = 2.2e-6                  # AO_env.py:220
    delta_t: float = 1e-3                           # AO_env.py:226
    outer_scale: float = 10.0                       # AO_env.py:230
    D_pupil_fiber: float = 0.5                      # AO_env.py:233
    num_pupil_pixels_fiber: int = 128               # AO_env.py:234 (effectively unused)
    num_focal_pixels_fiber: int = 128               # AO_env.py:235
    multimode_fiber_core_radius: float = 25 * 1e-6  # AO_env.py:237
    singlemode_fiber_core_radius: float = 4.5 * 1e-6  # AO_env.py:238
    fiber_NA: float = 0.14                          # AO_env.py:239
    fiber_length: float = 10.0                      # AO_env.py:240
    f_number: float = 50.0                          # AO_env.py:243
    num_lenslets: int = 12                          # AO_env.py:244
    sh_diameter: float = 5e-3                       # AO_env.py:245
    stellar_magnitude: float = -5.0                 # AO_env.py:246
    focal_q: int = 4                                # AO_env.py:314
    focal_num_airy: int = 30                        # AO_env.py:314
    action_rms_fraction: float = 0.1                # AO_env.py:120  (0.1 * wavelength_sci)
    ssim_ref_peak: float = 2.8                      # AO_env.py:492
    ssim_alpha: float = 0.8                         # AO_env.py:497

    @property
    def fiber_focal_length(self) -> float:          # AO_env.py:388
        return self.D_pupil_fiber / (2 * self.fiber_NA)

    @property
    def fiber_window(self) -> float:                # AO_env.py:381
        return 2.1 * self.multimode_fiber_core_radius

    @property
    def pupil_pixel(self) -> float:
        return self.telescope_diameter / self.num_pupil_pixels


def coerce_velocity(atm_type: str, velocity_value, verbose: bool = True):
    """AO_env.py:200-208 — same coercions, same printed messages."""
    if atm_type in ("quasi_static", "semi_dynamic") and velocity_value != 0:
        if verbose:
            print("In " + atm_type + " atmospheric condition, the velocity value should be zero.")
            print("therefore velocity value is changed to zero")
        velocity_value = 0
    elif atm_type == "dynamic" and velocity_value == 0:
        if verbose:
            print("In " + atm_type + " atmospheric condition, the velocity value cannot be zero.")
            print("therefore velocity value is changed to 1 m/s")
        velocity_value = 1
    return velocity_value
