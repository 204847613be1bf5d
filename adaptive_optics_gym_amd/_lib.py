"""ctypes binding of ``libaogym.so`` (the C-ABI declared in ``include/aogym.h``).

There is no CPU fallback: if the shared library is missing the import of the env classes still works (so
CPU-only tooling can inspect them) but the first call raises ``RuntimeError`` naming the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaogym.so")
ABI_VERSION = 19

AOG_REWARD = {"strehl_ratio": 0, "smf_ssim": 1}
AOG_PRECISION = {"fast": 0, "fp64": 1}
AOG_KERNEL = {"auto": 0, "valu": 1, "mfma": 2}
AOG_SCREENS = {"twoband": 0, "hcipy16": 1}
AOG_EXTRUDE = {"auto": 0, "f64": 1}
AOG_PROF = {"fused": 0, "screen_rows": 1, "screen_cols": 2, "pack": 3, "extrude": 4, "sh_field": 5, "sh_rows_fwd": 6, "sh_cols": 7, "sh_rows_inv": 8}


class AogConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "num_envs", "n_pupil", "n_modes", "obs_dim", "n_ap", "n_wfs_tables", "n_sci_tables",
        "n_fiber_modes", "reward_type", "sh_operation", "max_steps", "flat_mirror_start", "has_rew_threshold",
        "precision", "kernel", "pixel_chunks", "atm_dynamic", "env_id_base", "reserved0")] + [(n, C.c_double) for n in (
        "wavelength_wfs", "wavelength_sci", "surface_rms_target", "rew_threshold", "ssim_ref_peak", "ssim_alpha")]


class AogTables(C.Structure):
    _fields_ = [("ap_index", C.POINTER(C.c_int32))] + [(n, C.POINTER(C.c_double)) for n in (
        "modes", "gram", "wfs_tables", "sci_tables", "wfs_coef", "sci_coef", "focal_m1", "focal_m2")] + [("n_focal", C.c_int32)]


class AogLayerTables(C.Structure):
    _fields_ = [("nz_vertical", C.c_int32), ("nz_horizontal", C.c_int32),
                ("stencil_vertical", C.POINTER(C.c_int32)), ("stencil_horizontal", C.POINTER(C.c_int32)),
                ("A_vertical", C.POINTER(C.c_double)), ("B_vertical", C.POINTER(C.c_double)),
                ("A_horizontal", C.POINTER(C.c_double)), ("B_horizontal", C.POINTER(C.c_double)),
                ("sqrt_cn_squared", C.c_double), ("pixel_pitch", C.c_double), ("delta_t", C.c_double)]


class AogLayerComposite(C.Structure):
    _fields_ = [("axis", C.c_int32), ("k_max", C.c_int32), ("n_old", C.c_int32), ("reserved0", C.c_int32),
                ("old_yx", C.POINTER(C.c_int32)), ("A", C.POINTER(C.c_double)), ("B", C.POINTER(C.c_double))]


class AogShTables(C.Structure):
    _fields_ = [("n_sub", C.c_int32), ("sub_slot", C.POINTER(C.c_int32))] + [(n, C.POINTER(C.c_double)) for n in (
        "centres", "slopes_ref", "reconstruction", "mla_phase", "transfer", "x_det")] + [(n, C.c_double) for n in (
        "field_amplitude", "image_scale", "gain", "leakage")] + [("fft_double", C.c_int32), ("reserved0", C.c_int32)]


class AogActor(C.Structure):  # mirrors aog_actor in include/aogym.h
    _fields_ = [("batch", C.c_int32), ("state_dim", C.c_int32), ("hidden_dim", C.c_int32), ("act_dim", C.c_int32),
                ("env_id_base", C.c_int32), ("reserved0", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("wo", C.c_void_p), ("bo", C.c_void_p), ("dropout_p", C.c_float), ("cov_var", C.c_float),
                ("seed", C.c_uint64), ("call_index", C.c_uint64)]


class AogInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "num_envs", "num_envs_padded", "n_ap", "n_ap_padded", "n_modes_padded", "pixel_chunks",
        "kernel", "n_sums", "reserved")] + [("device_bytes", C.c_int64)]


# every symbol include/aogym.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "aog_abi_version": (C.c_int, []),
    "aog_last_error": (C.c_char_p, []),
    "aog_build_id": (C.c_char_p, []),
    "aog_struct_size": (C.c_int64, [C.c_int]),
    "aog_create": (C.c_int, [C.POINTER(AogConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "aog_destroy": (None, [C.c_void_p]),
    "aog_get_info": (C.c_int, [C.c_void_p, C.POINTER(AogInfo)]),
    "aog_upload_tables": (C.c_int, [C.c_void_p, C.POINTER(AogTables)]),
    "aog_set_screens_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aog_set_screens_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aog_upload_layer": (C.c_int, [C.c_void_p, C.POINTER(AogLayerTables)]),
    "aog_upload_layer_composite": (C.c_int, [C.c_void_p, C.POINTER(AogLayerComposite)]),
    "aog_set_extrusion_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "aog_set_wind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "aog_set_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "aog_set_extrusion_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "aog_set_rng_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "aog_get_screens_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aog_generate_screens": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "aog_set_screen_method": (C.c_int, [C.c_void_p, C.c_int]),
    "aog_upload_sh": (C.c_int, [C.c_void_p, C.POINTER(AogShTables)]),
    "aog_sh_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_sh_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_state_bytes": (C.c_int64, [C.c_void_p]),
    "aog_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "aog_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "aog_get_phase_screen": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aog_actor_act": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_device_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "aog_set_return_accumulator": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aog_get_actuators": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_set_actuators": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aog_step": (C.c_int, [C.c_void_p] * 9),
    "aog_step_pipelined": (C.c_int, [C.c_void_p] * 10),
    "aog_focal_image": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aog_focal_images": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aog_selftest_poisson": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p]),
    "aog_selftest_barrier_timeout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aog_selftest_sincos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aog_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "aog_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "aog_profile_block": (C.c_int, [C.c_void_p, C.c_int]),
    "aog_profile_read_kernel": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or adaptive_optics_gym_amd/build.py) — there is no CPU fallback for the device path.")
    try:  # let torch's bundled ROCm libraries (HIP runtime, hipFFT/rocFFT; same SONAMEs) win if torch is going to be used
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.aog_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libaogym.so ABI {lib.aog_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    # the binary must come from the sources beside it (build.py::source_id; developer builds carry "+FLAG" and are accepted on their base id)
    if os.path.isdir(os.path.join(_HERE, "csrc")) and os.environ.get("AOG_SKIP_BUILD_ID_CHECK") != "1":
        from .build import source_id

        have, want = lib.aog_build_id().decode(), source_id()
        if have.split("+")[0] != want:
            raise RuntimeError(f"libaogym.so was built from other sources (build id {have}, sources {want}): run "
                               "`python -m adaptive_optics_gym_amd.build` (or __graft_entry__.build())")
    for which, cls in enumerate((AogConfig, AogTables, AogLayerTables, AogShTables, AogActor, AogInfo, AogLayerComposite)):
        if lib.aog_struct_size(which) != C.sizeof(cls):
            raise RuntimeError(f"{cls.__name__}: ctypes layout is {C.sizeof(cls)} bytes, the library's struct {lib.aog_struct_size(which)}")
    _lib = lib
    return lib


class AogError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        msg = load().aog_last_error().decode("utf-8", "replace")
        if "win_size exceeds image extent" in msg:
            raise ValueError(msg)  # what skimage raises in the reference for smf_ssim with obs_dim=2 (AO_env.py:495)
        raise AogError(f"libaogym error {rc}: {msg}")
