"""ctypes binding of libfc3d_hip.so (declared in include/fc3d.h).

There is no fallback: if the library is missing or a symbol is absent the
import of the operators fails with an explicit error."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# FC3D_LIB_PATH: another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("FC3D_LIB_PATH") or os.path.join(HERE, "libfc3d_hip.so")


class LayerGrid(ctypes.Structure):
    """fc3d_layer_grid"""
    _fields_ = [("coeff", c_float), ("inv_z0", c_float), ("step", c_float), ("centre0", c_float),
                ("n_layers", c_int32)]


ABI_VERSION = 212  # FC3D_VERSION of include/fc3d.h this binding was written against (a stale library is refused)

P = c_void_p      # device pointers travel as integers
I64 = c_int64

# name -> (restype, argtypes); mirrors include/fc3d.h one to one
SIGNATURES = {
    "fc3d_version": (c_int, []),
    "fc3d_status_string": (c_char_p, [c_int]),
    "fc3d_code_pattern_fwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_code_pattern_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_code_pattern_fwd": (c_int, [P, P, I64, I64, I64, P, P, c_int, c_int, c_int, c_int, c_int, c_int,
                                      LayerGrid, P, P, P, POINTER(c_int)]),
    "fc3d_code_pattern_shaded_fwd": (c_int, [P, P, I64, I64, I64, P, c_int, c_int, c_int, LayerGrid, P, P, P,
                                             POINTER(c_int)]),
    "fc3d_code_pattern_shaded_conv_fwd": (c_int, [P, P, I64, I64, I64, P, P, c_int, c_int, c_int, c_int, c_int, LayerGrid,
                                                  P, P, P, POINTER(c_int)]),
    "fc3d_code_pattern_norm": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "fc3d_code_pattern_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "fc3d_code_pattern_bwd_dot_supported": (c_int, [c_int, c_int, c_int]),
    "fc3d_code_pattern_bwd_dot": (c_int, [P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_bilinear_bwd_img_dot_count": (c_int, [c_int, c_int, c_int]),
    "fc3d_bilinear_bwd_img_dot": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "fc3d_pose_transforms": (c_int, [P, P, P, I64, c_int, P, P, P, P, c_size_t]),
    "fc3d_project_fwd": (c_int, [P, P, I64, I64, I64, I64, P, c_int, c_int, c_int, P]),
    "fc3d_project_bwd": (c_int, [P, P, P, I64, I64, I64, I64, P, c_int, c_int, c_int, P]),
    "fc3d_project_bwd_transform_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_project_bwd_transform": (c_int, [P, P, P, I64, I64, I64, I64, P, c_int, c_int, c_int, P, P]),
    "fc3d_bilinear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_bilinear_bwd_grid": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_bilinear_bwd_img_workspace": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "fc3d_bilinear_tap_index": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_bilinear_bwd_img": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "fc3d_visible_mask": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "fc3d_warp_p2c_fwd": (c_int, [P, P, P, I64, I64, I64, I64, P, I64, I64, I64, I64, P, P, c_int, c_int, c_int,
                                  P, P, P, P, c_int, P]),
    "fc3d_warp_p2c_pose_fwd": (c_int, [P, P, P, I64, I64, I64, I64, P, I64, I64, I64, I64, P, P, I64, c_int, c_int,
                                       c_int, P, P, P, P, c_int, c_int, P, P, P, P]),
    "fc3d_warp_p2c_pose_depth_fwd": (c_int, [P, P, P, P, P, P, P, P, I64, c_int, c_int, c_int, P, P, P, P, c_int, c_int,
                                             P, P, P, P]),
    "fc3d_warp_project_splat": (c_int, [P, P, I64, I64, I64, I64, P, P, I64, c_int, c_int, c_int, P, P, c_int, P, P, P]),
    "fc3d_warp_sample": (c_int, [P, P, P, I64, I64, I64, I64, P, c_int, c_int, c_int, P, c_int, P, P, P]),
    "fc3d_points_fwd": (c_int, [P, P, P, P, P, I64, c_int, c_int, c_int, P]),
    "fc3d_points_bwd": (c_int, [P, P, P, P, P, I64, c_int, c_int, c_int, P]),
    "fc3d_reproject_fwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "fc3d_reproject_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_reproj_loss_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_reproj_loss_fwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "fc3d_reproj_loss_fwd_static": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "fc3d_reproj_loss_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "fc3d_zp_to_zc_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "fc3d_zp_to_zc_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "fc3d_zp_to_zc_bwd_full_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_zp_to_zc_bwd_full": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "fc3d_points_bwd_rays": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "fc3d_reflectance_fwd": (c_int, [P, P, I64, P, c_int, c_int, c_int, P]),
    "fc3d_capture_warp_fwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, c_int, P, P]),
    "fc3d_capture_post_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_capture_post": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fc3d_capture_fwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "fc3d_capture_fwd": (c_int, [P, P, P, P, P, P, P, P, I64, P, P, P, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int,
                                 P, P, P]),
    "fc3d_box_filter": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "fc3d_profile_enable": (c_int, [c_int]),
    "fc3d_profile_report": (c_int, [c_char_p, c_size_t]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Loads the shared library once and types every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python depth-estimation_amd/fc3d/build.py` "
            "(needs hipcc); the fc3d operators have no fallback implementation")
    # torch must be loaded first: its wheel carries its own libamdhip64.so.7, and the dynamic
    # loader then resolves this library's dependency of the same soname to that copy.  Loaded
    # the other way round the process ends up with two HIP runtimes that do not share streams
    # or devices.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.fc3d_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports version {lib.fc3d_version()}, expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().fc3d_status_string(status).decode()
        raise RuntimeError(f"{what} failed with status {status}: {msg}")
