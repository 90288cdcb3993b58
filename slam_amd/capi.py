"""ctypes binding of include/slam_hip.h (libslam_hip.so, built in-tree by slam_amd/csrc/Makefile).

There is no CPU fallback: if the library is missing this module raises, and every call that fails
raises SlamHipError with the library's own error text.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libslam_hip.so")

OK, EINVAL, EHIP, ENOMEM, ECAP = 0, -1, -2, -3, -4
MAX_LEVELS = 8


class SlamHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("slam_hip error %d: %s" % (code, text))
        self.code = code


class LevelDesc(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32),
                ("pitch", C.c_size_t), ("frame_stride", C.c_size_t), ("sub", C.c_float)]


class MatchJob(C.Structure):
    _fields_ = [("frame", C.c_int32), ("level", C.c_int32), ("pw", C.c_int32), ("ph", C.c_int32),
                ("patch_off", C.c_int64), ("pmask_off", C.c_int64),
                ("ax", C.c_int32), ("ay", C.c_int32), ("aw", C.c_int32), ("ah", C.c_int32),
                ("amask_off", C.c_int64), ("stride", C.c_int32), ("use_exact_area_mask", C.c_int32),
                ("score_off", C.c_int64)]


class MatchResult(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("response", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("status", C.c_int32)]


class TrackJob(C.Structure):
    _fields_ = [("frame", C.c_int32), ("projw", C.c_int32), ("projh", C.c_int32), ("flags", C.c_int32),
                ("proj_off", C.c_int64), ("mask_off", C.c_int64),
                ("px", C.c_float), ("py", C.c_float), ("threshold", C.c_float), ("stride", C.c_int32),
                ("amask_off", C.c_int64), ("ax", C.c_int32), ("ay", C.c_int32), ("aw", C.c_int32), ("ah", C.c_int32)]


TRACK_COARSE_AREA = 1


class TrackResult(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("response", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("levels_run", C.c_int32), ("status", C.c_int32)]


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("response", C.c_float)]


class DescJob(C.Structure):
    _fields_ = [("frame", C.c_int32), ("px", C.c_float), ("py", C.c_float), ("rx", C.c_int32), ("ry", C.c_int32),
                ("reserved", C.c_int32), ("out_off", C.c_int64)]


class DescResult(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("status", C.c_int32)]


class WarpJob(C.Structure):
    _fields_ = [("desc_off", C.c_int64), ("proj_off", C.c_int64), ("mask_off", C.c_int64), ("dw", C.c_int32), ("dh", C.c_int32),
                ("ow", C.c_int32), ("oh", C.c_int32), ("m", C.c_float * 9), ("reserved", C.c_int32)]


class DiscJob(C.Structure):
    _fields_ = [("mask_off", C.c_int64), ("w", C.c_int32), ("h", C.c_int32), ("circle_begin", C.c_int32), ("circle_count", C.c_int32)]


class LineJob(C.Structure):
    _fields_ = [("desc_off", C.c_int64), ("amask_off", C.c_int64), ("frame", C.c_int32), ("dw", C.c_int32), ("dh", C.c_int32),
                ("ax", C.c_int32), ("ay", C.c_int32), ("aw", C.c_int32), ("ah", C.c_int32), ("threshold", C.c_float),
                ("stride", C.c_int32), ("reserved", C.c_int32)]


class EllipseJob(C.Structure):
    _fields_ = [("mask_off", C.c_int64), ("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("ux", C.c_float), ("uy", C.c_float), ("q11", C.c_float), ("q12", C.c_float), ("q22", C.c_float),
                ("valid", C.c_int32)]


ELLIPSE_MAX = 300

# every symbol include/slam_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "slam_hip_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "slam_hip_destroy": (C.c_int, [_P]),
    "slam_hip_set_stream": (C.c_int, [_P, _P]),
    "slam_hip_synchronize": (C.c_int, [_P]),
    "slam_hip_set_match_lds": (C.c_int, [_P, C.c_int]),
    "slam_hip_set_walk_lds": (C.c_int, [_P, C.c_int]),
    "slam_hip_set_walk_chunks": (C.c_int, [_P, C.c_int]),
    "slam_hip_last_error_string": (C.c_char_p, [_P]),
    "slam_hip_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "slam_hip_free": (C.c_int, [_P, _P]),
    "slam_hip_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "slam_hip_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "slam_hip_pyramid_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(_P)]),
    "slam_hip_pyramid_destroy": (C.c_int, [_P, _P]),
    "slam_hip_pyramid_build_d": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t]),
    "slam_hip_pyramid_build": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t]),
    "slam_hip_pyramid_level": (C.c_int, [_P, C.c_int, C.POINTER(LevelDesc)]),
    "slam_hip_pyramid_download": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_size_t]),
    "slam_hip_match_batch": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_match_batch_d": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "slam_hip_match_feature_in_area": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t,
                                                 _P, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t,
                                                 C.c_int, C.c_int, _P, C.POINTER(MatchResult)]),
    "slam_hip_track_batch": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_track_batch_d": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_track_batch_noalloc_d": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_fast_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "slam_hip_fast_detect_d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                         C.c_int, _P, C.c_int, _P, _P, C.c_size_t]),
    "slam_hip_fast_detect": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_int, _P, C.c_int, _P]),
    "slam_hip_get_sorted_keypoints": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                C.POINTER(C.c_float), _P, C.c_int, C.POINTER(C.c_int32)]),
    "slam_hip_sort_keypoints_d": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "slam_hip_select_new_points_d": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, C.c_int,
                                               C.c_int, C.c_int, _P, _P]),
    "slam_hip_extract_descriptors_d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, _P, C.c_int,
                                                 _P, C.c_size_t, _P]),
    "slam_hip_add_new_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                          C.c_float, C.c_int, C.c_int, _P, C.POINTER(C.c_int32), _P, _P, C.c_size_t, _P]),
    "slam_hip_project_setup": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.POINTER(C.c_int32)]),
    "slam_hip_warp_batch_d": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int, _P, C.c_size_t, _P, C.c_int]),
    "slam_hip_project": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, _P, _P, _P, _P, _P, C.c_size_t, _P, C.POINTER(C.c_int32)]),
    "slam_hip_project_track_batch": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_disc_masks_d": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, C.c_size_t, _P, C.c_int]),
    "slam_hip_line_track_batch_d": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_line_track_batch": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_match_lines_batch": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "slam_hip_ellipse_setup": (C.c_int, [C.c_float, C.c_float, _P, C.c_float, C.c_float, C.POINTER(EllipseJob)]),
    "slam_hip_ellipse_masks_d": (C.c_int, [_P, _P, C.c_int, _P, C.c_size_t, _P, C.c_int]),
    "slam_hip_track_ellipse_batch": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "slam_hip_debug_score_formulas": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P]),
    "slam_hip_debug_resize": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_size_t, _P, C.c_int, C.c_int, C.c_size_t]),
    "slam_hip_debug_walk_deferred": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "slam_hip_debug_walk_plan": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_int32)]),
    "slam_hip_version": (C.c_char_p, []),
}

_lib = None


def build(force=False):
    """Compile libslam_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    """Load the library and bind every declared symbol (no GPU is touched by loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libslam_hip.so is not built (%s); run slam_amd.capi.build() or "
                              "`make -C slam_amd/csrc` — there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)       # AttributeError if the export is missing
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(ctx, rc, allow=()):
    if rc != OK and rc not in allow:
        txt = lib().slam_hip_last_error_string(ctx)
        raise SlamHipError(rc, txt.decode() if txt else "")
    return rc
