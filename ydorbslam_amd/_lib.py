"""Loader for libydorb.so (the C-ABI product library).  Fails loudly; never substitutes a CPU path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class YdorbError(RuntimeError):
    pass


def library_path():
    # YDORB_LIB: an alternative build of the same library (kernel experiments); the default is the in-tree product build
    return os.environ.get("YDORB_LIB") or os.path.join(_HERE, "libydorb.so")


def build_library(jobs=4):
    """Compile every HIP source for gfx950 into ydorbslam_amd/libydorb.so (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc")])
    return library_path()


class YdKeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


class YdExtractorConfig(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_fast_thr", C.c_int32), ("min_fast_thr", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("flags", C.c_int32)]


# every symbol include/ydorb/c_api.h declares: (restype, argtypes)
_VP, _I, _Z = C.c_void_p, C.c_int32, C.c_size_t
SYMBOLS = {
    "ydorb_last_error": (C.c_char_p, []),
    "ydorb_device_count": (C.c_int, []),
    "ydorb_version": (C.c_char_p, []),
    "ydorb_extractor_create": (C.c_int, [C.POINTER(YdExtractorConfig), C.POINTER(_VP)]),
    "ydorb_extractor_destroy": (None, [_VP]),
    "ydorb_extractor_tables": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "ydorb_extractor_max_keypoints": (C.c_int, [_VP]),
    "ydorb_extract": (C.c_int, [_VP, _VP, _I, _I, _I, _VP, _VP, _I, C.POINTER(_I)]),
    "ydorb_extract_batch": (C.c_int, [_VP, _VP, _I, _I, _I, _Z, _I, _VP, _VP, _I, _VP]),
    "ydorb_extract_batch_device": (C.c_int, [_VP, _VP, _I, _I, _I, _Z, _I, _VP, _VP, _I, _VP, _VP]),
    "ydorb_extractor_synchronize": (C.c_int, [_VP]),
    "ydorb_extractor_pyramid": (C.c_int, [_VP, _I, _I, C.POINTER(_VP), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ydorb_extractor_read_level": (C.c_int, [_VP, _I, _I, _VP, _Z]),
    "ydorb_extractor_read_pyramid": (C.c_int, [_VP, _I, _VP, _VP, _I]),
    "ydorb_extractor_debug_read": (C.c_int, [_VP, _I, _I, _I, _VP, _Z, C.POINTER(_Z)]),
    "ydorb_extractor_set_profiling": (C.c_int, [_VP, _I]),
    "ydorb_extractor_stage_times": (C.c_int, [_VP, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_I)]),
    "ydorb_matcher_create": (C.c_int, [_I, C.POINTER(_VP)]),
    "ydorb_matcher_destroy": (None, [_VP]),
    "ydorb_descriptor_distance": (C.c_int, [_VP, _VP]),
    "ydorb_descriptor_distance_rows": (C.c_int, [_VP, _VP, _VP, _I, _VP]),
    "ydorb_frame_keypoints_in_area": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_float, _I, _I, _VP, _I, C.POINTER(_I)]),
    "ydorb_search_by_projection": (C.c_int, [_VP, _I, _VP, _VP, _VP, _I, C.c_float, _I, _I, _VP, _VP, C.POINTER(_I)]),
    "ydorb_search_by_bow": (C.c_int, [_VP, _I, _VP, _VP, C.c_float, _I, _VP, C.POINTER(_I)]),
    "ydorb_fuse_search": (C.c_int, [_VP, _VP, _VP, _VP, _I, _VP, _I, _VP, C.POINTER(_I)]),
    "ydorb_window_search": (C.c_int, [_VP, _VP, _VP, _VP, _I, _VP, _I, _I, _VP, C.POINTER(_I)]),
    "ydorb_search_for_triangulation": (C.c_int, [_VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _I, _I, _I, _VP, C.POINTER(_I)]),
    "ydorb_distinctive_descriptors": (C.c_int, [_VP, _VP, _VP, _I, _VP]),
    "ydorb_vocabulary_create": (C.c_int, [_VP, _I, C.POINTER(_VP)]),
    "ydorb_vocabulary_destroy": (None, [_VP]),
    "ydorb_vocabulary_transform": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "ydorb_stereo_matches": (C.c_int, [_VP, _VP, _VP, _I, C.c_float, C.c_float, _I, _VP, _VP, _VP, _VP, _VP]),
    "ydorb_match_consecutive_device": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, C.c_float, _VP, _I, _VP, _I, _VP, _VP, _VP]),
    "ydorb_match_pairs_device": (C.c_int, [_VP, _VP, _VP, _VP, _I, _I, _I, C.c_float, _VP, _I, _VP, _I, _VP, _VP, _VP]),
    "ydorb_hamming_topk": (C.c_int, [_VP, _VP, _I, _VP, _I, _VP, _VP, _VP]),
    "ydorb_hamming_topk_device": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "ydorb_matcher_synchronize": (C.c_int, [_VP]),
    "ydorb_matcher_set_profiling": (C.c_int, [_VP, _I]),
    "ydorb_matcher_stage_times": (C.c_int, [_VP, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(_I)]),
    "ydorb_ba_default_options": (None, [_VP]),
    "ydorb_ba_release": (C.c_int, [_I]),
    "ydorb_ba_solve": (C.c_int, [_VP, _VP, _VP]),
    "ydorb_ba_solve_batch": (C.c_int, [_VP, _I, _VP, _VP, _I, _VP]),
    "ydorb_ba_dense_solve": (C.c_int, [_I, _VP, _I, _VP, _VP, C.POINTER(_I)]),
    "ydorb_pose_optimize": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
}

BA_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)


class YdBaProblem(C.Structure):
    _fields_ = [("n_poses", _I), ("n_points", _I), ("n_edges", _I), ("poses", _VP), ("pose_fixed", _VP), ("points", _VP),
                ("edge_pose", _VP), ("edge_point", _VP), ("edge_meas", _VP), ("edge_inv_sigma2", _VP),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double), ("stop", _VP)]


class YdPoseBatch(C.Structure):
    _fields_ = [("n_frames", _I), ("device", _I), ("edge_start", _VP), ("poses", _VP), ("points", _VP), ("meas", _VP),
                ("inv_sigma2", _VP), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double)]


class YdBaOptions(C.Structure):
    _fields_ = [("iters1", _I), ("iters2", _I), ("chi2_mono", C.c_double), ("chi2_stereo", C.c_double),
                ("delta_mono", C.c_double), ("delta_stereo", C.c_double), ("max_trials", _I), ("device", _I),
                ("allreduce", BA_ALLREDUCE_FN), ("allreduce_user", _VP), ("d_comm_buf", _VP), ("comm_doubles", C.c_int64),
                ("rank", _I), ("world", _I), ("flags", _I), ("reserved", _I)]


BA_SINGLE_STAGE, BA_NO_ROBUST, BA_PHASE_TIMES = 1, 2, 4


class YdBaResult(C.Structure):
    _fields_ = [("n_trials", _I), ("n_iterations", _I), ("n_log", _I), ("stopped", _I), ("log_chi2", C.c_double * 32),
                ("log_lambda", C.c_double * 32), ("log_trials", _I * 32), ("log_stage", _I * 32), ("edge_outlier", _VP),
                ("ms_total", C.c_float), ("ms_errors", C.c_float), ("ms_build", C.c_float), ("ms_schur", C.c_float),
                ("ms_solve", C.c_float), ("ms_update", C.c_float)]


class YdFrameSetDev(C.Structure):
    _fields_ = [("d_kps", _VP), ("d_desc", _VP), ("d_n", _VP), ("n_frames", _I), ("cap", _I)]


class YdFrameView(C.Structure):
    _fields_ = [("kps", _VP), ("desc", _VP), ("right_x", _VP), ("n", _I), ("min_x", C.c_float), ("max_x", C.c_float),
                ("min_y", C.c_float), ("max_y", C.c_float)]


class YdFeatureVector(C.Structure):
    _fields_ = [("node_ids", _VP), ("node_start", _VP), ("feat", _VP), ("n_nodes", _I)]


class YdBowSide(C.Structure):
    _fields_ = [("kps", _VP), ("desc", _VP), ("valid", _VP), ("n", _I), ("fv", YdFeatureVector)]


class YdTriSide(C.Structure):
    _fields_ = [("kps", _VP), ("desc", _VP), ("right_x", _VP), ("has_map_point", _VP), ("n", _I), ("fv", YdFeatureVector)]


class YdVocabularyTree(C.Structure):
    _fields_ = [("n_nodes", _I), ("levels", _I), ("child_begin", _VP), ("child_ids", _VP), ("node_desc", _VP), ("node_weight", _VP),
                ("node_word", _VP), ("weighting", _I), ("norm", _I)]


class YdStereoSide(C.Structure):
    _fields_ = [("extractor", _VP), ("first_frame", _I), ("frame_step", _I), ("kps", _VP), ("desc", _VP), ("n", _VP), ("cap", _I),
                ("reserved", _I)]


STEREO_INDEX_BY_KEYPOINT = 1
STEREO_DEVICE_POINTERS = 2


def lib():
    """The loaded C-ABI library.  Raises YdorbError when it has not been built — there is no fallback."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise YdorbError("%s is missing: build it with ydorbslam_amd.build_library(); the hot path has no CPU fallback" % path)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; loading it
        # first lets libydorb.so bind to that same copy (same SONAME) instead of bringing /opt/rocm's beside it,
        # which leaves torch with "No HIP GPUs are available".  Plumbing only: nothing here computes with torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            L = C.CDLL(path)
        except OSError as e:
            raise YdorbError("cannot load %s: %s" % (path, e))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise YdorbError("ydorb error %d: %s" % (rc, lib().ydorb_last_error().decode()))
