"""ctypes binding of the C-ABI (include/phdslam.h) exported by the in-tree libphdslam.so.

The library is hand-written HIP for gfx950; there is no CPU fallback.  Loading fails loudly if
the shared object is missing (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHD_LIB: an A/B build of the same library (make variant NAME=..., tools/ab_bench.sh); default: the product
LIB_PATH = os.environ.get("PHD_LIB") or os.path.join(_HERE, "libphdslam.so")

GAUSSIAN = np.dtype([("cov", np.float32, 4), ("mean", np.float32, 2), ("weight", np.float32)])
POSE = np.dtype([("px", np.float32), ("py", np.float32), ("ptheta", np.float32),
                 ("vx", np.float32), ("vy", np.float32), ("vtheta", np.float32)])
MEAS = np.dtype([("range", np.float32), ("bearing", np.float32), ("label", np.int32)])
NOISE = np.dtype([("n_alpha", np.float32), ("n_encoder", np.float32)])

PHD_OK = 0
ERR_NAMES = {-1: "INVALID_ARG", -2: "NO_DEVICE", -3: "HIP", -4: "UNSUPPORTED", -5: "CAPACITY", -6: "NAN",
             -7: "IO", -8: "PARSE"}
K_PREDICT, K_UPDATE_MERGE, K_WEIGHTS, K_COUNT = 0, 1, 2, 3


class SlamConfig(C.Structure):
    """src/slamtypes.h:142-250 — same field order and 324-byte layout."""
    _fields_ = [
        ("debug", C.c_uint8),
        ("x0", C.c_float), ("y0", C.c_float), ("z0", C.c_float), ("roll0", C.c_float), ("pitch0", C.c_float),
        ("yaw0", C.c_float), ("vx0", C.c_float), ("vy0", C.c_float), ("vz0", C.c_float), ("vroll0", C.c_float),
        ("vpitch0", C.c_float), ("vyaw0", C.c_float),
        ("followTrajectory", C.c_uint8),
        ("ax", C.c_float), ("ay", C.c_float), ("az", C.c_float), ("aroll", C.c_float), ("apitch", C.c_float),
        ("ayaw", C.c_float), ("dt", C.c_float),
        ("minRange", C.c_float), ("maxRange", C.c_float), ("maxBearing", C.c_float),
        ("stdRange", C.c_float), ("stdBearing", C.c_float), ("clutterRate", C.c_float), ("clutterDensity", C.c_float),
        ("pd", C.c_float), ("stdVxMap", C.c_float), ("stdVyMap", C.c_float), ("stdAxMap", C.c_float),
        ("stdAyMap", C.c_float), ("covVxBirth", C.c_float), ("covVyBirth", C.c_float), ("ps", C.c_float),
        ("tau", C.c_float), ("beta", C.c_float),
        ("particlesPerFeature", C.c_int32), ("imageWidth", C.c_int32), ("imageHeight", C.c_int32),
        ("stdU", C.c_float), ("stdV", C.c_float), ("disparityBirth", C.c_float), ("stdDBirth", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("u0", C.c_float), ("v0", C.c_float),
        ("n_particles", C.c_int32), ("nPredictParticles", C.c_int32), ("subdividePredict", C.c_int32),
        ("resampleThresh", C.c_float), ("birthWeight", C.c_float), ("birthNoiseFactor", C.c_float),
        ("gateBirths", C.c_uint8), ("gateMeasurements", C.c_uint8),
        ("gateThreshold", C.c_float), ("minExpectedFeatureWeight", C.c_float), ("minSeparation", C.c_float),
        ("maxFeatures", C.c_int32), ("minFeatureWeight", C.c_float),
        ("particleWeighting", C.c_int32), ("daughterMixtureType", C.c_int32), ("nSamples", C.c_int32),
        ("maxCardinality", C.c_int32), ("filterType", C.c_int32), ("distanceMetric", C.c_int32),
        ("maxSteps", C.c_int32), ("featureModel", C.c_int32), ("motionType", C.c_int32), ("mapEstimate", C.c_int32),
        ("cphdDistType", C.c_int32), ("nu", C.c_float),
        ("labeledMeasurements", C.c_uint8),
        ("l", C.c_float), ("h", C.c_float), ("a", C.c_float), ("b", C.c_float),
        ("stdAlpha", C.c_float), ("stdEncoder", C.c_float),
        ("saveAllMaps", C.c_uint8), ("savePrediction", C.c_uint8),
    ]


class Options(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("map_capacity", C.c_int32), ("max_measurements", C.c_int32),
                ("survivor_capacity", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("global_particles", C.c_int32), ("global_offset", C.c_int32)]


class StepReport(C.Structure):
    """phd_step_report: status word, high-water marks, nEff and resample decision of the last weights routine"""
    _fields_ = [("status", C.c_uint32), ("max_survivors", C.c_int32), ("max_map", C.c_int32), ("neff", C.c_float),
                ("did_resample", C.c_int32)]


class SnapshotView(C.Structure):
    """phd_snapshot_view: pointers into the slot's pinned block (valid until the slot is captured again)"""
    _fields_ = [("expected", C.c_void_p), ("map", C.c_void_p), ("poses", C.c_void_p), ("log_weights", C.c_void_p),
                ("resample_idx", C.c_void_p), ("cardinality", C.c_void_p), ("n_map", C.c_int32), ("particle", C.c_int32),
                ("n_particles", C.c_int32), ("cardinality_len", C.c_int32), ("report", StepReport)]


class Control(C.Structure):
    _fields_ = [("alpha", C.c_float), ("v_encoder", C.c_float)]


# every symbol include/phdslam.h declares: name -> (restype, argtypes)
_vp, _i, _f, _d, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64
_cfgp, _optp = C.POINTER(SlamConfig), C.POINTER(Options)
SYMBOLS = {
    "phd_last_error": (C.c_char_p, []),
    "phd_version": (C.c_char_p, []),
    "phd_create": (_i, [_cfgp, _optp, C.POINTER(_vp)]),
    "phd_destroy": (_i, [_vp]),
    "phd_set_config": (_i, [_vp, _cfgp]),
    "phd_seed": (_i, [_vp, _u64]),
    "phd_n_particles": (_i, [_vp]),
    "phd_map_capacity": (_i, [_vp]),
    "phd_set_particles": (_i, [_vp, _vp, _vp, _i]),
    "phd_get_particles": (_i, [_vp, _vp, _vp]),
    "phd_set_maps": (_i, [_vp, _vp, _vp]),
    "phd_get_map_sizes": (_i, [_vp, _vp]),
    "phd_get_maps": (_i, [_vp, _vp, _sz, _vp]),
    "phd_set_map": (_i, [_vp, _i, _vp, _i]),
    "phd_get_map": (_i, [_vp, _i, _vp, _i, _vp]),
    "phd_predict_ackerman": (_i, [_vp, Control, _vp]),
    "phd_update": (_i, [_vp, _vp, _i]),
    "phd_predict_update": (_i, [_vp, Control, _vp, _vp, _i]),
    "phd_neff": (_i, [_vp, _vp]),
    "phd_resample": (_i, [_vp, _vp, _i, _vp]),
    "phd_resample_if_needed": (_i, [_vp, _d, _i, _vp, _vp]),
    "phd_expected_pose": (_i, [_vp, _vp]),
    "phd_map_estimate": (_i, [_vp, _vp, _i, _vp, _vp]),
    "phd_expected_map": (_i, [_vp, _vp, _i, _vp]),
    "phd_cardinality_length": (_i, [_vp]),
    "phd_get_cardinalities": (_i, [_vp, _vp]),
    "phd_set_cardinalities": (_i, [_vp, _vp]),
    "phd_cardinality_estimate": (_i, [_vp, _vp, _vp]),
    "phd_gm_reduce": (_i, [_vp, _vp, C.c_int64, _f, _vp, _i, _vp]),
    "phd_expected_map_concat_dev": (_i, [_vp, _vp, _vp]),
    "phd_gm_reduce_dev": (_i, [_vp, _vp, C.c_int64, _i, _f, _vp, _i, _vp]),
    "phd_debug_gm_rounds": (_i, [_vp]),
    "phd_debug_copy_free_resamples": (_i, [_vp]),
    "phd_debug_update_instantiation": (_i, [_vp]),
    "phd_predict_ackerman_dev": (_i, [_vp, Control, _vp]),
    "phd_update_dev": (_i, [_vp, _vp, _i]),
    "phd_logweights_dev": (_i, [_vp, C.POINTER(_vp)]),
    "phd_raw_logweights_dev": (_i, [_vp, C.POINTER(_vp)]),
    "phd_update_local_dev": (_i, [_vp, _vp, _i]),
    "phd_global_normalize": (_i, [_vp, _vp, _i, _vp]),
    "phd_global_resample_indices": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "phd_apply_parents": (_i, [_vp, _vp]),
    "phd_particle_pack_bytes": (_sz, [_vp]),
    "phd_export_particles_dev": (_i, [_vp, _vp, _i, _vp]),
    "phd_import_particles_dev": (_i, [_vp, _vp, _i, _vp]),
    "phd_import_particles_sel_dev": (_i, [_vp, _vp, _vp, _i, _vp]),
    "phd_finish_resample": (_i, [_vp]),
    "phd_global_resample_begin": (_i, [_vp, _vp, _d, _i, _i, _vp, _vp, _vp, _vp]),
    "phd_step_local_dev": (_i, [_vp, Control, _vp, _vp, _i]),
    "phd_global_resample_end": (_i, [_vp, _vp]),
    "phd_global_resample_launch": (_i, [_vp, _vp, _d, _vp]),
    "phd_global_resample_launch_normalized": (_i, [_vp, _vp, _d, _vp]),
    "phd_global_resample_plan": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "phd_state_snapshot": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "phd_snapshot_capture": (_i, [_vp, _i]),
    "phd_snapshot_send": (_i, [_vp, _i, _i]),
    "phd_snapshot_wait": (_i, [_vp, _i, _vp]),
    "phd_host_alloc": (_vp, [C.c_size_t]),
    "phd_host_free": (None, [_vp]),
    "phd_step_report_get": (_i, [_vp, _vp]),
    "phd_set_particle_count": (_i, [_vp, _i]),
    "phd_export_shard_dev": (_i, [_vp, _vp, _vp]),
    "phd_export_shard_current_dev": (_i, [_vp, _vp, _vp]),
    "phd_step_local_rows_dev": (_i, [_vp, Control, _vp, _vp, _i, _vp, _vp]),
    "phd_set_rows_target": (_i, [_vp, _vp]),
    "phd_global_resample_gathered": (_i, [_vp, _vp, _d, _i, _i, _i, _vp]),
    "phd_peer_view_get": (_i, [_vp, _vp]),
    "phd_global_resample_pull": (_i, [_vp, _vp, _i, _i]),
    "phd_set_raw_target": (_i, [_vp, _vp]),
    "phd_global_resample_auto_supported": (_i, [_vp, _i]),
    "phd_global_resample_launch_auto": (_i, [_vp, _vp, _d]),
    "phd_global_resample_pull_auto": (_i, [_vp, _vp, _i, _i]),
    "phd_set_frozen": (_i, [_vp, _i]),
    "phd_step_dev": (_i, [_vp, Control, _vp, _vp, _i, _d, _i]),
    "phd_sync": (_i, [_vp]),
    "phd_stream": (_vp, [_vp]),
    "phd_timing_enable": (_i, [_vp, _i]),
    "phd_timing_read": (_i, [_vp, _vp, _vp]),
    "phd_timing_reset": (_i, [_vp]),
    "phd_debug_enable": (_i, [_vp, _i]),
    "phd_debug_get_stamps": (_i, [_vp, _vp]),
    "phd_debug_get_survivors": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "phd_debug_get_weight_increments": (_i, [_vp, _vp]),
    "phd_device_status": (_i, [_vp, _vp, _vp, _vp]),
    "phd_update_residency": (_i, [_vp, _vp, _vp]),
    "phd_config_defaults": (_i, [_cfgp]),
    "phd_config_load": (_i, [C.c_char_p, _cfgp, C.c_char_p, _sz, _vp]),
    "phd_load_measurements": (_i, [C.c_char_p, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    "phd_load_controls": (_i, [C.c_char_p, _i, _vp, _sz, _vp]),
    "phd_write_state_log": (_i, [C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "phd_ospa": (_i, [_vp, _i, _vp, _i, _d, _d, _vp]),
    "phd_evaluate_state_log": (_i, [C.c_char_p, _vp, _vp, _i, _d, _d, _vp]),
    "phd_load_timestamps": (_i, [C.c_char_p, _vp, _sz, _vp]),
    "phd_load_trajectory": (_i, [C.c_char_p, _vp, _sz, _vp]),
    "phd_write_state_log7": (_i, [C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i]),
    "phd_write_state_log_cphd": (_i, [C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i]),
    "phd_write_state_log7_cphd": (_i, [C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i]),
}

_lib = None


def build():
    """Compile libphdslam.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libphdslam.so is not built (%s): the HIP extension is required, there is no "
                               "fallback.  Run __graft_entry__.build()." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (SONAME
        # libamdhip64.so.7).  Loaded first, it satisfies this library's NEEDED entry and both share
        # one runtime; loaded second, it would come up as a second runtime that sees no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class PhdError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().phd_last_error()
        RuntimeError.__init__(self, "%s failed: PHD_ERR_%s (%d): %s" % (
            where, ERR_NAMES.get(code, "?"), code, msg.decode() if msg else ""))


def check(code, where):
    if code != PHD_OK:
        raise PhdError(code, where)


def ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def default_config(**over):
    """loadConfig defaults (src/main.cpp:960-1048) overridden by the values of the reference's
    cfg/config.cfg:46-159 — the configuration every BASELINE.json workload uses."""
    cfg = SlamConfig()
    check(lib().phd_config_defaults(C.byref(cfg)), "phd_config_defaults")
    vals = dict(motionType=1, maxRange=15.0, maxBearing=3.141593, stdRange=0.25, stdBearing=0.008727,
                clutterRate=20.0, pd=0.95, l=1.415, h=0.38, a=1.89, b=0.5, stdEncoder=1.0, stdAlpha=0.034907,
                dt=0.1, filterType=0, featureModel=0, particleWeighting=0, distanceMetric=0, n_particles=200,
                subdividePredict=1, resampleThresh=0.5, birthWeight=1e-4, birthNoiseFactor=1.0,
                minSeparation=10.0, minFeatureWeight=1e-6, nPredictParticles=1, maxCardinality=255, mapEstimate=1)
    vals.update(over)
    for k, v in vals.items():
        setattr(cfg, k, v)
    f32 = np.float32
    # src/main.cpp:1065-1066, evaluated in float like the reference
    cfg.clutterDensity = float(f32(cfg.clutterRate) / (f32(2) * f32(cfg.maxBearing) * f32(cfg.maxRange)))
    if "clutterDensity" in over:
        cfg.clutterDensity = over["clutterDensity"]
    return cfg
