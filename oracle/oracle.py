"""ctypes wrapper around oracle/libscphd_cpu.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libscphd_cpu.so")

GAUSSIAN = np.dtype([("cov", np.float32, 4), ("mean", np.float32, 2), ("weight", np.float32)])
POSE = np.dtype([("px", np.float32), ("py", np.float32), ("ptheta", np.float32),
                 ("vx", np.float32), ("vy", np.float32), ("vtheta", np.float32)])
MEAS = np.dtype([("range", np.float32), ("bearing", np.float32), ("label", np.int32)])
assert GAUSSIAN.itemsize == 28 and POSE.itemsize == 24 and MEAS.itemsize == 12


class OConfig(C.Structure):
    _fields_ = [("dt", C.c_float),
                ("minRange", C.c_float), ("maxRange", C.c_float), ("maxBearing", C.c_float),
                ("stdRange", C.c_float), ("stdBearing", C.c_float),
                ("clutterDensity", C.c_float), ("pd", C.c_float),
                ("birthWeight", C.c_float), ("birthNoiseFactor", C.c_float),
                ("minFeatureWeight", C.c_float), ("minSeparation", C.c_float),
                ("resampleThresh", C.c_float),
                ("l", C.c_float), ("h", C.c_float), ("a", C.c_float), ("b", C.c_float),
                ("subdividePredict", C.c_int32), ("distanceMetric", C.c_int32),
                ("labeledMeasurements", C.c_int32), ("particleWeighting", C.c_int32), ("mergeSums", C.c_int32)]


def default_config(**over):
    """Parameter values of the reference's cfg/config.cfg:46-159."""
    f32 = np.float32
    max_range, max_bearing, clutter_rate = f32(15.0), f32(3.141593), f32(20.0)
    cfg = OConfig(dt=0.1, minRange=0.0, maxRange=max_range, maxBearing=max_bearing,
                  stdRange=0.25, stdBearing=0.008727,
                  # src/main.cpp:1065-1066: clutterRate/(2*maxBearing*maxRange) in float
                  clutterDensity=float(clutter_rate / (f32(2) * max_bearing * max_range)),
                  pd=0.95, birthWeight=1e-4, birthNoiseFactor=1.0,
                  minFeatureWeight=1e-6, minSeparation=10.0, resampleThresh=0.5,
                  l=1.415, h=0.38, a=1.89, b=0.5,
                  subdividePredict=1, distanceMetric=0, labeledMeasurements=0, particleWeighting=0, mergeSums=0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _sources():
    return [os.path.join(_HERE, f) for f in ("scphd_cpu.c", "cphd_cpu.c", "scphd_cpu.h", "Makefile")]


def build(force=False):
    """the portable checker library (oracle/Makefile: -O3 -march=x86-64-v3 -ffp-contract=off): what the tests load"""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in _sources()):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


NATIVE_CFLAGS = "-O3 -march=native -ffp-contract=off -fopenmp -fPIC -std=c11"


def host_cpu():
    """(model name, flags line) of the CPU this process runs on"""
    model, flags = "unknown", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
            if model != "unknown" and flags:
                break
    except OSError:
        pass
    return model, flags


def build_native():
    """bench.py's cpu_baseline leg: the same sources compiled -O3 -march=native ON THE HOST THAT RUNS THEM.
    The file name is keyed on the CPU model + feature flags + compiler flags, so a library built on another machine
    (the snapshot travels) is never picked up.  -> (path, info dict)"""
    import hashlib
    import platform
    model, flags = host_cpu()
    tag = hashlib.sha1((model + "|" + flags + "|" + NATIVE_CFLAGS).encode()).hexdigest()[:12]
    d = os.path.join(_HERE, "_host")
    try:
        os.makedirs(d, exist_ok=True)
        if not os.access(d, os.W_OK):
            raise OSError("read-only")
    except OSError:                                    # a read-only checkout: build beside the temp files instead
        import tempfile
        d = os.path.join(tempfile.gettempdir(), "phd_oracle_host_%d" % os.getuid())
        os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "libscphd_cpu.native.%s.so" % tag)
    srcs = [os.path.join(_HERE, "scphd_cpu.c"), os.path.join(_HERE, "cphd_cpu.c")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in _sources()):
        tmp = path + ".tmp.%d" % os.getpid()
        subprocess.check_call(["gcc"] + NATIVE_CFLAGS.split() + ["-shared", "-o", tmp] + srcs + ["-lm"])
        os.replace(tmp, path)
    try:
        cc = subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0]
    except Exception:
        cc = "gcc"
    return path, {"compile_flags": NATIVE_CFLAGS, "compiler": cc, "compiled_on_cpu": model,
                  "compiled_on_host": platform.node(), "library": os.path.relpath(path, os.path.dirname(_HERE))}


_lib = None


def use_library(path=None):
    """load another build of the oracle (bench.py: the host-native one); None: back to the portable checker"""
    global _lib, _LIB_PATH
    _LIB_PATH = path or os.path.join(_HERE, "libscphd_cpu.so")
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if _LIB_PATH == os.path.join(_HERE, "libscphd_cpu.so"):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, f32, f64 = C.c_void_p, C.c_int, C.c_float, C.c_double
        cp = C.POINTER(OConfig)
        L.o_safe_log.restype = f32; L.o_safe_log.argtypes = [f32]
        L.o_wrap_angle.restype = f32; L.o_wrap_angle.argtypes = [f32]
        L.o_det_exp.restype = f64; L.o_det_exp.argtypes = [f32]
        L.o_omp_max_threads.restype = i32; L.o_omp_max_threads.argtypes = []
        L.o_tmp_release.restype = None; L.o_tmp_release.argtypes = []
        L.o_predict_ackerman.restype = None; L.o_predict_ackerman.argtypes = [vp, i32, f32, f32, vp, cp]
        L.o_predicted_measurement.restype = None; L.o_predicted_measurement.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.o_classify.restype = None; L.o_classify.argtypes = [vp, i32, vp, cp, vp]
        L.o_births.restype = None; L.o_births.argtypes = [vp, vp, i32, cp, vp]
        L.o_preupdate.restype = None; L.o_preupdate.argtypes = [vp, vp, i32, vp, i32, cp, vp, vp]
        L.o_update.restype = None; L.o_update.argtypes = [vp, vp, vp, vp, i32, i32, cp, vp, vp, vp]
        L.o_mahal_dist.restype = f32; L.o_mahal_dist.argtypes = [vp, vp]
        L.o_hellinger_dist.restype = f32; L.o_hellinger_dist.argtypes = [vp, vp]
        L.o_merge.restype = i32; L.o_merge.argtypes = [vp, i32, cp, vp, vp]
        L.o_gm_reduce.restype = i32; L.o_gm_reduce.argtypes = [vp, i32, f32, vp]
        L.o_update_particle.restype = i32
        L.o_update_particle.argtypes = [vp, vp, i32, vp, i32, cp, vp, vp, vp, vp, vp, vp]
        L.o_update_particle_ex.restype = i32
        L.o_update_particle_ex.argtypes = [vp, vp, i32, vp, i32, cp, vp, vp, vp, vp, vp, vp, vp]
        L.o_merge_follow.restype = i32; L.o_merge_follow.argtypes = [vp, vp, i32, cp, vp, vp]
        L.o_cphd_update_particle_ex.restype = i32
        L.o_cphd_update_particle_ex.argtypes = [vp, vp, i32, vp, i32, cp, f32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.o_normalize_weights.restype = None; L.o_normalize_weights.argtypes = [vp, vp, i32]
        L.o_neff.restype = f32; L.o_neff.argtypes = [vp, i32]
        L.o_resample.restype = None; L.o_resample.argtypes = [vp, i32, vp, i32, i32, vp]
        L.o_expected_pose.restype = None; L.o_expected_pose.argtypes = [vp, vp, i32, vp]
        L.o_expected_map.restype = i32; L.o_expected_map.argtypes = [vp, vp, vp, i32, f32, vp]
        L.o_cphd_terms.restype = None
        L.o_cphd_terms.argtypes = [vp, i32, vp, i32, f32, f32, f32, f32, f32, vp, vp, vp, vp]
        L.o_cphd_update_particle.restype = i32
        L.o_cphd_update_particle.argtypes = [vp, vp, i32, vp, i32, cp, f32, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.o_cphd_step.restype = i32
        L.o_cphd_step.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, i32, cp, f32, vp, i32, f64, i32, vp, vp, vp, vp, vp, i32]
        L.o_argmax_weight.restype = i32; L.o_argmax_weight.argtypes = [vp, i32]
        L.o_step.restype = i32
        L.o_step.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, i32, cp, f64, i32, vp, vp, vp, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------------------------------------
def wrap_angle(a):
    return float(lib().o_wrap_angle(float(np.float32(a))))


def det_exp(x):
    return float(lib().o_det_exp(float(np.float32(x))))


def predict_ackerman(poses, alpha, v_encoder, noise, cfg):
    poses = _c(poses, POSE).copy()
    noise = None if noise is None else _c(noise, np.float32)
    lib().o_predict_ackerman(_p(poses), len(poses), float(alpha), float(v_encoder), _p(noise), C.byref(cfg))
    return poses


def predicted_measurement(pose, mean):
    pose = _c(pose, POSE).reshape(1)
    mean = _c(mean, np.float32)
    r = C.c_float(); b = C.c_float()
    lib().o_predicted_measurement(_p(pose), _p(mean), C.byref(r), None, C.byref(b), None, None)
    return r.value, b.value


def classify(gmap, pose, cfg):
    gmap = _c(gmap, GAUSSIAN)
    pose = _c(pose, POSE).reshape(1)
    cls = np.zeros(len(gmap), np.int8)
    lib().o_classify(_p(gmap), len(gmap), _p(pose), C.byref(cfg), _p(cls))
    return cls


def births(pose, z, cfg):
    pose = _c(pose, POSE).reshape(1)
    z = _c(z, MEAS)
    out = np.zeros(len(z), GAUSSIAN)
    lib().o_births(_p(pose), _p(z), len(z), C.byref(cfg), _p(out))
    return out


def preupdate(pose, feat, z, cfg):
    pose = _c(pose, POSE).reshape(1)
    feat = _c(feat, GAUSSIAN)
    z = _c(z, MEAS)
    n, M = len(feat), len(z)
    pd = np.zeros(n, np.float32)
    pre = np.zeros((M, n), GAUSSIAN)
    lib().o_preupdate(_p(pose), _p(feat), n, _p(z), M, C.byref(cfg), _p(pd), _p(pre))
    return pd, pre


def update(feat, pd, pre, births_, cfg):
    feat = _c(feat, GAUSSIAN)
    n, M = len(feat), len(births_)
    slab = np.zeros(n * (M + 1) + M, GAUSSIAN)
    flag = np.zeros(len(slab), np.uint8)
    dlw = np.zeros(1, np.float32)
    lib().o_update(_p(feat), _p(_c(pd, np.float32)), _p(_c(pre, GAUSSIAN)), _p(_c(births_, GAUSSIAN)),
                   n, M, C.byref(cfg), _p(slab), _p(flag), _p(dlw))
    return slab, flag, float(dlw[0])


def mahal_dist(a, b):
    a = _c(a, GAUSSIAN).reshape(1); b = _c(b, GAUSSIAN).reshape(1)
    return float(lib().o_mahal_dist(_p(a), _p(b)))


def hellinger_dist(a, b):
    a = _c(a, GAUSSIAN).reshape(1); b = _c(b, GAUSSIAN).reshape(1)
    return float(lib().o_hellinger_dist(_p(a), _p(b)))


def merge(comps, cfg, with_margin=False):
    comps = _c(comps, GAUSSIAN)
    out = np.zeros(max(len(comps), 1), GAUSSIAN)
    margin = np.zeros(2, np.float32)
    n = lib().o_merge(_p(comps), len(comps), C.byref(cfg), _p(out), _p(margin))
    return (out[:n], margin) if with_margin else out[:n]


FOLLOW_STATS = ("dist_flips", "dist_unexplained", "dist_worst_ratio", "order_flips", "order_unexplained", "order_worst_ratio",
                "max_dist_move", "n_decisions", "nan_decisions")


def merge_follow(ref, comps, cfg):
    """TEST DIAGNOSTIC: the merge of `comps` (the oracle's survivors) taking every decision from the merge of `ref` (the
    device's survivors, same length and order), with a first-order proof for every decision `comps` alone would have
    taken differently (o_merge_follow).  -> (merged map in ref's output order, dict of FOLLOW_STATS)"""
    ref = _c(ref, GAUSSIAN); comps = _c(comps, GAUSSIAN)
    assert len(ref) == len(comps)
    out = np.zeros(max(len(comps), 1), GAUSSIAN)
    stats = np.zeros(9, np.float64)
    n = lib().o_merge_follow(_p(ref), _p(comps), len(comps), C.byref(cfg), _p(out), _p(stats))
    return out[:n], dict(zip(FOLLOW_STATS, stats.tolist()))


def gm_reduce(comps, min_distance):
    comps = _c(comps, GAUSSIAN)
    out = np.zeros(max(len(comps), 1), GAUSSIAN)
    n = lib().o_gm_reduce(_p(comps), len(comps), float(min_distance), _p(out))
    return out[:n]


def update_particle(pose, gmap, z, cfg, with_slab=False):
    """-> dict(map, dlogw, survivors, slab_idx, margin[, slab_all: the unpruned slab + the nearly-in-range features,
    the array slab_idx indexes])"""
    pose = _c(pose, POSE).reshape(1)
    gmap = _c(gmap, GAUSSIAN)
    z = _c(z, MEAS)
    n, M = len(gmap), len(z)
    cap = n * (M + 1) + M + n + 1
    out = np.zeros(cap, GAUSSIAN)
    surv = np.zeros(cap, GAUSSIAN)
    sidx = np.zeros(cap, np.int32)
    slab_all = np.zeros(cap, GAUSSIAN) if with_slab else None
    ns = C.c_int(0)
    dlw = np.zeros(1, np.float32)
    margin = np.zeros(2, np.float32)
    nm = lib().o_update_particle_ex(_p(pose), _p(gmap), n, _p(z), M, C.byref(cfg), _p(out), _p(dlw),
                                    _p(surv), _p(sidx), C.byref(ns), _p(margin), _p(slab_all))
    r = dict(map=out[:nm].copy(), dlogw=float(dlw[0]), survivors=surv[:ns.value].copy(),
             slab_idx=sidx[:ns.value].copy(), margin=margin)
    if with_slab:
        r["slab_all"] = slab_all
    return r


def normalize_weights(logw, dlogw=None):
    logw = _c(logw, np.float32).copy()
    d = None if dlogw is None else _c(dlogw, np.float32)
    lib().o_normalize_weights(_p(logw), _p(d), len(logw))
    return logw


def neff(logw):
    logw = _c(logw, np.float32)
    return float(lib().o_neff(_p(logw), len(logw)))


def resample(logw, uniforms, n_new=None):
    logw = _c(logw, np.float32)
    u = _c(np.atleast_1d(uniforms), np.float64)
    n_new = len(logw) if n_new is None else n_new
    idx = np.zeros(n_new, np.int32)
    lib().o_resample(_p(logw), len(logw), _p(u), len(u), n_new, _p(idx))
    return idx


def expected_pose(poses, logw):
    poses = _c(poses, POSE); logw = _c(logw, np.float32)
    out = np.zeros(1, POSE)
    lib().o_expected_pose(_p(poses), _p(logw), len(poses), _p(out))
    return out[0]


def expected_map(maps_concat, sizes, logw, min_distance):
    """computeExpectedMap (src/main.cpp:290-316): maps_concat = all particle maps back to back"""
    maps_concat = _c(maps_concat, GAUSSIAN); sizes = _c(sizes, np.int32); logw = _c(logw, np.float32)
    out = np.zeros(max(len(maps_concat), 1), GAUSSIAN)
    n = lib().o_expected_map(_p(maps_concat), _p(sizes), _p(logw), len(sizes), float(min_distance), _p(out))
    return out[:n]


def cphd_terms(cn_prior, S, w_all, pdw, birth_weight, clutter_rate, clutter_density):
    """-> dict(lz[M], r1, cn[cn_len], lY0): the cardinality-dependent terms of one particle's CPHD update"""
    cn_prior = _c(cn_prior, np.float32); S = _c(S, np.float32)
    M = len(S)
    lz = np.zeros(max(M, 1), np.float32); cn = np.zeros(len(cn_prior), np.float32)
    r1 = np.zeros(1, np.float32); ly0 = np.zeros(1, np.float32)
    lib().o_cphd_terms(_p(cn_prior), len(cn_prior), _p(S), M, float(w_all), float(pdw), float(birth_weight),
                       float(clutter_rate), float(clutter_density), _p(lz), _p(r1), _p(cn), _p(ly0))
    return dict(lz=lz[:M], r1=float(r1[0]), cn=cn, lY0=float(ly0[0]))


def cphd_set_reference_esf(on):
    """True: leave-one-out ESFs by M separate recursions (the .bak's O(M^3) structure); False: the O(M^2) form"""
    lib().o_cphd_set_reference_esf(int(bool(on)))


def cphd_update_particle(pose, gmap, z, cfg, clutter_rate, cn_prior):
    """-> dict(map, dlogw, cn, survivors, slab_idx, r1, margin (merge distance, seed weight gap), prune_margin, slab_all)"""
    pose = _c(pose, POSE).reshape(1)
    gmap = _c(gmap, GAUSSIAN); z = _c(z, MEAS); cn_prior = _c(cn_prior, np.float32)
    n, M = len(gmap), len(z)
    cap = n * (M + 1) + M + n + 1
    out = np.zeros(cap, GAUSSIAN); surv = np.zeros(cap, GAUSSIAN); sidx = np.zeros(cap, np.int32)
    ns = C.c_int(0)
    dlw = np.zeros(1, np.float32); r1 = np.zeros(1, np.float32); cn = np.zeros(len(cn_prior), np.float32)
    margin = np.zeros(3, np.float32); slab_all = np.zeros(cap, GAUSSIAN)
    nm = lib().o_cphd_update_particle_ex(_p(pose), _p(gmap), n, _p(z), M, C.byref(cfg), float(clutter_rate), _p(cn_prior),
                                         len(cn_prior), _p(out), _p(dlw), _p(cn), _p(surv), _p(sidx), C.byref(ns), _p(r1),
                                         _p(margin), _p(slab_all))
    return dict(map=out[:nm].copy(), dlogw=float(dlw[0]), cn=cn, survivors=surv[:ns.value].copy(),
                slab_idx=sidx[:ns.value].copy(), r1=float(r1[0]), margin=margin[:2].copy(), prune_margin=float(margin[2]),
                slab_all=slab_all)


def argmax_weight(logw):
    logw = _c(logw, np.float32)
    return int(lib().o_argmax_weight(_p(logw), len(logw)))


def step(poses, logw, maps, sizes, cap, alpha, v_encoder, noise, z, cfg, uniform, force_resample, n_threads=0):
    """Whole filter step on fixed-capacity slabs (maps: [N, cap] GAUSSIAN).  Inputs are copied."""
    poses = _c(poses, POSE).copy(); logw = _c(logw, np.float32).copy()
    maps = _c(maps, GAUSSIAN).reshape(-1); sizes = _c(sizes, np.int32)
    N = len(poses)
    z = _c(z, MEAS)
    noise = None if noise is None else _c(noise, np.float32)
    maps_out = np.zeros(N * cap, GAUSSIAN); sizes_out = np.zeros(N, np.int32)
    idx = np.zeros(N, np.int32); ne = np.zeros(1, np.float32)
    rc = lib().o_step(_p(poses), _p(logw), _p(maps), _p(sizes), N, cap, float(alpha), float(v_encoder), _p(noise),
                      _p(z), len(z), C.byref(cfg), float(uniform), int(force_resample),
                      _p(maps_out), _p(sizes_out), _p(idx), _p(ne), int(n_threads))
    return dict(rc=rc, poses=poses, logw=logw, maps=maps_out.reshape(N, cap), sizes=sizes_out, idx=idx,
                neff=float(ne[0]))


def make_stepper(poses, logw, maps, sizes, cap, alpha, v_encoder, noise, z, cfg, uniform, force_resample, clutter_rate=None,
                 cn=None):
    """bench.py's cpu_baseline: -> f(n_threads) running ONE whole step (o_step, or o_cphd_step when cn is given) on the same
    inputs with every buffer allocated and touched beforehand (the timed call is the C routine, not numpy's page faults)"""
    poses0 = _c(poses, POSE); logw0 = _c(logw, np.float32)
    maps = _c(maps, GAUSSIAN).reshape(-1); sizes = _c(sizes, np.int32)
    N = len(poses0)
    z = _c(z, MEAS)
    noise = None if noise is None else _c(noise, np.float32)
    maps_out = np.ones(N * cap, GAUSSIAN); sizes_out = np.zeros(N, np.int32)
    idx = np.zeros(N, np.int32); ne = np.zeros(1, np.float32)
    po = poses0.copy(); lw = logw0.copy()
    if cn is not None:
        cn = _c(cn, np.float32); cn_out = np.ones_like(cn)
    L = lib()

    def run(n_threads):
        po[:] = poses0; lw[:] = logw0              # the step updates poses and weights in place
        if cn is not None:
            return L.o_cphd_step(_p(po), _p(lw), _p(maps), _p(sizes), N, cap, float(alpha), float(v_encoder), _p(noise), _p(z),
                                 len(z), C.byref(cfg), float(clutter_rate), _p(cn), cn.shape[1], float(uniform),
                                 int(force_resample), _p(maps_out), _p(sizes_out), _p(cn_out), _p(idx), _p(ne), int(n_threads))
        return L.o_step(_p(po), _p(lw), _p(maps), _p(sizes), N, cap, float(alpha), float(v_encoder), _p(noise), _p(z), len(z),
                        C.byref(cfg), float(uniform), int(force_resample), _p(maps_out), _p(sizes_out), _p(idx), _p(ne),
                        int(n_threads))
    return run


def cphd_step(poses, logw, maps, sizes, cap, alpha, v_encoder, noise, z, cfg, clutter_rate, cn, uniform, force_resample,
              n_threads=0):
    """o_step with the CPHD update; cn: [N, cn_len] log cardinalities.  Inputs are copied."""
    poses = _c(poses, POSE).copy(); logw = _c(logw, np.float32).copy()
    maps = _c(maps, GAUSSIAN).reshape(-1); sizes = _c(sizes, np.int32)
    cn = _c(cn, np.float32)
    N = len(poses)
    z = _c(z, MEAS)
    noise = None if noise is None else _c(noise, np.float32)
    maps_out = np.zeros(N * cap, GAUSSIAN); sizes_out = np.zeros(N, np.int32); cn_out = np.zeros_like(cn)
    idx = np.zeros(N, np.int32); ne = np.zeros(1, np.float32)
    rc = lib().o_cphd_step(_p(poses), _p(logw), _p(maps), _p(sizes), N, cap, float(alpha), float(v_encoder), _p(noise),
                           _p(z), len(z), C.byref(cfg), float(clutter_rate), _p(cn), cn.shape[1], float(uniform),
                           int(force_resample), _p(maps_out), _p(sizes_out), _p(cn_out), _p(idx), _p(ne), int(n_threads))
    return dict(rc=rc, poses=poses, logw=logw, maps=maps_out.reshape(N, cap), sizes=sizes_out, cn=cn_out, idx=idx,
                neff=float(ne[0]))
