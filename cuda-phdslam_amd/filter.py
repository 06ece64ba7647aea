"""Host-side mirror of the reference's filter interface (src/phdfilter.h:10-34) over the C-ABI.

    reference (C++)                                  here
    -----------------------------------------------  --------------------------------------------
    setDeviceConfig(config)                          PhdFilter(cfg, ...) / .set_config(cfg)
    initRandomNumberGenerators()                     .seed(s)
    phdPredict(particles, control)                   .predict(control, noise=None)
    phdUpdateSynth(particles, Z)                     .update(Z)
    nEff test + resampleParticles (main.cpp)         .neff() / .resample(u) / .resample_if_needed(u)
    recoverSlamState(particles, pose, cn)            .expected_pose() / .map_estimate()
    SynthSLAM fields                                 .set_particles/.get_particles/.set_maps/.get_maps

Every compute call runs the hand-written gfx950 kernels; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import GAUSSIAN, MEAS, NOISE, POSE, Control, Options, check, lib, ptr


def _ctrl(control):
    """control = (v_encoder, alpha) like one line of controls.txt (src/main.cpp:182)"""
    if isinstance(control, Control):
        return control
    v, a = control
    return Control(alpha=float(a), v_encoder=float(v))


class PhdFilter:
    def __init__(self, cfg=None, n_particles=None, map_capacity=256, max_measurements=256, survivor_capacity=0,
                 device=0, stream=None, global_particles=0, global_offset=0):
        self.cfg = cfg if cfg is not None else L.default_config()
        opt = Options(n_particles=int(n_particles or self.cfg.n_particles), map_capacity=int(map_capacity),
                      max_measurements=int(max_measurements), survivor_capacity=int(survivor_capacity),
                      device=int(device), stream=C.c_void_p(stream) if stream else None,
                      global_particles=int(global_particles), global_offset=int(global_offset))
        h = C.c_void_p()
        check(lib().phd_create(C.byref(self.cfg), C.byref(opt), C.byref(h)), "phd_create")
        self._h = h
        self.cap = lib().phd_map_capacity(h)

    @property
    def n(self):
        """current particle count (n_particles; grows by n_predict_particles per predict until a resample)"""
        return lib().phd_n_particles(self._h)

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().phd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_config(self, cfg):
        check(lib().phd_set_config(self._h, C.byref(cfg)), "phd_set_config")
        self.cfg = cfg

    def seed(self, s):
        check(lib().phd_seed(self._h, int(s)), "phd_seed")

    def sync(self):
        check(lib().phd_sync(self._h), "phd_sync")

    @property
    def stream(self):
        return lib().phd_stream(self._h)

    # -- state ---------------------------------------------------------------------------------
    def set_particles(self, poses=None, log_weights=None):
        p = None if poses is None else np.ascontiguousarray(poses, POSE)
        w = None if log_weights is None else np.ascontiguousarray(log_weights, np.float32)
        check(lib().phd_set_particles(self._h, ptr(p), ptr(w), self.n), "phd_set_particles")

    def get_particles(self):
        p = np.zeros(self.n, POSE)
        w = np.zeros(self.n, np.float32)
        check(lib().phd_get_particles(self._h, ptr(p), ptr(w)), "phd_get_particles")
        return p, w

    def set_maps(self, maps, sizes=None):
        """maps: list of per-particle GAUSSIAN arrays, or an [N, G] array with sizes[N]"""
        if sizes is None:
            sizes = np.array([len(m) for m in maps], np.int32)
            concat = np.concatenate([np.ascontiguousarray(m, GAUSSIAN) for m in maps]) if len(maps) else np.zeros(0, GAUSSIAN)
        else:
            sizes = np.ascontiguousarray(sizes, np.int32)
            maps = np.ascontiguousarray(maps, GAUSSIAN)
            concat = np.concatenate([maps[p, :sizes[p]] for p in range(len(sizes))])
        concat = np.ascontiguousarray(concat, GAUSSIAN)
        check(lib().phd_set_maps(self._h, ptr(concat), ptr(sizes)), "phd_set_maps")

    def map_sizes(self):
        s = np.zeros(self.n, np.int32)
        check(lib().phd_get_map_sizes(self._h, ptr(s)), "phd_get_map_sizes")
        return s

    def get_maps(self):
        sizes = self.map_sizes()
        concat = np.zeros(int(sizes.sum()), GAUSSIAN)
        check(lib().phd_get_maps(self._h, ptr(concat), len(concat), ptr(sizes)), "phd_get_maps")
        off = np.concatenate([[0], np.cumsum(sizes)])
        return [concat[off[p]:off[p + 1]] for p in range(self.n)]

    def set_map(self, particle, gmap):
        g = np.ascontiguousarray(gmap, GAUSSIAN)
        check(lib().phd_set_map(self._h, int(particle), ptr(g), len(g)), "phd_set_map")

    def get_map(self, particle):
        out = np.zeros(self.cap, GAUSSIAN)
        n = C.c_int32(0)
        check(lib().phd_get_map(self._h, int(particle), ptr(out), self.cap, C.byref(n)), "phd_get_map")
        return out[:n.value].copy()

    # -- hot path ------------------------------------------------------------------------------
    def predict(self, control, noise=None):
        """phdPredict(particles, control); noise[N] = (n_alpha, n_encoder) or None (device RNG)"""
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).view(NOISE).reshape(-1)
        k = max(1, int(self.cfg.nPredictParticles))
        if nz is not None and len(nz) != self.n * k:
            raise ValueError("noise must have n_particles * n_predict_particles entries")
        check(lib().phd_predict_ackerman(self._h, _ctrl(control), ptr(nz)), "phd_predict_ackerman")

    def predict_update(self, control, noise, z):
        """phdPredict + phdUpdateSynth of one step in ONE launch: the results of predict(control, noise) followed by update(z)"""
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).view(NOISE).reshape(-1)
        z = np.ascontiguousarray(z, MEAS)
        check(lib().phd_predict_update(self._h, _ctrl(control), ptr(nz), ptr(z), len(z)), "phd_predict_update")

    def update(self, z):
        """phdUpdateSynth(particles, Z)"""
        z = np.ascontiguousarray(z, MEAS)
        check(lib().phd_update(self._h, ptr(z), len(z)), "phd_update")

    def neff(self):
        v = C.c_float(0)
        check(lib().phd_neff(self._h, C.byref(v)), "phd_neff")
        return v.value

    def resample(self, uniforms):
        u = np.ascontiguousarray(np.atleast_1d(uniforms), np.float64)
        idx = np.zeros(self.n, np.int32)
        check(lib().phd_resample(self._h, ptr(u), len(u), ptr(idx)), "phd_resample")
        return idx[:self.n]  # a grown (shotgun) particle set shrinks back to n_particles

    def resample_if_needed(self, uniform, had_measurements=True):
        did = C.c_int32(0)
        idx = np.zeros(self.n, np.int32)
        check(lib().phd_resample_if_needed(self._h, float(uniform), int(had_measurements), C.byref(did), ptr(idx)),
              "phd_resample_if_needed")
        return bool(did.value), idx[:self.n]

    def expected_pose(self):
        p = np.zeros(1, POSE)
        check(lib().phd_expected_pose(self._h, ptr(p)), "phd_expected_pose")
        return p[0]

    def map_estimate(self):
        out = np.zeros(self.cap, GAUSSIAN)
        n = C.c_int32(0)
        who = C.c_int32(0)
        check(lib().phd_map_estimate(self._h, ptr(out), self.cap, C.byref(n), C.byref(who)), "phd_map_estimate")
        return out[:n.value].copy(), who.value

    def state_snapshot(self):
        """recoverSlamState in one call and one host synchronisation: (expected pose, arg-max particle's map, its index,
        all poses, all log-weights) — expected_pose() + map_estimate() + get_particles() in a single round trip"""
        e = np.zeros(1, POSE)
        out = np.zeros(self.cap, GAUSSIAN)
        n, who = C.c_int32(0), C.c_int32(0)
        poses = np.zeros(self.n, POSE)
        lw = np.zeros(self.n, np.float32)
        rep = L.StepReport()
        check(lib().phd_state_snapshot(self._h, ptr(e), ptr(out), self.cap, C.byref(n), C.byref(who), ptr(poses), ptr(lw),
                                       C.byref(rep)), "phd_state_snapshot")
        self.last_report = rep
        return e[0], out[:n.value].copy(), who.value, poses, lw

    # the same snapshot without the host synchronisation (include/phdslam.h: phd_snapshot_capture / _send / _wait)
    def snapshot_capture(self, slot):
        check(lib().phd_snapshot_capture(self._h, int(slot)), "phd_snapshot_capture")

    def snapshot_send(self, slot, want_resample_idx=False):
        check(lib().phd_snapshot_send(self._h, int(slot), 1 if want_resample_idx else 0), "phd_snapshot_send")

    def snapshot_wait(self, slot):
        """-> (expected pose, arg-max particle's map, its index, poses, log-weights, resample indices or None, report): copies
        of the slot's pinned block"""
        v = L.SnapshotView()
        check(lib().phd_snapshot_wait(self._h, int(slot), C.byref(v)), "phd_snapshot_wait")
        n = v.n_particles

        def arr(addr, dtype, count):
            return np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype, count=count).copy()
        idx = arr(v.resample_idx, np.int32, n) if v.resample_idx else None
        self.last_cardinality = arr(v.cardinality, np.float32, v.cardinality_len) if v.cardinality else None   # (CPHD: cn_estimate)
        self.last_report = v.report
        return (arr(v.expected, POSE, 1)[0], arr(v.map, GAUSSIAN, v.n_map), v.particle, arr(v.poses, POSE, n),
                arr(v.log_weights, np.float32, n), idx, v.report)

    def step_report(self):
        """status word, high-water marks, nEff and resample decision of the last weights routine: one download"""
        rep = L.StepReport()
        check(lib().phd_step_report_get(self._h, C.byref(rep)), "phd_step_report_get")
        return rep

    def set_particle_count(self, n):
        check(lib().phd_set_particle_count(self._h, int(n)), "phd_set_particle_count")

    def expected_map(self, capacity=None):
        """EAP map (config map_estimate & 2): computeExpectedMap, src/main.cpp:290-316, on the device"""
        capacity = int(capacity or 4 * self.cap)
        while True:
            out = np.zeros(max(capacity, 1), GAUSSIAN)
            n = C.c_int32(0)
            rc = lib().phd_expected_map(self._h, ptr(out), capacity, C.byref(n))
            if rc == -5 and n.value > capacity:
                capacity = n.value
                continue
            check(rc, "phd_expected_map")
            return out[:n.value].copy()

    # -- CPHD variant (filter_type = 1): per-particle log cardinality distributions --------------
    def cardinalities(self):
        k = lib().phd_cardinality_length(self._h)
        out = np.zeros((self.n, k), np.float32)
        check(lib().phd_get_cardinalities(self._h, ptr(out)), "phd_get_cardinalities")
        return out

    def set_cardinalities(self, cn):
        k = lib().phd_cardinality_length(self._h)
        cn = np.ascontiguousarray(cn, np.float32)
        assert cn.shape == (self.n, k), (cn.shape, (self.n, k))
        check(lib().phd_set_cardinalities(self._h, ptr(cn)), "phd_set_cardinalities")

    def cardinality_estimate(self):
        out = np.zeros(lib().phd_cardinality_length(self._h), np.float32)
        who = C.c_int32(0)
        check(lib().phd_cardinality_estimate(self._h, ptr(out), C.byref(who)), "phd_cardinality_estimate")
        return out, who.value

    def gm_reduce(self, comps, min_distance):
        """reduceGaussianMixture (src/gm_reduce.cpp:57-134) of an arbitrary mixture on this filter's device"""
        comps = np.ascontiguousarray(comps, dtype=GAUSSIAN)
        out = np.zeros(max(len(comps), 1), GAUSSIAN)
        n = C.c_int32(0)
        check(lib().phd_gm_reduce(self._h, ptr(comps), len(comps), float(min_distance), ptr(out), len(out), C.byref(n)),
              "phd_gm_reduce")
        return out[:n.value].copy()

    def expected_map_concat_dev(self):
        """-> (device pointer of the [6][total] weighted concatenation, total)"""
        d = C.c_void_p()
        total = C.c_int64(0)
        check(lib().phd_expected_map_concat_dev(self._h, C.byref(d), C.byref(total)), "phd_expected_map_concat_dev")
        return d.value or 0, total.value

    def gm_reduce_dev(self, d_planes, total, n_planes, min_distance, capacity=4096):
        while True:
            out = np.zeros(max(capacity, 1), GAUSSIAN)
            n = C.c_int32(0)
            rc = lib().phd_gm_reduce_dev(self._h, ptr(d_planes), int(total), int(n_planes), float(min_distance), ptr(out),
                                         capacity, C.byref(n))
            if rc == -5 and n.value > capacity:
                capacity = n.value
                continue
            check(rc, "phd_gm_reduce_dev")
            return out[:n.value].copy()

    def gm_rounds(self):
        return lib().phd_debug_gm_rounds(self._h)

    # -- device-resident variants (raw device pointers as ints) --------------------------------
    def predict_dev(self, control, d_noise):
        check(lib().phd_predict_ackerman_dev(self._h, _ctrl(control), ptr(d_noise)), "phd_predict_ackerman_dev")

    def update_dev(self, d_z, n_meas):
        check(lib().phd_update_dev(self._h, ptr(d_z), int(n_meas)), "phd_update_dev")

    def step_dev(self, control, d_noise, d_z, n_meas, uniform, force_resample=False):
        check(lib().phd_step_dev(self._h, _ctrl(control), ptr(d_noise), ptr(d_z), int(n_meas), float(uniform),
                                 int(force_resample)), "phd_step_dev")

    def set_frozen(self, freeze):
        check(lib().phd_set_frozen(self._h, int(freeze)), "phd_set_frozen")

    # -- instrumentation -----------------------------------------------------------------------
    def timing(self, enable):
        check(lib().phd_timing_enable(self._h, int(enable)), "phd_timing_enable")

    def timing_reset(self):
        check(lib().phd_timing_reset(self._h), "phd_timing_reset")

    def timing_read(self):
        ms = np.zeros(L.K_COUNT, np.float64)
        n = np.zeros(L.K_COUNT, np.int64)
        check(lib().phd_timing_read(self._h, ptr(ms), ptr(n)), "phd_timing_read")
        return ms, n

    def debug(self, enable=True):
        check(lib().phd_debug_enable(self._h, int(enable)), "phd_debug_enable")

    def stamps(self):
        """phase stamps of the last update (after debug(2)): [n, 32] ticks of 10 ns (0..11 phase boundaries, 12..15 merge-round
        sums, 16.. finer sums inside the rounds and pass 1)"""
        out = np.zeros(self.n * 32 + 8, np.uint64)
        check(lib().phd_debug_get_stamps(self._h, ptr(out)), "phd_debug_get_stamps")
        self.weight_stamps = out[self.n * 32:]     # phases of the weights/resample kernel
        return out[:self.n * 32].reshape(self.n, 32)

    def survivors(self, particle):
        """pruned update components (+ nearly-in-range features) of one particle in slab order"""
        n = C.c_int32(0)
        check(lib().phd_debug_get_survivors(self._h, int(particle), None, None, 0, C.byref(n)), "phd_debug_get_survivors")
        out = np.zeros(max(n.value, 1), GAUSSIAN)
        sidx = np.zeros(max(n.value, 1), np.int32)
        check(lib().phd_debug_get_survivors(self._h, int(particle), ptr(out), ptr(sidx), len(out), C.byref(n)),
              "phd_debug_get_survivors")
        return out[:n.value], sidx[:n.value]

    def weight_increments(self):
        d = np.zeros(self.n, np.float32)
        check(lib().phd_debug_get_weight_increments(self._h, ptr(d)), "phd_debug_get_weight_increments")
        return d

    def residency(self):
        """-> dict(workgroups_per_cu, lds_bytes): how the update kernel sits on a CU (phd_update_residency)"""
        n = C.c_int32(0)
        b = C.c_uint64(0)
        check(lib().phd_update_residency(self._h, C.byref(n), C.byref(b)), "phd_update_residency")
        return dict(workgroups_per_cu=n.value, lds_bytes=b.value)

    def status(self, raise_on_overflow=True):
        st = C.c_uint32(0)
        ms = C.c_int32(0)
        mm = C.c_int32(0)
        rc = lib().phd_device_status(self._h, C.byref(st), C.byref(ms), C.byref(mm))
        if rc != 0 and raise_on_overflow:
            check(rc, "phd_device_status")
        return dict(status=st.value, max_survivors=ms.value, max_map=mm.value)


# ----------------------------------------------------------------------------------------------
# host-side boundary helpers (no device)
# ----------------------------------------------------------------------------------------------
def load_config(path):
    cfg = L.SlamConfig()
    ddir = C.create_string_buffer(4096)
    nsteps = C.c_int32(-1)
    check(lib().phd_config_load(path.encode(), C.byref(cfg), ddir, 4096, C.byref(nsteps)), "phd_config_load")
    return cfg, ddir.value.decode(), nsteps.value


def load_measurements(path, triples=False):
    ns = C.c_size_t(0)
    nt = C.c_size_t(0)
    check(lib().phd_load_measurements(path.encode(), int(triples), None, 0, None, 0, C.byref(ns), C.byref(nt)),
          "phd_load_measurements")
    out = np.zeros(max(nt.value, 1), MEAS)
    sizes = np.zeros(max(ns.value, 1), np.int32)
    check(lib().phd_load_measurements(path.encode(), int(triples), ptr(out), len(out), ptr(sizes), len(sizes),
                                      C.byref(ns), C.byref(nt)), "phd_load_measurements")
    sizes = sizes[:ns.value]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [out[off[k]:off[k + 1]] for k in range(len(sizes))]


def load_controls(path, has_header=-1):
    n = C.c_size_t(0)
    check(lib().phd_load_controls(path.encode(), int(has_header), None, 0, C.byref(n)), "phd_load_controls")
    out = np.zeros(max(n.value, 1), np.dtype([("alpha", np.float32), ("v_encoder", np.float32)]))
    check(lib().phd_load_controls(path.encode(), int(has_header), ptr(out), len(out), C.byref(n)), "phd_load_controls")
    return out[:n.value]


def write_state_log(directory, step, expected_pose, gmap, log_weights, poses, max_cardinality=255):
    e = np.ascontiguousarray(expected_pose, POSE).reshape(1)
    g = np.ascontiguousarray(gmap, GAUSSIAN)
    w = np.ascontiguousarray(log_weights, np.float32)
    p = np.ascontiguousarray(poses, POSE)
    check(lib().phd_write_state_log(directory.encode(), int(step), ptr(e), ptr(g), len(g), ptr(w), ptr(p), len(p),
                                    int(max_cardinality)), "phd_write_state_log")
