"""ctypes binding of the multi-device host (include/phdslam_multi.h, libphdslam_multi.so = libphdslam.so + RCCL): ONE filter
sharded over the GPUs of this process, driven from C++.  Mirrors PhdFilter's interface for the global particle set.

The Python multi-process glue (dist.py, one process per GPU over torch.distributed) stays for launches under
torch.distributed.run; this is the host the `phdslam --devices N` binary uses."""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import GAUSSIAN, MEAS, NOISE, POSE, Control, PhdError, StepReport, check, ptr
from .filter import _ctrl

_HERE = os.path.dirname(os.path.abspath(__file__))
MULTI_LIB_PATH = os.path.join(_HERE, "libphdslam_multi.so")

TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER_COPY = 0, 1, 2
EXCHANGE_AUTO, EXCHANGE_GATHERED, EXCHANGE_ALLTOALL, EXCHANGE_PULL = 0, 1, 2, 3
EXCHANGE_NAMES = {1: "gathered", 2: "alltoall", 3: "pull"}


class MultiOptions(C.Structure):
    _fields_ = [("n_shards", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("map_capacity", C.c_int32),
                ("max_measurements", C.c_int32), ("survivor_capacity", C.c_int32), ("transport", C.c_int32),
                ("exchange", C.c_int32), ("gathered_limit_bytes", C.c_size_t), ("flags", C.c_uint32)]


FLAG_NO_PEER_ACCESS = 1


_vp, _i, _d, _sz, _u64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_uint64
# every symbol include/phdslam_multi.h declares
MULTI_SYMBOLS = {
    "phd_multi_create": (_i, [C.POINTER(L.SlamConfig), C.POINTER(MultiOptions), C.POINTER(_vp)]),
    "phd_multi_destroy": (_i, [_vp]),
    "phd_multi_n_shards": (_i, [_vp]),
    "phd_multi_n_particles": (_i, [_vp]),
    "phd_multi_n_particles_now": (_i, [_vp]),
    "phd_multi_uses_rccl": (_i, [_vp]),
    "phd_multi_exchange_is_gathered": (_i, [_vp]),
    "phd_multi_exchange": (_i, [_vp]),
    "phd_multi_shard": (_vp, [_vp, _i]),
    "phd_multi_seed": (_i, [_vp, _u64]),
    "phd_multi_set_config": (_i, [_vp, C.POINTER(L.SlamConfig)]),
    "phd_multi_set_frozen": (_i, [_vp, _i]),
    "phd_multi_sync": (_i, [_vp]),
    "phd_multi_set_particles": (_i, [_vp, _vp, _vp, _i]),
    "phd_multi_get_particles": (_i, [_vp, _vp, _vp]),
    "phd_multi_set_maps": (_i, [_vp, _vp, _vp]),
    "phd_multi_get_map_sizes": (_i, [_vp, _vp]),
    "phd_multi_get_maps": (_i, [_vp, _vp, _sz, _vp]),
    "phd_multi_step": (_i, [_vp, Control, _vp, _vp, _i, _d, _i, _vp]),
    "phd_multi_upload_inputs": (_i, [_vp, _vp, _vp, _i]),
    "phd_multi_step_resident": (_i, [_vp, Control, _d, _i, _vp]),
    "phd_multi_update": (_i, [_vp, C.POINTER(Control), _vp, _vp, _i]),
    "phd_multi_resample": (_i, [_vp, _d]),
    "phd_multi_state_snapshot": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "phd_multi_expected_map": (_i, [_vp, _vp, _i, _vp]),
    "phd_multi_timing_enable": (_i, [_vp, _i]),
    "phd_multi_timing_reset": (_i, [_vp]),
    "phd_multi_timing_read": (_i, [_vp, _vp, _vp]),
}
PHASES = ("local_step", "all_gather", "weights", "plan_export", "send_recv", "import")

_mlib = None


def mlib():
    global _mlib
    if _mlib is None:
        L.lib()                                       # libphdslam.so first (and torch's HIP runtime / RCCL, if torch is there)
        if not os.path.exists(MULTI_LIB_PATH):
            raise RuntimeError("libphdslam_multi.so is not built (%s): run __graft_entry__.build()" % MULTI_LIB_PATH)
        M = C.CDLL(MULTI_LIB_PATH)
        for name, (res, args) in MULTI_SYMBOLS.items():
            fn = getattr(M, name)
            fn.restype = res
            fn.argtypes = args
        _mlib = M
    return _mlib


class MultiFilter:
    """One filter of cfg.n_particles particles sharded over `n_shards` shards (devices[k] = HIP ordinal of shard k;
    shards that share a device exchange by device copies instead of RCCL)."""

    def __init__(self, cfg, n_shards=0, devices=None, map_capacity=256, max_measurements=256, survivor_capacity=0,
                 transport=TRANSPORT_AUTO, exchange=EXCHANGE_AUTO, gathered_limit_bytes=0, flags=0):
        self.cfg = cfg
        dv = None
        if devices is not None:
            dv = (C.c_int32 * len(devices))(*devices)
            n_shards = n_shards or len(devices)
        opt = MultiOptions(n_shards=int(n_shards), devices=dv, map_capacity=int(map_capacity),
                           max_measurements=int(max_measurements), survivor_capacity=int(survivor_capacity),
                           transport=int(transport), exchange=int(exchange), gathered_limit_bytes=int(gathered_limit_bytes),
                           flags=int(flags))
        h = C.c_void_p()
        check(mlib().phd_multi_create(C.byref(cfg), C.byref(opt), C.byref(h)), "phd_multi_create")
        self._h = h
        self.n = mlib().phd_multi_n_particles(h)
        self.n_shards = mlib().phd_multi_n_shards(h)
        self.cap = int(map_capacity)

    def close(self):
        if getattr(self, "_h", None):
            mlib().phd_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_now(self):
        """global particle count now (grows by n_predict_particles per predict until the resample that follows)"""
        return mlib().phd_multi_n_particles_now(self._h)

    @property
    def uses_rccl(self):
        return bool(mlib().phd_multi_uses_rccl(self._h))

    @property
    def gathered(self):
        return bool(mlib().phd_multi_exchange_is_gathered(self._h))

    @property
    def exchange(self):
        """"gathered" | "pull" | "alltoall": how a resampling step moves particles between shards"""
        return EXCHANGE_NAMES[mlib().phd_multi_exchange(self._h)]

    def seed(self, s):
        check(mlib().phd_multi_seed(self._h, int(s)), "phd_multi_seed")

    def set_frozen(self, freeze):
        check(mlib().phd_multi_set_frozen(self._h, int(freeze)), "phd_multi_set_frozen")

    def sync(self):
        check(mlib().phd_multi_sync(self._h), "phd_multi_sync")

    def set_particles(self, poses=None, log_weights=None):
        p = None if poses is None else np.ascontiguousarray(poses, POSE)
        w = None if log_weights is None else np.ascontiguousarray(log_weights, np.float32)
        check(mlib().phd_multi_set_particles(self._h, ptr(p), ptr(w), self.n_now), "phd_multi_set_particles")

    def get_particles(self):
        p = np.zeros(self.n_now, POSE)
        w = np.zeros(self.n_now, np.float32)
        check(mlib().phd_multi_get_particles(self._h, ptr(p), ptr(w)), "phd_multi_get_particles")
        return p, w

    def set_maps(self, maps, sizes):
        sizes = np.ascontiguousarray(sizes, np.int32)
        maps = np.ascontiguousarray(maps, GAUSSIAN)
        concat = np.ascontiguousarray(np.concatenate([maps[p, :sizes[p]] for p in range(len(sizes))]), GAUSSIAN)
        check(mlib().phd_multi_set_maps(self._h, ptr(concat), ptr(sizes)), "phd_multi_set_maps")

    def map_sizes(self):
        s = np.zeros(self.n_now, np.int32)
        check(mlib().phd_multi_get_map_sizes(self._h, ptr(s)), "phd_multi_get_map_sizes")
        return s

    def get_maps(self):
        sizes = self.map_sizes()
        concat = np.zeros(max(int(sizes.sum()), 1), GAUSSIAN)
        check(mlib().phd_multi_get_maps(self._h, ptr(concat), len(concat), ptr(sizes)), "phd_multi_get_maps")
        off = np.concatenate([[0], np.cumsum(sizes)])
        return [concat[off[p]:off[p + 1]] for p in range(len(sizes))]

    def step(self, control, noise, z, uniform, force_resample=False):
        """-> did_resample"""
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).view(NOISE).reshape(-1)
        zz = np.ascontiguousarray(z, MEAS)
        did = C.c_int32(0)
        check(mlib().phd_multi_step(self._h, _ctrl(control), ptr(nz), ptr(zz), len(zz), float(uniform), int(force_resample),
                                    C.byref(did)), "phd_multi_step")
        return bool(did.value)

    def update(self, control, noise, z):
        """predict (control None: no motion) + update + global normalisation, no resample"""
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).view(NOISE).reshape(-1)
        zz = np.ascontiguousarray(z, MEAS)
        cu = None if control is None else C.byref(_ctrl(control))
        check(mlib().phd_multi_update(self._h, cu, ptr(nz), ptr(zz), len(zz)), "phd_multi_update")

    def resample(self, uniform):
        check(mlib().phd_multi_resample(self._h, float(uniform)), "phd_multi_resample")

    def upload_inputs(self, noise, z):
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).view(NOISE).reshape(-1)
        zz = np.ascontiguousarray(z, MEAS)
        check(mlib().phd_multi_upload_inputs(self._h, ptr(nz), ptr(zz), len(zz)), "phd_multi_upload_inputs")

    def step_resident(self, control, uniform, force_resample=False):
        check(mlib().phd_multi_step_resident(self._h, _ctrl(control), float(uniform), int(force_resample), None),
              "phd_multi_step_resident")

    def shard_handle(self, k):
        """shard k's phd_filter handle (inspection: per-kernel timing, status)"""
        return C.c_void_p(mlib().phd_multi_shard(self._h, int(k)))

    def timing(self, enable):
        check(mlib().phd_multi_timing_enable(self._h, int(enable)), "phd_multi_timing_enable")

    def timing_reset(self):
        check(mlib().phd_multi_timing_reset(self._h), "phd_multi_timing_reset")

    def timing_read(self):
        """-> ({phase: mean microseconds per step}, steps)"""
        us = np.zeros(len(PHASES), np.float64)
        k = C.c_int64(0)
        check(mlib().phd_multi_timing_read(self._h, ptr(us), C.byref(k)), "phd_multi_timing_read")
        return {nm: float(us[i]) / max(k.value, 1) for i, nm in enumerate(PHASES)}, int(k.value)

    def state_snapshot(self):
        e = np.zeros(1, POSE)
        out = np.zeros(self.cap, GAUSSIAN)
        n, who = C.c_int32(0), C.c_int32(0)
        poses = np.zeros(self.n_now, POSE)
        lw = np.zeros(self.n_now, np.float32)
        rep = StepReport()
        check(mlib().phd_multi_state_snapshot(self._h, ptr(e), ptr(out), self.cap, C.byref(n), C.byref(who), ptr(poses), ptr(lw),
                                              C.byref(rep)), "phd_multi_state_snapshot")
        self.last_report = rep
        return e[0], out[:n.value].copy(), who.value, poses, lw

    def expected_map(self, capacity=4096):
        while True:
            out = np.zeros(max(capacity, 1), GAUSSIAN)
            n = C.c_int32(0)
            rc = mlib().phd_multi_expected_map(self._h, ptr(out), capacity, C.byref(n))
            if rc == -5 and n.value > capacity:
                capacity = n.value
                continue
            check(rc, "phd_multi_expected_map")
            return out[:n.value].copy()
