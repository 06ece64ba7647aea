"""Multi-GPU host glue: particles shard contiguously over ranks, one process per
GPU, torch.distributed (backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).

Per step and rank:
    predict -> update + prune + merge of the local shard      (no communication)
    all-gather of the un-normalised log-weights                (RCCL; N floats in total)
    identical normalise / nEff / resample-index routine on every rank (bit-identical results)
    particle migration for parents that live on another rank   (all_to_all of packed particles,
                                                                point-to-point over xGMI)
Small shards (all ranks' packed particles together <= ShardedFilter.gathered_limit bytes) use the "gathered" exchange
instead: ONE fixed-size all-gather of every rank's whole shard (the raw weight rides in the row header), then each rank
normalises, draws the indices and fills its slots from the gathered rows on the device — no host round trip at all.

The exchange logic is written against a small backend interface so that the same code runs on the
gfx950 filter (GpuShard) and — in the world_size-2 gloo tests — on a numpy stand-in.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except ImportError:  # pragma: no cover
    torch = None
    dist = None


def shard_range(n_global, world, rank):
    """contiguous blocks: rank r owns [r*n/P, (r+1)*n/P)"""
    if n_global % world:
        raise ValueError("n_global must be divisible by the world size")
    n = n_global // world
    return rank * n, n


def plan_migration(idx, n_global, world, rank):
    """Given the global parent indices (identical on every rank) return this rank's part of the
    exchange:
        local_parent[j]  : local index of slot j's parent, or -1 if the parent is remote
        send[r]          : local particle indices to export to rank r (in the order r expects)
        recv_slots[r]    : local slots filled by what rank r sends, in slot order
        recv_rows[r]     : for each of those slots, the row of the receive buffer (rows grouped by source rank,
                           ascending) that fills it
    Slot j of rank q (global index g = q*n + j) receives particle idx[g]; rank idx[g] // n owns it.
    A parent travels ONCE per destination rank however many of its slots it fills (resampling is called when the
    weights have degenerated: a few parents fill most slots): both sides walk the destination's slots in order and skip
    a parent equal to the previous one from the same source — systematic indices are non-decreasing, so that is every
    repeat; were they not, the rule is still the same on both sides.  The same plan as phd_global_resample_plan
    (csrc/phd_api.cpp).
    """
    off, n = shard_range(n_global, world, rank)
    idx = np.asarray(idx, np.int64)
    owner = idx // n
    dest_rank = np.arange(n_global) // n
    local_parent = np.full(n, -1, np.int32)
    mine = idx[off:off + n]
    is_local = owner[off:off + n] == rank
    local_parent[is_local] = (mine[is_local] - off).astype(np.int32)

    def first_of_run(a):
        keep = np.ones(len(a), bool)
        keep[1:] = a[1:] != a[:-1]
        return keep

    send, recv_slots, recv_rows = [], [], []
    row0 = 0
    for r in range(world):
        if r == rank:
            send.append(np.zeros(0, np.int32))
            recv_slots.append(np.zeros(0, np.int32))
            recv_rows.append(np.zeros(0, np.int32))
            continue
        # what rank r needs from me: its slots whose parent I own, in slot order, runs of one parent collapsed
        need = idx[(dest_rank == r) & (owner == rank)]
        send.append((need[first_of_run(need)] - off).astype(np.int32))
        # what I need from rank r: my slots whose parent r owns, in slot order; a run shares one row
        got = np.nonzero(owner[off:off + n] == r)[0]
        new = first_of_run(mine[got])
        recv_slots.append(got.astype(np.int32))
        recv_rows.append((row0 + np.cumsum(new) - 1).astype(np.int32))
        row0 += int(new.sum())
    return local_parent, send, recv_slots, recv_rows


class ShardedFilter:
    """One rank's view of a particle filter sharded over `world` ranks."""

    def __init__(self, backend, n_global, rank, world, group=None):
        self.b = backend
        self.n_global = n_global
        self.rank = rank
        self.world = world
        self.group = group
        self.offset, self.n = shard_range(n_global, world, rank)
        # a one-rank group needs no collective; tests set this on a one-rank RCCL group to run the collective code
        # paths (buffers, split sizes, empty exchanges) on a single GPU
        self.collectives = world > 1
        # all ranks' packed particles up to this many bytes: whole-shard all-gather; above: all-to-all of the migrants
        self.gathered_limit = 32 << 20

    def gather_logweights(self):
        raw = self.b.raw_logweights()                       # tensor [n] on the backend's device
        if not self.collectives:
            return raw.clone()
        if dist.get_backend(self.group) == "gloo" and raw.is_cuda:
            # test transport (several ranks sharing one GPU): stage through host memory
            host = torch.empty(self.n_global, dtype=raw.dtype)
            dist.all_gather_into_tensor(host, raw.cpu().contiguous(), group=self.group)
            return host.to(raw.device)
        # the gather buffer lives as long as the filter (one allocation, not one per step); the kernels that read it
        # are stream-ordered before the next step's gather overwrites it
        allw = getattr(self, "_allw", None)
        if allw is None or allw.device != raw.device:
            allw = self._allw = torch.empty(self.n_global, dtype=raw.dtype, device=raw.device)
        dist.all_gather_into_tensor(allw, raw, group=self.group)
        return allw

    def normalize(self, all_logw, want_neff=True):
        """-> nEff of the global particle set (None if not wanted: saves a host synchronisation);
        the shard's normalised weights are adopted"""
        return self.b.global_normalize(all_logw, want_neff)

    def _exchange(self, out_buf, send_counts, recv_counts, pack):
        """rows of out_buf grouped by destination rank -> rows grouped by source rank"""
        if dist.get_backend(self.group) == "gloo":
            # gloo has no all_to_all: pairwise exchange through host memory (CPU tests, and several
            # ranks sharing one GPU); rank order breaks the send/recv symmetry
            host_out = out_buf.cpu()
            so = np.concatenate([[0], np.cumsum(send_counts)])
            pieces = []
            for r in range(self.world):
                piece = torch.empty((recv_counts[r], pack), dtype=torch.uint8)
                if r != self.rank:
                    mine = host_out[so[r]:so[r + 1]].contiguous()
                    if self.rank < r:
                        dist.send(mine, r, group=self.group)
                        dist.recv(piece, r, group=self.group)
                    else:
                        dist.recv(piece, r, group=self.group)
                        dist.send(mine, r, group=self.group)
                pieces.append(piece)
            return torch.cat(pieces).to(out_buf.device)
        # a shard receives at most one particle per slot: one receive buffer of n rows for the whole run
        in_buf = getattr(self, "_recv", None)
        if in_buf is None or in_buf.shape[1] != pack or in_buf.device != out_buf.device:
            in_buf = self._recv = torch.empty((max(self.n, 1), pack), dtype=torch.uint8, device=out_buf.device)
        dist.all_to_all_single(in_buf[:sum(recv_counts)], out_buf, output_split_sizes=recv_counts,
                               input_split_sizes=send_counts, group=self.group)
        return in_buf

    def gathered(self):
        """True when the resampling exchange is the whole-shard all-gather (the same answer on every rank)"""
        return (self.collectives and hasattr(self.b, "export_shard")
                and self.n_global * self.b.pack_bytes() <= self.gathered_limit)

    def _gather_rows(self, rows):
        """[n, pack] rows of this rank -> [n_global, pack] rows of all ranks, rank order"""
        if dist.get_backend(self.group) == "gloo":
            # test transport (CPU stand-in, or several ranks sharing one GPU): through host memory
            host = torch.empty((self.n_global, rows.shape[1]), dtype=torch.uint8)
            dist.all_gather_into_tensor(host.view(-1), rows.cpu().contiguous().view(-1), group=self.group)
            return host.to(rows.device)
        allrows = getattr(self, "_allrows", None)              # allocated once; stream-ordered reuse as for _allw
        if allrows is None or allrows.shape[1] != rows.shape[1] or allrows.device != rows.device:
            allrows = self._allrows = torch.empty((self.n_global, rows.shape[1]), dtype=torch.uint8, device=rows.device)
            if hasattr(self.b, "set_rows_target"):
                # from the next step on the local step writes its rows straight into this rank's segment: the all-gather
                # is in place (RCCL skips the self copy)
                self.b.set_rows_target(allrows[self.rank * self.n:(self.rank + 1) * self.n])
        dist.all_gather_into_tensor(allrows.view(-1), rows.reshape(-1), group=self.group)
        return allrows

    def resample_gathered(self, uniform, weights_in_rows=True, want_idx=False, rows=None):
        """whole-shard all-gather form of resample(): export -> ONE collective -> normalise + indices + import on the
        device.  weights_in_rows: normalise the raw weights carried by the rows (forced resample straight after the
        local step: no separate weight gather, no normalize()); False: indices from what normalize() left.
        rows: what the backend's step_local_rows() returned (the local step already wrote the export rows)."""
        if rows is None:
            rows = self.b.export_shard()
        return self.b.resample_gathered(self._gather_rows(rows), uniform, self.world, self.rank, weights_in_rows, want_idx)

    def resample(self, uniform, all_raw_logw=None):
        """global systematic resample + migration.  Returns the global parent indices.
        all_raw_logw: the gathered UN-normalised weights — normalisation and indices then come from one launch
        (forced resample; skip normalize()).  Without it the indices come from what normalize() left on every rank:
        call normalize() after the last change of the weights (the C++ host, csrc/phd_multi.cpp, gathers the current weights
        itself when no normalisation precedes)"""
        if self.gathered():
            return self.resample_gathered(uniform, all_raw_logw is not None, want_idx=True)
        if self.collectives and hasattr(self.b, "resample_begin"):
            # library path: one call plans (in C++) and exports, one collective, one call imports and commits
            send_counts, recv_counts, send, idx = self.b.resample_begin(uniform, self.world, self.rank, all_raw_logw)
            recv = self._exchange(send[:sum(send_counts)], send_counts, recv_counts, self.b.pack_bytes())
            self.b.resample_end(recv)
            return idx
        idx = self.b.global_resample_indices(uniform)       # numpy [n_global], identical on all ranks
        local_parent, send, recv_slots, recv_rows = plan_migration(idx, self.n_global, self.world, self.rank)
        if self.collectives:
            send_counts = [len(s) for s in send]
            recv_counts = [int(len(np.unique(r))) for r in recv_rows]
            order = np.concatenate(send) if sum(send_counts) else np.zeros(0, np.int32)
            out_buf = self.b.export_particles(order)        # tensor [n_send, pack] (uint8)
            in_buf = self._exchange(out_buf, send_counts, recv_counts, self.b.pack_bytes())
            self.b.apply_parents(local_parent)
            slots = np.concatenate(recv_slots) if sum(recv_counts) else np.zeros(0, np.int32)
            rows = np.concatenate(recv_rows) if sum(recv_counts) else np.zeros(0, np.int32)
            self.b.import_particles(slots, in_buf, rows)
        else:
            self.b.apply_parents(local_parent)
        self.b.finish_resample()
        return idx


def _expected_map(self, min_distance):
    """EAP map of the GLOBAL particle set (computeExpectedMap, src/main.cpp:290-316): every rank
    concatenates its own weighted maps on its device, the shards' concatenations are all-gathered
    (ragged: padded to the longest) in rank order = global particle order, and each rank reduces
    the same global mixture — identical results everywhere, no rank is special."""
    planes = self.b.expected_map_concat()                   # tensor [6, total_r]
    if self.collectives:
        gloo = dist.get_backend(self.group) == "gloo"
        dev = planes.device
        t = torch.tensor([planes.shape[1]], dtype=torch.int64, device="cpu" if gloo else dev)
        totals = torch.empty(self.world, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(totals, t, group=self.group)
        totals = [int(x) for x in totals.cpu()]
        width = max(totals)
        if width == 0:
            return self.b.gm_reduce_planes(planes, min_distance)
        mine = torch.zeros((6, width), dtype=planes.dtype, device="cpu" if gloo else dev)
        mine[:, :planes.shape[1]] = planes.cpu() if gloo else planes
        allp = torch.empty(self.world * 6 * width, dtype=planes.dtype, device=mine.device)
        dist.all_gather_into_tensor(allp, mine.contiguous().view(-1), group=self.group)
        allp = allp.view(self.world, 6, width)
        planes = torch.cat([allp[r, :, :totals[r]] for r in range(self.world)], dim=1).contiguous().to(dev)
    return self.b.gm_reduce_planes(planes, min_distance)


ShardedFilter.expected_map = _expected_map


class GpuShard:
    """ShardedFilter backend over the gfx950 filter (C-ABI calls; torch only wraps device memory)."""

    def __init__(self, flt, n_global):
        import ctypes as C
        from ._lib import check, lib, ptr
        self._C, self._check, self._lib, self._ptr = C, check, lib, ptr
        self.f = flt
        self.n_global = n_global
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._all = None

    # The filter enqueues on its own HIP stream unless it was created on torch's current stream
    # (bench.py does that: RCCL collectives and kernels are then ordered by the stream itself).
    # With different streams the hand-offs are ordered by host synchronisation.
    def _same_stream(self):
        return int(self.f.stream or 0) == int(torch.cuda.current_stream().cuda_stream)

    def _torch_to_filter(self):
        if not self._same_stream():
            torch.cuda.current_stream().synchronize()

    def _filter_to_torch(self):
        if not self._same_stream():
            self.f.sync()

    def _wrap(self, ptr_value, n, dtype=torch.float32):
        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr_value), False), "version": 2}
        return torch.as_tensor(h, device=self.device)

    def raw_logweights(self):
        cached = getattr(self, "_raw", None)
        if cached is None or cached[0] != self.f.n:          # the device buffer is fixed for the filter's lifetime
            p = self._C.c_void_p()
            self._check(self._lib().phd_raw_logweights_dev(self.f._h, self._C.byref(p)), "phd_raw_logweights_dev")
            cached = self._raw = (self.f.n, self._wrap(p.value, self.f.n))
        self._filter_to_torch()
        return cached[1]

    def update_local_dev(self, d_z, n_meas):
        self._check(self._lib().phd_update_local_dev(self.f._h, self._ptr(d_z), int(n_meas)), "phd_update_local_dev")

    def global_normalize(self, all_logw, want_neff=True):
        self._all = all_logw
        self._torch_to_filter()
        ne = self._C.c_float(0)
        self._check(self._lib().phd_global_normalize(self.f._h, self._ptr(all_logw.data_ptr()), self.n_global,
                                                     self._C.byref(ne) if want_neff else None), "phd_global_normalize")
        return ne.value if want_neff else None

    def global_resample_indices(self, uniform):
        u = np.ascontiguousarray(np.atleast_1d(uniform), np.float64)
        idx = np.zeros(self.n_global, np.int32)
        self._check(self._lib().phd_global_resample_indices(self.f._h, None, self.n_global, self._ptr(u), len(u),
                                                            self._ptr(idx)), "phd_global_resample_indices")
        return idx

    def pack_bytes(self):
        return int(self._lib().phd_particle_pack_bytes(self.f._h))

    def export_particles(self, which):
        which = np.ascontiguousarray(which, np.int32)
        buf = torch.empty((len(which), self.pack_bytes()), dtype=torch.uint8, device=self.device)
        self._check(self._lib().phd_export_particles_dev(self.f._h, self._ptr(which), len(which),
                                                         self._ptr(buf.data_ptr())), "phd_export_particles_dev")
        self._filter_to_torch()
        return buf

    def apply_parents(self, local_parent):
        lp = np.ascontiguousarray(local_parent, np.int32)
        self._check(self._lib().phd_apply_parents(self.f._h, self._ptr(lp)), "phd_apply_parents")

    def import_particles(self, slots, buf, rows=None):
        """slot slots[k] <- row rows[k] of buf (rows None: row k)"""
        slots = np.ascontiguousarray(slots, np.int32)
        rows = None if rows is None else np.ascontiguousarray(rows, np.int32)
        self._torch_to_filter()
        self._check(self._lib().phd_import_particles_sel_dev(self.f._h, self._ptr(slots), None if rows is None else self._ptr(rows),
                                                             len(slots), self._ptr(buf.data_ptr())), "phd_import_particles_sel_dev")

    def finish_resample(self):
        self._check(self._lib().phd_finish_resample(self.f._h), "phd_finish_resample")

    def step_local_dev(self, control, d_noise, d_z, n_meas):
        from .filter import _ctrl
        self._check(self._lib().phd_step_local_dev(self.f._h, _ctrl(control), self._ptr(d_noise), self._ptr(d_z), int(n_meas)),
                    "phd_step_local_dev")

    def resample_begin(self, uniform, world, rank, all_raw_logw=None):
        C = self._C
        sc = (C.c_int32 * world)()
        rc = (C.c_int32 * world)()
        buf = C.c_void_p()
        idx = np.zeros(self.n_global, np.int32)
        self._torch_to_filter()
        raw = self._ptr(all_raw_logw.data_ptr()) if all_raw_logw is not None else None
        self._check(self._lib().phd_global_resample_begin(self.f._h, raw, float(uniform), int(world), int(rank), sc, rc,
                                                          C.byref(buf), self._ptr(idx)), "phd_global_resample_begin")
        sc, rc = list(sc), list(rc)
        self._filter_to_torch()
        pack = self.pack_bytes()
        n_send = max(sum(sc), 1)
        cached = getattr(self, "_send", None)                 # the library's send buffer only moves when it grows
        if cached is None or cached[0] != buf.value or cached[1] < n_send:
            rows = max(n_send, self.f.n)
            cached = self._send = (buf.value, rows, self._wrap(buf.value, rows * pack // 4).view(torch.uint8).view(rows, pack))
        return sc, rc, cached[2], idx

    def step_local_rows(self, control, d_noise, d_z, n_meas):
        """local step whose outputs land directly in the export rows (one launch); complete it with resample_gathered"""
        from .filter import _ctrl
        C = self._C
        p, nb = C.c_void_p(), C.c_size_t()
        self._check(self._lib().phd_step_local_rows_dev(self.f._h, _ctrl(control), self._ptr(d_noise), self._ptr(d_z), int(n_meas),
                                                        C.byref(p), C.byref(nb)), "phd_step_local_rows_dev")
        return self._rows_tensor(p.value, nb.value)

    def set_rows_target(self, segment):
        """step_local_rows() writes its rows into `segment` ([n, pack] uint8 on the device, kept alive here) from now on"""
        assert segment.is_contiguous() and segment.numel() == self.f.n * self.pack_bytes()
        self._rows_target = segment
        self._check(self._lib().phd_set_rows_target(self.f._h, self._ptr(segment.data_ptr())), "phd_set_rows_target")

    def export_shard(self):
        C = self._C
        p, nb = C.c_void_p(), C.c_size_t()
        self._check(self._lib().phd_export_shard_dev(self.f._h, C.byref(p), C.byref(nb)), "phd_export_shard_dev")
        return self._rows_tensor(p.value, nb.value)

    def _rows_tensor(self, address, n_bytes):
        cached = getattr(self, "_rows", None)                # the library's row buffer only moves when it grows
        if cached is None or cached[0] != address or cached[1] != n_bytes:
            pack = self.pack_bytes()
            cached = self._rows = (address, n_bytes, self._wrap(address, n_bytes // 4).view(torch.uint8).view(n_bytes // pack, pack))
        self._filter_to_torch()
        return cached[2]

    def resample_gathered(self, allrows, uniform, world, rank, weights_in_rows, want_idx):
        idx = np.zeros(self.n_global, np.int32) if want_idx else None
        self._torch_to_filter()
        self._check(self._lib().phd_global_resample_gathered(self.f._h, self._ptr(allrows.data_ptr()), float(uniform), int(world),
                                                             int(rank), int(bool(weights_in_rows)),
                                                             self._ptr(idx) if want_idx else None),
                    "phd_global_resample_gathered")
        return idx

    def resample_end(self, recv):
        self._torch_to_filter()
        self._check(self._lib().phd_global_resample_end(self.f._h, self._ptr(recv.data_ptr())), "phd_global_resample_end")

    def expected_map_concat(self):
        d, total = self.f.expected_map_concat_dev()          # synchronises the filter's stream
        if total == 0:
            return torch.zeros((6, 0), dtype=torch.float32, device=self.device)
        return self._wrap(d, 6 * total).view(6, total).clone()

    def gm_reduce_planes(self, planes, min_distance):
        total = planes.shape[1]
        self._torch_to_filter()
        return self.f.gm_reduce_dev(planes.data_ptr() if total else None, total, 6, min_distance)
