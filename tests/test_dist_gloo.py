"""Multi-rank path on CPU: world_size-2 (and 4) gloo runs of the same exchange code the GPU path uses
(cuda-phdslam_amd/dist.py), over a numpy stand-in backend.  Checks that after an all-gather +
global systematic resample + particle migration every rank holds exactly the particles a single
process would hold."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O

D = importlib.import_module("cuda-phdslam_amd.dist")


class NumpyShard:
    """ShardedFilter backend on host arrays (test stand-in for GpuShard)."""

    def __init__(self, poses, maps, raw_logw, n_global):
        self.poses = poses.copy()          # [n] float32 id per particle
        self.maps = [m.copy() for m in maps]
        self.raw = raw_logw.astype(np.float32)
        self.n_global = n_global
        self.logw = None
        self._all = None
        self._next = None

    def raw_logweights(self):
        return torch.from_numpy(self.raw.copy())

    def global_normalize(self, all_logw, want_neff=True):
        a = O.normalize_weights(all_logw.numpy())
        self._all = a
        return O.neff(a)

    def global_resample_indices(self, uniform):
        return O.resample(self._all, uniform)

    def pack_bytes(self):
        return 4 + 4 + 4 * 8  # id, size, 8 floats of payload

    def export_particles(self, which):
        buf = np.zeros((len(which), self.pack_bytes()), np.uint8)
        for k, p in enumerate(which):
            rec = np.zeros(10, np.float32)
            rec[0] = self.poses[p]
            rec[1] = len(self.maps[p])
            rec[2:2 + len(self.maps[p])] = self.maps[p]
            buf[k] = rec.view(np.uint8)
        return torch.from_numpy(buf)

    def apply_parents(self, local_parent):
        n = len(self.poses)
        self._next = ([None] * n, np.zeros(n, np.float32))
        for j, s in enumerate(local_parent):
            if s >= 0:
                self._next[0][j] = self.maps[s].copy()
                self._next[1][j] = self.poses[s]

    def import_particles(self, slots, buf):
        b = buf.numpy()
        for k, j in enumerate(slots):
            rec = b[k].view(np.float32)
            self._next[1][j] = rec[0]
            self._next[0][j] = rec[2:2 + int(rec[1])].copy()

    def finish_resample(self):
        self.maps, self.poses = self._next
        self.logw = np.full(len(self.poses), -np.log(self.n_global), np.float32)

    # the whole-shard ("gathered") exchange: rows of every local particle, the raw weight in the header
    def export_shard(self):
        rows = self.export_particles(range(len(self.poses))).numpy().copy()
        rows.view(np.float32)[:, 9] = self.raw                 # last payload word is free (maps hold < 8 values)
        return torch.from_numpy(rows)

    def resample_gathered(self, allrows, uniform, world, rank, weights_in_rows, want_idx):
        b = allrows.numpy()
        n = len(self.poses)
        if weights_in_rows:
            self._all = O.normalize_weights(b.view(np.float32)[:, 9].copy())
        idx = O.resample(self._all, uniform)
        self._next = ([None] * n, np.zeros(n, np.float32))
        self.import_particles(range(n), torch.from_numpy(b[idx[rank * n:(rank + 1) * n]]))
        self.finish_resample()
        return idx if want_idx else None


def _global_set(n_global, seed):
    rng = np.random.default_rng(seed)
    ids = np.arange(n_global, dtype=np.float32)
    maps = [rng.normal(size=rng.integers(0, 8)).astype(np.float32) for _ in range(n_global)]
    raw = rng.normal(0, 2.5, n_global).astype(np.float32)
    return ids, maps, raw


def _worker(rank, world, port, n_global, seed, uniform, out_dir, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, maps, raw = _global_set(n_global, seed)
    off, n = D.shard_range(n_global, world, rank)
    shard = NumpyShard(ids[off:off + n], maps[off:off + n], raw[off:off + n], n_global)
    sf = D.ShardedFilter(shard, n_global, rank, world)
    if exchange == "alltoall":
        sf.gathered_limit = 0
    assert sf.gathered() == (exchange != "alltoall")
    allw = sf.gather_logweights()
    assert np.array_equal(allw.numpy(), raw)                       # all-gather reassembles the global vector
    neff = sf.normalize(allw)
    if exchange == "gathered_rows":
        idx = sf.resample_gathered(uniform, weights_in_rows=True, want_idx=True)   # the bench's forced-resample step
    else:
        idx = sf.resample(uniform)                                 # "gathered": indices from what normalize() left
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ids=shard.poses, idx=idx, neff=neff, logw=shard.logw,
             sizes=np.array([len(m) for m in shard.maps]), flat=np.concatenate(shard.maps + [np.zeros(0, np.float32)]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_global,seed,exchange", [(2, 64, 1, "alltoall"), (2, 256, 2, "alltoall"), (4, 128, 3, "alltoall"),
                                                          (2, 64, 1, "gathered"), (4, 128, 3, "gathered_rows")])
def test_global_resample_and_migration(tmp_path, world, n_global, seed, exchange):
    uniform = 0.37
    mp.spawn(_worker, args=(world, _free_port(), n_global, seed, uniform, str(tmp_path), exchange), nprocs=world, join=True)
    ids, maps, raw = _global_set(n_global, seed)
    ref_lw = O.normalize_weights(raw)
    ref_idx = O.resample(ref_lw, uniform)
    n = n_global // world
    moved = 0
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(d["idx"], ref_idx)                   # identical indices on every rank
        assert abs(float(d["neff"]) - O.neff(ref_lw)) < 1e-6
        want = ref_idx[r * n:(r + 1) * n]
        assert np.array_equal(d["ids"], ids[want])                 # the right parents arrived
        assert np.all(d["logw"] == np.float32(-np.log(n_global)))
        off = np.concatenate([[0], np.cumsum(d["sizes"])])
        for j, g in enumerate(want):
            assert np.array_equal(d["flat"][off[j]:off[j + 1]], maps[g])
        moved += int(np.sum(want // n != r))
    assert moved > 0                                               # the test exercised remote parents


def test_plan_migration_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 4, 8):
        n_global = 64 * world
        idx = np.sort(rng.integers(0, n_global, n_global))         # non-decreasing like systematic resampling
        n = n_global // world
        plans = [D.plan_migration(idx, n_global, world, r) for r in range(world)]
        for r, (lp, send, recv) in enumerate(plans):
            # every slot is filled exactly once: locally or by exactly one remote rank
            filled = np.zeros(n, int)
            filled[lp >= 0] += 1
            for q in range(world):
                filled[recv[q]] += 1
                # what r expects from q is what q plans to send to r, in the same order
                assert len(plans[q][1][r]) == len(recv[q])
                if len(recv[q]):
                    assert np.array_equal(plans[q][1][r] + q * n, idx[r * n + recv[q]])
            assert np.all(filled == 1)
            assert np.array_equal(lp[lp >= 0] + r * n, idx[r * n:(r + 1) * n][lp >= 0])
