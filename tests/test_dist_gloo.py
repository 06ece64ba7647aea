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

    def import_particles(self, slots, buf, rows=None):
        b = buf.numpy()
        for k, j in enumerate(slots):
            rec = b[k if rows is None else rows[k]].view(np.float32)
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
        check_plans(idx, n_global, world)
    # degenerate weights, the case resampling exists for: three parents fill every slot; each travels once per destination
    world, n = 4, 64
    idx = np.sort(np.concatenate([np.full(150, 5), np.full(100, 70), np.full(6, 200)]))
    plans = check_plans(idx, world * n, world)
    assert [len(s) for s in plans[0][1]] == [0, 1, 1, 0]           # rank 0 owns particle 5: needed by ranks 1 and 2 (slots 64..149)
    assert [len(s) for s in plans[1][1]] == [0, 0, 1, 1]           # rank 1 owns particle 70: fills 150..249 (ranks 2 and 3)
    assert sum(len(s) for s in plans[3][2]) == 58 and len(np.unique(np.concatenate(plans[3][3]))) == 1   # slots 192..249 <- one row


def check_plans(idx, n_global, world):
    n = n_global // world
    plans = [D.plan_migration(idx, n_global, world, r) for r in range(world)]
    for r, (lp, send, recv, rows) in enumerate(plans):
        # every slot is filled exactly once: locally or by exactly one remote rank
        filled = np.zeros(n, int)
        filled[lp >= 0] += 1
        row0 = 0
        for q in range(world):
            filled[recv[q]] += 1
            if len(recv[q]):
                # the row a slot takes, counted inside q's block of the receive buffer, is where q put that slot's parent
                rel = rows[q] - row0
                assert rel.min() == 0 and rel.max() == len(plans[q][1][r]) - 1 and np.all(np.diff(rel) >= 0)
                assert np.array_equal(plans[q][1][r][rel] + q * n, idx[r * n + recv[q]])
            else:
                assert len(plans[q][1][r]) == 0
            row0 += len(plans[q][1][r])
        assert np.all(filled == 1)
        assert np.array_equal(lp[lp >= 0] + r * n, idx[r * n:(r + 1) * n][lp >= 0])
    return plans


# ----------------------------------------------------------------------------------------------
# the split bench.py --gpus N uses for BASELINE.json configs[3]: ONE generated particle set, rank r takes the contiguous
# slice [r n, (r + 1) n), the same measurement set and uniform on every rank (synthetic.shard_workload)
# ----------------------------------------------------------------------------------------------
def _split_worker(rank, world, port, N, G, M, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    P = importlib.import_module("cuda-phdslam_amd")
    from parity_utils import oracle_config_from
    ocfg = oracle_config_from(P.default_config())
    full = S.make_workload(N, G, M, seed=0x5EED0004, clustered=True)       # every rank generates the same set ...
    w = S.shard_workload(full, world, rank)                                 # ... and takes its slice
    off, n = D.shard_range(N, world, rank)
    assert (w["offset"], w["N"], w["N_global"]) == (off, n, N)
    # the local step of the shard on the CPU oracle: predict + update of its own particles, raw = logw + dlogw
    pred = O.predict_ackerman(w["poses"], 0.05, 2.0, w["noise"][0], ocfg)
    raw = np.array([w["logw"][p] + np.float32(O.update_particle(pred[p], w["maps"][p, :w["sizes"][p]], w["z"][0], ocfg)["dlogw"])
                    for p in range(n)], np.float32)
    shard = NumpyShard(np.arange(off, off + n, dtype=np.float32), [np.zeros(0, np.float32)] * n, raw, N)
    sf = D.ShardedFilter(shard, N, rank, world)
    sf.gathered_limit = 0
    allw = sf.gather_logweights()
    sf.normalize(allw)
    idx = sf.resample(float(w["uniform"][0]))
    np.savez(os.path.join(out_dir, "split%d.npz" % rank), raw=raw, allw=allw.numpy(), idx=idx, ids=shard.poses, z=w["z"][0].view(np.uint8),
             u=w["uniform"][0], first_pose=w["poses"]["px"][:1])
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_split_of_one_particle_set(tmp_path):
    world, N, G, M = 2, 32, 16, 8
    mp.spawn(_split_worker, args=(world, _free_port(), N, G, M, str(tmp_path)), nprocs=world, join=True)
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    P = importlib.import_module("cuda-phdslam_amd")
    from parity_utils import oracle_config_from
    ocfg = oracle_config_from(P.default_config())
    full = S.make_workload(N, G, M, seed=0x5EED0004, clustered=True)
    pred = O.predict_ackerman(full["poses"], 0.05, 2.0, full["noise"][0], ocfg)
    ref_raw = np.array([full["logw"][p] + np.float32(O.update_particle(pred[p], full["maps"][p, :G], full["z"][0], ocfg)["dlogw"])
                        for p in range(N)], np.float32)
    ref_idx = O.resample(O.normalize_weights(ref_raw), float(full["uniform"][0]))
    d = [np.load(tmp_path / ("split%d.npz" % r)) for r in range(world)]
    n = N // world
    for r in range(world):
        assert np.array_equal(d[r]["raw"], ref_raw[r * n:(r + 1) * n])           # the shard's step == the same particles of the one filter
        assert np.array_equal(d[r]["allw"], ref_raw)                               # the gathered vector is the one filter's
        assert np.array_equal(d[r]["idx"], ref_idx)                                # identical global indices on every rank
        assert np.array_equal(d[r]["ids"], ref_idx[r * n:(r + 1) * n].astype(np.float32))   # and the right parents arrived
        assert np.array_equal(d[r]["z"], d[0]["z"]) and d[r]["u"] == d[0]["u"]     # one scan, one uniform
    assert d[0]["first_pose"][0] == full["poses"]["px"][0] and d[1]["first_pose"][0] == full["poses"]["px"][n]
