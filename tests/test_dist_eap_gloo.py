"""The ragged all-gather of the multi-rank expected-map path (ShardedFilter.expected_map) on CPU:
world_size-2 and -3 gloo runs over a numpy stand-in backend whose reduction is the oracle; every
rank must return the map a single process computes from the global particle set
(computeExpectedMap, src/main.cpp:290-316)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from test_dist_gloo import _free_port

D = importlib.import_module("cuda-phdslam_amd.dist")


def _global_maps(n_global, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-30, 30, (12, 2))
    maps = []
    for p in range(n_global):
        k = int(rng.integers(0, 9)) if p % 5 else 0            # some particles carry empty maps
        g = np.zeros(k, O.GAUSSIAN)
        pick = rng.integers(0, 12, k)
        g["mean"] = (c[pick] + 0.1 * rng.standard_normal((k, 2))).astype(np.float32)
        s = rng.uniform(0.05, 0.3, (k, 2))
        g["cov"][:, 0] = s[:, 0] ** 2
        g["cov"][:, 3] = s[:, 1] ** 2
        g["cov"][:, 1] = g["cov"][:, 2] = 0.3 * s[:, 0] * s[:, 1]
        g["weight"] = rng.uniform(0.01, 1.0, k).astype(np.float32)
        maps.append(g)
    logw = O.normalize_weights(rng.normal(0, 1.5, n_global).astype(np.float32))
    return maps, logw


class EapShard:
    def __init__(self, maps, logw):
        self.maps, self.logw = maps, logw

    def expected_map_concat(self):
        cat = np.concatenate(self.maps) if len(self.maps) else np.zeros(0, O.GAUSSIAN)
        planes = np.zeros((6, len(cat)), np.float32)
        f = np.concatenate([np.full(len(m), np.float32(O.det_exp(lw)), np.float32) for m, lw in zip(self.maps, self.logw)]
                           + [np.zeros(0, np.float32)])
        planes[0] = cat["weight"] * f
        planes[1], planes[2] = cat["mean"][:, 0], cat["mean"][:, 1]
        planes[3], planes[4], planes[5] = cat["cov"][:, 0], cat["cov"][:, 1], cat["cov"][:, 3]
        return torch.from_numpy(planes)

    def gm_reduce_planes(self, planes, min_distance):
        p = planes.numpy()
        g = np.zeros(p.shape[1], O.GAUSSIAN)
        g["weight"] = p[0]
        g["mean"][:, 0], g["mean"][:, 1] = p[1], p[2]
        g["cov"][:, 0], g["cov"][:, 1], g["cov"][:, 2], g["cov"][:, 3] = p[3], p[4], p[4], p[5]
        return O.gm_reduce(g, min_distance)


def _worker(rank, world, port, n_global, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    maps, logw = _global_maps(n_global, seed)
    off, n = D.shard_range(n_global, world, rank)
    sf = D.ShardedFilter(EapShard(maps[off:off + n], logw[off:off + n]), n_global, rank, world)
    got = sf.expected_map(10.0)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_global,seed", [(2, 32, 1), (3, 48, 2)])
def test_sharded_expected_map(tmp_path, world, n_global, seed):
    mp.spawn(_worker, args=(world, _free_port(), n_global, seed, str(tmp_path)), nprocs=world, join=True)
    maps, logw = _global_maps(n_global, seed)
    sizes = np.array([len(m) for m in maps], np.int32)
    ref = O.expected_map(np.concatenate(maps), sizes, logw, 10.0)
    assert len(ref) > 5
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got.tobytes() == ref.tobytes()
