"""The RCCL code paths of cuda-phdslam_amd/dist.py on the one GPU a test box has: a ONE-rank "nccl" group with
`ShardedFilter.collectives` forced on, so that the collectives the N > 1 bench issues — all_gather_into_tensor on the
library's wrapped raw-weight buffer, all_to_all_single with split sizes on the wrapped send buffer (empty and
non-empty), the padded all-gather of the expected-map planes — run through RCCL on device memory, on the filter's
stream, and must give what the plain single-filter step gives.  (tests/test_gpu_dist.py covers the two-rank logic
with gloo; RCCL itself needs one GPU per rank.)"""
import importlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    D = importlib.import_module("cuda-phdslam_amd.dist")
    N, G, M, u = 64, 24, 10, 0.37
    w = S.make_workload(N, G, M, seed=5)
    cfg = P.default_config(n_particles=N)
    ts = torch.cuda.Stream(device=dev)                       # as bench.py: one stream for kernels and collectives
    torch.cuda.set_stream(ts)
    d_z = torch.from_numpy(w["z"][0].view(np.uint8).copy()).to(dev)
    d_noise = torch.from_numpy(w["noise"][0].copy()).to(dev)
    res = {}
    for mode in ("plain", "rccl", "rccl_gathered"):
        f = P.PhdFilter(cfg, n_particles=N, map_capacity=4 * G, max_measurements=M, stream=ts.cuda_stream,
                        global_particles=N, global_offset=0)
        f.set_particles(w["poses"], w["logw"])
        f.set_maps(w["maps"], w["sizes"])
        shard = D.GpuShard(f, N)
        sf = D.ShardedFilter(shard, N, 0, 1)
        sf.collectives = mode != "plain"
        if mode == "rccl":
            sf.gathered_limit = 0                            # all-to-all form
        for k in range(3):                                   # several steps: the cached buffers are reused
            if mode == "rccl_gathered":
                # local step into the export rows (k odd: step, then a separate export) -> all_gather_into_tensor of the
                # whole shard -> normalise + indices + import, no host sync
                assert sf.gathered()
                rows = None
                if k % 2 == 0:
                    rows = shard.step_local_rows((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M)
                else:
                    shard.step_local_dev((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M)
                idx = sf.resample_gathered(u, weights_in_rows=True, want_idx=(k == 2), rows=rows)
                continue
            shard.step_local_dev((2.0, 0.05), d_noise.data_ptr(), d_z.data_ptr(), M)
            allw = sf.gather_logweights()
            if mode == "rccl":
                assert allw.is_cuda and allw.data_ptr() == sf._allw.data_ptr()
                idx = sf.resample(u, all_raw_logw=allw)      # begin -> all_to_all_single -> end
            else:
                sf.normalize(allw, want_neff=False)
                idx = sf.resample(u)
        eap = sf.expected_map(cfg.minSeparation)
        poses, lw = f.get_particles()
        maps = f.get_maps()
        f.status()
        res[mode] = dict(idx=np.asarray(idx), lw=lw, poses=poses, eap=eap, flat=np.concatenate(maps),
                         sizes=np.array([len(m) for m in maps]))
        if mode == "rccl":
            # a non-empty exchange through RCCL: rows to "rank 0" come back as they were sent
            pack = shard.pack_bytes()
            rows = torch.randint(0, 255, (7, pack), dtype=torch.uint8, device=dev)
            back = sf._exchange(rows, [7], [7], pack)
            torch.cuda.synchronize()
            assert torch.equal(back[:7], rows)
            empty = sf._exchange(rows[:0], [0], [0], pack)   # nothing migrates: the usual case right after a resample
            torch.cuda.synchronize()
            assert empty.shape[1] == pack
        f.close()
    a = res["plain"]
    ok = all(np.array_equal(a["idx"], b["idx"]) and a["lw"].tobytes() == b["lw"].tobytes()
             and a["poses"].tobytes() == b["poses"].tobytes() and a["flat"].tobytes() == b["flat"].tobytes()
             and np.array_equal(a["sizes"], b["sizes"]) and a["eap"].tobytes() == b["eap"].tobytes()
             for b in (res["rccl"], res["rccl_gathered"]))
    b = res["rccl"]
    np.savez(os.path.join(out_dir, "ok.npz"), ok=ok, n_eap=len(b["eap"]))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_paths_on_a_one_rank_rccl_group(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(os.path.join(str(tmp_path), "ok.npz"))
    assert bool(r["ok"]) and int(r["n_eap"]) > 0


def test_bench_multi_rank_path_prints_the_json_line_last():
    """bench.py's N > 1 path on a one-rank RCCL group: RCCL's version banner (C stdio, flushed at exit on a pipe) must not
    follow the JSON line — the driver reads the last line of stdout"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PHD_BENCH_ONE_RANK_RCCL="1", PHD_BENCH_EXCHANGE="alltoall", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "0",
                        "--preroll-ms", "0", "--no-secondary"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["multi_gpu_exchange"] == "alltoall"
