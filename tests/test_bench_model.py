"""The byte and flop models behind bench.py's roofline entries, pinned to hand-computed values — the
figures the round-1 review recomputed: 31 662 472 algorithmic bytes per launch at 256 x 64 x 32, 3.19 GFLOP at 4096 x 256 x 64."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_bytes_and_flops_match_the_survey_formulas():
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    bench = importlib.import_module("bench")
    # per particle: map read 28 G, pose 24, update components written and read back 2 * 28 (G (M + 1) + M), merged map 28 G,
    # weight 8; + the measurement set once (12 M + 8)
    def by_hand(N, G, M):
        return N * (28 * G + 24 + 2 * 28 * (G * (M + 1) + M) + 28 * G + 8) + 12 * M + 8
    assert by_hand(256, 64, 32) == 31662472
    for (N, G, M) in ((256, 64, 32), (4096, 256, 64), (16384, 256, 64), (1, 64, 32)):
        assert S.algorithmic_bytes(N, G, M) == by_hand(N, G, M), (N, G, M)
    # 150 G + 45 G M + 30 M flops per particle
    assert bench.algorithmic_flops(4096, 256, 64) == 4096 * (150 * 256 + 45 * 256 * 64 + 30 * 64)
    assert abs(bench.algorithmic_flops(4096, 256, 64) - 3.185e9) < 1e7


def test_bench_gpus_n_without_a_launcher_refuses_missing_devices():
    """`python3 bench.py --gpus N` needs no launcher: it drives N devices from one process (libphdslam_multi.so).  On a box with
    fewer devices — this CPU container has none — it exits non-zero and says so; the GPU side of
    the same path is tests/test_gpu_multi.py::test_bench_gpus_2_unlaunched_runs_the_cpp_host."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PHD_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
