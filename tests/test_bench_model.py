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


def test_line_is_compact():
    """The driver's record of a bench run is the LAST stdout line, and it keeps a bounded tail of stdout: round 5's line had grown to
    21.7 KB (riders, notes, thread scans) and BENCH_r05.json.parsed was null.  bench.py now prints compact_line(record) and writes
    the record itself to profiles/bench_last.json.  Here: round 5's full 21.7 KB record (the canned input) through that same
    function -> <= 4096 bytes, json round trip, every key of the contract present and non-null where the record had it."""
    import json
    bench = importlib.import_module("bench")
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_invocation.json")))
    assert len(json.dumps(full)) > 20000                      # the canned record is the one that broke the driver's parser
    full["value_general"] = 4300.123456789
    full["config"]["update_kernel_instantiation"] = {"index": 20, "fast_path": True}   # (the key the last commit of round 5 added)
    for k, r in enumerate(full["secondary"]):
        r["rider"] = "cfg%d_a_rider_with_a_long_label" % k
    text = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(text) <= bench.LINE_LIMIT == 4096, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "value_general", "riders"):
        assert key in line, key
    assert line["value"] == float("%.6g" % full["value"]) and line["unit"] == "steps/s" and line["vs_baseline"] is None
    assert len(line["config"]["workload"]) <= 120 and "model" not in line["config"]
    assert (line["config"]["N"], line["config"]["G"], line["config"]["M"]) == (4096, 256, 64)
    assert line["config"]["instantiation"] == 20 and line["config"]["fast_path"] is True
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_us", "library_build"):
        assert line["roofline"].get(key) is not None, key
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert line["cpu_baseline"].get(key) is not None, key
    assert len(line["cpu_baseline"]["sample"]) <= 120
    assert len(line["riders_steps_per_s"]) == len(full["secondary"])
    # no string of the line is longer than what the driver keeps of any string
    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(t) for t in strings(line)) <= 120


def test_emit_prints_the_compact_line_last_and_writes_the_record(tmp_path, capsys, monkeypatch):
    import json
    bench = importlib.import_module("bench")
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_invocation.json")))
    rec = tmp_path / "bench_last.json"
    monkeypatch.setattr(bench, "RECORD_PATH", str(rec))
    bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    line = json.loads(out[-1])
    assert len(out[-1]) <= 4096 and line["value"] == float("%.6g" % full["value"])
    assert json.load(open(rec))["secondary"] == full["secondary"]        # nothing of the record is lost
