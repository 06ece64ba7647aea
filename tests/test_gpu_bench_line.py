"""bench.py as the driver runs it: the LAST stdout line is the compact record (<= 4096 bytes) with the contract's keys — `roofline`
and `cpu_baseline` included — and the full record lands in the side file (BENCH_r05.json.parsed was null because the
line had grown to 21.7 KB)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_last_line_is_the_compact_record(tmp_path):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PHD_BENCH_SHARE_GPU", "PHD_LAYOUT"):
        env.pop(k, None)
    rec = tmp_path / "bench_last.json"
    env["PHD_BENCH_RECORD"] = str(rec)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "2",
                        "--preroll-ms", "50"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) <= 4096, len(last)
    line = json.loads(last)
    assert line["metric"].startswith("PHD-update steps/sec") and line["unit"] == "steps/s" and line["n_gpus"] == 1
    assert line["steps"] == 5 and line["warmup"] == 2 and line["value"] > 0 and line["vs_baseline"] is None
    assert (line["config"]["N"], line["config"]["G"], line["config"]["M"]) == (4096, 256, 64)
    assert line["config"]["one_launch_per_step"] is True and line["config"]["fast_path"] is True
    # the same filter without the layout / scan-length specialisation, same run: what a scan of arbitrary length gets
    assert line["value_general"] and 0.5 * line["value"] < line["value_general"] < 1.2 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and roof["kernel"] == "phd_update_merge_kernel"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and roof["kernel_avg_us"] > 0 and roof["library_build"]
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["unit"] == "steps/s"
    if cpu["quota_cpus"]:
        assert cpu["cores"] <= int(-(-cpu["quota_cpus"] // 1))           # the thread scan stops at the cgroup quota
    assert line["hip_runtime_version"]
    # every rider of the run is in the record file, one steps/s figure each in the line
    full = json.load(open(rec))
    assert len(full["secondary"]) == len(line["riders_steps_per_s"]) >= 4
    assert full["roofline"]["note"] and full["cpu_baseline"]["thread_scan_s_per_step"]
