#!/usr/bin/env python3
"""Start-up stress of the multi-device host: N consecutive `phdslam --devices 1` runs (a one-rank RCCL
communicator: ncclCommInitAll under the watchdog of phd_multi_create), each with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP.
Records the time to the first output line and the wall time of every start; a start that produces nothing for `--stall` seconds
gets a native backtrace of all its threads (rocgdb attached from THIS process — a fresh one, never a re-exec) and its NCCL log
saved before it is killed.

    python tools/stress_start.py [--n 200] [--stall 45] [--out gpurun_out/stress_start.txt]
"""
import argparse
import os
import select
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--stall", type=float, default=45.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress_start.txt"))
    ap.add_argument("--extra", default="--devices 1", help="driver arguments under test")
    args = ap.parse_args()
    from test_gpu_driver import PKG, write_dataset
    d = tempfile.mkdtemp(prefix="phd_stress_")
    cfg_path = write_dataset(d, n_steps=3, n_particles=48)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    env = dict(os.environ, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,BOOTSTRAP")
    first, total, stalls = [], [], 0
    log = open(args.out, "w")
    for i in range(args.n):
        o = os.path.join(d, "run%d" % i)
        os.makedirs(o, exist_ok=True)
        cmd = [os.path.join(PKG, "bin", "phdslam"), cfg_path, "synth", "--out", o, "--seed", "9", "--capacity", "256"] + args.extra.split()
        t0 = time.perf_counter()
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
        buf, t_first, stalled = b"", None, False
        last = time.perf_counter()
        while True:
            r, _, _ = select.select([p.stdout], [], [], 1.0)
            now = time.perf_counter()
            if r:
                chunk = os.read(p.stdout.fileno(), 65536)
                if not chunk:
                    break
                buf += chunk
                last = now
                if t_first is None and b"sharded filter" in buf:
                    t_first = now - t0
            elif now - last > args.stall:
                stalled = True
                break
        if stalled:
            stalls += 1
            bt = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(p.pid), "-batch", "-ex", "thread apply all bt"],
                                capture_output=True, text=True, timeout=120)
            log.write("==== start %d STALLED after %.1f s without output; output so far:\n%s\n---- backtrace:\n%s\n%s\n" %
                      (i, args.stall, buf.decode(errors="replace")[-6000:], bt.stdout[-12000:], bt.stderr[-2000:]))
            p.kill()
            p.wait()
            continue
        rc = p.wait()
        dt = time.perf_counter() - t0
        if rc != 0:
            log.write("==== start %d FAILED rc=%d:\n%s\n" % (i, rc, buf.decode(errors="replace")[-4000:]))
        first.append(t_first if t_first is not None else float("nan"))
        total.append(dt)
        log.write("start %3d: first line after %.2f s, done in %.2f s, rc %d\n" % (i, first[-1], dt, rc))
        log.flush()
    import numpy as np
    f, t = np.array(first), np.array(total)
    summary = ("%d starts of `phdslam %s` (48 particles, 3 steps): %d stalled (> %.0f s silent), %d completed; time to the first "
               "output line: median %.2f s, p99 %.2f s, max %.2f s; wall time per start: median %.2f s, max %.2f s" %
               (args.n, args.extra, stalls, args.stall, len(t), np.nanmedian(f), np.nanpercentile(f, 99), np.nanmax(f),
                np.median(t), t.max()))
    log.write(summary + "\n")
    log.close()
    print(summary)


if __name__ == "__main__":
    main()
