#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh for the update+merge kernel."""
import csv, ctypes, datetime, glob, json, os, sys
cfg, tag = sys.argv[1], sys.argv[2]
_L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-phdslam_amd", "libphdslam.so"))
_L.phd_version.restype = ctypes.c_char_p
BUILD_ID = _L.phd_version().decode().split("build ")[-1]          # bench.py refuses counters of another build
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s_cfg%s_%s/**/*counter_collection.csv" % (c, cfg, tag), recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if "phd_update_merge_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
                vals.append(float(row["Counter_Value"]))
    res[c] = (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
print(res)
if res["FETCH_SIZE"][0] is not None and res["WRITE_SIZE"][0] is not None:
    # rocprofv3 reports both in KiB.  gfx950: FETCH_SIZE counts 64 B per 128-B request for wide
    # (16 B/lane) streaming reads (guide: "double it"); this kernel's plane reads are 4 B/lane, a width
    # the guide calls uncalibrated, so both the raw and the doubled figure are kept.
    fetch, write = res["FETCH_SIZE"][0] * 1024, res["WRITE_SIZE"][0] * 1024
    out = {"config": int(cfg), "kernel": "phd_update_merge_kernel", "build": tag, "build_id": BUILD_ID, "date": datetime.date.today().isoformat(),
           "dispatches_averaged": res["FETCH_SIZE"][1],
           "fetch_bytes_raw": fetch, "fetch_bytes_doubled": 2 * fetch, "write_bytes": write,
           "hbm_bytes_per_launch": 2 * fetch + write,
           "note": "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (upper bound for 4 B/lane reads)"}
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open("gpurun_out/pmc_traffic_cfg%s.json" % cfg, "w"), indent=1)
    print(json.dumps(out))
