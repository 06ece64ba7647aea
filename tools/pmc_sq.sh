#!/bin/bash
# Issue/stall breakdown of the update+merge kernel from SQ counters (one rocprofv3 --pmc pass, kernel trace only).
# Writes gpurun_out/pmc_sq_cfg<cfg>.json (copied to profiles/: bench.py reads `valu_issue_fraction` from it and prints
# the file, build tag and date beside the number).
# usage (repo root on the GPU box): bash tools/pmc_sq.sh <config id> <build tag>
cfg=${1:-3}; tag=${2:-r02}
steps=20; [ "$cfg" = "2" ] && steps=60
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_cfg${cfg}_$tag -- \
  python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --bare --steps $steps --warmup 5 --preroll-ms 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_cfg${cfg}_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$cfg" "$tag" <<'PY'
import csv, ctypes, glob, json, sys, collections, datetime
_L = ctypes.CDLL('cuda-phdslam_amd/libphdslam.so'); _L.phd_version.restype = ctypes.c_char_p
BUILD_ID = _L.phd_version().decode().split('build ')[-1]          # bench.py refuses counters of another build
cfg, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob("gpurun_out/pmc_sq_cfg%s_%s/**/*counter_collection.csv" % (cfg, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        if "phd_update_merge_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_sq_cfg%s_%s/**/*kernel_trace.csv" % (cfg, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        if "phd_update_merge_kernel" in row.get("Kernel_Name", ""):
            dur.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
n = len(next(iter(acc.values()))) if acc else 0
print("config %s, per launch (mean of %d):" % (cfg, n))
wc = m.get("SQ_WAVE_CYCLES", 0) or 1
for k in sorted(m):
    print("  %-22s %14.0f  %6.1f %% of wave cycles" % (k, m[k], 100 * m[k] / wc))
if m:
    # units (MI355X_MICROARCH.md, constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
    # over waves; SQ_BUSY_CYCLES counts cycles summed over the 32 shader engines (8 XCDs x 4) -> / 32 = the kernel's
    # duration in shader cycles.  VALU issue fraction = cycles some wave of a SIMD spends issuing VALU / SIMD cycles:
    #   4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * SQ_BUSY_CYCLES / 32)
    kcyc = m["SQ_BUSY_CYCLES"] / 32.0
    out = {"config": int(cfg), "kernel": "phd_update_merge_kernel", "build": tag, "build_id": BUILD_ID, "date": datetime.date.today().isoformat(),
           "dispatches_averaged": n, "kernel_avg_us": (sum(dur) / len(dur)) if dur else None,
           "kernel_shader_cycles": kcyc,
           "valu_issue_fraction": 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * kcyc),
           "any_issue_fraction": 4.0 * m["SQ_ACTIVE_INST_ANY"] / (1024.0 * kcyc),
           "mean_waves_per_simd": 4.0 * m["SQ_WAVE_CYCLES"] / (1024.0 * kcyc),
           "formula": "valu_issue_fraction = 4 * SQ_ACTIVE_INST_VALU / (1024 * SQ_BUSY_CYCLES / 32)"}
    # the formula's ceiling is MEASURED (tools/valu_ceiling.sh, profiles/valu_ceiling.json): 1.91 on a saturating stream of
    # independent full-rate vector instructions (a SIMD issues one per 2.1 cycles once two waves interleave) - not 1.0
    try:
        ceil = json.load(open("profiles/valu_ceiling.json"))
        out["valu_issue_ceiling_measured"] = ceil["formula_reading_at_saturation"]
        out["valu_issue_fraction_of_ceiling"] = out["valu_issue_fraction"] / ceil["formula_reading_at_saturation"]
        if m.get("SQ_INSTS_VALU"):
            out["valu_instructions_per_cycle_per_simd"] = m["SQ_INSTS_VALU"] / (1024.0 * kcyc)
            out["valu_instructions_per_cycle_per_simd_ceiling"] = ceil["full_rate_ipc_per_simd"]
    except Exception as e:
        out["valu_issue_ceiling_measured"] = None
    out.update(m)
    json.dump(out, open("gpurun_out/pmc_sq_cfg%s.json" % cfg, "w"), indent=1)
    print(json.dumps(out))
PY
