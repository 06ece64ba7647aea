#!/bin/bash
# Issue/stall breakdown of the update+merge kernel from SQ counters (one rocprofv3 --pmc pass, kernel trace only).
# usage (repo root on the GPU box): bash tools/pmc_sq.sh <config id> <tag>
cfg=${1:-3}; tag=${2:-r01}
steps=20; [ "$cfg" = "2" ] && steps=60
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_cfg${cfg}_$tag -- \
  python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $steps --warmup 5 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_cfg${cfg}_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$cfg" "$tag" <<'PY'
import csv, glob, sys, collections
cfg, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_sq_cfg%s_%s/**/*counter_collection.csv" % (cfg, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        if "phd_update_merge_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("config %s, per launch (mean of %d):" % (cfg, len(next(iter(acc.values()))) if acc else 0))
wc = m.get("SQ_WAVE_CYCLES", 0) or 1
for k in sorted(m):
    print("  %-22s %14.0f  %6.1f %% of wave cycles" % (k, m[k], 100 * m[k] / wc))
PY
