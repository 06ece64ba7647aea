#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each with --kernel-trace only.
# usage (repo root on the GPU box): bash tools/pmc_traffic.sh <config id> <tag>
cfg=${1:-2}; tag=${2:-r02}
steps=60; [ "$cfg" = "3" ] && steps=20
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_cfg${cfg}_$tag -- \
    python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --bare --steps $steps --warmup 5 --preroll-ms 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_cfg${cfg}_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $cfg $tag
