#!/bin/bash
# Calibrates the ceiling `valu_issue_fraction` (tools/pmc_sq.sh) is quoted against: tools/probes/valu_issue_probe — streams of
# independent vector instructions, one class per launch, at 1 … 8 waves per SIMD — timed plain, then under rocprofv3 with EXACTLY
# the counter set and the formula of tools/pmc_sq.sh, then with the instruction counters.  tools/valu_ceiling.py joins the three.
# usage (repo root on the GPU box): bash tools/valu_ceiling.sh <tag>      -> gpurun_out/valu_ceiling_<tag>.txt / .json
tag=${1:-r05}
P=$GRAFT_REPO_ROOT/tools/probes/valu_issue_probe
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
# the probe is built here from its source (the binary is not tracked: tools/probes/.gitignore)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $P $GRAFT_REPO_ROOT/tools/probes/valu_issue_probe.hip || exit 1
$P --json $O/valu_probe_plain_$tag.json > $O/valu_probe_plain_$tag.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/valu_probe_pmc1_$tag -- $P --json $O/valu_probe_pmc1_$tag.json > $O/valu_probe_pmc1_$tag.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/valu_probe_pmc2_$tag -- $P --json $O/valu_probe_pmc2_$tag.json > $O/valu_probe_pmc2_$tag.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/valu_ceiling.py $tag | tee $O/valu_ceiling_$tag.txt
