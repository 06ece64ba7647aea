#!/bin/bash
# Instruction-cache and instruction-mix counters of the update+merge kernel (diagnostic; two rocprofv3 --pmc passes,
# kernel trace only).  Prints per-launch means to gpurun_out/pmc_icache_cfg<cfg>_<tag>.txt.
# usage (repo root on the GPU box): bash tools/pmc_icache.sh <config id> <tag>
cfg=${1:-3}; tag=${2:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc_avail_$tag.txt 2>&1
pass() {
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ic$1_cfg${cfg}_$tag -- \
    python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --bare --steps 20 --warmup 5 --preroll-ms 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_ic$1_cfg${cfg}_$tag.log 2>&1
}
pass 1 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
pass 2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
pass 3 "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
pass 4 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64"
pass 5 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"
cd $GRAFT_REPO_ROOT
python3 - "$cfg" "$tag" <<'PY' > gpurun_out/pmc_icache_cfg${cfg}_$tag.txt
import csv, glob, sys, collections
cfg, tag = sys.argv[1], sys.argv[2]
for p in (1, 2, 3, 4, 5):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_ic%d_cfg%s_%s/**/*counter_collection.csv" % (p, cfg, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "phd_update_merge_kernel" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("pass %d (config %s):" % (p, cfg))
    for k in sorted(acc):
        print("  %-28s %16.0f  (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
PY
cat gpurun_out/pmc_icache_cfg${cfg}_$tag.txt
grep -i -E "error|invalid|not supported" gpurun_out/pmc_ic*_cfg${cfg}_$tag.log | head
