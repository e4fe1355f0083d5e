#!/bin/bash
# Dev tool (GPU box): counter passes over tools/scatbench.py (the scattering pipeline), each in its own run.
#   usage: bash tools/scat_pmc.sh [rays]        -> gpurun_out/scatpmc/
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/scatpmc
N=${1:-20000000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_WAVE32_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VSKIPPED SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  echo "== pass $i: $set"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/scatbench.py --rays $N --variants 1 --reps 2 > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "front" if "scatter_front" in k else "cont" if "trace_queue" in k else None
        if short: agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:6s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
