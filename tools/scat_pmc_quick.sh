#!/bin/bash
# Dev tool (GPU box): two counter passes over the scattering pipeline at 2^24 rays -> gpurun_out/scatpmcq/summary.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/scatpmcq
N=${1:-16777216}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/scatbench.py --rays $N --variants 1 --reps 2 > $OUT/p$i.log 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - $OUT $N <<'PY' > $OUT/summary.txt
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "front" if "scatter_front" in k else "cont" if "trace_queue" in k else None
        if short: agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
n = float(sys.argv[2])
for (k, c), v in sorted(agg.items()):
    m = sum(v) / len(v)
    print(f"{k:6s} {c:28s} n={len(v):3d} mean={m:.6g}  per ray (x64 lanes/ray for wave instructions): {m * 64 / n:.5g}")
PY
cat $OUT/summary.txt
