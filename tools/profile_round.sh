#!/bin/bash
# Round profile (GPU box): for each bench workload, `rocprofv3 --kernel-trace --stats` of the bench
# command itself and the PMC passes (each in its own run: --pmc never together with other trace
# domains than --kernel-trace), then the plain bench line, into gpurun_out/prof/<workload>/.  tools/stamp_profiles.py turns the
# (the seventh pass — fp32 instruction classes — is for the fp32 leg of point1e7)
# CSVs into profiles/rNN/*.json and profiles/pmc_per_launch.json (stamped with ort_build_id()).
#   usage: bash tools/profile_round.sh [workloads...]      (default: point1e7 ring1e8 full1e9)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof
WL=${@:-point1e7 ring1e8 full1e9}
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  case $w in
    point1e7) STEPS="--steps 200 --warmup 20"; PSTEPS="--steps 24 --warmup 2";;
    ring1e8)  STEPS="--steps 40 --warmup 4";   PSTEPS="--steps 8 --warmup 1";;
    full1e9)  STEPS="--steps 4 --warmup 1";    PSTEPS="--steps 1 --warmup 0";;
  esac
  D=$OUT/$w; mkdir -p $D
  echo "== $w: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python3 $R/bench.py --workload $w $STEPS --no-cpu-baseline > $D/bench_under_rocprof.json 2> $D/trace.err || exit 1
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP64" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_STALL_sum TCC_ATOMIC_sum" \
             "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_INT32"; do
    i=$((i+1))
    echo "== $w: pmc pass $i: $set"
    # point1e7: with the fp32 and fast-fp64 legs, so that their kernels are counted too (bench.py's `fp32` / `fast_fp64` objects)
    LEGS="--no-fp32 --no-fast"; [ $w = point1e7 ] && LEGS=""
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/pmc$i -o p -- python3 $R/bench.py --workload $w $PSTEPS --no-cpu-baseline $LEGS > $D/pmc$i.json 2> $D/pmc$i.err || echo "pass $i failed (see pmc$i.err)"
  done
done
# the plain bench lines last, quoting the counters just collected (bench.py reads
# profiles/pmc_per_launch.json only when its build id is the library's)
python3 $R/tools/stamp_profiles.py _box $OUT > $OUT/stamp_on_box.log 2>&1 || exit 1
rm -rf $R/profiles/_box
for w in $WL; do
  case $w in
    point1e7) STEPS="--steps 200 --warmup 20";;
    ring1e8)  STEPS="--steps 40 --warmup 4";;
    full1e9)  STEPS="--steps 4 --warmup 1";;
  esac
  echo "== $w: plain bench"
  python3 $R/bench.py --workload $w $STEPS > $OUT/$w/bench.json 2> $OUT/$w/bench.err || exit 1
done
echo done
