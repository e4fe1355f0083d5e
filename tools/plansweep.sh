#!/bin/bash
# Dev tool (GPU box): head share x tail range length of plan_ranges against the step time of a bench workload.
#   usage: bash tools/plansweep.sh workload "pct ..." "tail_batches ..." [head_blocks]
W=${1:-ring1e8}; PCT=${2:-"80 86 92"}; TB=${3:-"6 12 24 48"}; HB=${4:-1280}
case $W in point1e7) ST="--steps 100 --warmup 10";; ring1e8) ST="--steps 40 --warmup 4";; full1e9) ST="--steps 3 --warmup 1";; esac
for pct in $PCT; do
  for tb in $TB; do
    ORT_DEV_HEAD_BLOCKS=$HB ORT_DEV_HEAD_PERCENT=$pct ORT_DEV_TAIL_BATCHES=$tb python bench.py --workload $W $ST --no-cpu-baseline --no-fast --no-strict > /tmp/ps.json 2>/dev/null || exit 1
    python - "$pct" "$tb" "$HB" <<'PY'
import json, sys
d = json.load(open('/tmp/ps.json'))
print(f"head_blocks {sys.argv[3]} pct {sys.argv[1]:>3} tail_batches {sys.argv[2]:>3}: fp64 {d['ms_per_step']:.4f} ms (kernel {d['roofline']['kernel_ms']:.4f})  fp32 {d['fp32']['ms_per_step']:.4f} ms (kernel {d['fp32']['roofline']['kernel_ms']:.4f})", flush=True)
PY
  done
done
