#!/bin/bash
# Dev tool (GPU box): the range plan of the queued kernels (plan_ranges: long ranges on ORT_DEV_HEAD_BLOCKS workgroups for
# ORT_DEV_HEAD_PERCENT of the rays, short ranges of ORT_DEV_TAIL_BATCHES batches for the rest) against the step time of the
# fp64 and fp32 legs of bench.py.   usage: bash tools/headsweep.sh [workload] > gpurun_out/headsweep.log
W=${1:-point1e7}
for hb in ${HB:-1280 1536 2048 2560}; do
  for pct in 80 86 92; do
    for tb in 4 6; do
      ORT_DEV_HEAD_BLOCKS=$hb ORT_DEV_HEAD_PERCENT=$pct ORT_DEV_TAIL_BATCHES=$tb python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --no-fast --no-strict > /tmp/hs.json 2>/dev/null || exit 1
      python - "$hb" "$pct" "$tb" <<'PY'
import json, sys
d = json.load(open('/tmp/hs.json'))
print(f"head_blocks {sys.argv[1]:>5} pct {sys.argv[2]:>3} tail_batches {sys.argv[3]}: fp64 {d['ms_per_step']:.4f} ms (kernel {d['roofline']['kernel_ms']:.4f})  fp32 {d['fp32']['ms_per_step']:.4f} ms (kernel {d['fp32']['roofline']['kernel_ms']:.4f})", flush=True)
PY
    done
  done
done
