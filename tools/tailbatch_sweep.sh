#!/bin/bash
# Dev tool (GPU box): the length of a tail range (ORT_DEV_TAIL_BATCHES) against the launch size, both loops.
#   usage: bash tools/tailbatch_sweep.sh "rays ..." "tail_batches ..."
RAYS=${1:-"20000000 33554432 67108864 134217728"}; TB=${2:-"6 12 24 48"}
for ph in 2 1; do
  for n in $RAYS; do
    for tb in $TB; do
      ORT_DEV_TAIL_BATCHES=$tb python bench.py --rays $n --phase $ph --steps 24 --warmup 4 --no-cpu-baseline --no-fast --no-strict > /tmp/tb.json 2>/dev/null || exit 1
      python - "$ph" "$n" "$tb" <<'PY'
import json, sys
d = json.load(open('/tmp/tb.json'))
print(f"phase {sys.argv[1]} rays {int(sys.argv[2]):>10} tail_batches {sys.argv[3]:>3}: fp64 kernel {d['roofline']['kernel_ms']:.4f} ms  fp32 kernel {d['fp32']['roofline']['kernel_ms']:.4f} ms", flush=True)
PY
    done
  done
done
