#!/bin/bash
# gpurun_out/prof + gpurun_out/records (tools/profile_round.sh, tools/round_records.sh, tools/batch_probe.py on the GPU box) -> profiles/rNN/
#   usage: bash tools/collect_records.sh r05
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
P=profiles/$1; S=gpurun_out/records
mkdir -p $P/study
python tools/stamp_profiles.py $1 > /dev/null
cp $S/study/*.json $P/study/
cp $S/gpu_tests.log $S/scatbench.log $S/sweep_profile.log $S/first_call.log $S/parity_soak.log $S/bench_force_dist.json $S/bench_single_process.json $S/pathbench.log $P/
python - "$S/bench_sweep.json" "$P/bench_sweep.json" <<'PY'
import json, sys
b = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
json.dump(b, open(sys.argv[2], "w"), indent=1)
PY
if [ -f $P/pull_nobin.log ]; then                       # keep the header (the reading), replace the table
  { sed -n '/^# VERDICT/,/^build /p' $P/pull_nobin.log | grep -v '^build '; grep -v amdgpu.ids $S/pull_nobin_raw.log; } > /tmp/pull.log && mv /tmp/pull.log $P/pull_nobin.log
fi
if [ -f $S/batch_probe.log ]; then
  { sed -n '/^# tools\/batch_probe.py/,/^# the batched literal/p' $P/batch_probe.log; grep -v '^# tools\|^# launch gaps\|^# the batched' $S/batch_probe.log; } > /tmp/bp.log && mv /tmp/bp.log $P/batch_probe.log
fi
python tools/stamp_profiles.py --check $P
