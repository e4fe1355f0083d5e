#!/bin/bash
# The round's SECONDARY records on the final build (GPU box), into gpurun_out/records/: every figure DESIGN / README quote
# beside the bench lines of tools/profile_round.sh.  Each tool prints the library's build id; `python tools/stamp_profiles.py
# --check profiles/rNN` refuses a copy that names another build.     usage: bash tools/round_records.sh [part ...]   (default: all)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/records
mkdir -p $O/study
cd $R
PARTS=${@:-tests paths sweeps soak dist}
for p in $PARTS; do
  case $p in
    tests)  ORT_STUDY_DIR=$O/study timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1 || exit 1;;
    paths)  timeout -k 10 600 python tools/pathbench.py > $O/pathbench.log 2>&1 || exit 1
            timeout -k 10 200 python tools/scatbench.py --rays 1000000,10000000,40000000,100000000 --variants 1 > $O/scatbench.log 2>&1 || exit 1
            timeout -k 10 300 python tools/expbench.py > $O/pull_nobin_raw.log 2>&1 || exit 1;;
    sweeps) timeout -k 10 200 python tools/sweep_profile.py > $O/sweep_profile.log 2>&1 || exit 1
            timeout -k 10 100 python tools/first_call.py > $O/first_call.log 2>&1 || exit 1
            timeout -k 10 600 python bench.py --sweep > $O/bench_sweep.json 2> $O/bench_sweep.err || exit 1;;
    soak)   timeout -k 10 600 python tools/parity_soak.py --systems 300 > $O/parity_soak.log 2>&1 || exit 1;;
    dist)   timeout -k 10 300 python bench.py --force-dist --no-fp32 --no-fast --no-strict > $O/bench_force_dist.json 2> $O/bench_force_dist.err || exit 1
            timeout -k 10 300 python bench.py --single-process --force-dist > $O/bench_single_process.json 2> $O/bench_single_process.err || exit 1;;
  esac
  echo "== $p done"
done
