#!/bin/bash
# HBM-side counter passes of the training decode step (64 rows x 202 tokens), one run per counter.
#   tools/profile_train.sh      writes gpurun_out/prof_train/pmc_<counter>.txt
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_train
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $ctr -d "$out/pmc_$ctr" -o pmc -- python3 "$root/tools/bench_train.py" 64 1000 100 > "$out/pmc_$ctr.log" 2>&1 || { echo "pmc $ctr failed"; exit 1; }
  python3 "$root/tools/pmc_query.py" "$(find "$out/pmc_$ctr" -name '*_results.db' | head -1)" > "$out/pmc_$ctr.txt" || exit 1
  echo "pmc $ctr done"
done
rm -rf "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE
