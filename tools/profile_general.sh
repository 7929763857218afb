#!/bin/bash
# Per-kernel totals (rocprofv3 --kernel-trace) of the paths for rows of more than 32 tokens:
#   the training decode step (64 rows x 202 tokens) and the inference forward (512 rows x 102 tokens).
#   tools/profile_general.sh      writes gpurun_out/prof_general/{train_64x202,fwd_512x102}_kernel_stats.txt
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_general
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {   # name, script, args...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --kernel-trace -d "$out/$name" -o trace -- python3 "$@" > "$out/$name.log" 2>&1 || { echo "$name failed"; exit 1; }
  python3 "$root/tools/rocpd_stats.py" "$(find "$out/$name" -name '*_results.db' | head -1)" > "$out/${name}_kernel_stats.txt" || exit 1
  rm -rf "$out/$name"
  echo "$name done"
}
run train_64x202 "$root/tools/bench_train.py" 64 1000 100
run fwd_512x102 "$root/tools/bench_general_fwd.py" 512 50
