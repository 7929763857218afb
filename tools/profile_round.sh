#!/bin/bash
# Round profile on the GPU box: kernel-trace summary of the default bench command, then the HBM-side counter passes
# (separate runs per counter, as the counters cannot share a pass with traces) for TSP-1000 and CVRP-1000.
#   tools/profile_round.sh <tag>      writes gpurun_out/prof_<tag>/*
set -u
tag=${1:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*_results.db' | head -1; }

timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/trace" -o bench -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_profiled.json" 2> "$out/bench_profiled.err" || { echo "trace run failed"; exit 1; }
python3 "$root/tools/rocprof_summary.py" "$(db "$out/trace")" "$out/kernel_stats.csv" || exit 1
echo "kernel stats done"
for prob in tsp cvrp; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $ctr -d "$out/pmc_${prob}_$ctr" -o pmc -- python3 "$root/bench.py" --problem $prob --steps 1 --warmup 0 --no-cpu-baseline --no-aux > "$out/pmc_${prob}_$ctr.json" 2> "$out/pmc_${prob}_$ctr.err" || { echo "pmc $prob $ctr failed"; exit 1; }
    python3 "$root/tools/pmc_query.py" "$(db "$out/pmc_${prob}_$ctr")" > "$out/pmc_${prob}_$ctr.txt" || exit 1
    echo "pmc $prob $ctr done"
  done
done
# the step kernels at a saturating batch (16384 rows x 1000 nodes)
for prob in tsp cvrp; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $ctr -d "$out/pmcsat_${prob}_$ctr" -o pmc -- python3 "$root/tools/bench_step_sat.py" $prob > "$out/pmcsat_${prob}_$ctr.log" 2>&1 || { echo "pmc sat $prob $ctr failed"; exit 1; }
    python3 "$root/tools/pmc_query.py" "$(db "$out/pmcsat_${prob}_$ctr")" > "$out/pmcsat_${prob}_$ctr.txt" || exit 1
    echo "pmc sat $prob $ctr done"
  done
done
rm -rf "$out"/trace "$out"/pmc_*_FETCH_SIZE "$out"/pmc_*_WRITE_SIZE "$out"/pmcsat_*_FETCH_SIZE "$out"/pmcsat_*_WRITE_SIZE
ls "$out"
