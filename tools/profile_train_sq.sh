#!/bin/bash
# Where the waves of the training step's kernels spend their cycles, from hardware counters (two rocprofv3 --pmc
# passes, counters only):   tools/profile_train_sq.sh     writes gpurun_out/prof_train/sq_<pass>.txt
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_train
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$out/sq_$name" -o pmc -- python3 "$root/tools/bench_train.py" 64 1000 100 > "$out/sq_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/sq_$name.log"; exit 1; }
  python3 "$root/tools/pmc_query.py" "$(find "$out/sq_$name" -name '*_results.db' | head -1)" > "$out/sq_$name.txt" || exit 1
  rm -rf "$out/sq_$name"
  echo "pass $name done"
}
pass issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pass wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
