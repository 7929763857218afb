#!/bin/bash
# Matrix-core utilisation of the policy kernel from hardware counters (one rocprofv3 --pmc pass, counters only):
#   tools/profile_policy_mfma.sh <tag>     writes gpurun_out/prof_<tag>/policy_mfma_counters.txt
# SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_f16), summed over
# the SIMDs; SQ_BUSY_CU_CYCLES the cycles a CU has waves, summed over the CUs (x 4 SIMDs for the same base).
set -u
tag=${1:-r02}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE \
  -d "$out/pmc_policy" -o pmc -- python3 "$root/tools/bench_policy.py" 512 200 > "$out/policy_mfma.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$out/policy_mfma.log"; exit 1; }
python3 "$root/tools/pmc_query.py" "$(find "$out/pmc_policy" -name '*_results.db' | head -1)" '%policy_kernel_f16%' > "$out/policy_mfma_counters.txt" || exit 1
rm -rf "$out/pmc_policy"
cat "$out/policy_mfma_counters.txt"
