#!/bin/bash
# Copies the outputs of tools/profile_round.sh <tag> (gpurun_out/prof_<tag>) into profiles/ under the round's names,
# regenerates profiles/pmc_traffic.json and prints the figures the documents quote.
set -eu
tag=${1:-r03}
root=$(cd "$(dirname "$0")/.." && pwd)
P=$root/gpurun_out/prof_$tag
for prob in tsp cvrp; do
  { echo "# FETCH_SIZE (KB per launch; x2 on gfx950)"; cat $P/pmc_${prob}_FETCH_SIZE.txt; echo "# WRITE_SIZE (KB per launch)"; cat $P/pmc_${prob}_WRITE_SIZE.txt; } > $root/profiles/${tag}_${prob}1000_b512_pmc_counters.txt
  { echo "# FETCH_SIZE (KB per launch; x2 on gfx950)"; cat $P/pmcsat_${prob}_FETCH_SIZE.txt; echo "# WRITE_SIZE (KB per launch)"; cat $P/pmcsat_${prob}_WRITE_SIZE.txt; } > $root/profiles/${tag}_${prob}1000_r16384_step_counters.txt
done
cp $P/kernel_stats.csv $root/profiles/${tag}_tsp_cvrp1000_b512_kernel_stats.csv
tail -1 $P/bench_profiled.json > $root/profiles/${tag}_tsp_cvrp1000_b512_bench_profiled.json
tail -1 $P/bench_unprofiled.json > $root/profiles/${tag}_tsp_cvrp1000_b512_bench_unprofiled.json
python3 $root/tools/make_pmc_traffic.py $tag | tr -d '\n '; echo
python3 - "$root" "$tag" <<'PY'
import json, sys, os
root, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
print("source hash", bench.kernel_source_hash(), json.load(open(os.path.join(root, 'profiles/pmc_traffic.json')))['kernel_source_hash'])
for f in ('unprofiled', 'profiled'):
    d = json.loads(open(os.path.join(root, f'profiles/{tag}_tsp_cvrp1000_b512_bench_{f}.json')).read())
    k = d['roofline_knn']; c = d['cvrp1000']; ck = c['roofline_knn']; i = d['ieee_roots_mode']; s = k['saturating_batch']
    print(f, 'tsp', round(d['value']), 'ms', round(d['ms_per_step'], 2), 'policy us', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'TF', round(d['roofline']['achieved'], 1), 'frac', round(d['roofline']['frac'], 4))
    print('   step us', round(k['avg_launch_ms'] * 1e3, 2), 'sat', round(s['avg_launch_ms'] * 1e3, 1), round(s['achieved']), round(s['frac'], 3), 'with tiles', round(s['achieved_with_tile_output'] / 8000, 3))
    print('   cvrp', round(c['value']), 'step', round(ck['avg_launch_ms'] * 1e3, 2), round(ck['saturating_batch']['avg_launch_ms'] * 1e3, 1), round(ck['saturating_batch']['frac'], 3))
    print('   ieee tsp', round(i['tsp']['value']), round(i['tsp']['step_kernel_ms'] * 1e3, 2), round(i['tsp']['step_kernel_saturating_batch']['avg_launch_ms'] * 1e3, 1), round(i['tsp']['step_kernel_saturating_batch']['frac'], 3),
          'cvrp', round(i['cvrp']['value']), round(i['cvrp']['step_kernel_ms'] * 1e3, 2), round(i['cvrp']['step_kernel_saturating_batch']['avg_launch_ms'] * 1e3, 1), round(i['cvrp']['step_kernel_saturating_batch']['frac'], 3))
    b = d['tsp10000_beam16']
    print('   beam', round(b['value'], 1), round(b['us_per_decode_step']), 'step', round(b['step_kernel']['avg_launch_ms'] * 1e3, 1), round(b['step_kernel']['frac'], 3), 'train ms', round(d['cvrp5000_train']['value'], 3),
          'wide us', round(d['cvrplib_k50_forward']['value']), round(d['cvrplib_k50_forward']['achieved']), 'general', round(d['cvrplib_k50_forward']['general_path_us']), 'cpu', d.get('cpu_baseline', {}).get('value'))
import csv
for row in csv.reader(open(os.path.join(root, f'profiles/{tag}_tsp_cvrp1000_b512_kernel_stats.csv'))):
    n = row[0]
    if n.startswith('void policy_kernel_f16<2') or 'policy_wide' in n or ('step_kernel<' in n and ('4, true, true, false' in n or '4, true, true, true' in n or '4, true, false, true' in n or '8, true, true, false' in n or '8, true, true, true' in n)):
        print('   csv', n[:62], row[1], row[3])
PY
