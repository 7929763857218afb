#!/bin/bash
# Build the library from a git revision (or the working tree: "wt") into dgl_amd/lib/libdgl_amd_<name>.so for A/B runs.
#   tools/build_variant.sh <name> <rev|wt> [extra hipcc flags...]
set -e
name=$1; rev=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$rev" = wt ]; then src=$root/dgl_amd/csrc; inc=$root/include
else tmp=$(mktemp -d); git -C "$root" archive "$rev" dgl_amd/csrc include | tar -x -C "$tmp"; src=$tmp/dgl_amd/csrc; inc=$tmp/include; fi
files=$(cd "$src" && ls api.cpp *.hip | grep -v -e bench_peaks -e check_)
(cd "$src" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden "$@" -x hip $files -o "$root/dgl_amd/lib/libdgl_amd_$name.so") 2>&1 | grep -E " error|Error" || true
ls -la "$root/dgl_amd/lib/libdgl_amd_$name.so"
