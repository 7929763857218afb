"""profiles/pmc_traffic.json from the counter passes of tools/profile_round.sh (gpurun_out/prof_<tag>/pmc*.txt):
HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes)
+ WRITE_SIZE, both in KB per launch; stamped with the hash of the kernels' sources (bench.kernel_source_hash)."""
import json, os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")

def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.{48}) (\S+)\s+([\d.]+)\s+n=(\d+)", line)
        if m:
            out[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return out

def entry(fetch, write, pat, comment):
    f = next(v for k, v in fetch.items() if pat in k)
    w = next(v for k, v in write.items() if pat in k)
    return {"fetch_size_kb": f[0], "write_size_kb": w[0], "launches": f[1],
            "bytes_per_launch": int(round((2 * f[0] + w[0]) * 1024)), "comment": comment}

rec = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/profile_round.sh): of `python3 bench.py "
               "--problem <p> --steps 1 --warmup 0 --no-cpu-baseline --no-aux` for the 512-row entries, of `python3 "
               "tools/bench_step_sat.py <p>` for the 16384-row entries; per-kernel averages by tools/pmc_query.py (KB per launch). "
               "FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes); WRITE_SIZE is "
               "taken as it is.  The counters sit on the fabric side of the L2: Infinity-Cache hits are included.",
       "kernel_source_hash": bench.kernel_source_hash(), "profile": f"gpurun_out/prof_{tag} -> profiles/{tag}_pmc_*.txt"}
for prob, knn in (("tsp", 30), ("cvrp", 15)):
    f, w = counters(os.path.join(src, f"pmc_{prob}_FETCH_SIZE.txt")), counters(os.path.join(src, f"pmc_{prob}_WRITE_SIZE.txt"))
    rec[f"{prob}1000_b512_k{knn}"] = {
        "policy_kernel_f16": entry(f, w, "policy_kernel_f16", "the 3.3 MB split-f16 weight stream is fetched once per XCD (8 L2s); tiles 0.8 MB"),
        "step_kernel": entry(f, w, f"{prob}_step_kernel", "state traffic + tile / index output")}
    f, w = counters(os.path.join(src, f"pmcsat_{prob}_FETCH_SIZE.txt")), counters(os.path.join(src, f"pmcsat_{prob}_WRITE_SIZE.txt"))
    rec[f"{prob}1000_b16384_k{knn}"] = {"step_kernel": entry(f, w, f"{prob}_step_kernel", "saturating batch")}
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps({k: {kk: vv["bytes_per_launch"] for kk, vv in v.items()} for k, v in rec.items() if isinstance(v, dict)}, indent=1))
