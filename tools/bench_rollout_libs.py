"""Greedy TSP rollout time between library builds / environment switches, interleaved child processes on one device.
usage: bench_rollout_libs.py lib1[:ENV=VAL,...],lib2[...] [R] [N] [problem]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
specs = sys.argv[1].split(";")
R = sys.argv[2] if len(sys.argv) > 2 else "512"
N = sys.argv[3] if len(sys.argv) > 3 else "1000"
code = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.join(%r, '..'))
from dgl_amd._lib import PROBLEM_TSP
from dgl_amd.engine import PackedPolicy, TspEngine
from dgl_amd.weights import random_state_dict
R, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda:0')
eng = TspEngine(torch.rand(R, N, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(1)), 1, 30)
pol = PackedPolicy(random_state_dict('TSP', 1), PROBLEM_TSP, dev)
start = torch.full((R,), 5, dtype=torch.int32, device=dev)
eng.greedy_rollout(pol, start); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): length = eng.greedy_rollout(pol, start)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(f"{dt * 1e3:.2f} ms per rollout, {R / dt:.0f} instances/s, mean length {length.mean().item():.5f}")
''' % HERE
for rep in range(3):
    for spec in specs:
        lib, _, envs = spec.partition(":")
        env = dict(os.environ, DGL_AMD_LIB=os.path.join(HERE, "..", "dgl_amd", "lib", lib))
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", code, R, N], env=env, capture_output=True, text=True)
        print(f"{spec}: {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
