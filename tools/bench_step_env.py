"""Step kernel timing under different environment switches (each in a child process), interleaved."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
R, N = sys.argv[1], sys.argv[2]
variants = sys.argv[3].split(";")
for rep in range(2):
    for v in variants:
        env = dict(os.environ)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("="); env[k] = val
        out = subprocess.run([sys.executable, os.path.join(HERE, "bench_step.py"), R, N, "40"], env=env, capture_output=True, text=True)
        print(f"[{v}] {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
