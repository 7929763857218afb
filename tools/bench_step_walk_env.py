"""bench_step_walk.py under different environment switches / library builds (child processes, interleaved).
  python tools/bench_step_walk_env.py "<args of bench_step_walk.py>" "VAR=val,VAR2=val;VAR=val;..." """
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1].split()
for rep in range(2):
    for v in sys.argv[2].split(";"):
        env = dict(os.environ)
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=", 1); env[k] = val
        out = subprocess.run([sys.executable, os.path.join(HERE, "bench_step_walk.py")] + args, env=env, capture_output=True, text=True)
        print(f"[{v}] {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
