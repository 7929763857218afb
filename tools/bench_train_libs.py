"""A/B of the training decode step (tools/bench_train.py) between library builds (DGL_AMD_LIB), interleaved child processes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1].split(",")
args = sys.argv[2:] or ["64"]
for rep in range(3):
    for lib in libs:
        env = dict(os.environ, DGL_AMD_LIB=os.path.join(HERE, "..", "dgl_amd", "lib", lib))
        out = subprocess.run([sys.executable, os.path.join(HERE, "bench_train.py")] + args, env=env, capture_output=True, text=True)
        print(f"{lib}: {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
