"""A/B of dgl_policy_forward between library builds (DGL_AMD_LIB), interleaved child processes on one device."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1].split(",")
R = sys.argv[2] if len(sys.argv) > 2 else "512"
for rep in range(3):
    for lib in libs:
        env = dict(os.environ, DGL_AMD_LIB=os.path.join(HERE, "..", "dgl_amd", "lib", lib))
        out = subprocess.run([sys.executable, os.path.join(HERE, "bench_policy.py"), R, "2000"], env=env, capture_output=True, text=True)
        print(f"{lib}: {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
