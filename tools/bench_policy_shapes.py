"""A/B of the policy kernel's workgroup shapes (DGL_POLICY_SHAPE) on live data: each shape runs in a child process
(the shape is read once per process), same device, interleaved repetitions."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
R = sys.argv[1] if len(sys.argv) > 1 else "512"
shapes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["12", "11"]
for rep in range(3):
    for sh in shapes:
        env = dict(os.environ, DGL_POLICY_SHAPE=sh)
        out = subprocess.run([sys.executable, os.path.join(HERE, "bench_policy.py"), R, "2000"], env=env, capture_output=True, text=True)
        print(f"shape {sh}: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
