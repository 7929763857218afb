"""Policy kernel, shape 12 (two rows per workgroup), with fewer and fewer workgroups: does a workgroup get faster
when fewer of them stream the weights at the same time?"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for R in (512, 256, 128, 64, 16):
    env = dict(os.environ, DGL_POLICY_SHAPE="12")
    out = subprocess.run([sys.executable, os.path.join(HERE, "bench_policy.py"), str(R), "2000"], env=env, capture_output=True, text=True)
    print(out.stdout.strip(), flush=True)
