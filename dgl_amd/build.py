"""Build the C-ABI shared library (hipcc, gfx950 only) in-tree.

python -m dgl_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libdgl_amd.so")
SOURCES = ["api.cpp", "tsp_state.hip", "cvrp_state.hip", "policy.hip", "policy16.hip", "train.hip", "augment.hip", "selfimprove.hip", "beam.hip", "rollout.hip"]
HEADERS = ["common.h", "select.h", "state_common.h", "policy_layout.h", "policy16_body.h", "tsp_step_body.h", "split_f16.h", os.path.join("..", "..", "include", "dgl_amd.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",            # every FMA is explicit: results must not depend on contraction
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
    "-x", "hip",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


PEAKS_LIB = os.path.join(HERE, "lib", "libdgl_peaks.so")        # measurement helpers of bench.py, not the product ABI
PEAKS_SRC = os.path.join(CSRC, "bench_peaks.hip")


def build_peaks(force: bool = False) -> str:
    if not force and os.path.exists(PEAKS_LIB) and os.path.getmtime(PEAKS_LIB) >= os.path.getmtime(PEAKS_SRC):
        return PEAKS_LIB
    os.makedirs(os.path.dirname(PEAKS_LIB), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-x", "hip",
           PEAKS_SRC, "-o", PEAKS_LIB]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libdgl_peaks.so")
    return PEAKS_LIB


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """One object per source, compiled side by side (a source is recompiled when it or any header is newer than its
    object), then one link.  `extra_flags` / `out`: variant builds for A/B runs (objects kept apart by the flags)."""
    if not force and not extra_flags and out == LIB and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "default"
    objdir = os.path.join(HERE, "lib", "obj", tag)
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    if verbose:
        cflags.insert(0, "-Rpass-analysis=kernel-resource-usage")

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), newest_header):
            return obj, 0, ""
        res = subprocess.run([hipcc()] + cflags + ["-c", path, "-o", obj], capture_output=True, text=True)
        return obj, res.returncode, res.stdout + res.stderr

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as pool:
        done = list(pool.map(compile_one, SOURCES))
    for obj, rc, log in done:
        if rc != 0:
            sys.stderr.write(log)
            raise RuntimeError(f"hipcc failed compiling {os.path.basename(obj)}")
        if verbose:
            sys.stderr.write(log)
    res = subprocess.run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden"] + [o for o, _, _ in done] +
                         ["-o", out], capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking " + os.path.basename(out))
    return out


if __name__ == "__main__":
    # python -m dgl_amd.build [--force] [--verbose] [--variant NAME -DFLAG ...]: a variant goes to lib/libdgl_amd_NAME.so
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(force="--force" in sys.argv, extra_flags=tuple(sys.argv[i + 2:]),
                    out=os.path.join(HERE, "lib", f"libdgl_amd_{sys.argv[i + 1]}.so")))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print(build_peaks(force="--force" in sys.argv))
