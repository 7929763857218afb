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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libdgl_amd.so")
    if verbose:
        sys.stderr.write(res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print(build_peaks(force="--force" in sys.argv))
