#!/usr/bin/env python3
"""Benchmark of the DGL greedy decode path on MI355X.

A *step* is one full greedy decode of one batch of synthetic instances (reset, N-1 policy
decisions with the fused state-update/kNN/tile kernel in between, tour length): one pass
of the hot path over one batch.  Headline workload: TSP-1000, 512 uniform instances per GPU,
k = 30 (BASELINE.json configs[1]); by default CVRP-1000 (configs[2]'s workload, k = kd = 15) is
run after it with the same step / warm-up counts and reported under the key ``cvrp1000`` of the
same JSON line (``--problem tsp`` / ``--problem cvrp`` run one of them alone).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  ``value`` is whole-job instances/s (all ranks; instances are
independent, so ranks shard them with no data-path collective: weak scaling); only the
tour-length statistics are all-reduced (RCCL) at the end.  ``roofline`` describes the
dominant kernel (the policy network), timed with HIP events on the launch stream inside the
timed region: ``achieved`` = algorithmic flops per launch / average launch time, ``frac`` = that
over the nominal dense f16 matrix peak.  (The kernel reaches binary32 accuracy by evaluating
every product as three f16 products; ``frac_of_executed_f16_work`` = 3 x frac is the matrix-core
utilisation of the executed work.)  ``policy_f32_kernel`` times the selectable f32-input-MFMA
kernel on the same tiles; ``roofline_knn`` the HBM-bound state-update + kNN + tile kernel, at
the workload's batch and at a batch large enough to fill the chip; ``measured_peaks`` what this
card delivers for a float4 triad / copy and for back-to-back matrix instructions on random
operands; ``tsp10000_beam16`` and ``cvrp5000_train`` are BASELINE.json's configurations 4 and 5 on one GPU (a beam-16
decode of TSP-10000 instances; a training decode step of the CVRP policy at knn = depot_knn = 100 on CVRP-5000 rows);
``ieee_roots_mode`` the two headline workloads and the step kernel's saturating batch once more with correctly rounded
square roots (the lines above run under the engines' default, the square-root model of the reference's torch build);
``cvrplib_k50_forward`` one policy evaluation on rows of 102 tokens (the CVRPLIB neighbourhoods, knn = depot_knn = 50).
``cpu_baseline`` is the oracle (the PyTorch-CPU restatement of the reference path) on a
bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X dense fp32 MFMA (v_mfma_f32_32x32x2_f32), spec
PEAK_F16_MATRIX_TFLOPS = 2516.8   # MI355X dense f16/bf16 MFMA (16 x the f32-input rate), spec
SPLIT_PASSES = 3                  # f16 products per f32-accurate product (hi*hi + hi*lo + lo*hi)
PEAK_HBM_GBS = 8000.0             # MI355X HBM3E, spec


def policy_flops_per_row(n: int, feat: int, anchors: int) -> float:
    """Forward flops of the 1+3-block transformer for one row of n tokens (2 flops per MAC):
    per block n*393216 (q,k,v,combine,FF) + 512 n^2 (QK^T and PV over 8 heads x 16)."""
    return 4 * (n * 393216 + 512 * n * n) + n * 2 * feat * 128 + 32768 * anchors + n * 256


def knn_bytes_per_row(nodes: int, k: int, cvrp: bool) -> float:
    """Algorithmic HBM bytes of the fused state-update + kNN + tile kernel per row and step.  The CVRP figure counts
    one demand read per node; the kernel reads the demand of the candidates only (about 24 m bytes really move)."""
    if cvrp:
        m = nodes + 1
        return 28 * m + m / 4 + 8 * k
    return 24 * nodes + nodes / 8 + 8 * k


KERNEL_SOURCES = ["policy16.hip", "policy16_body.h", "split_f16.h", "policy.hip", "policy_layout.h", "tsp_state.hip",
                  "tsp_step_body.h", "cvrp_state.hip", "select.h", "state_common.h", "common.h"]


def kernel_source_hash() -> str:
    """Hash of the sources of the kernels whose counter traffic profiles/pmc_traffic.json records."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "dgl_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(problem, N, B, knn, kernel):
    """HBM-side bytes per launch of ``kernel`` from the committed counter pass of this workload (the counters
    cannot be read from inside the process).  None when no pass was recorded for it, or when the kernels' sources
    have changed since the pass was recorded (the record carries their hash)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except OSError:
        return None
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    entry = rec.get(f"{problem}{N}_b{B}_k{knn}", {}).get(kernel)
    return entry["bytes_per_launch"] if entry else None


def measured_peaks(dev):
    """What this card delivers: streaming kernels over 1 GiB arrays and back-to-back matrix instructions on random
    operands (dgl_amd/csrc/bench_peaks.hip), quoted beside the nominal figures."""
    import ctypes
    from dgl_amd import build
    lib = ctypes.CDLL(build.build_peaks())
    vp, f, i64, i32 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int64, ctypes.c_int32
    lib.dglb_triad.argtypes = [vp, vp, vp, f, i64, i32, i32, vp]
    lib.dglb_copy.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.dglb_mfma_f16.argtypes = [i32, i32, vp, vp]
    lib.dglb_mfma_f32.argtypes = [i32, i32, vp, vp]
    lib.dglb_mfma_f16_16.argtypes = [i32, i32, vp, vp]
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 28                                               # 1 GiB per array: far beyond the 256 MiB Infinity Cache
    a, b, c = (torch.rand(n, device=dev) for _ in range(3))
    sink = torch.zeros(4, device=dev)

    def timed(fn, reps):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    # the best of {1, 4 float4 per thread} x {plain, non-temporal} accesses
    variants = [(u, nt) for u in (1, 4) for nt in (0, 1)]
    t_triad = min(timed(lambda: lib.dglb_triad(a.data_ptr(), b.data_ptr(), c.data_ptr(), 1.5, n, u, nt, st), 8) for u, nt in variants)
    t_copy = min(timed(lambda: lib.dglb_copy(a.data_ptr(), b.data_ptr(), n, u, nt, st), 8) for u, nt in variants)
    del a, b, c
    blocks, it16, it32 = 2048, 4000, 1000
    # ~0.5 s of back-to-back launches on random operands first: the clock the chip settles on under this load
    t16 = timed(lambda: lib.dglb_mfma_f16(blocks, it16, sink.data_ptr(), st), 60)
    t16b = timed(lambda: lib.dglb_mfma_f16_16(blocks, it16, sink.data_ptr(), st), 60)
    t32 = timed(lambda: lib.dglb_mfma_f32(blocks, it32, sink.data_ptr(), st), 20)
    fl16 = blocks * 4 * it16 * 4 * 2.0 * 32 * 32 * 16
    fl32 = blocks * 4 * it32 * 4 * 2.0 * 32 * 32 * 2
    return {"hbm_triad_GBs": 3 * 4 * n / t_triad / 1e9, "hbm_copy_GBs": 2 * 4 * n / t_copy / 1e9,
            "mfma_f16_32x32x16_TFLOPs": fl16 / t16 / 1e12, "mfma_f16_16x16x32_TFLOPs": blocks * 4 * it16 * 8 * 2.0 * 16 * 16 * 32 / t16b / 1e12,
            "mfma_f32_32x32x2_TFLOPs": fl32 / t32 / 1e12,
            "nominal": {"hbm_GBs": PEAK_HBM_GBS, "mfma_f16_TFLOPs": PEAK_F16_MATRIX_TFLOPS,
                        "mfma_f32_TFLOPs": PEAK_FP32_MATRIX_TFLOPS},
            "note": "float4 triad / copy over 1 GiB arrays; register-resident matrix instructions on random operands, "
                    "8+ waves per SIMD-second of warm-up; roofline fractions are quoted against the NOMINAL peaks"}


def run_problem(problem, args, world, rank, dev, dist):
    """Times args.steps greedy rollouts of one problem on this rank's batch; returns this rank's result fields
    (rank 0 gets the kernel-level fields)."""
    from dgl_amd._lib import ARITH_F32, ARITH_SPLIT_F16, PROBLEM_CVRP, PROBLEM_TSP
    from dgl_amd.engine import CvrpEngine, PackedPolicy, TspEngine
    from dgl_amd.weights import random_state_dict

    cvrp = problem == "cvrp"
    N, B = args.nodes, args.batch
    knn = args.knn or (15 if cvrp else 30)
    gen = torch.Generator().manual_seed(789 + rank)
    if cvrp:
        depot = torch.rand(B, 1, 2, generator=gen)
        nodes = torch.rand(B, N, 2, generator=gen)
        demand = torch.randint(1, 200, (B, N), generator=gen) / 500.0      # CVRP/CVRPEnv.py:42-48
        xy = torch.cat((depot, nodes), 1).to(dev)
        dem = torch.cat((torch.zeros(B, 1), demand), 1).to(dev)
        eng = CvrpEngine(xy, dem, 1, knn, knn)
        policy = PackedPolicy(random_state_dict("CVRP", 789), PROBLEM_CVRP, dev)
        start_node = int(torch.randperm(N, generator=gen)[0])
        start = torch.full((B,), start_node + 2 + N, dtype=torch.int32, device=dev)   # CVRP/CVRPTester.py:183
        n_tok, feat, anchors = 2 * knn + 2, 9, 2
    else:
        xy = torch.rand(B, N, 2, generator=gen).to(dev)
        eng = TspEngine(xy, 1, knn)
        policy = PackedPolicy(random_state_dict("TSP", 789), PROBLEM_TSP, dev)
        start = torch.full((B,), int(torch.randperm(N, generator=gen)[0]), dtype=torch.int32, device=dev)
        n_tok, feat, anchors = knn + 1, 7, 1
    decisions = N - 1

    ev_policy, ev_knn, ev_null = [], [], []

    def rollout(sample: bool):
        eng.overflow.zero_()
        eng.reset()
        eng.step(start)
        s = 0
        while eng.t < N:
            probe = sample and (s % 16 == 8) and (eng.kd if cvrp else eng.k) == knn
            if probe:
                # an empty event pair measures what an event boundary itself adds to an interval
                n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n0.record(); n1.record()
                ev_null.append((n0, n1))
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            eng.forward(policy, want_logits=False, want_prob=False)
            if probe:
                e1.record()
            eng.step(eng.selected)
            if probe:
                e2.record()
                ev_policy.append((e0, e1))
                ev_knn.append((e1, e2))
            s += 1
        return eng.tour_length()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        rollout(False)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        length = rollout(True)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # the range guard of the split-f16 arithmetic, read once per run (random-init weights stay far inside the range)
    assert int(eng.overflow.item()) == 0, "an activation left the binary16 range: rerun with DGL_ARITHMETIC=f32"

    stats = torch.tensor([float(length.double().sum()), float(B), elapsed], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(stats[:2], op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0])
    mean_len = float(stats[0] / stats[1])
    if rank != 0:
        return None

    ms_null = sum(a.elapsed_time(b) for a, b in ev_null) / max(1, len(ev_null))
    raw_policy = sum(a.elapsed_time(b) for a, b in ev_policy) / max(1, len(ev_policy))
    raw_knn = sum(a.elapsed_time(b) for a, b in ev_knn) / max(1, len(ev_knn))
    ms_policy, ms_knn = raw_policy - ms_null, raw_knn - ms_null
    flops = policy_flops_per_row(n_tok, feat, anchors) * B
    split = eng.arithmetic == ARITH_SPLIT_F16
    peak_tf = PEAK_F16_MATRIX_TFLOPS if split else PEAK_FP32_MATRIX_TFLOPS
    ach_tf = flops / (ms_policy * 1e-3) / 1e12 if ms_policy > 0 else 0.0
    bytes_knn = knn_bytes_per_row(N, 2 * knn if cvrp else knn, cvrp) * B
    ach_gbs = bytes_knn / (ms_knn * 1e-3) / 1e9 if ms_knn > 0 else 0.0
    kname = "policy_kernel_f16" if split else "policy_kernel_f32"
    from dgl_amd import sqrt_model as _sqrt_model
    sqrt_model_name = _sqrt_model.current(dev)
    out = {
        "metric": f"instances/sec, {problem.upper()}-{N} greedy decode (node selection)",
        "value": world * B * args.steps / elapsed,
        "unit": "instances/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f16 split-operand MFMA x3, f32 accumulate)" if split else "f32", "data": "synthetic",
        "config": {"workload": f"{problem.upper()}-{N} greedy decode, {B} uniform instances per GPU, "
                               f"k={knn}" + (f"/{knn}" if cvrp else "") + ", random-init weights",
                   "instances_per_gpu": B, "nodes": N, "knn": knn, "parallelism": f"instance-sharded x{world}",
                   "sqrt_model": sqrt_model_name},
        "decode_steps_per_s": args.steps * decisions / elapsed,
        "row_decisions_per_s": world * B * args.steps * decisions / elapsed,
        "mean_tour_length": mean_len,
        "roofline": {"kernel": kname, "bound": "mfma",
                     "achieved": ach_tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach_tf / peak_tf,
                     "traffic": pmc_traffic(problem, N, B, knn, kname),
                     "traffic_unit": "HBM-side bytes per launch, rocprofv3 PMC pass recorded in profiles/pmc_traffic.json "
                                     "(null when the kernel sources changed since)",
                     "avg_launch_ms": ms_policy, "flops_per_launch": flops,
                     "samples": len(ev_policy), "event_interval_ms": raw_policy, "empty_event_interval_ms": ms_null,
                     "peak_note": ("nominal dense f16 MFMA peak; `achieved` counts ALGORITHMIC flops (one per f32-accurate "
                                   f"multiply-add), the kernel executes {SPLIT_PASSES} f16 products for each: "
                                   "frac_of_executed_f16_work is the matrix-core utilisation of the executed work"
                                   if split else "nominal dense f32-input MFMA peak"),
                     "frac_of_executed_f16_work": (SPLIT_PASSES * ach_tf / PEAK_F16_MATRIX_TFLOPS) if split else None},
        "roofline_knn": {"kernel": "step_kernel", "bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS,
                         "traffic": pmc_traffic(problem, N, B, knn, "step_kernel"),
                         "avg_launch_ms": ms_knn, "event_interval_ms": raw_knn, "bytes_per_launch": bytes_knn, "rows": B},
    }
    if cvrp:
        out["roofline_knn"]["bytes_note"] = ("per-row figure 28 m + m/4 + 8 (k + kd) with a demand read per node; the "
                                             "kernel reads the candidates' demand only, about 24 m bytes move")
    if not args.no_aux and world == 1:
        out["roofline_knn"]["saturating_batch"] = sat = knn_saturating(dev, problem, N, knn)
        sat["traffic"] = pmc_traffic(problem, N, sat["rows"], knn, "step_kernel")
        if split and not cvrp:                   # the f32-input kernel on tiles of the same shape, outside the timed region
            eng.reset(); eng.step(start)
            eng.arithmetic = ARITH_F32
            for _ in range(3):
                eng.forward(policy, want_logits=False, want_prob=False)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                eng.forward(policy, want_logits=False, want_prob=False)
            b.record(); torch.cuda.synchronize()
            eng.arithmetic = ARITH_SPLIT_F16
            ms32 = a.elapsed_time(b) / 10
            out["policy_f32_kernel"] = {"kernel": "policy_kernel_f32", "avg_launch_ms": ms32,
                                        "achieved": flops / (ms32 * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS,
                                        "unit": "TFLOP/s", "frac": flops / (ms32 * 1e-3) / 1e12 / PEAK_FP32_MATRIX_TFLOPS}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(problem, N, knn, args.cpu_sample)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--problem", choices=["both", "tsp", "cvrp"], default="both",
                    help="both: TSP is the headline line, CVRP's results ride along under the key cvrp<nodes>")
    ap.add_argument("--nodes", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=512, help="instances per GPU")
    ap.add_argument("--knn", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64, help="instances decoded by the CPU baseline")
    ap.add_argument("--no-aux", action="store_true", help="skip the saturating-batch kNN and measured-peak measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
            sys.exit(2)
    # one rank per GPU; DGL_BENCH_BACKEND=gloo lets several ranks rehearse the multi-rank path on one card
    backend = os.environ.get("DGL_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from dgl_amd import build
    if local_rank == 0:
        build.build()                               # no-op when the in-tree library is current
    if dist is not None:
        dist.barrier()                              # the other ranks load the library only after that

    first = "cvrp" if args.problem == "cvrp" else "tsp"
    out = run_problem(first, args, world, rank, dev, dist)
    other = run_problem("cvrp", args, world, rank, dev, dist) if args.problem == "both" else None
    if rank == 0:
        if other is not None:
            out[f"cvrp{args.nodes}"] = {k: other[k] for k in ("metric", "value", "unit", "ms_per_step", "config",
                                                              "decode_steps_per_s", "mean_tour_length", "roofline",
                                                              "roofline_knn", "cpu_baseline") if k in other}
        if not args.no_aux and world == 1:
            out["ieee_roots_mode"] = bench_ieee_roots_mode(args, dev)
            out["tsp10000_beam16"] = bench_tsp10000_beam(dev)
            out["cvrp5000_train"] = bench_cvrp5000_train(dev)
            out["cvrplib_k50_forward"] = bench_wide_forward(dev)
            out["measured_peaks"] = mp = measured_peaks(dev)
            for o in (out, out.get(f"cvrp{args.nodes}")):
                if o:
                    o["roofline"]["frac_of_measured_f16_peak"] = o["roofline"]["achieved"] / mp["mfma_f16_32x32x16_TFLOPs"]
                    o["roofline_knn"]["frac_of_measured_copy_rate"] = o["roofline_knn"]["achieved"] / mp["hbm_copy_GBs"]
        out["kernel_source_hash"] = kernel_source_hash()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def knn_saturating(dev, problem: str, N: int, knn: int, rows: int = 16384):
    """The HBM-bound kernel at a batch that fills the chip (same kernel, same N)."""
    from dgl_amd.engine import CvrpEngine, TspEngine
    g = torch.Generator().manual_seed(1)
    cvrp = problem == "cvrp"
    if cvrp:
        xy = torch.rand(rows, N + 1, 2, generator=g).to(dev)
        dem = torch.cat((torch.zeros(rows, 1), torch.randint(1, 200, (rows, N), generator=g) / 500.0), 1).to(dev)
        eng = CvrpEngine(xy, dem, 1, knn, knn)
        sel = torch.full((rows,), 7 + 2 + N, dtype=torch.int32, device=dev)
    else:
        xy = torch.rand(rows, N, 2, generator=g).to(dev)
        eng = TspEngine(xy, 1, knn)
        sel = torch.full((rows,), 7, dtype=torch.int32, device=dev)
    eng.reset()
    eng.step(sel)
    nxt = [torch.full((rows,), 100 + i + (0 if not cvrp else 0), dtype=torch.int32, device=dev) for i in range(12)]
    torch.cuda.synchronize()
    evs = []
    for i in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.step(nxt[i])
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
    ms = sum(ms) / len(ms)
    by = knn_bytes_per_row(N, 2 * knn if cvrp else knn, cvrp) * rows
    gbs = by / (ms * 1e-3) / 1e9
    # the normalised [n][12] float tile each row also writes is not part of the per-row figure `achieved` is
    # computed from; reported beside it
    tokens = (2 * knn + 2) if cvrp else (knn + 1)
    tile_by = tokens * 12 * 4 * rows
    return {"rows": rows, "avg_launch_ms": ms, "achieved": gbs, "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": by,
            "tile_output_bytes": tile_by, "achieved_with_tile_output": (by + tile_by) / (ms * 1e-3) / 1e9}


def bench_ieee_roots_mode(args, dev):
    """The headline workloads once more with correctly rounded square roots (DGL_SQRT_MODEL=ieee, dgl_amd/sqrt_model.py).
    The lines above run under the engines' default, the model of the reference's own torch build (MKL's AVX-512 sqrt,
    torch.mean's summation order): the state is then the reference's bit for bit and the >= 99 % identical-tours tests
    against the reference's tours hold; correctly rounded roots are the host-independent alternative, about 2 % faster
    per rollout and 19 % at a saturating batch of the step kernel."""
    import copy
    from dgl_amd import sqrt_model
    a = copy.copy(args)
    a.no_aux, a.no_cpu_baseline, a.steps, a.warmup = True, True, min(args.steps, 2), 1
    sqrt_model.use("ieee", dev)
    try:
        res = {}
        for problem in (("tsp", "cvrp") if args.problem == "both" else (args.problem,)):
            r = run_problem(problem, a, 1, 0, dev, None)
            sat = knn_saturating(dev, problem, r["config"]["nodes"], r["config"]["knn"])
            res[problem] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                            "policy_kernel_ms": r["roofline"]["avg_launch_ms"], "step_kernel_ms": r["roofline_knn"]["avg_launch_ms"],
                            "step_kernel_saturating_batch": {k: sat.get(k) for k in ("rows", "avg_launch_ms", "achieved", "frac")}}
    finally:
        sqrt_model.use(sqrt_model.default_model(), dev)
    res["note"] = ("same workloads with correctly rounded square roots (DGL_SQRT_MODEL=ieee); the headline runs under the "
                   "reference-bits model " + sqrt_model.default_model())
    return res


def bench_tsp10000_beam(dev, beam: int = 16, instances: int = 32, N: int = 10000, knn: int = 30):
    """BASELINE.json configs[3]: TSP-10000 beam-search decode (TSP/test.py:125-129: beam 16 at this size) on the
    device-resident beam path - per decision the policy, the expansion over beam x k candidates and one pass over the
    state that follows the back-pointers; nothing returns to the host.  One warm-up rollout cut short, one timed."""
    from dgl_amd._lib import PROBLEM_TSP
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd.weights import random_state_dict
    g = torch.Generator().manual_seed(4)
    R = instances * beam
    xy = torch.rand(instances, N, 2, generator=g).to(dev)
    eng = TspEngine(xy, beam, knn)
    pol = PackedPolicy(random_state_dict("TSP", 789), PROBLEM_TSP, dev)
    start = torch.full((R,), int(torch.randperm(N, generator=g)[0]), dtype=torch.int32, device=dev)
    # the step kernel alone at this size (greedy form: state update + neighbourhood + tile for R rows of N nodes)
    eng.reset(); eng.step(start)
    evs = []
    for i in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sel = torch.full((R,), 50 + i, dtype=torch.int32, device=dev)
        a.record(); eng.step(sel); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms_step = sorted(a.elapsed_time(b) for a, b in evs[2:])
    ms_step = sum(ms_step) / len(ms_step)
    by = knn_bytes_per_row(N, knn, False) * R
    eng.beam_rollout(pol, start, beam)                       # warm-up (allocates the history buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    length = eng.beam_rollout(pol, start, beam)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": f"instances/sec, TSP-{N} beam-{beam} decode", "value": instances / dt, "unit": "instances/s",
            "rows": R, "ms_per_rollout": dt * 1e3, "us_per_decode_step": dt / (N - 1) * 1e6,
            "best_of_beam_mean_length": float(length.view(instances, beam).min(1).values.mean()),
            "config": {"workload": f"TSP-{N}, {instances} uniform instances x beam {beam} ({R} rows), k={knn}, "
                                   f"random-init weights, device-resident beam path"},
            "step_kernel": {"avg_launch_ms": ms_step, "bytes_per_launch": by, "achieved": by / (ms_step * 1e-3) / 1e9,
                            "unit": "GB/s", "frac": by / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, "rows": R, "nodes": N,
                            "note": "greedy-form step (in-place update) timed with an event pair per launch"}}


def bench_wide_forward(dev, rows: int = 512, N: int = 1000, k: int = 50, iters: int = 20):
    """One policy evaluation on rows of 2 k + 2 = 102 tokens, the neighbourhoods the reference solves the CVRPLIB
    instances with (CVRP/test_cvrplib.py:326-327: knn = depot_knn = 50): dgl_policy_forward's one-launch kernel for
    33..128 tokens, and the general path's sequence of launches on the same tiles beside it."""
    import ctypes
    from dgl_amd._lib import PROBLEM_CVRP, check, ptr, stream_ptr
    from dgl_amd.engine import CvrpEngine, PackedPolicy, general_workspace
    from dgl_amd.weights import random_state_dict
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(rows, N + 1, 2, generator=g).to(dev)
    dem = torch.cat((torch.zeros(rows, 1), torch.randint(1, 200, (rows, N), generator=g) / 500.0), 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, k, k)
    pol = PackedPolicy(random_state_dict("CVRP", 1), PROBLEM_CVRP, dev)
    eng.reset()
    eng.step(torch.full((rows,), 5 + 2 + N, dtype=torch.int32, device=dev))
    io = eng.policy_io(want_logits=False, want_prob=False)
    ws = general_workspace(eng, io.R, io.n)

    def wide():
        check(eng.lib.dgl_policy_forward(ptr(pol.packed), ctypes.byref(io), stream_ptr()), "dgl_policy_forward")

    def general():
        check(eng.lib.dgl_policy_general(ctypes.byref(pol.w), None, ctypes.byref(io), 0, 0.0, ptr(ws), 0, stream_ptr()),
              "dgl_policy_general")

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    n = io.n
    flops = rows * policy_flops_per_row(n, 9, 2)
    ms_wide, ms_general = timed(wide), timed(general)
    return {"metric": "us per policy forward, rows of 102 tokens", "value": ms_wide * 1e3, "unit": "us", "higher_is_better": False,
            "rows": rows, "tokens_per_row": n, "achieved": flops / ms_wide / 1e9, "unit_achieved": "TFLOP/s (algorithmic)",
            "frac": flops / ms_wide / 1e9 / PEAK_F16_MATRIX_TFLOPS,
            "general_path_us": ms_general * 1e3,
            "config": {"workload": f"CVRP policy forward, {rows} rows x {n} tokens (knn = depot_knn = {k}), random-init weights"}}


def bench_cvrp5000_train(dev, rows: int = 64, N: int = 5000, knn: int = 100, iters: int = 20):
    """BASELINE.json configs[4]: one teacher-forced training decode step of the CVRP policy at the reference's training
    neighbourhoods (CVRP/train.py:130-131: knn = depot_knn = 100, i.e. 202 tokens per row): forward in mode 'train',
    backward, Adam (CVRP/CVRPTrainer.py:268-272), on rows of a CVRP-5000 state whose distances include the demand
    (CVRP/CVRPEnv.py:519).  Algorithmic flops: 3 x the forward pass (backward = two products per forward product)."""
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import CvrpEngine
    from dgl_amd.train import PolicyTrainer
    from dgl_amd.weights import random_state_dict
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(rows, N + 1, 2, generator=g).to(dev)
    dem = torch.cat((torch.zeros(rows, 1), torch.randint(1, 200, (rows, N), generator=g) / 500.0), 1).to(dev)
    eng = CvrpEngine(xy, dem, 1, knn, knn, demand_in_distance=True)
    tr = PolicyTrainer(random_state_dict("CVRP", 789), PROBLEM_CVRP, dev)
    eng.reset()
    eng.step(torch.full((rows,), 5 + N + 1, dtype=torch.int32, device=dev))
    evs = []
    for i in range(14):                                       # the step kernel with demand_in_distance at this size
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sel = torch.full((rows,), 50 + i, dtype=torch.int32, device=dev)
        a.record(); eng.step(sel); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms_step = sorted(a.elapsed_time(b) for a, b in evs[2:])
    ms_step = sum(ms_step) / len(ms_step)
    by = knn_bytes_per_row(N, 2 * knn, True) * rows
    moves = [torch.full((rows,), 100 + i, dtype=torch.int32, device=dev) for i in range(iters + 3)]

    def one(i):
        # one decode step of CVRPTrainer._train_one_batch: env.step (state update, both neighbourhoods, tile), then
        # model(...) in mode 'train', loss.backward(), optimizer.step() (CVRP/CVRPTrainer.py:268-276)
        eng.step(moves[i])
        tr.zero_grad(); tr.forward_backward(eng, eng.idx_last[:, 0]); tr.optimizer_step()

    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        one(3 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    n = 2 * knn + 2
    flops = 3 * rows * policy_flops_per_row(n, 9, 2)
    return {"metric": "ms per training decode step (env.step + forward + backward + Adam)", "value": ms, "unit": "ms",
            "higher_is_better": False, "rows": rows, "tokens_per_row": n, "steps_timed": iters,
            "achieved": flops / (ms * 1e-3) / 1e12, "unit_achieved": "TFLOP/s (algorithmic: 3 x forward)",
            "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F16_MATRIX_TFLOPS,
            "config": {"workload": f"CVRP-{N} training decode step, {rows} rows x {n} tokens (knn = depot_knn = {knn}), "
                                   f"distances over (x, y, demand), random-init weights, Adam lr 1e-4"},
            "step_kernel": {"avg_launch_ms": ms_step, "bytes_per_launch": by, "achieved": by / (ms_step * 1e-3) / 1e9,
                            "unit": "GB/s", "frac": by / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, "rows": rows, "nodes": N + 1,
                            "note": "demand_in_distance step, both neighbourhoods of 100 out of 5001 keys"}}


def cpu_baseline(problem: str, N: int, knn: int, sample: int):
    """The oracle (PyTorch-CPU restatement of the reference path: dense distance matrix, full
    argsort, [R,N] probability scatter, top-k) decoding `sample` instances of the same workload."""
    from dgl_amd.weights import random_state_dict
    from oracle.dgl_oracle import CVRPOracle, TSPOracle, cvrp_rollout, tsp_rollout
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(789)
    with torch.no_grad():
        if problem == "cvrp":
            depot, nodes = torch.rand(sample, 2, generator=g), torch.rand(sample, N, 2, generator=g)
            demand = torch.randint(1, 200, (sample, N), generator=g) / 500.0
            orc = CVRPOracle(random_state_dict("CVRP", 789), knn, knn)
            t0 = time.perf_counter()
            cvrp_rollout(orc, depot, nodes, demand, torch.tensor([3]))
        else:
            xy = torch.rand(sample, N, 2, generator=g)
            orc = TSPOracle(random_state_dict("TSP", 789), knn)
            t0 = time.perf_counter()
            tsp_rollout(orc, xy, torch.tensor([3]))
        dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "instances/s", "cores": threads, "kind": "port",
            "sample": f"{sample} instances of {problem.upper()}-{N}, one full greedy decode, "
                      f"torch {torch.__version__} CPU, {threads} threads, {dt:.1f} s"}


if __name__ == "__main__":
    main()
