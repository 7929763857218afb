"""Two ranks on one card: the multi-GPU paths (instance-sharded evaluation with the statistics all-reduce; the
training step's gradient all-reduce) rehearsed with two gloo processes that both use cuda:0, against the same work in
one process.  The children are started by conftest.py when the session starts - before this process has touched the
GPU - and only collected here."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_card_match_one_process(device, multirank_children):
    out_dir, procs = multirank_children
    for name, p in procs.items():
        rc = p.wait(timeout=600)
        log = open(os.path.join(out_dir, name + ".log")).read()
        assert rc == 0, f"{name} exited with {rc}:\n{log[-3000:]}"
    one = json.load(open(os.path.join(out_dir, "eval_w1.json")))
    two = json.load(open(os.path.join(out_dir, "eval_w2.json")))
    # instances are decoded by the same kernels whichever rank holds them; the sums are float64
    assert abs(one["score"] - two["score"]) < 1e-9 * abs(one["score"]) and abs(one["gap"] - two["gap"]) < 1e-9
    assert 5.0 < one["score"] < 20.0
    g1, g2 = np.load(os.path.join(out_dir, "grad_w1.npy")), np.load(os.path.join(out_dir, "grad_w2.npy"))
    p1, p2 = np.load(os.path.join(out_dir, "params_w1.npy")), np.load(os.path.join(out_dir, "params_w2.npy"))
    # Each rank's gradient is a sum over its own rows; one process sums all eight rows inside its matrix products, in
    # another order: equal up to float rounding of a 248-term sum, not bit for bit.
    scale = np.abs(g1).max()
    err = np.abs(g1 - g2).max() / scale
    print(f"two ranks vs one process: score {two['score']:.6f} vs {one['score']:.6f}; all-reduced gradient differs by "
          f"{err:.2e} of its largest entry; parameters after Adam by {np.abs(p1 - p2).max():.2e}")
    assert scale > 0 and err < 2e-5
    assert np.abs(p1 - p2).max() <= 2.1e-4          # Adam moves an entry by at most lr = 1e-4 per step, in either direction
