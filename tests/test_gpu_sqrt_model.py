"""The "mkl_avx512" square-root model (include/dgl_amd.h: dgl_set_sqrt_model): with it the device carries the state of
the reference as it ran in the container that made the golden vectors - torch built against MKL on an AVX-512 CPU - bit
for bit: square roots, reset moments, token tiles; and the greedy tours of the 512-instance populations follow."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture
def mkl(device):
    from dgl_amd import sqrt_model
    sqrt_model.use("mkl_avx512", device)
    yield sqrt_model
    sqrt_model.use(sqrt_model.default_model(), device)


def test_square_roots_bit_for_bit(device, mkl):
    g = np.load(os.path.join(GOLDEN, "torch_sqrt_mkl.npz"))
    x = torch.from_numpy(g["x"]).to(device)
    got = mkl.apply(x).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), g["sqrt"].view(np.uint32))
    ieee = np.sqrt(g["x"])
    assert (g["sqrt"] != ieee).sum() > 400                  # the fixture does contain the roots where the two differ
    mkl.use("ieee", device)
    assert np.array_equal(mkl.apply(x).cpu().numpy().view(np.uint32), ieee.view(np.uint32))


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n1000_b2", "tsp_n1000_b128_nn_tours", "tsp_n20_b4"])
def test_tsp_reset_moments_bit_for_bit(name, device, mkl):
    from dgl_amd.engine import TspEngine
    case, _ = load_golden(name)
    eng = TspEngine(torch.from_numpy(case["xy"]).to(device), 1, 30)
    eng.reset()
    assert np.array_equal(eng.state["avg"].cpu().numpy(), case["avg0"])
    assert np.array_equal(eng.state["sd"].cpu().numpy(), case["std0"])


@pytest.mark.parametrize("name", ["cvrp_n60_b4", "cvrp_n100_b8", "cvrp_n1000_b2", "cvrp_n1000_b128_nn_tours", "cvrp_n20_b4"])
def test_cvrp_reset_moments_bit_for_bit(name, device, mkl):
    from test_gpu_cvrp import _engine
    case, _ = load_golden(name)
    eng = _engine(case, device)
    eng.reset()
    assert np.array_equal(eng.state["avg"].cpu().numpy(), case["avg0"])
    assert np.array_equal(eng.state["sd"].cpu().numpy(), case["std0"])


@pytest.mark.parametrize("name", ["tsp_n50_b4", "tsp_n100_b8", "tsp_n1000_b2", "tsp_n20_b4"])
def test_tsp_state_bit_for_bit_along_the_reference_tour(name, device, mkl):
    """Driven with the reference's own tour: the dense state (coordinates, distance to the current node, running mean and
    standard deviation, directions) equals the reference's snapshots bit for bit, down to the last steps of the tour,
    and so do the normalised token tiles where the fixture holds them."""
    from dgl_amd.engine import TspEngine
    case, _ = load_golden(name)
    eng = TspEngine(torch.from_numpy(case["xy"]).to(device), 1, int(case["meta/knn"]))
    tour = torch.from_numpy(case["tour"]).to(device=device, dtype=torch.int32)
    N = eng.N
    snaps = sorted(int(k[4:].split("/")[0]) for k in case if k.startswith("snap") and k.endswith("/data"))
    eng.reset()
    checked = tiles = 0
    for t in range(1, N):
        eng.step(tour[:, t - 1].contiguous())
        if t in snaps:
            data = eng.data().cpu().numpy()
            g = case[f"snap{t}/data"]
            assert np.array_equal(data.view(np.uint32), g.view(np.uint32)), f"state differs at step {t}"
            checked += 1
        if "tile" in case:
            k = eng.k
            got = eng.tile.view(eng.R, -1, 12)[:, :k + 1, :7].cpu().numpy()
            want = case["tile"][t - 1].reshape(eng.R, -1, 7)[:, :k + 1]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"tile differs at step {t}"
            tiles += 1
    print(f"{name}: {checked} state snapshots and {tiles} token tiles equal to the reference's bit for bit")
    assert checked >= 2


@pytest.mark.parametrize("name,min_identical", [("tsp_n1000_b512_nn_tours", 0.99), ("tsp_n1000_b128_clustered_nn_tours", 0.97)])
def test_tsp_population_tours_under_the_references_square_roots(name, min_identical, device, mkl):
    from dgl_amd.engine import PackedPolicy, TspEngine
    from dgl_amd._lib import PROBLEM_TSP
    case, weights = load_golden(name)
    eng = TspEngine(torch.from_numpy(case["xy"]).to(device), 1, int(case["meta/knn"]))
    pol = PackedPolicy(weights, PROBLEM_TSP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B)
    eng.greedy_rollout(pol, start)
    same = (eng.state["tour"].cpu().numpy() == case["tour"]).all(axis=1)
    tie = case["ref_tie"] if "ref_tie" in case else np.zeros_like(same)
    print(f"{name} (mkl_avx512 roots): identical tours {int(same.sum())}/{len(same)} ({100 * same.mean():.2f} %); without the "
          f"{int(tie.sum())} instances whose reference trajectory passes a neighbourhood tie {int((same & ~tie).sum())}/{int((~tie).sum())}")
    assert same.mean() >= min_identical


@pytest.mark.parametrize("name,min_identical", [("cvrp_n1000_b512_nn_tours", 0.99), ("cvrp_n1000_b128_clustered_nn_tours", 0.95)])
def test_cvrp_population_routes_under_the_references_square_roots(name, min_identical, device, mkl):
    from dgl_amd._lib import PROBLEM_CVRP
    from dgl_amd.engine import PackedPolicy
    from test_gpu_cvrp import _engine
    case, weights = load_golden(name)
    eng = _engine(case, device)
    pol = PackedPolicy(weights, PROBLEM_CVRP, device)
    start = torch.from_numpy(case["start"]).to(device=device, dtype=torch.int32).repeat(eng.B) + 2 + eng.N
    eng.greedy_rollout(pol, start)
    nodes, flags = eng.state["nodes"].cpu().numpy(), eng.state["flags"].cpu().numpy()
    same = (nodes == case["nodes"]).all(axis=1) & (flags == case["flags"]).all(axis=1)
    tie = case["ref_tie"] if "ref_tie" in case else np.zeros_like(same)
    print(f"{name} (mkl_avx512 roots): identical routes {int(same.sum())}/{len(same)} ({100 * same.mean():.2f} %); without the "
          f"{int(tie.sum())} instances whose reference trajectory passes a neighbourhood tie {int((same & ~tie).sum())}/{int((~tie).sum())}")
    assert same.mean() >= min_identical


@pytest.mark.parametrize("name", ["cvrp_n60_b4", "cvrp_n100_b8", "cvrp_n1000_b2", "cvrp_n20_b4"])
def test_cvrp_state_bit_for_bit_along_the_reference_route(name, device, mkl):
    """The CVRP counterpart: dense state snapshots (and capacities) along the reference's own route, and the normalised
    token tiles of both neighbourhoods where the fixture holds them, bit for bit."""
    from test_gpu_cvrp import _actions, _engine
    case, _ = load_golden(name)
    eng = _engine(case, device)
    acts = torch.from_numpy(_actions(case)).to(device)
    N = eng.N
    snaps = sorted(int(k[4:].split("/")[0]) for k in case if k.startswith("snap") and k.endswith("/data"))
    eng.reset()
    checked = tiles = 0
    for t in range(1, N):
        eng.step(acts[:, t - 1].contiguous())
        if t in snaps:
            assert np.array_equal(eng.data().cpu().numpy().view(np.uint32), case[f"snap{t}/data"].view(np.uint32)), t
            assert np.array_equal(eng.state["capacity"].cpu().numpy(), case[f"snap{t}/capacity"]), t
            checked += 1
        if "tile" in case:
            kl, kd = eng.kl, eng.kd
            n = kl + kd + 2
            got = eng.tile.view(eng.R, -1, 12)[:, :n, :9].cpu().numpy()
            want = case["tile"][t - 1].reshape(eng.R, -1, 9)[:, :n]
            # the demand column of the last group is demand / capacity after the in-place capacity nudge: same bits too
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"tile differs at step {t}"
            tiles += 1
    print(f"{name}: {checked} state snapshots and {tiles} token tiles equal to the reference's bit for bit")
    assert checked >= 2


def test_square_roots_follow_the_recorded_instruction_on_random_inputs(device, mkl):
    """dgl_sqrt_apply (the kernels' closed form of VRSQRT14) against the NumPy restatement of MKL's AVX-512 float sqrt
    over the recording of the instruction
    (tools/make_rsqrt14_table.py::model_sqrt, itself equal to torch.sqrt on the fixture above): four million inputs
    over sixty binades, the exact powers of two and four among them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from make_rsqrt14_table import model_sqrt
    from dgl_amd.sqrt_model import rsqrt14_table_full
    rng = np.random.default_rng(5)
    x = (rng.random(4_000_000, dtype=np.float32) + np.float32(1.0)) * np.exp2(rng.integers(-30, 30, 4_000_000)).astype(np.float32)
    x[:64] = np.exp2(np.arange(-32, 32)).astype(np.float32)
    want = model_sqrt(x, rsqrt14_table_full())
    got = mkl.apply(torch.from_numpy(x).to(device)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (want != np.sqrt(x)).mean() > 0.003
    # whole waves take the closed form through the lanes' registers, a ragged tail through memory: 63 inputs are all tail
    for lo in range(0, 63 * 50, 63):
        got = mkl.apply(torch.from_numpy(x[lo:lo + 63].copy()).to(device)).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want[lo:lo + 63].view(np.uint32))
    # zeros, negative numbers, infinities and NaN: as under the other model
    odd = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, -1e-30, 2.0, 0.0, 3.0] * 13, np.float32)
    with np.errstate(invalid="ignore"):
        want_odd = np.where((odd == 2.0) | (odd == 3.0), model_sqrt(odd, rsqrt14_table_full()), np.sqrt(odd))
    for part in (odd, odd[:50]):                        # (two whole waves and two lanes / a ragged wave)
        got = mkl.apply(torch.from_numpy(part.copy()).to(device)).cpu().numpy()
        ok = (got.view(np.uint32) == want_odd[:part.size].view(np.uint32)) | (np.isnan(got) & np.isnan(want_odd[:part.size]))
        assert ok.all(), (part[~ok], got[~ok])
