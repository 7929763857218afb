"""Neighbourhood selection and state recurrences of the TSP step kernel vs the plain-C oracle
on synthetic inputs, including the edge cases (ragged N, fewer unvisited nodes than k,
duplicate points = exact key ties, clustered points, large k)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _correctly_rounded_roots(ieee_roots):
    """The plain-C oracle takes correctly rounded square roots: so does the device in this file."""
    yield


def _run_case(device, xy, visit_order, knn, check_until=None, rows=None, model="ieee"):
    """xy [B,N,2] float32 numpy; visit_order [B,N] permutation; drives the kernel with that order and
    checks idx/avg/sd against the C oracle after every step (rows: only these rows are compared).
    model: the square-root model of both sides (dgl_amd/sqrt_model.py, oracle/cref.py::use_model)."""
    from dgl_amd import sqrt_model
    from oracle import cref
    lib = cref.load()
    sqrt_model.use(model, device)
    cref.use_model(lib, model)
    try:
        _compare_with_c_oracle(lib, device, xy, visit_order, knn, check_until, rows, exact_reset=model == "ieee")
    finally:
        cref.use_model(lib, "ieee")
        sqrt_model.use("ieee", device)


def _compare_with_c_oracle(lib, device, xy, visit_order, knn, check_until, rows, exact_reset):
    from dgl_amd.engine import TspEngine
    B, N = xy.shape[0], xy.shape[1]
    rows = range(B) if rows is None else rows
    sel = np.asarray(list(rows))
    eng = TspEngine(torch.from_numpy(xy).to(device), 1, knn)
    eng.reset()
    avg = eng.state["avg"].cpu().numpy().copy()
    sd = eng.state["sd"].cpu().numpy().copy()
    ref_avg, ref_sd = np.empty_like(avg), np.empty_like(sd)
    for b in rows:
        lib.dglref_row_mean_std(np.ascontiguousarray(xy[b]), N, ref_avg[b], ref_sd[b])
    # (the oracle's reset sums in binary64; under the reference-bits model the device sums the means in torch.mean's
    # binary32 order and its distances differ in the last bit here and there)
    np.testing.assert_allclose(avg[sel], ref_avg[sel], rtol=3e-7 if exact_reset else 3e-6)
    np.testing.assert_allclose(sd[sel], ref_sd[sel], rtol=3e-7 if exact_reset else 3e-6)
    order = torch.from_numpy(visit_order.astype(np.int32)).to(device)
    visited = np.zeros((B, N), np.uint8)
    steps = N - 1 if check_until is None else check_until
    for t in range(1, steps + 1):
        eng.step(order[:, t - 1].contiguous())
        idx = eng.idx.cpu().numpy()
        gavg, gsd = eng.state["avg"].cpu().numpy(), eng.state["sd"].cpu().numpy()
        for b in rows:
            cur = int(visit_order[b, t - 1])
            visited[b, cur] = 1
            d = np.empty(N, np.float32)
            # oracle recurrences start from the kernel's own previous state: bit-exact comparison
            lib.dglref_step_features(np.ascontiguousarray(xy[b]), N, cur, N - t, 0, avg[b], sd[b], d)
            ref_idx = np.full(knn, -1, np.int32)
            k = lib.dglref_tsp_knn(np.ascontiguousarray(xy[b]), N, cur, visited[b], knn, ref_idx)
            assert k == min(knn, N - t)
            assert (idx[b] == ref_idx).all(), (t, b, idx[b], ref_idx)
        assert np.array_equal(gavg[sel], avg[sel], equal_nan=True), t
        assert np.array_equal(gsd[sel], sd[sel], equal_nan=True), t
        avg, sd = gavg.copy(), gsd.copy()


@pytest.mark.parametrize("N,knn", [(26, 30), (63, 30), (64, 5), (65, 30), (129, 30), (200, 64), (200, 100), (333, 1)])
def test_uniform(device, N, knn):
    rng = np.random.default_rng(N * 1000 + knn)
    xy = rng.random((3, N, 2), dtype=np.float32)
    order = np.stack([rng.permutation(N) for _ in range(3)])
    _run_case(device, xy, order, knn)


@pytest.mark.parametrize("N,knn,until", [(26, 30, None), (65, 30, None), (200, 64, None), (333, 1, None), (513, 30, None),
                                         (777, 100, 60), (1100, 30, 50), (2500, 65, 10), (5000, 30, 12)])
def test_uniform_under_the_reference_bits_model(device, N, knn, until):
    """The default square-root model (MKL's AVX-512 root, evaluated in closed form by the kernels) against the C oracle
    with the same root looked up in the recording of the instruction: recurrences and neighbourhoods bit for bit, one
    wave per row, shared rows, wide neighbourhoods, shared compaction; and coincident points (roots of zero, ties)."""
    rng = np.random.default_rng(N * 7 + knn)
    xy = rng.random((2, N, 2), dtype=np.float32)
    order = np.stack([rng.permutation(N) for _ in range(2)])
    _run_case(device, xy, order, knn, check_until=until, model="mkl_avx512")
    if N == 200:
        dup = np.ascontiguousarray(np.repeat(rng.random((2, 20, 2), dtype=np.float32), 10, axis=1)[:, rng.permutation(200)])
        _run_case(device, dup, order, 30, model="mkl_avx512")


@pytest.mark.parametrize("N,knn,until", [(513, 30, None), (600, 64, 80), (777, 100, 60), (1100, 30, 50)])
def test_uniform_rows_shared_by_four_waves(device, N, knn, until):
    """From 512 nodes on (and with few rows) four waves share a row: ragged quarters, k > 64, and for N = 513 the
    whole tour down to the last node."""
    rng = np.random.default_rng(N * 1000 + knn)
    xy = rng.random((2, N, 2), dtype=np.float32)
    order = np.stack([rng.permutation(N) for _ in range(2)])
    _run_case(device, xy, order, knn, check_until=until)


@pytest.mark.parametrize("B", [8192, 16384])
def test_many_rows_shared_by_four_waves(device, B):
    """The saturating-batch variant of the same kernel (8192 rows and more; from 16 million row-nodes on the state
    arrays are streamed past the L2 with non-temporal loads and stores: 16384 rows of 1000 nodes)."""
    rng = np.random.default_rng(99)
    N = 512 if B == 8192 else 1000
    xy = rng.random((B, N, 2), dtype=np.float32)
    order = np.stack([rng.permutation(N)[:4] for _ in range(B)])
    order = np.concatenate((order, np.zeros((B, N - 4), dtype=order.dtype)), axis=1)
    _run_case(device, xy, order, 30, check_until=2, rows=None if B == 8192 else range(0, B, 16))


def test_duplicate_points_tie_break(device):
    """Many coincident points: equal keys; the lowest index must win."""
    rng = np.random.default_rng(7)
    base = rng.random((2, 12, 2), dtype=np.float32)
    xy = np.repeat(base, 10, axis=1)            # 120 nodes, 10 copies of each location
    perm = rng.permutation(120)
    xy = np.ascontiguousarray(xy[:, perm])
    order = np.stack([rng.permutation(120) for _ in range(2)])
    _run_case(device, xy, order, 30)


def test_clustered(device):
    rng = np.random.default_rng(11)
    centers = rng.random((4, 2), dtype=np.float32)
    xy = (centers[rng.integers(0, 4, size=(2, 500))] + 0.01 * rng.standard_normal((2, 500, 2))).astype(np.float32)
    xy = np.clip(xy, 0, 1)
    order = np.stack([rng.permutation(500) for _ in range(2)])
    _run_case(device, xy, order, 30, check_until=120)


@pytest.mark.parametrize("N", [300, 1000, 2500])
def test_more_than_64_candidates(device, N):
    """A tight cluster whose nodes all sit in 29 of the 64 index classes (index mod 64): the 30th smallest class
    minimum is then a far node, more than 64 keys lie under it, and the selection has to take its second pass.
    One wave per row, four waves per row, and four waves that also share the compaction."""
    rng = np.random.default_rng(N)
    xy = (0.5 + 0.5 * rng.random((2, N, 2))).astype(np.float32)
    near = np.array([j for j in range(N) if j % 64 < 29][:87])
    xy[:, near] = (0.1 + 0.02 * rng.random((2, near.size, 2))).astype(np.float32)
    order = np.stack([np.concatenate((rng.permutation(near), np.setdiff1d(np.arange(N), near))) for _ in range(2)])
    _run_case(device, xy, order, 30, check_until=25)


@pytest.mark.parametrize("N,knn,until", [(1000, 100, 30), (1000, 200, 12), (1000, 256, 8), (2500, 65, 10), (5000, 100, 16)])
def test_wide_neighbourhoods_on_shared_rows(device, N, knn, until):
    """More than 64 neighbours on rows that several waves share (four waves; eight from 4096 nodes on): one compaction
    pass under a bound taken from the minima of the 256 / 512 scan classes (the reference trains with knn = 100,
    CVRP/train.py:130-131)."""
    rng = np.random.default_rng(N + knn)
    xy = rng.random((2, N, 2), dtype=np.float32)
    order = np.stack([rng.permutation(N) for _ in range(2)])
    _run_case(device, xy, order, knn, check_until=until)


def test_wide_neighbourhood_falls_back_when_the_class_minima_do_not_bound(device):
    """156 near nodes that all sit in 20 of the 256 scan classes: the 100th smallest class minimum is a far node, more
    than 256 keys lie under it, and the selection takes the general path; and duplicate points (equal roots at the
    edge of the neighbourhood) through the one-pass path."""
    rng = np.random.default_rng(3)
    N = 2000
    xy = (0.5 + 0.5 * rng.random((2, N, 2))).astype(np.float32)
    near = np.array([j for j in range(500) if j % 64 < 20])
    xy[:, near] = (0.1 + 0.02 * rng.random((2, near.size, 2))).astype(np.float32)
    order = np.stack([np.concatenate((rng.permutation(near), np.setdiff1d(np.arange(N), near))) for _ in range(2)])
    _run_case(device, xy, order, 100, check_until=12)
    base = rng.random((2, 100, 2), dtype=np.float32)
    dup = np.ascontiguousarray(np.repeat(base, 8, axis=1)[:, rng.permutation(800)])
    _run_case(device, dup, np.stack([rng.permutation(800) for _ in range(2)]), 100, check_until=10)


@pytest.mark.parametrize("N,until", [(5000, 40), (18512, 10)])
def test_large_n(device, N, until):
    """18512 nodes: the largest instance of the reference's TSPLIB table (d18512, utils/utils.py:47-125)."""
    rng = np.random.default_rng(5)
    xy = rng.random((1, N, 2), dtype=np.float32)
    order = rng.permutation(N)[None]
    _run_case(device, xy, order, 30, check_until=until)


def test_reset_size_limit(device):
    """The reset stages an instance's coordinates in LDS: 19200 nodes fit, one more is refused with a message."""
    import ctypes
    from dgl_amd import _lib
    from dgl_amd.engine import TspEngine
    eng = TspEngine(torch.rand(1, 19200, 2, device=device), 1, 30)
    eng.reset()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.state["avg"]).all() and (eng.state["sd"] > 0).all()
    big = TspEngine(torch.rand(1, 19201, 2, device=device), 1, 30)
    assert _lib.load().dgl_tsp_reset(ctypes.byref(big.state["struct"]), 0) == -1
    assert b"19200" in _lib.load().dgl_last_error_string()


def test_bad_arguments(device):
    import ctypes
    from dgl_amd import _lib
    from dgl_amd.engine import TspEngine
    eng = TspEngine(torch.rand(2, 40, 2, device=device), 1, 30)
    lib = _lib.load()
    st = eng.state["struct"]
    sel = torch.zeros(2, dtype=torch.int32, device=device)
    assert lib.dgl_tsp_step(ctypes.byref(st), 0, 1, 30, 0, 0, 0) == -1           # NULL selected
    assert b"selected" in lib.dgl_last_error_string()
    assert lib.dgl_tsp_step(ctypes.byref(st), sel.data_ptr(), 0, 30, 0, 0, 0) == -1   # t out of range
    assert lib.dgl_tsp_step(ctypes.byref(st), sel.data_ptr(), 41, 30, 0, 0, 0) == -1
    bad = _lib.TspState(2, 40, 1, 3, st.xy, st.avg, st.sd, st.visited, st.tour)
    assert lib.dgl_tsp_reset(ctypes.byref(bad), 0) == -1
