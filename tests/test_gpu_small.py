"""Instances of at most 25 points (depot included): torch.cdist evaluates their distances pair by pair,
sqrt(fma(dy, dy, dx dx)) with the scalar sqrt, instead of through the matrix product it takes for larger ones.  The
golden-vector cases of these sizes run in the TSP / CVRP / beam / square-root-model test files; here: the training
reset's distances over (x, y, demand) and the boundary between the two evaluations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [20, 24, 25])
def test_cvrp_training_state_over_three_coordinates(N, device, ieee_roots):
    """CVRP/CVRPEnv.py:519: the training reset measures distances over (x, y, demand).  N = 24 is the last size on the
    pair-by-pair evaluation (25 points with the depot), N = 25 the first on the matrix product.  From the oracle's reset
    moments (the device sums them in another order) every step's running statistics, capacities and both
    neighbourhoods must equal the oracle's bit for bit along a random feasible route."""
    from dgl_amd.engine import CvrpEngine
    from oracle.dgl_oracle import CVRPOracle, random_weights
    B, knn = 8, 8
    M = N + 1
    gen = torch.Generator().manual_seed(500 + N)
    depot, nodes = torch.rand(B, 2, generator=gen), torch.rand(B, N, 2, generator=gen)
    demand = torch.randint(1, 200, (B, N), generator=gen) / 500.0
    orc = CVRPOracle(random_weights("CVRP", 5), knn, knn)
    orc.reset(depot, nodes, demand, train=True, chain=True)
    xy = torch.cat((depot[:, None, :], nodes), 1)
    dem = torch.cat((torch.zeros(B, 1), demand), 1)
    eng = CvrpEngine(xy.to(device), dem.to(device), 1, knn, knn, demand_in_distance=True)
    eng.reset()
    np.testing.assert_allclose(eng.state["avg"].cpu().numpy(), orc.avg.numpy(), rtol=3e-7)
    np.testing.assert_allclose(eng.state["sd"].cpu().numpy(), orc.std.numpy(), rtol=3e-7)
    eng.state["avg"].copy_(orc.avg)
    eng.state["sd"].copy_(orc.std)
    order = torch.stack([torch.randperm(N, generator=gen) + 1 for _ in range(B)])
    cap = np.ones(B, np.float32)
    dem_np = dem.numpy()
    for t in range(1, N):
        act = torch.zeros(B, dtype=torch.long)
        for b in range(B):
            node = int(order[b, t - 1])
            via = t == 1 or cap[b] + np.float32(1e-6) < dem_np[b, node]
            cap[b] = np.float32(1.0) - dem_np[b, node] if via else cap[b] - dem_np[b, node]
            act[b] = node + (M if via else 0)
        eng.step(act.to(device=device, dtype=torch.int32))
        orc.step(act.clone())
        unv = np.ones((B, M), bool)
        unv[np.arange(B)[:, None], orc.nodes.numpy()] = False
        assert np.array_equal(eng.state["avg"].cpu().numpy()[unv], orc.avg.numpy()[unv]), t
        assert np.array_equal(eng.state["sd"].cpu().numpy()[unv], orc.std.numpy()[unv]), t
        il, idp = orc.knn_indices()
        kl, kd = il.shape[1], idp.shape[1]
        assert eng.kl == kl and eng.kd == kd
        assert np.array_equal(eng.idx_last.cpu().numpy()[:, :kl], il.numpy()), t
        assert np.array_equal(eng.idx_dep.cpu().numpy()[:, :kd], idp.numpy()), t
        # CVRPModel.get_knn_state nudges near-zero capacities before it divides by them (CVRP/CVRPModel.py:164-166)
        small = cap < np.float32(1e-5)
        cap[small] = cap[small] + np.float32(1e-7)
        orc.capacity = torch.from_numpy(cap.copy())
        assert np.array_equal(eng.state["capacity"].cpu().numpy(), cap), t


@pytest.mark.parametrize("N", [20, 25, 26])
def test_nearest_neighbour_labels_on_small_instances(N, device):
    """The nearest-neighbour construction the trainers start from (TSP/TSPEnv.py:140-169, CVRP/CVRPEnv.py:340-409) on
    both sides of the 25-point boundary: labels equal to the oracle's loops over torch.cdist's matrix."""
    from oracle import dgl_oracle as O
    from dgl_amd import CVRPEnv, TSPEnv
    from test_gpu_selfimprove import CVRP_INFO, TSP_INFO
    B = 16
    torch.manual_seed(300 + N)
    env = TSPEnv(mode="train", pomo_size=1, append_information=TSP_INFO, device=device, data_path="")
    env.load_problems(0, B, N, 1)
    env.greedy_search()
    sol, length = O.tsp_nearest_neighbour(env.problems)
    assert torch.equal(env.solution.cpu(), sol)
    np.testing.assert_allclose(env.solution_len.cpu().numpy(), length.numpy(), rtol=3e-6)
    cenv = CVRPEnv(mode="train", pomo_size=1, append_information=CVRP_INFO, device=device)
    cenv.load_problems(0, B, N - 1)                            # N points with the depot
    cenv.greedy_search()
    nodes, flags = O.cvrp_nearest_neighbour(cenv.depot_node_xy, cenv.depot_node_demand)
    assert torch.equal(cenv.solution.cpu(), nodes) and torch.equal(cenv.solution_flag.cpu(), flags)
