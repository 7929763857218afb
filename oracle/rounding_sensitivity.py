"""How reproducible are the reference's own greedy tours?  (test infrastructure: CPU only, never part of the product path)

The oracle - bit-equal to the reference on every golden case - decodes a population twice: as it is, and with the reset's
row mean / standard deviation (TSP/TSPEnv.py:228-235) moved by one ulp at random, which is what another summation order
of torch.mean (another SIMD width, another thread count) does.  On uniform instances no tour changes; on clustered
instances (oracle/make_goldens.py: seven Gaussian clusters, sigma 0.04) about a quarter of them do: min-max normalising
the features of 31 neighbours whose values agree to three digits amplifies the last bit a thousandfold.  "Tours
identical to the reference's" is therefore asserted on the uniform populations only; on clustered ones the mean length
is (tests/test_gpu_tsp.py, tests/test_gpu_cvrp.py).

    python oracle/rounding_sensitivity.py [instances per population]
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden                      # noqa: E402
from oracle.dgl_oracle import TSPOracle, tsp_rollout  # noqa: E402


class OneUlpOff(TSPOracle):
    """The oracle with avg / std of the reset moved by -1, 0 or +1 ulp (seeded)."""

    def reset(self, xy, rows_per_instance=1, chain=False):
        super().reset(xy, rows_per_instance, chain)
        g = torch.Generator().manual_seed(1)
        for name in ("avg", "std"):
            v = getattr(self, name)
            step = (torch.randint(0, 3, v.shape, generator=g) - 1).to(torch.int32)
            setattr(self, name, (v.view(torch.int32) + step).view(torch.float32))


def tours_kept(name: str, m: int):
    """(#instances of the first m whose tour survives the perturbation, relative change of the mean length)"""
    case, weights = load_golden(name)
    xy, start = torch.from_numpy(case["xy"][:m]), torch.from_numpy(case["start"]).long()
    with torch.no_grad():
        a = tsp_rollout(TSPOracle(weights, int(case["meta/knn"])), xy, start)
        b = tsp_rollout(OneUlpOff(weights, int(case["meta/knn"])), xy, start)
    kept = int((a["tour"] == b["tour"]).all(dim=1).sum())
    la, lb = a["length"].mean().item(), b["length"].mean().item()
    return kept, abs(la - lb) / la


if __name__ == "__main__":
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for name in ("tsp_n1000_b128_nn_tours", "tsp_n1000_b128_clustered_nn_tours"):
        kept, rel = tours_kept(name, m)
        print(f"{name}: {kept}/{m} tours unchanged by a one-ulp perturbation of the reset statistics, mean length moves by {rel:.1e}")
