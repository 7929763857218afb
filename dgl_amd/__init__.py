"""MI355X-native decode step of the DGL TSP/CVRP solver.

Host classes keep the reference's names and signatures (TSPModel, TSPEnv, CVRPModel, CVRPEnv,
Beamsearch, TSPTester, CVRPTester, TSPTrainer, CVRPTrainer); every computation on the decode path runs in hand-written
gfx950 HIP kernels reached through the C ABI of include/dgl_amd.h (dgl_amd/lib/libdgl_amd.so).
"""
__all__ = ["TSPModel", "TSPEnv", "CVRPModel", "CVRPEnv", "Beamsearch", "TSPTester", "CVRPTester", "TSPTrainer",
           "CVRPTrainer"]


def __getattr__(name):
    if name in ("TSPModel", "TSPEnv"):
        from . import tsp
        return getattr(tsp, name)
    if name in ("CVRPModel", "CVRPEnv"):
        from . import cvrp
        return getattr(cvrp, name)
    if name == "Beamsearch":
        from .beamsearch import Beamsearch
        return Beamsearch
    if name in ("TSPTester", "CVRPTester"):
        from . import tester
        return getattr(tester, name)
    if name in ("TSPTrainer", "CVRPTrainer"):
        from . import trainer
        return getattr(trainer, name)
    raise AttributeError(name)
