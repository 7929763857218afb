"""ctypes binding of the C-ABI library (include/dgl_amd.h).

The library is built in-tree by ``dgl_amd.build`` (hipcc, gfx950).  Loading fails loudly
when it is missing: there is no CPU or PyTorch fallback for any of these entry points.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGL_AMD_LIB", os.path.join(_HERE, "lib", "libdgl_amd.so"))

TILE_STRIDE = 12
MAX_TOKENS = 32
MAX_WIDE_TOKENS = 128
STATIC_WORKSPACE_FLOATS = 16 * 1024 * 1024   # DGL_STATIC_WORKSPACE_FLOATS
PROBLEM_TSP = 0
PROBLEM_CVRP = 1
ARITH_SPLIT_F16 = 0    # f16 split-operand matrix cores, f32 accumulate (default)
ARITH_F32 = 1          # f32-input matrix instructions (bitwise FMA chains)


class TspState(Structure):
    _fields_ = [("B", c_int32), ("N", c_int32), ("rows_per_instance", c_int32), ("R", c_int32),
                ("xy", c_void_p), ("avg", c_void_p), ("sd", c_void_p), ("visited", c_void_p),
                ("tour", c_void_p)]


class CvrpState(Structure):
    _fields_ = [("B", c_int32), ("N", c_int32), ("rows_per_instance", c_int32), ("R", c_int32),
                ("xy", c_void_p), ("demand", c_void_p), ("avg", c_void_p), ("sd", c_void_p),
                ("capacity", c_void_p), ("visited", c_void_p), ("nodes", c_void_p), ("flags", c_void_p),
                ("demand_in_distance", c_int32)]


class BlockWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("wq", "wk", "wv", "wc", "bc", "w1", "b1", "w2", "b2")]


class PolicyWeights(Structure):
    _fields_ = [("problem", c_int32), ("emb_w", c_void_p), ("emb_b", c_void_p),
                ("block", BlockWeights * 4), ("anchor_w", c_void_p * 2), ("anchor_b", c_void_p * 2),
                ("final_w", c_void_p), ("final_b", c_void_p)]


class PolicyIO(Structure):
    _fields_ = [("problem", c_int32), ("R", c_int32), ("n", c_int32), ("k_last", c_int32), ("N", c_int32),
                ("idx_stride", c_int32), ("idx_dep_stride", c_int32), ("tile_row_stride", c_int32),
                ("arithmetic", c_int32), ("out_stride", c_int32),
                ("tile", c_void_p), ("idx_last", c_void_p), ("idx_depot", c_void_p), ("infeasible", c_void_p),
                ("logits_out", c_void_p), ("prob_out", c_void_p), ("selected_out", c_void_p),
                ("overflow_out", c_void_p)]


# name -> (restype, argtypes); every symbol include/dgl_amd.h declares
SIGNATURES = {
    "dgl_version": (c_int32, []),
    "dgl_last_error_string": (c_char_p, []),
    "dgl_tsp_reset": (c_int32, [POINTER(TspState), c_void_p]),
    "dgl_tsp_step": (c_int32, [POINTER(TspState), c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "dgl_tsp_knn": (c_int32, [POINTER(TspState), c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "dgl_tsp_state_data": (c_int32, [POINTER(TspState), c_int32, c_void_p, c_void_p]),
    "dgl_tsp_step_beam": (c_int32, [POINTER(TspState), POINTER(TspState), c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dgl_tsp_beam_backtrack": (c_int32, [POINTER(TspState), c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "dgl_tsp_tour_length": (c_int32, [POINTER(TspState), c_void_p, c_void_p]),
    "dgl_tsp_permute": (c_int32, [POINTER(TspState), POINTER(TspState), c_void_p, c_int32, c_int32, c_void_p]),
    "dgl_cvrp_reset": (c_int32, [POINTER(CvrpState), c_void_p]),
    "dgl_cvrp_step": (c_int32, [POINTER(CvrpState), c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "dgl_cvrp_knn": (c_int32, [POINTER(CvrpState), c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "dgl_cvrp_state_data": (c_int32, [POINTER(CvrpState), c_int32, c_void_p, c_void_p]),
    "dgl_cvrp_step_beam": (c_int32, [POINTER(CvrpState), POINTER(CvrpState), c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                     c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dgl_cvrp_beam_backtrack": (c_int32, [POINTER(CvrpState), c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "dgl_cvrp_route_length": (c_int32, [POINTER(CvrpState), c_void_p, c_void_p]),
    "dgl_cvrp_permute": (c_int32, [POINTER(CvrpState), POINTER(CvrpState), c_void_p, c_int32, c_int32, c_void_p]),
    "dgl_policy_packed_floats": (c_int64, [c_int32]),
    "dgl_policy_pack": (c_int32, [POINTER(PolicyWeights), c_void_p, c_void_p]),
    "dgl_policy_forward": (c_int32, [c_void_p, POINTER(PolicyIO), c_void_p]),
    "dgl_tsp_greedy_decode": (c_int32, [POINTER(TspState), c_void_p, POINTER(PolicyIO), c_int32, c_int32, c_int32, c_void_p]),
    "dgl_policy_general_workspace_floats": (c_int64, [c_int32, c_int32]),
    "dgl_policy_general": (c_int32, [POINTER(PolicyWeights), POINTER(PolicyWeights), POINTER(PolicyIO), c_void_p,
                                     c_float, c_void_p, c_void_p, c_void_p]),
    "dgl_policy_general_static": (c_int32, [POINTER(PolicyWeights), POINTER(PolicyWeights), POINTER(PolicyIO), c_void_p,
                                            c_float, c_void_p, c_void_p]),
    "dgl_augment": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dgl_tsp_nearest_neighbour": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "dgl_cvrp_nearest_neighbour": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "dgl_cvrp_solution_augment": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p]),
    "dgl_adam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float, c_float,
                                c_float, c_float, c_float, c_void_p]),
    "dgl_policy_scatter_probs": (c_int32, [POINTER(PolicyIO), c_void_p, c_void_p, c_int32, c_void_p]),
    "dgl_beam_advance": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p]),
    "dgl_beam_advance_compact": (c_int32, [POINTER(PolicyIO), c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                           c_int32, c_void_p, c_void_p, c_void_p]),
    "dgl_set_sqrt_model": (c_int32, [c_int32]),
    "dgl_sqrt_apply": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
}


class DglError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load libdgl_amd.so and attach signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime (libamdhip64.so.7).  It must be the one already
    # resident when this library is loaded, so that both resolve to a single runtime
    # instance and share devices, streams and allocations.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DglError(f"{LIB_PATH} is missing: build it with `python -m dgl_amd.build` "
                       "(there is no fallback path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("DGL_AMD_ALLOW_MISSING") == "1" and not hasattr(lib, name):
            continue                     # (A/B tools only: an older build of the library lacks the newest entry points)
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().dgl_last_error_string()
        raise DglError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
