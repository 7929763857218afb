"""The C-ABI library loads and exports exactly what include/dgl_amd.h declares (no GPU needed)."""
import os
import re
import subprocess

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "dgl_amd.h")).read()
    return sorted(set(re.findall(r"^DGL_API[^;(]*?\b(dgl_[a-z_0-9]+)\s*\(", text, flags=re.M)))


def test_header_declares_entry_points():
    names = _declared()
    assert "dgl_tsp_step" in names and "dgl_policy_forward" in names and len(names) >= 15


def test_library_exports_every_declared_symbol():
    from dgl_amd import _lib, build
    build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (dgl_[a-z_0-9]+)", out))
    assert set(_declared()) <= exported, set(_declared()) - exported
    # nothing but the declared C ABI is visible
    assert exported == set(_declared()), exported ^ set(_declared())


def test_ctypes_binding_covers_header():
    from dgl_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()                     # attaches every signature; AttributeError if one is missing
    assert lib.dgl_version() == 100
    assert lib.dgl_policy_packed_floats(0) == lib.dgl_policy_packed_floats(1) > 800000


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "dgl_amd.h")).read()
    for name in _declared():
        if name in ("dgl_version", "dgl_last_error_string", "dgl_policy_packed_floats", "dgl_policy_pack",
                    "dgl_policy_general_static"):      # no counterpart in the reference (says so in its comment)
            continue
        pos = text.index(name + "(")
        comment = text[text.rfind("/*", 0, pos):pos]
        assert re.search(r"\.py:\d+", comment), f"{name} has no reference file:line citation"


def test_fused_decode_refuses_instances_of_at_most_25_points():
    """torch.cdist evaluates at most 25 points pair by pair; the step, reset and nearest-neighbour kernels follow it
    there (tests/test_gpu_small.py), the fused decode kernel implements the matrix-product distances only and
    refuses such instances before anything is launched."""
    import ctypes
    from dgl_amd import _lib
    lib = _lib.load()
    fake = 4096                                   # never dereferenced: the size check comes first
    st = _lib.TspState(2, 25, 1, 2, fake, fake, fake, fake, fake)
    io = _lib.PolicyIO()
    assert lib.dgl_tsp_greedy_decode(ctypes.byref(st), fake, ctypes.byref(io), 1, 25, 10, None) == -2   # DGL_ERR_UNSUPPORTED
    assert b"25 points" in lib.dgl_last_error_string()


def test_general_path_refuses_bad_token_counts_and_misaligned_workspace():
    """dgl_policy_general checks its arguments before anything is launched: 2 <= n <= 256 tokens per row, the TSP token
    layout, and a 16-byte aligned workspace (its kernels read it 16 bytes at a time)."""
    import ctypes
    from dgl_amd import _lib
    lib = _lib.load()
    fake = 4096                                   # never dereferenced: the checks come first
    w = _lib.PolicyWeights()
    w.problem = _lib.PROBLEM_TSP
    io = _lib.PolicyIO()
    io.problem, io.R, io.n, io.k_last, io.N = _lib.PROBLEM_TSP, 4, 51, 50, 100
    io.idx_stride, io.tile_row_stride, io.out_stride = 50, 51 * _lib.TILE_STRIDE, 64
    io.tile, io.idx_last = fake, fake
    args = (ctypes.byref(w), None, ctypes.byref(io), None, ctypes.c_float(0.0))
    assert lib.dgl_policy_general(*args, fake + 4, None, None) == -1              # DGL_ERR_INVALID_ARGUMENT
    assert b"aligned" in lib.dgl_last_error_string()
    io.n, io.k_last = 257, 256
    assert lib.dgl_policy_general(*args, fake, None, None) == -1
    assert b"n=257" in lib.dgl_last_error_string()
    io.n, io.k_last = 51, 40                                                      # TSP: the anchor is the last token
    assert lib.dgl_policy_general(*args, fake, None, None) == -1
    assert lib.dgl_policy_general_workspace_floats(64, 202) > lib.dgl_policy_general_workspace_floats(10, 202) > 0
