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
