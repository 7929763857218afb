"""Which square root the state kernels reproduce (include/dgl_amd.h: dgl_set_sqrt_model).

    "mkl_avx512"  what torch.sqrt / torch.cdist return on a CPU build of torch linked against MKL on an AVX-512 machine
                  - the setting the golden vectors of this repository were captured under - together with the reset's
                  means in torch.mean's own summation order: the device then carries the reference's state bit for bit.
                  The default, of the engines and of the library.
    "ieee"        correctly rounded roots: the reference's algorithm independent of any host's math library; about 1 %
                  faster rollouts, and what the C and PyTorch oracles of the tests restate.  DGL_SQRT_MODEL=ieee selects it.

The setting is per device and process-wide; states reset after a change follow the new model.

The kernels evaluate the AVX-512 instruction VRSQRT14 (the estimate MKL's square root starts from) in closed form from 64
pairs of integers (dgl_amd/csrc/common.h: DGL_RSQRT14_SEG).  The recording of the instruction those pairs were solved from
ships as data/rsqrt14_avx512.npz (tools/make_rsqrt14_table.py wrote it on an AVX-512 machine); rsqrt14_table() /
rsqrt14_segments() below expand it and solve it again, which is how the tests re-check the constants."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib

IEEE, MKL_AVX512 = 0, 1
DEFAULT = "mkl_avx512"
_TABLE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "rsqrt14_avx512.npz")
_current = {}         # device index -> model name


def rsqrt14_table_full() -> np.ndarray:
    """VRSQRT14 over [1, 4) as 2^24 uint16 (index = exponent parity << 23 | mantissa; value = (bits - 0x3F000000) >> 7,
    the single entry 65536 - rsqrt14(1) = 1 - stored as 65535)."""
    z = np.load(_TABLE_FILE)
    pos, val = z["pos"], z["val"]
    table = np.repeat(val, np.diff(np.append(pos, 1 << 24)).astype(np.int64))
    assert table.size == 1 << 24 and table.dtype == np.uint16
    return table


def rsqrt14_table() -> np.ndarray:
    """The recording reduced to 2^16 uint16, index = exponent parity << 15 | mantissa >> 8: the instruction's estimate is
    constant over every block of 256 consecutive inputs except for the exact powers of 4 (input index 0, estimate exactly
    1; MKL's refinement returns the exact root for them from either estimate), which the assertion re-checks."""
    full = rsqrt14_table_full()
    blocks = full.reshape(1 << 16, 256)
    compact = blocks[:, 1].copy()
    same = blocks[:, 1:] == compact[:, None]
    assert same.all() and (blocks[1:, 0] == compact[1:]).all(), "VRSQRT14 varies inside a block of 256 inputs"
    return compact


def rsqrt14_segments(table: np.ndarray = None) -> np.ndarray:
    """The 64 integer pairs (C, S) with table[seg << 10 | m] == (C - S m) >> 9 for every m < 1024, seg = exponent parity
    << 5 | upper 5 mantissa bits, solved from the recording: for every segment exactly one pair fits all 1024 values
    (asserted).  Returned as int64 [64, 2] in the kernels' order, index = (float bits >> 18) & 63 (the lowest bit of the
    biased exponent is the inverse of the parity), as {C, -S}."""
    t = (rsqrt14_table() if table is None else table).astype(np.int64).reshape(64, 1024)
    m = np.arange(1024, dtype=np.int64)
    out = np.zeros((64, 2), np.int64)
    for seg in range(64):
        fits = []
        for s in range(1, 2048):
            lo, hi = (t[seg] * 512 + s * m).max(), ((t[seg] + 1) * 512 + s * m).min()
            if lo < hi:
                fits.append((int(lo), int(hi), s))
        assert len(fits) == 1 and fits[0][1] - fits[0][0] == 1, f"segment {seg}: {len(fits)} linear fits"
        out[seg ^ 32] = (fits[0][0], -fits[0][2])
    return out


def use(model: str, device=None) -> None:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    lib = _lib.load()
    with torch.cuda.device(index):
        if model == "ieee":
            _lib.check(lib.dgl_set_sqrt_model(IEEE), "dgl_set_sqrt_model")
        elif model == "mkl_avx512":
            _lib.check(lib.dgl_set_sqrt_model(MKL_AVX512), "dgl_set_sqrt_model")
        else:
            raise ValueError(f"square-root model {model!r}: expected 'ieee' or 'mkl_avx512'")
    _current[index] = model


def default_model() -> str:
    """The model the engines select: DGL_SQRT_MODEL from the environment, else the reference-bits model."""
    return os.environ.get("DGL_SQRT_MODEL", "") or DEFAULT


def apply_environment(device) -> None:
    """Called when an engine is created: selects default_model() on the engine's device unless a model has been chosen
    there already (once per device; an explicit ``use`` before or afterwards wins)."""
    index = torch.device(device).index
    index = torch.cuda.current_device() if index is None else index
    if index not in _current:
        use(default_model(), torch.device("cuda", index))


def current(device=None) -> str:
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    return _current.get(index, DEFAULT)


def apply(x: torch.Tensor) -> torch.Tensor:
    """The current model's square root of a float32 device tensor (for checks against recorded torch.sqrt values)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().dgl_sqrt_apply(x.data_ptr(), out.data_ptr(), x.numel(), _lib.stream_ptr()), "dgl_sqrt_apply")
    return out
