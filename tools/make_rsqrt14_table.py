"""Tabulate the AVX-512 instruction VRSQRT14SS over all 2^24 (exponent parity, mantissa) inputs of [1, 4) and write the
table, run-length encoded, to dgl_amd/data/rsqrt14_avx512.npz.  Needs gcc and a CPU with AVX-512F (the instruction's
result is architecturally defined: every AVX-512 CPU returns the same bits).

The decode kernels' square-root model "mkl_avx512" reproduces, bit for bit, what torch.sqrt and torch.cdist return on
CPU when torch is built against MKL and runs on an AVX-512 machine (the container the golden vectors were made in); the
recording is what its constants were solved from (dgl_amd/sqrt_model.py::rsqrt14_segments: the instruction is piecewise
linear, 64 segments with one integer pair each) and what the tests check them against.  MKL's vector sqrt refines the
14-bit estimate y = rsqrt14(x) once,
    s = x y;  result = fma(fma(-s, s, x), y / 2, s),
which is one ulp below the correctly rounded root on 0.6 % of all inputs.  The result of rsqrt14 has 16 significant
bits and scales exactly with the exponent, so a 16-bit table over two binades describes it completely.  The script
also checks the whole model against torch.sqrt on a few million inputs when torch is importable.

usage: python tools/make_rsqrt14_table.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "dgl_amd", "data", "rsqrt14_avx512.npz")

C_SRC = r'''
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "wb");
    for (uint32_t i = 0; i < (1u << 24); ++i) {
        uint32_t bits = 0x3F800000u + i, yb;
        float x, y;
        memcpy(&x, &bits, 4);
        __m128 v = _mm_set_ss(x);
        y = _mm_cvtss_f32(_mm_rsqrt14_ss(v, v));
        memcpy(&yb, &y, 4);
        uint32_t t = yb - 0x3F000000u;                 /* y in (0.5, 1]: 16 significant bits, the low 7 are zero */
        if (t & 0x7F) return 2;
        t >>= 7;
        if (t > 65535 && i != 0) return 3;             /* only rsqrt14(1.0) = 1.0 needs a 17th bit: special-cased by users */
        uint16_t u = t > 65535 ? 65535 : (uint16_t)t;
        fwrite(&u, 2, 1, f);
    }
    fclose(f);
    return 0;
}
'''


def expand(pos: np.ndarray, val: np.ndarray) -> np.ndarray:
    """The full table (2^24 uint16) from its run-length encoding."""
    return np.repeat(val, np.diff(np.append(pos, 1 << 24)).astype(np.int64))


def main():
    with tempfile.TemporaryDirectory() as d:
        src, exe, raw = os.path.join(d, "t.c"), os.path.join(d, "t"), os.path.join(d, "t.bin")
        open(src, "w").write(C_SRC)
        subprocess.run(["gcc", "-O2", "-mavx512f", src, "-o", exe], check=True)
        subprocess.run([exe, raw], check=True)
        table = np.fromfile(raw, dtype=np.uint16)
    assert table.size == 1 << 24
    change = np.flatnonzero(np.diff(table.astype(np.int32)) != 0) + 1
    pos = np.concatenate(([0], change)).astype(np.uint32)
    val = table[pos]
    assert np.array_equal(expand(pos, val), table)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, pos=pos, val=val)
    print(f"{len(pos)} runs -> {os.path.getsize(OUT) / 1e3:.0f} KB")
    try:
        import torch
    except ImportError:
        return
    torch.manual_seed(7)
    e = torch.randint(-60, 10, (3_000_000,)).float()
    x = torch.cat([(torch.rand(3_000_000) + 1) * torch.exp2(e), torch.rand(2_000_000) * 2, torch.zeros(4)]).float()
    got = model_sqrt(x.numpy(), table)
    ref = torch.sqrt(x).numpy()
    ieee = np.sqrt(x.numpy())
    print(f"model vs torch.sqrt: {int((got != ref).sum())} of {x.numel()} differ; torch.sqrt vs the correctly rounded root: "
          f"{int((ref != ieee).sum())} differ")


def model_sqrt(x: np.ndarray, table: np.ndarray) -> np.ndarray:
    """MKL's AVX-512 float sqrt restated with numpy (binary32 FMAs evaluated in binary64 and rounded once: the products
    of two binary32 numbers are exact there, and a sum of such a product and a binary32 number rounds correctly from
    binary64 to binary32 except in double-rounding cases that the comparison above would show)."""
    x = x.astype(np.float32)
    b = x.view(np.uint32).astype(np.int64)
    e = (b >> 23) - 127
    p = e & 1
    q = (e - p) // 2
    i = (p << 23) | (b & 0x7FFFFF)
    yb = np.where(i == 0, 0x3F800000, 0x3F000000 + (table[i].astype(np.int64) << 7)) - (q << 23)
    y = yb.astype(np.uint32).view(np.float32)
    s = (x * y).astype(np.float32)
    h = (np.float32(0.5) * y).astype(np.float32)
    d = (x.astype(np.float64) - s.astype(np.float64) * s.astype(np.float64)).astype(np.float32)
    r = (d.astype(np.float64) * h.astype(np.float64) + s.astype(np.float64)).astype(np.float32)
    return np.where(x == 0, np.float32(0), r)


if __name__ == "__main__":
    main()
