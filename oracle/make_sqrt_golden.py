"""torch.sqrt of this container's torch build (CPU, MKL, AVX-512) on 100 000 inputs, as a fixture for the device's
"mkl_avx512" square-root model (dgl_amd/sqrt_model.py): wide exponent range, values like squared distances and
variances, exact squares, zeros.  Runs in the build container only.

usage: python oracle/make_sqrt_golden.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "torch_sqrt_mkl.npz")


def main():
    g = torch.Generator().manual_seed(2024)
    e = torch.randint(-80, 20, (40_000,), generator=g).float()
    wide = (torch.rand(40_000, generator=g) + 1) * torch.exp2(e)
    unit = torch.rand(40_000, generator=g) * 2                       # squared distances in the unit square
    small = torch.rand(15_000, generator=g) * 1e-3                   # variances late in a tour
    squares = (torch.rand(4_990, generator=g) * 3).square()
    x = torch.cat([wide, unit, small, squares, torch.zeros(10)]).float().contiguous()
    r = torch.sqrt(x)
    ieee = np.sqrt(x.numpy())
    np.savez_compressed(OUT, x=x.numpy(), sqrt=r.numpy(), **{"meta/torch": np.array(torch.__version__),
                                                             "meta/cpu": np.array(torch.backends.cpu.get_cpu_capability())})
    print(f"{x.numel()} inputs, {int((r.numpy() != ieee).sum())} roots differ from the correctly rounded ones -> "
          f"{os.path.getsize(OUT) / 1e3:.0f} KB")


if __name__ == "__main__":
    main()
