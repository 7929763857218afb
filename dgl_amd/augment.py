"""mix_aug of the reference (utils/utils.py:790-918): the host draws the angles exactly as
``run_aug('mix', x)`` does (Rotate_aug on the even instances first, then Reflect_aug on the odd
ones, each one ``torch.rand((b, 1)) * 2 * np.pi`` from the global CPU generator), the device
kernel (dgl_augment) applies them."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def draw_mix_angles(B: int):
    """(cos_sin [B][2], kind [B]) for run_aug('mix', x) with aug_all=False, aug_num=None: instance 0 stays as it is
    (utils/utils.py:913-917), even instances rotate, odd instances reflect."""
    n_even, n_odd = (B + 1) // 2, B // 2
    th_rot = torch.rand((n_even, 1)) * 2 * np.pi
    th_ref = torch.rand((n_odd, 1)) * 2 * np.pi
    cs = torch.zeros(B, 2)
    cs[0::2, 0], cs[0::2, 1] = torch.cos(th_rot)[:, 0], torch.sin(th_rot)[:, 0]
    cs[1::2, 0], cs[1::2, 1] = torch.cos(2 * th_ref)[:, 0], torch.sin(2 * th_ref)[:, 0]
    kind = torch.ones(B, dtype=torch.int32)
    kind[1::2] = 2
    kind[0] = 0
    return cs, kind


def mix_augment(xy: torch.Tensor) -> torch.Tensor:
    """run_aug('mix', xy) for xy [B][M][2] on the device."""
    B, M = xy.shape[0], xy.shape[1]
    cs, kind = draw_mix_angles(B)
    cs, kind = cs.to(xy.device), kind.to(xy.device)
    out = torch.empty_like(xy)
    check(_lib.load().dgl_augment(ptr(xy.contiguous()), B, M, ptr(cs), ptr(kind), ptr(out), stream_ptr()), "dgl_augment")
    return out
