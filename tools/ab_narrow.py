"""A/B of the fp32 direct kernels on the narrow (N <= 64 ... 128) shapes of the step: 256-row tiles
(conv_tall) and XCD-banded row tiles for single-column grids (conv_xcd1), against the 128-row kernel.

    python tools/ab_narrow.py
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib, ops      # noqa: E402

L = lib.load()
dev = torch.device("cuda:0")
B = 32
# H, Cin, Cout, k, stride, pad, reflect
SHAPES = [(256, 44, 64, 7, 1, 3, True), (256, 24, 64, 7, 1, 3, True), (256, 8, 64, 7, 1, 3, True),
          (256, 4, 64, 7, 1, 3, True), (256, 64, 64, 3, 1, 1, False), (256, 4, 64, 3, 1, 1, False),
          (256, 64, 128, 3, 2, 1, False), (128, 128, 256, 3, 2, 1, False)]


def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def setopt(**kw):
    for k, v in kw.items():
        lib.check(L.mmh_set_option(k.encode(), v), "opt")


CFG = [("base", dict(conv_tall=0, conv_xcd1=0)), ("xcd1", dict(conv_tall=0, conv_xcd1=1)),
       ("tall", dict(conv_tall=1, conv_xcd1=0)), ("tall+xcd1", dict(conv_tall=1, conv_xcd1=1))]
for (H, Cin, Cout, k, s, p, refl) in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    setopt(conv_tall=0, conv_xcd1=0)
    y0 = ops.raw_conv_fprop(x, w, None, s, p, refl, 0)
    dy = torch.randn_like(y0)
    dx0 = ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl) if Cin >= 32 else None
    fl = 2.0 * y0.numel() * Cin * k * k
    res = {}
    for rep in range(3):
        for name, kw in CFG:
            setopt(**kw)
            fn = lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0)
            y = fn()
            assert torch.equal(y, y0), (name, "fprop differs")
            torch.cuda.synchronize()
            res.setdefault((name, "fprop"), []).append(timeit(fn))
            if dx0 is not None:
                fd = lambda: ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl)
                dx = fd()
                assert torch.equal(dx, dx0), (name, "dgrad differs")
                torch.cuda.synchronize()
                res.setdefault((name, "dgrad"), []).append(timeit(fd))
    line = f"{Cin:3d}->{Cout:3d} k{k} s{s} @{H}: "
    for what in ("fprop", "dgrad"):
        if (CFG[0][0], what) not in res:
            continue
        line += f"\n    {what}: " + "  ".join(
            f"{name} {statistics.median(res[(name, what)]) * 1e3:7.1f} us ({fl / statistics.median(res[(name, what)]) / 1e9:5.1f} TF)"
            for name, _ in CFG)
    print(line, flush=True)
    del x, w, y0, dy
