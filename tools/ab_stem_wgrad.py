"""fp32 wgrad of the 7x7 stems: LDS-band kernel (conv_stem.hip) against the generic direct wgrad, B=32 @256x256.

    python tools/ab_stem_wgrad.py
"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import ops      # noqa: E402
dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for Cin in (44, 24, 8, 4):
    x = torch.randn(32, 256, 256, Cin, device=dev); dy = torch.randn(32, 256, 256, 64, device=dev)
    fl = 2.0 * dy.numel() * Cin * 49
    out = []
    for flag in (False, True, False, True):
        ops.USE_STEM_WGRAD = flag
        fw = lambda: ops.raw_conv_wgrad(x, dy, 7, 1, 3, True)
        fw(); torch.cuda.synchronize()
        m = statistics.median([timeit(fw) for _ in range(3)])
        out.append(f"{'band' if flag else 'generic'} {m*1e3:7.1f} us {fl/m/1e9:6.1f} TF")
    print(f"{Cin}->64 k7: " + " | ".join(out), flush=True)
