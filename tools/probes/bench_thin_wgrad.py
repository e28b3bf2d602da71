"""mmh_conv7_thin_wgrad (the Generator head's weight gradient, 64 -> 4 channels 7x7) at the step's two shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, H in ((32, 256), (4, 512), (2, 64)):
    x = torch.randn(B, H, H, 64, device=dev); dy = torch.randn(B, H, H, 4, device=dev)
    print(f"B={B} {H}x{H}: {t(lambda: ops.raw_conv_wgrad(x, dy, 7, 1, 3, True)):.0f} us")
