"""The general 16-bit 3x3 kernel (conv_lp16g) on the stride-2 / transposed convs of the step (B=32): fprop and dgrad."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = 32
for (H, Cin, Cout, stride) in ((256, 64, 128, 2), (128, 128, 256, 2), (256, 64, 64, 1)):
    Ho = H // stride
    x = ops.lp16_twin(torch.randn(B, H, H, Cin, device=dev), True)
    dy = ops.lp16_twin(torch.randn(B, Ho, Ho, Cout, device=dev), True)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    d = ops.conv_desc(B, H, H, Cin, Cout, 3, stride, 1, False)
    flop = 2.0 * B * Ho * Ho * Cin * Cout * 9
    variants = {"fprop": lambda: ops.raw_conv_lp16g(d, 0, x, w, bias, 1, True, out16=True),
                "dgrad": lambda: ops.raw_conv_lp16g(d, 1, dy, w, None, 0, True, out16=True)}
    res = {k: [] for k in variants}
    for f in variants.values(): f()
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    print(f"{Cin}->{Cout} s{stride} @{H}: " + " | ".join(f"{k}: {statistics.median(v)*1e3:.0f} us ({flop/statistics.median(v)/1e9:.0f} TF)" for k, v in res.items()), flush=True)
