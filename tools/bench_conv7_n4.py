"""The 7x7 convolutions with four output channels at the training shapes (B=32, 256x256): the Generator head's fprop and a
Discriminator stem's gradient towards the generated image - conv_thin.hip (vector ALU) against conv7_n4.hip (16-column
MFMA over an LDS halo); interleaved rounds in one process."""
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
B, H = 32, 256
x = torch.randn(B, H, H, 64, device=dev); xb = ops.lp16_twin(x, True)
w = torch.randn(7, 7, 64, 4, device=dev) * 0.05; bias = torch.randn(4, device=dev)
def head(n4, twin):
    def f():
        ops.USE_CONV7_N4 = n4
        r = ops.raw_conv_fprop(x, w, bias, 1, 3, True, 2, bf16=True)
        ops.USE_CONV7_N4 = True
        return r
    return f
d = ops.conv_desc(B, H, H, 64, 4, 7, 1, 3, True)
y = torch.empty(B, H, H, 4, device=dev)
variants = {"head fprop, vector ALU (fp32 x)": head(False, False), "head fprop, cvt + conv7_n4": head(True, True),
            "head fprop, conv7_n4 alone (16-bit x given)": lambda: ops.raw_conv7_n4(d, 0, xb, w, bias, y, 2, True)}
for Cin in (8, 24):
    ws = torch.randn(7, 7, Cin, 64, device=dev) * 0.05
    dy = ops.lp16_twin(torch.randn(B, H, H, 64, device=dev), True)
    def dg(n4, ws=ws, dy=dy, Cin=Cin):
        def f():
            ops.USE_CONV7_N4 = n4
            r = ops.raw_conv_dgrad_thin(dy, ws, (B, H, H, Cin), True)
            ops.USE_CONV7_N4 = True
            return r
        return f
    variants[f"stem dgrad {Cin}->64 towards 4 channels, vector ALU"] = dg(False)
    variants[f"stem dgrad {Cin}->64 towards 4 channels, conv7_n4 + fold"] = dg(True)
res = {k: [] for k in variants}
for f in variants.values(): f()
torch.cuda.synchronize()
for r in range(5):
    for k, f in variants.items(): res[k].append(timeit(f))
for k, v in res.items():
    print(f"{k:62s} {statistics.median(v) * 1e3:8.0f} us")
