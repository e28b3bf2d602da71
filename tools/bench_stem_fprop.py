"""The 7x7 stems' 16-bit fprop at the training shapes (B=32, 256x256, 64 output channels, 16-bit output + ReLU as in the
generation path): the flat-K im2col kernel (conv_lp16f) against conv_stem16.hip (LDS-resident halo, column taps flattened)."""
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
for Cin in (3, 8, 24, 44):
    x16p = ops.lp16_pad8(torch.randn(B, H, H, Cin, device=dev), True)
    w = torch.randn(7, 7, Cin, 64, device=dev) * 0.05; bias = torch.randn(64, device=dev)
    flop = 2.0 * B * H * H * 49 * Cin * 64
    def run(new):
        def f():
            ops.USE_STEM_FPROP16 = new
            r = ops.raw_conv_lp16_flat(ops.conv_desc(B, H, H, Cin, 64, 7, 1, 3, True), None, w, bias, 1, True, out16=True, x16p=x16p)
            ops.USE_STEM_FPROP16 = True
            return r
        return f
    variants = {"flat-K (conv_lp16f)": run(False), "conv_stem16": run(True)}
    a, b_ = variants["flat-K (conv_lp16f)"]().float(), variants["conv_stem16"]().float()
    err = float((a - b_).abs().max() / a.abs().max())
    res = {k: [] for k in variants}
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    print(f"{Cin:2d}->64 (C8 {x16p.shape[3]}): " + " | ".join(f"{k}: {statistics.median(v)*1e3:.0f} us ({flop/statistics.median(v)/1e9:.0f} TF)" for k, v in res.items()) + f" | max diff {err:.1e}", flush=True)
