"""The 7x7 stems' 16-bit weight gradient at the training shapes (B=32, 256x256, 64 output channels): the first-generation
implicit-GEMM wgrad against wgrad_stem.hip (column taps flattened into a transposed-read operand); interleaved rounds."""
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
dy16 = ops.lp16_twin(torch.randn(B, H, H, 64, device=dev), True)
for Cin in (4, 8, 24, 44):
    x16p = ops.lp16_pad8(torch.randn(B, H, H, Cin, device=dev), True)
    d = ops.conv_desc(B, H, H, Cin, 64, 7, 1, 3, True)
    flop = 2.0 * B * H * H * 49 * Cin * 64
    variants = {"first generation": lambda: ops.raw_conv_wgrad_lp16_gen1(x16p, dy16, Cin, 7, 1, 3, True, True),
                "wgrad_stem": lambda: ops.raw_wgrad_stem_lp16(d, x16p, dy16, True)}
    a, b_ = variants["first generation"](), variants["wgrad_stem"]()
    err = float((a - b_).abs().max() / a.abs().max())
    res = {k: [] for k in variants}
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    print(f"{Cin:2d}->64 (C8 {x16p.shape[3]}): " + " | ".join(f"{k}: {statistics.median(v)*1e3:.0f} us ({flop/statistics.median(v)/1e9:.0f} TF)" for k, v in res.items()) + f" | max diff {err:.1e}", flush=True)
