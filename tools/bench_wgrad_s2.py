"""fp32 stride-2 wgrads 64 -> 128 @256x256 and 128 -> 256 @128x128 (B=32): strip-streaming kernel (wgrad_s2.hip) against the
generic implicit GEMM (both include their slab reduction)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 32))
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for H, Cin, Cout in ((256, 64, 128), (128, 128, 256)):
    x = torch.randn(B, H, H, Cin, device=dev)
    dy = torch.randn(B, H // 2, H // 2, Cout, device=dev)
    flop = 2.0 * B * (H // 2) ** 2 * Cout * Cin * 9
    fn = lambda: ops.raw_conv_wgrad(x, dy, 3, 2, 1, False)
    res, outs = {0: [], 1: []}, {}
    for v in (0, 1):
        lib.call("mmh_set_option", b"wgrad_s2_strip", v); outs[v] = fn().clone(); torch.cuda.synchronize()
    rel = float((outs[1].double() - outs[0].double()).abs().sum() / outs[0].double().abs().sum())
    for _ in range(5):
        for v in (0, 1):
            lib.call("mmh_set_option", b"wgrad_s2_strip", v); res[v].append(timeit(fn))
    m = {v: statistics.median(res[v]) for v in res}
    print(f"s2 wgrad {Cin}->{Cout} @{H}: implicit GEMM {m[0] * 1e3:.0f} us = {flop / m[0] / 1e9:.1f} TF ({flop / m[0] / 1e9 / 157.3:.2f}) | "
          f"strip stream {m[1] * 1e3:.0f} us = {flop / m[1] / 1e9:.1f} TF ({flop / m[1] / 1e9 / 157.3:.2f}) | rel diff {rel:.1e}", flush=True)
lib.call("mmh_set_option", b"wgrad_s2_strip", 1)
