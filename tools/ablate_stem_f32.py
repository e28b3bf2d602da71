"""Ablations of conv_stem_f32_kernel (mmh_set_option("stem_f32_dbg")): 1 no filter DMA after the first phase, 2 no halo DMA after
the first tile, 4 no fragment reads in the k loop, 8 no epilogue.  Results are wrong under every bit."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B, H = 32, 256
Cin = int(os.environ.get("CIN", 44))
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(7, 7, Cin, 64, device=dev) * 0.05; bias = torch.randn(64, device=dev)
flop = 2.0 * B * H * H * 64 * Cin * 49
def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
fn = lambda: ops.raw_conv_fprop(x, w, bias, 1, 3, True, 0)
vals = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 8, 7, 15]
res = {v: [] for v in vals}
for v in vals:
    lib.call("mmh_set_option", b"stem_f32_dbg", v); fn(); torch.cuda.synchronize()
for _ in range(3):
    for v in vals:
        lib.call("mmh_set_option", b"stem_f32_dbg", v); res[v].append(timeit(fn))
lib.call("mmh_set_option", b"stem_f32_dbg", 0)
for v in vals:
    m = statistics.median(res[v]); print(f"Cin {Cin} dbg {v:2d}: {m * 1e3:.0f} us = {flop / m / 1e9:.1f} TF", flush=True)
