"""Ablations of dgrad_s2_kernel (mmh_set_option("dgrad_s2_dbg")): 1 no stores, 2 no halo DMA after the first tile, 4 no
filter DMA after the first tap, 8 scheduling barrier behind the fragment prefetch.  Results are wrong under 1/2/4."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B, H = 32, 256
w = torch.randn(3, 3, 64, 128, device=dev) * 0.05
dy = torch.randn(B, H // 2, H // 2, 128, device=dev)
flop = 2.0 * B * (H // 2) ** 2 * 128 * 64 * 9
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
fn = lambda: ops.raw_conv_dgrad(dy, w, (B, H, H, 64), 2, 1, False)
vals = [int(v) for v in sys.argv[1:]] or [0, 8, 1, 2, 4, 7, 15]
res = {v: [] for v in vals}
for v in vals:
    lib.call("mmh_set_option", b"dgrad_s2_dbg", v); fn(); torch.cuda.synchronize()
for _ in range(5):
    for v in vals:
        lib.call("mmh_set_option", b"dgrad_s2_dbg", v); res[v].append(timeit(fn))
lib.call("mmh_set_option", b"dgrad_s2_dbg", 0)
for v in vals:
    m = statistics.median(res[v]); print(f"dbg {v:2d}: {m * 1e3:.0f} us = {flop / m / 1e9:.1f} TF", flush=True)
