"""Probe: are the 16-bit norm backward passes bound by HBM, by the infinity cache, or by the kernel?

Times mmh_norm_bwd_reduce / mmh_norm_bwd_apply on a B=32, 64x64x256 bf16 feature map with the operands
hot in the 256 MB infinity cache and after a 1 GiB flush.  Round-3 result on MI355X:
reduce 35 us hot / 60 us cold, apply 40 / 68 us, so the in-step launches (35 / 40 us in the rocprof
summary) already run at the hot rate; sweeping the blocks-per-group of either kernel from 1/8x to 4x
of the shipped geometry moved neither number by more than 2 us.
"""
import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch, ctypes as C
from mmhand_amd import ops, lib as L
dev = torch.device("cuda:0")
B, H, W, Cc = 32, 64, 64, 256
x16 = torch.randn(B, H, W, Cc, device=dev).bfloat16(); g16 = torch.randn(B, H, W, Cc, device=dev).bfloat16()
kb = torch.randint(0, 255, (B, H, W, Cc // 4), device=dev, dtype=torch.uint8)
mean = torch.randn(B, Cc, device=dev) * 0.1; invstd = torch.rand(B, Cc, device=dev) + 0.5
rows = H * W
ws = torch.empty(64 << 20, device=dev)
s1 = torch.empty(B, Cc, device=dev); s2 = torch.empty(B, Cc, device=dev)
dx = torch.empty_like(x16)
big = torch.empty(1 << 28, device=dev)      # 1 GiB: flushes the 256 MB infinity cache
def reduce_():
    L.call("mmh_norm_bwd_reduce", ops._ptr(g16), ops._ptr(kb), ops._ptr(x16), ops._ptr(mean), ops._ptr(invstd), B, rows, Cc, 2, 0.5,
           ops._ptr(s1), ops._ptr(s2), ops._ptr(ws), ws.numel() * 4, L.BF16, L.BF16, ops._stream())
def apply_():
    L.call("mmh_norm_bwd_apply", ops._ptr(g16), ops._ptr(kb), ops._ptr(x16), ops._ptr(mean), ops._ptr(invstd), None, ops._ptr(s1), ops._ptr(s2),
           float(rows), B, rows, Cc, 2, 0.5, ops._ptr(dx), L.BF16, L.BF16, L.BF16, ops._stream())
def t(fn, flush):
    ts = []
    for _ in range(8):
        if flush: big.zero_()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
reduce_(); apply_(); torch.cuda.synchronize()
print("reduce hot %.1f us  cold %.1f us" % (t(reduce_, False), t(reduce_, True)))
print("apply  hot %.1f us  cold %.1f us" % (t(apply_, False), t(apply_, True)))
def both(): reduce_(); apply_()
print("reduce+apply hot %.1f us  cold(before reduce) %.1f us" % (t(both, False), t(both, True)))

