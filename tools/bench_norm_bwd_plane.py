"""One-pass InstanceNorm backward (mmh_norm_bwd_fused) against reduce + apply at the 16-bit step's PATBlock shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0"); B = 32
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for H, Cc, masked, g32, dx32 in ((64, 256, True, False, False), (64, 512, True, False, False), (64, 256, False, True, False),
                                 (32, 256, True, False, False), (32, 512, True, False, False), (16, 512, True, False, False)):
    W = H
    rows = H * W
    x = torch.randn(B, H, W, Cc, device=dev).bfloat16(); g = torch.randn(B, H, W, Cc, device=dev)
    g = g if g32 else g.bfloat16()
    mean = torch.zeros(B, Cc, device=dev); invstd = torch.ones(B, Cc, device=dev)
    kb = None
    if masked:
        _, kb = ops.raw_scale_shift_act(x, invstd, mean, None, True, 0.5, 1, None, keep_bits=True, out_lp=True)
    mk = 2 if masked else 0
    ws = ops._ws(L.load().mmh_norm_bwd_ws_bytes(B, rows, Cc), x)
    s1 = torch.empty(B, Cc, device=dev); s2 = torch.empty(B, Cc, device=dev)
    dx = torch.empty((B, H, W, Cc), dtype=torch.float32 if dx32 else torch.bfloat16, device=dev)
    td = ops._tdt
    def red(): L.call("mmh_norm_bwd_reduce", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), B, rows, Cc, mk, 0.5, ops._ptr(s1), ops._ptr(s2), ops._ptr(ws), ws.numel() * 4, td(g), td(x), ops._stream())
    def app(): L.call("mmh_norm_bwd_apply", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), None, ops._ptr(s1), ops._ptr(s2), float(rows), B, rows, Cc, mk, 0.5, ops._ptr(dx), td(g), td(x), td(dx), ops._stream())
    def fus(): L.call("mmh_norm_bwd_fused", ops._ptr(g), ops._ptr(kb), ops._ptr(x), ops._ptr(mean), ops._ptr(invstd), None, float(rows), B, rows, Cc, mk, 0.5, ops._ptr(s1), ops._ptr(s2), ops._ptr(dx), td(g), td(x), td(dx), ops._stream())
    n = x.numel()
    byt = n * ((4 if g32 else 2) + 2 + (0.25 if masked else 0) + (4 if dx32 else 2))
    tr, ta, tf = t(red), t(app), t(fus)
    print(f"{H}x{W} C={Cc} masked={masked} g32={g32}: reduce {tr:.1f} us + apply {ta:.1f} us = {tr + ta:.1f}; one pass {tf:.1f} us "
          f"({byt / tf / 1e6:.2f} TB/s of its {byt / 1e6:.0f} MB)")
