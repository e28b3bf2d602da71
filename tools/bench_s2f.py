"""The stride-2 16-bit fprop kernel (conv_s2_lp16.hip) against the general kernel at the step's shapes (B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, H, Cin, Cout in ((32, 256, 64, 128), (32, 128, 128, 256), (4, 512, 64, 128), (4, 256, 128, 256)):
    x16 = torch.randn(B, H, H, Cin, device=dev).bfloat16(); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    d = lambda: ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False)
    fl = 2.0 * B * (H // 2) ** 2 * Cin * Cout * 9
    res = []
    for on in (0, 1):
        L.check(L.load().mmh_set_option(b"lp16_s2f", 2 * on), "opt")
        us = t(lambda: ops.raw_conv_lp16g(d(), 0, x16, w, None, 0, True, out16=True))
        res.append(f"{'s2f' if on else 'general'} {us:.1f} us = {fl / us / 1e6:.0f} TF ({fl / us / 1e6 / 2500:.3f})")
    for dbg in (1, 2, 3):
        L.check(L.load().mmh_set_option(b"lp16_dbg", dbg), "opt")
        us = t(lambda: ops.raw_conv_lp16g(d(), 0, x16, w, None, 0, True, out16=True))
        res.append(f"dbg{dbg} {us:.1f} us")
    L.check(L.load().mmh_set_option(b"lp16_dbg", 0), "opt")
    L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "opt")
    print(f"B={B} {H}x{H} {Cin}->{Cout} stride 2 fprop, 16-bit out: " + "; ".join(res))

for B, H in ((32, 256), (64, 256), (4, 512)):
    Cin, Cout = 64, 128
    dy16 = torch.randn(B, H // 2, H // 2, Cout, device=dev).bfloat16(); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    d = lambda: ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False)
    fl = 2.0 * B * (H // 2) ** 2 * Cin * Cout * 9
    res = []
    for on in (0, 1):
        L.check(L.load().mmh_set_option(b"lp16_s2f", on), "opt")
        us = t(lambda: ops.raw_conv_lp16g(d(), 1, dy16, w, None, 0, True, out16=True))
        res.append(f"{'s2d' if on else 'general'} {us:.1f} us = {fl / us / 1e6:.0f} TF ({fl / us / 1e6 / 2500:.3f})")
    L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "opt")
    print(f"B={B} {H}x{H} {Cin}->{Cout} stride 2 DGRAD, 16-bit out: " + "; ".join(res))
