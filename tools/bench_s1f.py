"""The 64 -> 64 stride-1 3x3 conv of VGG19 (conv1_2; zero padding) in 16 bits: the general kernel against the register-resident-
weights kernel of conv_s2_lp16.hip in its stride-1 form, fprop and input gradient; max difference between the two."""
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
for B, H in ((32, 256), (4, 512), (2, 64)):
    x16 = torch.randn(B, H, H, 64, device=dev).bfloat16(); w = torch.randn(3, 3, 64, 64, device=dev) * 0.05
    bias = torch.randn(64, device=dev)
    d = lambda: ops.conv_desc(B, H, H, 64, 64, 3, 1, 1, False)
    fl = 2.0 * B * H * H * 64 * 64 * 9
    for mode, name in ((0, "fprop (+bias, ReLU)"), (1, "dgrad")):
        res, ys = [], []
        for on in (0, 1):
            L.check(L.load().mmh_set_option(b"lp16_s2f", on), "opt")
            f = lambda: ops.raw_conv_lp16g(d(), mode, x16, w, bias if mode == 0 else None, L.ACT_RELU if mode == 0 else 0, True, out16=True)
            ys.append(f().float().clone())
            us = t(f)
            res.append(f"{'stride-1 s2f' if on else 'general'} {us:.1f} us = {fl / us / 1e6:.0f} TF ({fl / us / 1e6 / 2500:.3f})")
        L.check(L.load().mmh_set_option(b"lp16_s2f", 1), "opt")
        print(f"B={B} {H}x{H} 64->64 {name}: " + "; ".join(res) + f"; max diff {float((ys[0] - ys[1]).abs().max()):.2e} of {float(ys[0].abs().max()):.1f}")
