"""Reflect dgrad of the 16-bit 3x3 stack as the training step calls it (dx in 16 bits): mmh_conv3x3_lp16 mode 2 (border
terms folded inside the halo kernel) against mode 1 + mmh_conv2d_dgrad_border; interleaved rounds in one process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, Cin, Cout) in ((32, 64, 256, 256), (32, 64, 512, 512), (32, 64, 256, 512), (4, 128, 256, 256)):
    dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dyb = ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    def run(fold):
        ops.USE_LP16_FOLD = fold
        return ops.raw_conv_dgrad(None, w, (B, H, H, Cin), 1, 1, True, bf16=True, dy16=dyb, out16=True)
    main = lambda: ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1, out16=True)
    def dbg(v):
        def f():
            lib.check(L.mmh_set_option(b"lp16_dbg", v), "set")
            r = run(True)
            lib.check(L.mmh_set_option(b"lp16_dbg", 0), "set")
            return r
        return f
    addend = torch.randn(B, H, H, Cin, device=dev)
    def run32(ad):
        ops.USE_LP16_FOLD = True
        return ops.raw_conv_dgrad(None, w, (B, H, H, Cin), 1, 1, True, bf16=True, dy16=dyb, out16=False, addend=ad)
    variants = {"mode 2, fp32 dx": lambda: run32(None), "mode 2, fp32 dx + addend (dgrad_add)": lambda: run32(addend),
                "main term only": main, "mode 1 + border": lambda: run(False), "mode 2 (fold)": lambda: run(True),
                "mode 2, folds switched off (timing only)": dbg(4), "no column folds": dbg(8), "no row folds": dbg(16),
                "column fragment from a conflict-free address (timing only)": dbg(64)}
    a, b_ = run(True).float(), run(False).float()
    print(f"B{B} {H}x{H} {Cout}->{Cin}: max |fold - border| / max = {float((a - b_).abs().max() / b_.abs().max()):.2e}")
    res = {k: [] for k in variants}
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    print("   " + " | ".join(f"{k}: {statistics.median(v)*1e3:.0f} us ({flop/statistics.median(v)/1e9:.0f} TF)" for k, v in res.items()), flush=True)
ops.USE_LP16_FOLD = True
