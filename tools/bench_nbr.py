"""The fold dgrad with the norm-backward sums in its epilogue (mmh_conv3x3_lp16_dgrad_nbr) against the plain fold dgrad
followed by mmh_norm_bwd_reduce, at the 16-bit step's shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib, ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


for B, H, Cin, Cout in ((32, 64, 512, 512), (32, 64, 256, 256), (64, 64, 256, 256), (32, 64, 256, 512)):
    dy16 = torch.randn(B, H, H, Cout, device=dev).bfloat16()
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    xn = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    mean = torch.zeros(B, Cin, device=dev); invstd = torch.ones(B, Cin, device=dev)
    bits = torch.randint(-32768, 32767, (B * H * H * Cin // 8,), device=dev, dtype=torch.int16)
    site = ops.NormBwdSite(xn, bits, mean, invstd, B, 0.5)
    ops.bump_weights_epoch()
    rows = H * H
    s1 = torch.empty(B, Cin, device=dev); s2 = torch.empty(B, Cin, device=dev)
    nws = lib.load().mmh_norm_bwd_ws_bytes(B, rows, Cin)
    ws = torch.empty(nws // 4 + 4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plain = lambda: ops.raw_conv3x3_lp16(dy16, w, None, True, lib.ACT_NONE, True, 2, out16=True)
    dx = plain()
    red = lambda: lib.call("mmh_norm_bwd_reduce", dx.data_ptr(), bits.data_ptr(), xn.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                           B, rows, Cin, 2, 0.5, s1.data_ptr(), s2.data_ptr(), ws.data_ptr(), ws.numel() * 4, lib.BF16, lib.BF16, st)
    fused = lambda: ops.raw_conv3x3_lp16(dy16, w, None, True, lib.ACT_NONE, True, 2, out16=True, nbr=site)
    tp, tr, tf = timeit(plain), timeit(red), timeit(fused)
    print(f"B={B} {H}x{H} {Cout}->{Cin}: fold dgrad {tp:.0f} us + reduce {tr:.0f} us = {tp + tr:.0f}; fused {tf:.0f} us "
          f"(saves {tp + tr - tf:.0f})", flush=True)
