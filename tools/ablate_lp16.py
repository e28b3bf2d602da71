"""Timing-only ablation of conv_lp16s_kernel (mmh_set_option "lp16_dbg"; results are wrong with any bit
set): 1 = no LDS-DMA after the first stage, 2 = B fragments not read from LDS, 4 = no MFMAs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
SHAPE = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = lib.load()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (Cin, Cout) in ((512, 512), (256, 256)):
    B, H = 32, 64
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    ops.bump_weights_epoch()
    xb = ops.lp16_twin(x, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    for bits in (0, 1, 2, 4, 3, 5, 6, 7):
        L.mmh_set_option(b"lp16_shape", SHAPE)
        L.mmh_set_option(b"lp16_dbg", bits)
        t = timeit(lambda: ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0))
        print(f"{Cin}->{Cout} dbg={bits} (noDMA={bits&1} noread={(bits>>1)&1} noMFMA={(bits>>2)&1}): {t*1e3:8.1f} us  {flop/t*1e-9:7.0f} TF-equiv", flush=True)
    L.mmh_set_option(b"lp16_dbg", 0)

print("wgrad_lp16_kernel:")
for (Cin, Cout) in ((512, 512), (256, 256)):
    B, H = 32, 64
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    xb = ops.lp16_twin(x, True); dyb = ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    for bits in (0, 1, 2, 4, 3, 5, 6, 7):
        L.mmh_set_option(b"lp16_dbg", bits)
        t = timeit(lambda: ops.raw_wgrad3x3_lp16(xb, dyb, True, True))
        print(f"wgrad {Cin}->{Cout} dbg={bits} (noDMA={bits&1} noread={(bits>>1)&1} noMFMA={(bits>>2)&1}): {t*1e3:8.1f} us  {flop/t*1e-9:7.0f} TF-equiv", flush=True)
    L.mmh_set_option(b"lp16_dbg", 0)
