"""Direct fp32 wgrad: the split-K slot count (wgrad_slots) on the stem and stride-2 shapes.

    python tools/ab_wgrad_slots.py
"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib, ops      # noqa: E402
L = lib.load(); dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (H, Cin, Cout, k, s, p, refl) in [(256, 44, 64, 7, 1, 3, True), (256, 24, 64, 7, 1, 3, True), (256, 8, 64, 7, 1, 3, True),
                                      (256, 64, 128, 3, 2, 1, False), (128, 128, 256, 3, 2, 1, False)]:
    x = torch.randn(32, H, H, Cin, device=dev)
    Ho = H // s
    dy = torch.randn(32, Ho, Ho, Cout, device=dev)
    fl = 2.0 * dy.numel() * Cin * k * k
    fw = lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl)
    out = []
    for slots in (768, 1024, 1536, 2304, 512):
        lib.check(L.mmh_set_option(b"wgrad_slots", slots), "o")
        fw(); torch.cuda.synchronize()
        m = statistics.median([timeit(fw) for _ in range(3)])
        out.append(f"{slots}: {m*1e3:7.1f} us {fl/m/1e9:6.1f} TF")
    lib.check(L.mmh_set_option(b"wgrad_slots", 768), "o")
    print(f"{Cin}->{Cout} k{k} s{s}: " + " | ".join(out), flush=True)
