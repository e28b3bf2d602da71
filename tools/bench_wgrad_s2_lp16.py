"""16-bit weight gradient of the 3x3 / stride-2 convs: the nine-tap halo kernel's stride-2 form (wgrad_lp16t_kernel<., 2>,
mmh_set_option("lp16_wgrad_s2", 1), default) against the flat (tap, channel)-row kernel (0), at the training shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib, ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


for B, H, Cin, Cout in ((32, 256, 64, 128), (64, 256, 64, 128), (32, 128, 128, 256), (64, 128, 128, 256), (4, 512, 64, 128)):
    x16 = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    dy16 = torch.randn(B, H // 2, H // 2, Cout, device=dev).bfloat16()
    mk = lambda: ops.conv_desc(B, H, H, Cin, Cout, 3, 2, 1, False)
    res = []
    for opt in (0, 1):
        lib.check(lib.load().mmh_set_option(b"lp16_wgrad_s2", opt), "mmh_set_option")
        res.append(timeit(lambda: ops.raw_wgrad_lp16_flat(mk(), x16, Cin, dy16, True)))
    fl = 2.0 * B * (H // 2) ** 2 * 9 * Cin * Cout
    mb = (x16.numel() + dy16.numel()) * 2 / 1e6
    print(f"B={B} {H}x{H} {Cin}->{Cout}: flat {res[0]:.0f} us ({fl / res[0] / 1e6:.0f} TF), halo {res[1]:.0f} us "
          f"({fl / res[1] / 1e6:.0f} TF = {fl / res[1] / 1e6 / 2500:.2f} of 2500; {mb:.0f} MB of operands = {mb / res[1]:.1f} TB/s)", flush=True)
