"""VGG19's conv1_1 (3 -> 64, 3x3, zero padding, full resolution, 16-bit ReLU output): the stem kernel's 3x3 form against the
flat-K kernel (MMH_STEM3=0)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib, ops
dev = torch.device("cuda:0")


def timeit(fn, iters=20, reps=5):
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


for B, H in ((32, 256), (4, 512)):
    x = torch.randn(B, H, H, 4, device=dev); x[..., 3] = 0
    w = torch.randn(3, 3, 4, 64, device=dev) * 0.1
    b = torch.randn(64, device=dev)
    x16p = ops.lp16_pad8(x, True)
    res = []
    for on in (False, True):
        ops.USE_STEM3 = on
        res.append(timeit(lambda: ops.raw_conv_lp16_flat(ops.conv_desc(B, H, H, 4, 64, 3, 1, 1, False), None, w, b, lib.ACT_RELU, True,
                                                         out16=True, x16p=x16p)))
    mb = B * H * H * (8 + 64) * 2 / 1e6
    print(f"B={B} {H}x{H} 4->64 k3: flat-K {res[0]:.0f} us, stem 3x3 form {res[1]:.0f} us ({mb:.0f} MB = {mb / res[1]:.2f} TB/s)", flush=True)
