"""The Generator head's weight gradient (64 -> 3 of 4 columns, 7x7, reflect): the 16-bit stem-wgrad form
(mmh_conv7_head_wgrad_lp16) against the fp32 vector-ALU kernel (mmh_conv7_thin_wgrad) at the training shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops

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


for B, H in ((32, 256), (4, 512)):
    x = torch.randn(B, H, H, 64, device=dev)
    g = torch.randn(B, H, H, 4, device=dev)
    g[..., 3] = 0
    x16 = ops.lp16_twin(x, True)
    new = timeit(lambda: ops.raw_head_wgrad16(x16, g, True))
    old = timeit(lambda: ops.raw_conv_wgrad(x, g, 7, 1, 3, True, bf16=False))
    print(f"B={B} {H}x{H}: head wgrad 16-bit {new:.0f} us, fp32 vector-ALU {old:.0f} us", flush=True)
