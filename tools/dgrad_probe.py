"""Is dgrad's weight-tile path (B_NMAJOR) slower than fprop's?  Time an fprop with the same GEMM
shape as the reflect dgrad (66x66 rows per image, zero padding) against the real dgrad."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for C in (512, 256):
    w = torch.randn(3, 3, C, C, device=dev) * 0.05
    xp = torch.randn(B, 66, 66, C, device=dev)          # fprop, zero pad 1 -> 66x66 output rows
    dy = torch.randn(B, 64, 64, C, device=dev)
    f = lambda: ops.raw_conv_fprop(xp, w, None, 1, 1, False, 0)
    d = lambda: ops.raw_conv_dgrad(dy, w, (B, 64, 64, C), 1, 1, True)
    f(); d(); torch.cuda.synchronize()
    rf, rd = [], []
    for _ in range(5):
        rf.append(timeit(f)); rd.append(timeit(d))
    print(f"C={C}: fprop on 66x66 rows {statistics.median(rf):.3f} ms | reflect dgrad (+fold) {statistics.median(rd):.3f} ms")
