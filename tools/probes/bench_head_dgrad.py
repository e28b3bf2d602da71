"""The Generator head's input gradient (64 <- 4 channels, 7x7 reflect): fp32 implicit GEMM + fold against the 16-bit stem kernel path."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, H in ((32, 256), (4, 512)):
    dy = torch.randn(B, H, H, 4, device=dev); w = torch.randn(7, 7, 64, 4, device=dev) * 0.05
    out = []
    for on in (False, True):
        ops.USE_HEAD_DGRAD16 = on
        for o16 in ((False, True) if on else (False,)):
            out.append(f"{'stem16 path' if on else 'fp32 path'}{' dx16' if o16 else ''}: {t(lambda: ops.raw_conv_dgrad(dy, w, (B, H, H, 64), 1, 3, True, bf16=True, out16=o16)):.0f} us")
    print(f"B={B} {H}x{H}: " + " | ".join(out))
