"""conv_lp16h2_kernel with its LDS-DMA switched off (timing only, results wrong; mmh_set_option lp16_dbg: 1 = no weight
stages, 2 = no halo stages): what the kernel costs without waiting for memory.  The first timing of a process runs slow."""
import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H = 32, 64
for (Cin, Cout) in ((512, 512), (256, 256)):
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    xb = ops.lp16_twin(x, True)
    f = lambda: ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0, out16=True)
    for dbg in (0, 0, 1, 2, 3):
        L.mmh_set_option(b"lp16_dbg", dbg); f(); torch.cuda.synchronize()
        ts = [timeit(f) for _ in range(4)]
        print(Cin, Cout, "dbg", dbg, "%.0f us" % (statistics.median(ts) * 1e3), flush=True)
L.mmh_set_option(b"lp16_dbg", 0)
