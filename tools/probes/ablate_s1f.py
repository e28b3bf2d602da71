import os, sys
sys.path.insert(0, "/root/repo")
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 32, 256
x16 = torch.randn(B, H, H, 64, device=dev).bfloat16(); w = torch.randn(3, 3, 64, 64, device=dev) * 0.05
d = lambda: ops.conv_desc(B, H, H, 64, 64, 3, 1, 1, False)
f = lambda: ops.raw_conv_lp16g(d(), 0, x16, w, None, 0, True, out16=True)
for dbg in (0, 1, 2, 3, 0):
    L.check(L.load().mmh_set_option(b"lp16_dbg", dbg), "opt")
    print("dbg", dbg, "%.1f us" % t(f), flush=True)
L.check(L.load().mmh_set_option(b"lp16_dbg", 0), "opt")
