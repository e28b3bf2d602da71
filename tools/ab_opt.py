"""Interleaved A/B of one option over fprop+dgrad of the K4 shapes, one process."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2:]]
L = lib.load(); dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, Cin, Cout, k, s, p, refl) in [(64, 512, 512, 3, 1, 1, True), (64, 256, 256, 3, 1, 1, True), (128, 128, 256, 3, 2, 1, False), (256, 64, 64, 3, 1, 1, False)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0); dy = torch.randn_like(y)
    flop = 2.0 * y.numel() * Cin * k * k
    for name, fn in (("fprop", lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0)),
                     ("dgrad", lambda: ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl)),
                     ("wgrad", lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl))):
        res = {v: [] for v in vals}
        for v in vals:
            lib.check(L.mmh_set_option(key.encode(), v), "set"); fn()
        torch.cuda.synchronize()
        for r in range(5):
            for v in vals:
                lib.check(L.mmh_set_option(key.encode(), v), "set"); res[v].append(timeit(fn))
        print(f"{Cin}->{Cout}@{H} s{s} {name}: " + " | ".join(f"{key}={v}: {statistics.median(res[v]):.3f} ms {flop/statistics.median(res[v])/1e9:6.1f} TF" for v in vals), flush=True)
