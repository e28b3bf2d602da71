"""Interleaved A/B of library tuning knobs in ONE process (rule: never compare across devices).
usage: python tools/ab_conv.py key v0 v1 [shape_index] [pass]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib

key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B = 32
SHAPES = [(64, 64, 512, 512, 3, 1, 1, True), (64, 64, 256, 256, 3, 1, 1, True), (64, 64, 512, 256, 3, 1, 1, True),
          (256, 256, 64, 64, 3, 1, 1, False)]
dev = torch.device("cuda:0")
L = lib.load()

def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (H, W, Cin, Cout, k, s, p, refl) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0); dy = torch.randn_like(y)
    flop = 2.0 * y.numel() * Cin * k * k
    fns = {"fprop": lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0),
           "dgrad": lambda: ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl),
           "wgrad": lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl)}
    for name, fn in fns.items():
        res = {v0: [], v1: []}
        for v in (v0, v1):
            lib.check(L.mmh_set_option(key.encode(), v), "set"); fn()
        torch.cuda.synchronize()
        for r in range(6):
            for v in (v0, v1):
                lib.check(L.mmh_set_option(key.encode(), v), "set")
                res[v].append(timeit(fn))
        m0, m1 = statistics.median(res[v0]), statistics.median(res[v1])
        print(f"{Cin}->{Cout}@{H} {name}: {key}={v0}: {m0:.3f} ms {flop/m0/1e9:6.1f} TF | {key}={v1}: {m1:.3f} ms {flop/m1/1e9:6.1f} TF | ratio {m0/m1:.3f}", flush=True)
