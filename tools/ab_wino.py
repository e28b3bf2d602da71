"""A/B a knob on the Winograd passes of the K4 shapes (one process)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2:]]
L = lib.load(); dev = torch.device("cuda:0"); B = 32; TILE = int(os.environ.get("TILE", "6"))
def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, Cin, Cout) in [(64, 512, 512), (64, 256, 256), (64, 512, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dy = torch.randn(B, H, H, Cout, device=dev)
    for name, fn in (("fprop", lambda: ops.raw_conv_fprop_wino(x, w, None, True, 0, TILE)),
                     ("dgrad", lambda: ops.raw_conv_dgrad_wino(dy, w, x.shape, True, TILE)),
                     ("wgrad", lambda: ops.raw_conv_wgrad_wino(x, dy, True, TILE))):
        res = {v: [] for v in vals}
        for v in vals:
            lib.check(L.mmh_set_option(key.encode(), v), "set"); fn()
        torch.cuda.synchronize()
        for r in range(5):
            for v in vals:
                lib.check(L.mmh_set_option(key.encode(), v), "set"); res[v].append(timeit(fn))
        print(f"{Cin}->{Cout}@{H} {name}: " + " | ".join(f"{key}={v}: {statistics.median(res[v]):.3f} ms" for v in vals), flush=True)
