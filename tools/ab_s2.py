"""A/B a knob on the stride-2 conv dgrad and ConvTranspose fprop shapes of the step."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2:]]
L = lib.load(); dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, Cin, Cout) in [(256, 64, 128), (128, 128, 256)]:
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dy = torch.randn(B, H // 2, H // 2, Cout, device=dev)
    xs = (B, H, H, Cin)
    wt = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05      # convT weight [kh,kw,CoutT=Cin,CinT=Cout]
    xt = torch.randn(B, H // 2, H // 2, Cout, device=dev)
    ref = {}
    for name, fn in (("s2 dgrad", lambda: ops.raw_conv_dgrad(dy, w, xs, 2, 1, False)),
                     ("convT fprop", lambda: ops.raw_convT_fprop(xt, wt, None))):
        res = {v: [] for v in vals}
        for v in vals:
            lib.check(L.mmh_set_option(key.encode(), v), "set"); out = fn(); torch.cuda.synchronize()
            if name not in ref: ref[name] = out.clone()
            else: assert torch.equal(ref[name], out), "knob changes the result"
        for r in range(5):
            for v in vals:
                lib.check(L.mmh_set_option(key.encode(), v), "set"); res[v].append(timeit(fn))
        print(f"{Cin}->{Cout}@{H} {name}: " + " | ".join(f"{key}={v}: {statistics.median(res[v]):.3f} ms" for v in vals), flush=True)
