"""The halo kernel (mmh_conv3x3_lp16, lp16_shape 19: conv_lp16h2_kernel) as the 16-bit training step calls it, beside the row-tile
kernel (17: conv_lp16p_kernel): fprop with bias and a 16-bit epilogue, dgrad main term with a 16-bit epilogue, on the PATBlock
shapes; interleaved rounds in one process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
def setshape(v): lib.check(L.mmh_set_option(b"lp16_shape", v), "set")
for (Cin, Cout) in ((256, 256), (512, 512), (512, 256)):
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    xb, dyb = ops.lp16_twin(x, True), ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    def mk(shape, fn):
        def run():
            setshape(shape)
            return fn()
        return run
    fp = lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True)
    dg = lambda: ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1, out16=True)
    fps = lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True, want_stats=True)
    variants = {"p(17) fprop": mk(17, fp), "h2(19) fprop": mk(19, fp), "h2 fprop + statistics": mk(19, fps),
                "p(17) dgrad": mk(17, dg), "h2(19) dgrad": mk(19, dg)}
    res = {k: [] for k in variants}
    for f in variants.values(): f()
    torch.cuda.synchronize()
    for r in range(5):
        for k, f in variants.items(): res[k].append(timeit(f))
    print(f"{Cin}->{Cout}: " + " | ".join(f"{k}: {statistics.median(v)*1e3:.0f} us ({flop/statistics.median(v)/1e9:.0f} TF)" for k, v in res.items()), flush=True)
setshape(19)
