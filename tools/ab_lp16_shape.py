"""conv_lp16 fprop: MFMA 32x32x16 vs 16x16x32 (one process, interleaved rounds), correctness of both."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from mmhand_amd import ops, lib
from oracle import ops_ref as R
L = lib.load(); dev = torch.device("cuda:0")
def setopt(k, v): lib.check(L.mmh_set_option(k.encode(), v), "set")
def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator().manual_seed(0)
x = torch.rand(2, 9, 11, 128, generator=g) * 2 - 1; w = (torch.rand(3, 3, 128, 256, generator=g) * 2 - 1) * 0.1
rb = lambda t: t.bfloat16().float()
yr = R.conv2d(rb(x), rb(w), None, 1, 1, True, 0)
for shape in (32, 16):
    setopt("lp16_shape", shape)
    y = ops.raw_conv3x3_lp16(ops.lp16_twin(x.to(dev), True), w.to(dev), None, True, 0, True, 0)
    print(f"shape {shape}: rel-L1 vs oracle {R.rel_l1(y, yr):.2e}")
for (Cin, Cout) in ((512, 512), (256, 256), (512, 256)):
    B, H = 32, 64
    xx = torch.randn(B, H, H, Cin, device=dev); ww = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    xb = ops.lp16_twin(xx, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    variants = [(32, 0), (16, 0), (32, 1), (16, 1)]
    res = {v: [] for v in variants}
    for v in variants:
        setopt("lp16_shape", v[0]); setopt("lp16_tap_inner", v[1]); ops.raw_conv3x3_lp16(xb, ww, None, True, 0, True, 0)
    torch.cuda.synchronize()
    for r in range(6):
        for v in variants:
            setopt("lp16_shape", v[0]); setopt("lp16_tap_inner", v[1])
            res[v].append(timeit(lambda: ops.raw_conv3x3_lp16(xb, ww, None, True, 0, True, 0)))
    print(f"{Cin}->{Cout}: " + " | ".join(f"shape{v[0]} tap_inner={v[1]}: {statistics.median(res[v])*1e3:.0f} us ({flop/statistics.median(res[v])/1e9:.0f} TF)" for v in variants), flush=True)
setopt("lp16_shape", 16); setopt("lp16_tap_inner", 1)
