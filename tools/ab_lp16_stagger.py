"""conv_lp16h2_kernel with the next stages' DMA issued by both waves of a SIMD behind the mid-k-step barrier
(mmh_set_option("lp16_dbg", 32): round 3's first build) against the staggered issue (0: wr = 0 waves there, wr = 1 waves after
half of half 1's multiplies): fprop, fprop + statistics, the complete reflect dgrad (mode 2), on the PATBlock shapes."""
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
for (Cin, Cout) in ((256, 256), (512, 512), (512, 256)):
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    xb, dyb = ops.lp16_twin(x, True), ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    fns = {"fprop": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True),
           "fprop+stats": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True, want_stats=True),
           "dgrad fold": lambda: ops.raw_conv3x3_lp16(dyb, w, None, True, 0, True, 2, out16=True)}
    row = []
    for name, fn in fns.items():
        res, outs = {32: [], 0: []}, {}
        for v in (32, 0):
            lib.call("mmh_set_option", b"lp16_dbg", v); outs[v] = fn().clone(); torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[32]), name
        for _ in range(5):
            for v in (32, 0):
                lib.call("mmh_set_option", b"lp16_dbg", v); res[v].append(timeit(fn))
        a, b = statistics.median(res[32]), statistics.median(res[0])
        row.append(f"{name}: together {a * 1e3:.0f} us ({flop / a / 1e9:.0f} TF) | staggered {b * 1e3:.0f} us ({flop / b / 1e9:.0f} TF)")
    print(f"{Cin}->{Cout}: " + " || ".join(row), flush=True)
lib.call("mmh_set_option", b"lp16_dbg", 0)
ops._pending_stats.clear()
