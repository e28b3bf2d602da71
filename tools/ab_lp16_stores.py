"""conv_lp16h2_kernel's 16-bit epilogue: 8-byte stores (lp16_dbg 128) against 16-byte stores after v_permlane16_swap (0); bit-identical results."""
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
def opt(k, v): lib.check(L.mmh_set_option(k, v), "set")
B, H = 32, 64
for (Cin, Cout) in ((256, 256), (512, 512), (512, 256)):
    x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
    xb, dyb = ops.lp16_twin(x, True), ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    fns = {"fprop": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True),
           "fprop+stats": lambda: ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True, want_stats=True),
           "dgrad(reflect fold)": lambda: ops.raw_conv3x3_lp16(dyb, w, None, True, 0, True, 2, out16=True)}
    line = []
    for name, fn in fns.items():
        outs, ts = {}, {}
        for tag, dbg in {"8B": 128, "16B": 0}.items():
            opt(b"lp16_dbg", dbg)
            poison = [torch.full((B, H, H, max(Cin, Cout)), float("nan"), device=dev) for _ in range(3)]; del poison
            outs[tag] = fn().clone(); fn(); torch.cuda.synchronize()
            ts[tag] = statistics.median(timeit(fn) for _ in range(5)) * 1e3
        opt(b"lp16_dbg", 0)
        same = torch.equal(outs["8B"], outs["16B"])
        line.append(f"{name}: 8-byte {ts['8B']:.0f} us, 16-byte {ts['16B']:.0f} ({flop / ts['16B'] / 1e6:.0f} TF){'' if same else '  RESULTS DIFFER'}")
    print(f"B={B} {H}x{H} {Cin}->{Cout}: " + " | ".join(line), flush=True)
