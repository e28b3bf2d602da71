"""fp32 fprop of the 7x7 stems @256x256 (B=32): halo-resident kernel (conv_stem_f32.hip) against the generic implicit GEMM."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 32)); H = 256
def timeit(fn, iters=6):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for Cin in (8, 24, 44):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(7, 7, Cin, 64, device=dev) * 0.05
    bias = torch.randn(64, device=dev)
    flop = 2.0 * B * H * H * 64 * Cin * 49
    fn = lambda: ops.raw_conv_fprop(x, w, bias, 1, 3, True, 0, want_stats=True)
    res, outs = {0: [], 1: []}, {}
    for v in (0, 1):
        lib.call("mmh_set_option", b"stem_f32", v); outs[v] = fn().clone(); torch.cuda.synchronize()
    rel = float((outs[1].double() - outs[0].double()).abs().sum() / outs[0].double().abs().sum())
    for _ in range(5):
        for v in (0, 1):
            lib.call("mmh_set_option", b"stem_f32", v); res[v].append(timeit(fn))
    m = {v: statistics.median(res[v]) for v in res}
    print(f"stem fprop {Cin}->64 @{H}: implicit GEMM {m[0] * 1e3:.0f} us = {flop / m[0] / 1e9:.1f} TF ({flop / m[0] / 1e9 / 157.3:.2f}) | "
          f"halo-resident {m[1] * 1e3:.0f} us = {flop / m[1] / 1e9:.1f} TF ({flop / m[1] / 1e9 / 157.3:.2f}) | rel diff {rel:.1e}", flush=True)
lib.call("mmh_set_option", b"stem_f32", 1)
ops._pending_stats.clear()
