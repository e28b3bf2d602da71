"""wgrad_lp16_kernel (0: two 64-KiB stages) against wgrad_lp16r_kernel (1: ring of five 32-KiB half stages) and
wgrad_lp16t_kernel (2: nine taps of a 64 x 128 tile resident; mmh_set_option lp16_wgrad_ring): results vs each other
(repeated runs) and time on the PATBlock shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
L = lib.load()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (B, H, W, Cin, Cout, refl) in ((2, 9, 11, 256, 256, True), (3, 17, 33, 256, 512, False), (32, 64, 64, 512, 512, True), (32, 64, 64, 256, 256, True),
                                   (32, 64, 64, 512, 256, True), (4, 128, 128, 256, 256, True)):
    x = torch.randn(B, H, W, Cin, device=dev); dy = torch.randn(B, H, W, Cout, device=dev)
    xb = ops.lp16_twin(x, True); dyb = ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * W * Cin * Cout * 9
    res = {}
    for ring in (0, 1, 2):
        L.mmh_set_option(b"lp16_wgrad_ring", ring)
        outs = [ops.raw_wgrad3x3_lp16(xb, dyb, refl, True) for _ in range(3)]
        t = timeit(lambda: ops.raw_wgrad3x3_lp16(xb, dyb, refl, True))
        res[ring] = outs
        print(f"B{B} {H}x{W} {Cin}->{Cout} ring={ring}: {t*1e3:8.1f} us  {flop/t*1e-9:7.0f} TF", flush=True)
    d = max((o - res[0][0]).abs().max().item() for o in res[0] + res[1] + res[2])
    print(f"   max |diff| over runs and kernels: {d:.3e} (max |dw| {res[0][0].abs().max().item():.1f})" + ("  <-- MISMATCH" if d > 1e-2 * res[0][0].abs().max().item() else ""))
L.mmh_set_option(b"lp16_wgrad_ring", 2)
