"""conv_lp16.hip (16-bit direct 3x3, both operands 16-bit, LDS-DMA) against the fp64 oracle (small) and
timed on the PATBlock shapes (B=32, 64x64) beside the first-generation bf16 kernel."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from mmhand_amd import ops, lib
from oracle import ops_ref as R
dev = torch.device("cuda:0")
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
g = torch.Generator().manual_seed(0)
for lp in (True, 2):
    rb = (lambda t: t.half().float()) if lp == 2 else (lambda t: t.bfloat16().float())
    for (B, H, W, Cin, Cout, refl) in ((2, 9, 11, 64, 256, True), (1, 16, 16, 128, 256, False), (3, 7, 5, 256, 512, True)):
        x = torch.rand(B, H, W, Cin, generator=g) * 2 - 1
        w = (torch.rand(3, 3, Cin, Cout, generator=g) * 2 - 1) * 0.1
        bias = torch.rand(Cout, generator=g)
        ops.bump_weights_epoch()
        wd = w.to(dev)
        y = ops.raw_conv3x3_lp16(ops.lp16_twin(x.to(dev), lp), wd, bias.to(dev), refl, 1, lp, 0)
        yr = R.conv2d(rb(x), rb(w), bias, 1, 1, refl, 1)
        print(f"lp={lp} fprop {B}x{H}x{W} {Cin}->{Cout} reflect={refl}: rel-L1 {R.rel_l1(y, yr):.2e}", flush=True)
        if Cin % 256 == 0:
            dy = torch.rand(B, H, W, Cout, generator=g) * 2 - 1
            dx = ops.raw_conv3x3_lp16(ops.lp16_twin(dy.to(dev), lp), wd, None, False, 0, lp, 1)
            _, dxr, _, _ = R.conv2d_grads(torch.zeros(B, H, W, Cin), rb(w), None, rb(dy), 1, 1, False)
            print(f"lp={lp} dgrad(zero pad) {Cout}->{Cin}: rel-L1 {R.rel_l1(dx, dxr):.2e}", flush=True)
for lp in (True, 2):
    rb = (lambda t: t.half().float()) if lp == 2 else (lambda t: t.bfloat16().float())
    for (B, H, W, Cin, Cout, refl) in ((2, 9, 11, 256, 256, True), (1, 16, 16, 512, 256, False), (3, 7, 5, 256, 512, True)):
        x = torch.rand(B, H, W, Cin, generator=g) * 2 - 1
        dy = torch.rand(B, H, W, Cout, generator=g) * 2 - 1
        dw = ops.raw_wgrad3x3_lp16(ops.lp16_twin(x.to(dev), lp), ops.lp16_twin(dy.to(dev), lp), refl, lp)
        _, _, dwr, _ = R.conv2d_grads(rb(x), torch.zeros(3, 3, Cin, Cout), None, rb(dy), 1, 1, refl)
        print(f"lp={lp} wgrad {B}x{H}x{W} {Cin}->{Cout} reflect={refl}: rel-L1 {R.rel_l1(dw, dwr):.2e}", flush=True)
for (Cin, Cout) in ((512, 512), (256, 256), (512, 256)):
    B, H = 32, 64
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    ops.bump_weights_epoch()
    xb = ops.lp16_twin(x, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    t2 = timeit(lambda: ops.raw_conv3x3_lp16(xb, w, None, True, 0, True, 0))
    tc = timeit(lambda: ops.lp16_twin(x, True))
    ops.USE_WINOGRAD_BF16 = False
    t1 = timeit(lambda: ops.raw_conv_fprop(x, w, None, 1, 1, True, 0, True))
    dy = torch.randn(B, H, H, Cout, device=dev); dyb = ops.lp16_twin(dy, True)
    tw2 = timeit(lambda: ops.raw_wgrad3x3_lp16(xb, dyb, True, True))
    ops.USE_LP16_V2 = False
    tw1 = timeit(lambda: ops.raw_conv_wgrad(x, dy, 3, 1, 1, True, True))
    ops.USE_WINOGRAD_BF16 = True
    tw1w = timeit(lambda: ops.raw_conv_wgrad(x, dy, 3, 1, 1, True, True))
    ops.USE_WINOGRAD_BF16 = False; ops.USE_LP16_V2 = True
    print(f"{Cin}->{Cout} wgrad: v2 {tw2*1e3:.0f} us ({flop/tw2/1e9:.0f} TF) | v1 direct {tw1*1e3:.0f} us | v1 F(2x2,3x3) {tw1w*1e3:.0f} us")
    print(f"{Cin}->{Cout} @64x64 B=32 fprop: v2 {t2*1e3:.0f} us ({flop/t2/1e9:.0f} TF) + twin {tc*1e3:.0f} us | v1 {t1*1e3:.0f} us ({flop/t1/1e9:.0f} TF)", flush=True)
