import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, Cin, Cout, k, s, p, refl) in [(64, 512, 512, 3, 1, 1, True), (64, 256, 256, 3, 1, 1, True), (64, 512, 256, 3, 1, 1, True),
                                       (128, 128, 256, 3, 2, 1, False), (256, 64, 64, 3, 1, 1, False)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0); dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * k * k
    for bf in (False, True):
        tf = timeit(lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0, bf16=bf))
        td = timeit(lambda: ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl, bf16=bf))
        print(f"{Cin}->{Cout}@{H} s{s} {'bf16' if bf else 'fp32'}: fprop {tf:.3f} ms {fl/tf/1e9:7.1f} TF | dgrad {td:.3f} ms {fl/td/1e9:7.1f} TF", flush=True)
