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
for (H, Cin, Cout, k, s, p, refl) in [(256, 44, 64, 7, 1, 3, True), (256, 4, 64, 7, 1, 3, True), (256, 8, 64, 7, 1, 3, True),
                                       (256, 24, 64, 7, 1, 3, True), (256, 4, 64, 3, 1, 1, False), (256, 64, 4, 7, 1, 3, True)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0); dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * k * k
    for bf in (False, True):
        tf = timeit(lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0, bf16=bf))
        tw = timeit(lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl, bf16=bf))
        print(f"{Cin}->{Cout} k{k} {'bf16' if bf else 'fp32'}: fprop {tf:.3f} ms {fl/tf/1e9:7.1f} TF | wgrad {tw:.3f} ms {fl/tw/1e9:7.1f} TF", flush=True)
