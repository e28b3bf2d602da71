"""Micro-benchmark of the conv kernels at the true hot-path shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (name, H, W, Cin, Cout, k, s, p, reflect)
SHAPES = [
    ("K4 256->256", 64, 64, 256, 256, 3, 1, 1, True),
    ("K4 512->512", 64, 64, 512, 512, 3, 1, 1, True),
    ("K4 512->256", 64, 64, 512, 256, 3, 1, 1, True),
    ("K3 64->128 s2", 256, 256, 64, 128, 3, 2, 1, False),
    ("K3 128->256 s2", 128, 128, 128, 256, 3, 2, 1, False),
    ("K1 44->64 7x7", 256, 256, 44, 64, 7, 1, 3, True),
    ("K1 4->64 7x7", 256, 256, 4, 64, 7, 1, 3, True),
    ("K2 64->4 7x7", 256, 256, 64, 4, 7, 1, 3, True),
    ("VGG 64->64", 256, 256, 64, 64, 3, 1, 1, False),
]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, H, W, Cin, Cout, k, s, p, refl in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0)
    dy = torch.randn_like(y)
    flop = 2.0 * y.numel() / Cout * Cout * Cin * k * k
    t_f = timeit(lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0))
    t_d = timeit(lambda: ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl))
    t_w = timeit(lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl))
    print(f"{name:18s} B={B} GF={flop/1e9:8.1f} | fprop {t_f:7.3f} ms {flop/t_f/1e9:6.1f} TF | "
          f"dgrad {t_d:7.3f} ms {flop/t_d/1e9:6.1f} TF | wgrad {t_w:7.3f} ms {flop/t_w/1e9:6.1f} TF", flush=True)
    del x, w, y, dy
