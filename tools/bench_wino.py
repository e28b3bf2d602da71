import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (H, Cin, Cout) in [(64, 512, 512), (64, 256, 256), (64, 512, 256), (256, 64, 64)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    fl = 2.0 * B * H * H * Cin * Cout * 9
    ops.USE_WINOGRAD = False
    fd = lambda: ops.raw_conv_fprop(x, w, None, 1, 1, True, 0)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    fw = lambda: ops.raw_conv_fprop_wino(x, w, None, True, 0, T)
    y = fd(); dy = torch.randn_like(y)
    gd = lambda: ops.raw_conv_dgrad(dy, w, x.shape, 1, 1, True)
    gw = lambda: ops.raw_conv_dgrad_wino(dy, w, x.shape, True, T)
    fd(); fw(); gd(); gw(); torch.cuda.synchronize()
    r1, r2 = [], []
    for _ in range(5):
        r1.append(timeit(gd)); r2.append(timeit(gw))
    hd = lambda: ops.raw_conv_wgrad(x, dy, 3, 1, 1, True)
    hw = lambda: ops.raw_conv_wgrad_wino(x, dy, True, T)
    hd(); hw(); torch.cuda.synchronize()
    r3, r4 = [], []
    for _ in range(5):
        r3.append(timeit(hd)); r4.append(timeit(hw))
    print(f"{Cin}->{Cout}@{H}: wgrad direct {statistics.median(r3):.3f} ms | winograd {statistics.median(r4):.3f} ms | speedup {statistics.median(r3)/statistics.median(r4):.2f}x", flush=True)
    print(f"{Cin}->{Cout}@{H}: dgrad direct {statistics.median(r1):.3f} ms | winograd {statistics.median(r2):.3f} ms | speedup {statistics.median(r1)/statistics.median(r2):.2f}x", flush=True)
    rd, rw = [], []
    for _ in range(5):
        rd.append(timeit(fd)); rw.append(timeit(fw))
    md, mw = statistics.median(rd), statistics.median(rw)
    print(f"{Cin}->{Cout}@{H}: direct {md:.3f} ms ({fl/md/1e9:.1f} TF) | winograd {mw:.3f} ms ({fl/mw/1e9:.1f} TF-equivalent) | speedup {md/mw:.2f}x", flush=True)
