"""Does the last partial round of workgroups explain the gap to the MFMA peak?  Time the dominant
fprop at batch sizes that give whole and fractional numbers of 768-workgroup rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for Cin, Cout in ((512, 512), (256, 256)):
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    for B in (6, 12, 24, 30, 32, 36, 48):
        x = torch.randn(B, 64, 64, Cin, device=dev)
        tiles = B * 4096 // 128 * (Cout // 128)
        t = timeit(lambda: ops.raw_conv_fprop(x, w, None, 1, 1, True, 0))
        fl = 2.0 * B * 4096 * Cin * Cout * 9
        print(f"{Cin}->{Cout} B={B:3d} tiles={tiles:5d} rounds768={tiles/768:5.2f} {t:7.3f} ms {fl/t/1e9:6.1f} TF", flush=True)
