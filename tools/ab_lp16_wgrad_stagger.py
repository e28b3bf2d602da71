"""wgrad_lp16t_kernel with the next stage's four DMA instructions issued by every wave behind the barrier
(mmh_set_option("lp16_wgrad_ring", 3)) against the staggered issue (2, the default: wr = 0 waves behind the barrier, wr = 1 waves two MFMA
groups later), on the PATBlock shapes; results bit-identical."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); L = lib.load()
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, H, W, Cin, Cout) in ((32, 64, 64, 512, 512), (32, 64, 64, 256, 256), (32, 64, 64, 512, 256), (3, 17, 33, 256, 512)):
    x = torch.randn(B, H, W, Cin, device=dev); dy = torch.randn(B, H, W, Cout, device=dev)
    xb = ops.lp16_twin(x, True); dyb = ops.lp16_twin(dy, True)
    flop = 2.0 * B * H * W * Cin * Cout * 9
    fn = lambda: ops.raw_wgrad3x3_lp16(xb, dyb, True, True)
    res, outs = {3: [], 2: []}, {}
    for v in (3, 2):
        L.mmh_set_option(b"lp16_wgrad_ring", v); outs[v] = fn().clone(); torch.cuda.synchronize()
    assert torch.equal(outs[2], outs[3])
    for _ in range(5):
        for v in (3, 2):
            L.mmh_set_option(b"lp16_wgrad_ring", v); res[v].append(timeit(fn))
    a, b = statistics.median(res[3]), statistics.median(res[2])
    print(f"B{B} {H}x{W} {Cin}->{Cout}: together {a*1e3:.0f} us ({flop/a/1e9:.0f} TF) | staggered {b*1e3:.0f} us ({flop/b/1e9:.0f} TF)", flush=True)
L.mmh_set_option(b"lp16_wgrad_ring", 2)
