import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
B, Cin, Cout = 32, 512, 512
x = torch.randn(B, 64, 64, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
def timeit(fn, iters=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
BF = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
fn = lambda: ops.raw_conv_fprop(x, w, None, 1, 1, True, 0, bf16=BF)
fl = 2.0 * B * 4096 * Cin * Cout * 9
res = {}
for r in range(5):
    for dbg in (0, 1, 2, 3):
        lib.check(L.mmh_set_option(b"conv_dbg", dbg), "set"); fn(); torch.cuda.synchronize()
        res.setdefault(dbg, []).append(timeit(fn))
for dbg, v in res.items():
    m = statistics.median(v)
    print(f"dbg={dbg} (1=no loads+addr, 2=no LDS stores/barriers, 4=loads issue but all OOB): {m:.3f} ms {fl/m/1e9:6.1f} TF")
