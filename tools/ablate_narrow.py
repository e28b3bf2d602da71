import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
lib.check(L.mmh_set_option(b"conv_tall", 0), "o")
for (H, Cin, Cout, k, s, p, refl) in [(256, 44, 64, 7, 1, 3, True), (256, 24, 64, 7, 1, 3, True), (256, 8, 64, 7, 1, 3, True), (256, 64, 64, 3, 1, 1, False), (256, 64, 128, 3, 2, 1, False)]:
    x = torch.randn(32, H, H, Cin, device=dev); w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    fn = lambda: ops.raw_conv_fprop(x, w, None, s, p, refl, 0)
    y = fn(); fl = 2.0 * y.numel() * Cin * k * k
    dy = torch.randn_like(y)
    fw = lambda: ops.raw_conv_wgrad(x, dy, k, s, p, refl)
    for nm, f, key in (("fprop", fn, b"conv_dbg"), ("wgrad", fw, b"conv_dbg")):
        out = []
        for dbg in (0, 8, 1, 4):
            rc = L.mmh_set_option(key, dbg)
            if rc: out.append("n/a"); continue
            f(); torch.cuda.synchronize()
            m = statistics.median([timeit(f) for _ in range(3)])
            out.append(f"dbg{dbg} {m*1e3:7.1f} us {fl/m/1e9:6.1f} TF")
        L.mmh_set_option(key, 0)
        print(f"{Cin}->{Cout} k{k} s{s} {nm}: " + " | ".join(out), flush=True)
