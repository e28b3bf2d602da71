"""A/B 1 vs 2 channels per thread in the F(6x6,3x3) transforms (knob wino6_vec bits) per stage."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib
dev = torch.device("cuda:0"); B = 32; t = 6
L = lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (H, C) in [(64, 512), (64, 256)]:
    x = torch.randn(B, H, H, C, device=dev); P = 64; tiles = B * (-(-H // t)) ** 2
    V = torch.empty(P, tiles, C, device=dev); y = torch.empty(B, H, H, C, device=dev)
    for v in (0, 7):
        lib.check(L.mmh_set_option(b"wino6_vec", v), "set")
        ti = timeit(lambda: lib.call("mmh_wino_input", x.data_ptr(), B, H, H, C, 1, t, lib.F32, V.data_ptr(), st()))
        to = timeit(lambda: lib.call("mmh_wino_output", V.data_ptr(), y.data_ptr(), None, B, H, H, C, 0, t, lib.F32, None, 0, st()))
        td = timeit(lambda: lib.call("mmh_wino_dy", x.data_ptr(), B, H, H, C, t, lib.F32, V.data_ptr(), st()))
        print(f"C={C} channels/thread={'2' if v else '1'}: input {ti*1e3:.0f} us | output {to*1e3:.0f} us | dy {td*1e3:.0f} us", flush=True)
