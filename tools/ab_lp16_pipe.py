"""conv_lp16s_kernel (lp16_shape 16) against the software-pipelined conv_lp16p_kernel (17): results vs each
other and time on the PATBlock shapes."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
L = lib.load()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (Cin, Cout, mode) in ((512, 512, 0), (256, 256, 0), (512, 256, 0), (256, 512, 0), (256, 256, 1), (512, 512, 1)):
    B, H = 32, 64
    x = torch.randn(B, H, H, Cin if mode == 0 else Cout, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    ops.bump_weights_epoch()
    xb = ops.lp16_twin(x, True)
    flop = 2.0 * B * H * H * Cin * Cout * 9
    res = {}
    for shape in (16, 17, 18):
        L.mmh_set_option(b"lp16_shape", shape)
        y = ops.raw_conv3x3_lp16(xb, w, None, mode == 0, 0, True, mode)
        t = timeit(lambda: ops.raw_conv3x3_lp16(xb, w, None, mode == 0, 0, True, mode))
        res[shape] = (y, t)
        print(f"{Cin}->{Cout} mode{mode} shape={shape}: {t*1e3:8.1f} us  {flop/t*1e-9:7.0f} TF", flush=True)
    d = max((res[16][0] - res[17][0]).abs().max().item(), (res[16][0] - res[18][0]).abs().max().item())
    print(f"   max |diff| between the two kernels: {d:.3e} (max |y| {res[16][0].abs().max().item():.2f})")
L.mmh_set_option(b"lp16_shape", 19)
