"""Can an HBM-bound row kernel run UNDER an MFMA/LDS-bound conv kernel on a second stream?  The nine-tap wgrad (one workgroup per
CU, 128 KiB of LDS, two 200-register waves per SIMD) on stream A, mmh_norm_bwd_apply / mmh_scale_shift_act on stream B:
serial time against concurrent time for N pairs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import lib, ops
dev = torch.device("cuda:0")
B, H, C = 32, 64, 512
x16 = torch.randn(B, H, H, C, device=dev).bfloat16()
dy16 = torch.randn(B, H, H, C, device=dev).bfloat16()
g16 = torch.randn(B, H, H, C, device=dev).bfloat16()
xn = torch.randn(B, H, H, C, device=dev).bfloat16()
bits = torch.randint(-32768, 32767, (B * H * H * C // 8,), device=dev, dtype=torch.int16)
mean = torch.zeros(B, C, device=dev); invstd = torch.ones(B, C, device=dev)
s1 = torch.zeros(B, C, device=dev); s2 = torch.zeros(B, C, device=dev)
dx = torch.empty_like(xn)
d = ops.conv_desc(B, H, H, C, C, 3, 1, 1, True); d.dtype = lib.BF16
import ctypes as Ct
nws = lib.load().mmh_wgrad3x3_lp16_ws_bytes(Ct.byref(d))
ws = torch.empty(nws // 4 + 4, device=dev)
dw = torch.zeros(3, 3, C, C, device=dev)
zp = ops.zero_page(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def wgrad(st):
    lib.call("mmh_wgrad3x3_lp16", Ct.byref(d), x16.data_ptr(), dy16.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, 0,
             zp.data_ptr(), st.cuda_stream)


def apply(st):
    lib.call("mmh_norm_bwd_apply", g16.data_ptr(), bits.data_ptr(), xn.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None,
             s1.data_ptr(), s2.data_ptr(), float(H * H), B, H * H, C, 2, 0.5, dx.data_ptr(), lib.BF16, lib.BF16, lib.BF16, st.cuda_stream)


def run(concurrent, n=8, k_apply=1):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    sa.wait_stream(cur); sb.wait_stream(cur)
    for _ in range(n):
        wgrad(sa)
        for _ in range(k_apply):
            apply(sb if concurrent else sa)
    cur.wait_stream(sa); cur.wait_stream(sb)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for k in (1, 3, 6):
    for _ in range(2):
        run(False, k_apply=k); run(True, k_apply=k)
    ser = statistics.median(run(False, k_apply=k) for _ in range(5))
    con = statistics.median(run(True, k_apply=k) for _ in range(5))
    print(f"per wgrad + {k} x norm_bwd_apply: serial {ser:.0f} us, two streams {con:.0f} us", flush=True)
