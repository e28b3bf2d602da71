"""fp32 Winograd-domain wgrad GEMMs at the step's three channel pairs (64 planes, 3872 tiles): the register-staged kernel
against the LDS-DMA ring kernel (wino_wgrad_dma.hip)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
P, T = 64, 3872
for Cin, Cout in ((512, 512), (512, 256), (256, 512), (256, 256)):
    V = torch.randn(P, T, Cin, device=dev); Y = torch.randn(P, T, Cout, device=dev); dU = torch.empty(P, Cin, Cout, device=dev)
    res, outs = {}, {}
    for v in (0, 1, 0, 1):
        lib.call("mmh_set_option", b"wino_wgrad_dma", v)
        nws = L.mmh_wino_wgrad_gemm_ws_bytes(T, Cin, Cout, P)
        ws = torch.empty(nws // 4 + 4, device=dev)
        fn = lambda: lib.call("mmh_wino_wgrad_gemm", V.data_ptr(), Y.data_ptr(), T, Cin, Cout, P, lib.F32, ws.data_ptr(), nws, dU.data_ptr(), st())
        res[v] = timeit(fn); outs[v] = dU.clone()
    rel = float((outs[1].double() - outs[0].double()).abs().sum() / outs[0].double().abs().sum())
    fl = 2.0 * P * T * Cin * Cout
    print(f"[{Cin}x{T}].[{T}x{Cout}] x64: register-staged {res[0] * 1e3:.0f} us = {fl / res[0] / 1e9:.1f} TF ({fl / res[0] / 1e9 / 157.3:.2f}) | "
          f"DMA ring {res[1] * 1e3:.0f} us = {fl / res[1] / 1e9:.1f} TF ({fl / res[1] / 1e9 / 157.3:.2f}) | rel diff {rel:.1e}", flush=True)
lib.call("mmh_set_option", b"wino_wgrad_dma", 1)
