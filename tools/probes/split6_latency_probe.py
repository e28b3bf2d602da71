"""Is the split-6 GEMM's DMA path bound by bandwidth or by HBM latency?  The same kernel on 8 / 16 / 64 planes: 8 planes of
operands are 108 MB and stay in the 256 MiB Infinity Cache across launches, 64 planes are 862 MB and come from HBM every time.
Per-plane time with the MFMAs switched off (dbg 4) and as built."""
import ctypes as C, os, sys, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
dev = torch.device("cuda:0")
fast = C.CDLL(os.path.join(HERE, "build", "split6_fast.so"))
fast.split6_gemm_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)
M, K, N = 3872, 512, 512
for P in (8, 16, 32, 64):
    A3 = (torch.randn(3, P, K // 32, M, 32, device=dev) * 0.5).bfloat16()
    B3 = (torch.randn(3, P, K // 32, N, 32, device=dev) * 0.05).bfloat16()
    Co = torch.empty(P, M, N, device=dev)
    mb = (A3.numel() + B3.numel()) * 2 / 1e6
    res = []
    for dbg, what in ((0, "as built"), (4, "no MFMAs"), (1, "no DMA")):
        t = timeit(lambda: fast.split6_gemm_fast(A3.data_ptr(), B3.data_ptr(), Co.data_ptr(), M, K, N, P, dbg, st))
        res.append(f"{what} {t:.0f} us = {t / P:.2f} us/plane")
    print(f"P={P:2d} (operands {mb:.0f} MB, output {Co.numel() * 4 / 1e6:.0f} MB): " + "; ".join(res), flush=True)
