import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib
L = lib.load(); dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
def setopt(k, v): lib.check(L.mmh_set_option(k.encode(), v), "set")
def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for K, N in ((512, 512), (256, 256)):
    P, tiles = 64, 32 * 121
    V = torch.randn(P, tiles, K, device=dev); U = torch.randn(P, K, N, device=dev); M = torch.empty(P, tiles, N, device=dev)
    row = []
    for occ in (3, 2, 1, 3):
        for lv in (2, 1):
            setopt("wino_gemm_occ", occ); setopt("wino_gemm_levels", lv)
            t = timeit(lambda: lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, K, N, P, lib.F32, st()))
            row.append(f"occ{occ} lv{lv}: {t*1e3:.0f} us ({P*2.0*tiles*K*N/t/1e9:.0f} TF)")
    print(f"[{tiles}x{K}].[{K}x{N}] x64: " + " | ".join(row), flush=True)
