"""Winograd-domain GEMM variants (one process): accumulation levels x column-tile width.
Speed: the 64 x [3872x512].[512x512] launch of the 512->512 fprop at B=32 (HIP events).
Accuracy: the whole F(6x6,3x3) conv (B=1, 36x36, 512->512) against the fp64 CPU oracle."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from mmhand_amd import ops, lib
from oracle import ops_ref as R
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
variants = [(1, 0), (2, 0), (1, 64), (2, 64)]
g = torch.Generator().manual_seed(0)
xs = (torch.rand(1, 36, 36, 512, generator=g) * 2 - 1)
ws = (torch.rand(3, 3, 512, 512, generator=g) * 2 - 1) * 0.05
ref = R.conv2d(xs, ws, None, 1, 1, True)
for K, N in ((512, 512), (256, 256), (512, 256)):
    B, H = 32, 64
    P, tiles = 64, B * 11 * 11
    V = torch.randn(P, tiles, K, device=dev); U = torch.randn(P, K, N, device=dev); M = torch.empty(P, tiles, N, device=dev)
    row = []
    for lv, bn in variants:
        setopt("wino_gemm_levels", lv); setopt("wino_gemm_bn", bn)
        t = timeit(lambda: lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, K, N, P, lib.F32, st()))
        row.append(f"levels={lv} bn={bn or 'auto'}: {t*1e3:.0f} us ({P*2.0*tiles*K*N/t/1e9:.0f} TF)")
    print(f"[{tiles}x{K}].[{K}x{N}] x64: " + " | ".join(row), flush=True)
for lv, bn in variants:
    setopt("wino_gemm_levels", lv); setopt("wino_gemm_bn", bn)
    ops.bump_weights_epoch()
    y = ops.raw_conv_fprop_wino(xs.to(dev), ws.to(dev), None, True, 0, 6)
    print(f"levels={lv} bn={bn or 'auto'}: F(6x6,3x3) 512->512 rel-L1 vs fp64 {R.rel_l1(y, ref):.2e}", flush=True)
setopt("wino_gemm_levels", 1); setopt("wino_gemm_bn", 0)
ops.USE_WINOGRAD = False
y = ops.raw_conv_fprop(xs.to(dev), ws.to(dev), None, 1, 1, True)
print(f"direct kernel: rel-L1 vs fp64 {R.rel_l1(y, ref):.2e}")
