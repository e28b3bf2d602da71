"""Probe: would the FORWARD / dgrad Winograd-domain GEMMs gain from the TN form the wgrad GEMM runs on (wino_wgrad_dma_kernel:
0.91 of the fp32 MFMA peak against the NN kernel's 0.78)?  The TN kernel computes C[m][n] = sum_k A[k][m] B[k][n] with both
operands k-major; the forward GEMM M = V.U is that with A = V^T [Cin][tiles] and B = U [Cin][Cout] - i.e. if the input transform
wrote V transposed.  Here the EXISTING TN entry point is simply called on the forward problem's shape (contraction 512, M = 3840
= 15 x 256 rows standing in for 3872, N = 512) beside the NN kernel on [3840 x 512].[512 x 512]: timing only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=10, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)
P = 64
for (M, K, N) in ((3840, 512, 512), (3840, 256, 256), (3840, 512, 256), (7680, 256, 256)):
    V = torch.randn(P, M, K, device=dev); U = torch.randn(P, K, N, device=dev) * 0.05
    Mo = torch.empty(P, M, N, device=dev)
    nn2 = lambda: L.call("mmh_wino_gemm_levels", V.data_ptr(), U.data_ptr(), Mo.data_ptr(), M, K, N, P, 2, st)
    nn1 = lambda: L.call("mmh_wino_gemm_levels", V.data_ptr(), U.data_ptr(), Mo.data_ptr(), M, K, N, P, 1, st)
    Vt = V.transpose(1, 2).contiguous()                 # [P][K][M]: "tiles" = K rows of M "input channels"
    ws_b = L.load().mmh_wino_wgrad_gemm_ws_bytes(K, M, N, P)
    ws = torch.empty(max(ws_b, 16) // 4 + 4, device=dev)
    tn = lambda: L.call("mmh_wino_wgrad_gemm", Vt.data_ptr(), U.data_ptr(), K, M, N, P, L.F32, ws.data_ptr(), ws.numel() * 4, Mo.data_ptr(), st)
    flop = 2.0 * P * M * K * N
    try:
        tn(); torch.cuda.synchronize()
        ref = torch.bmm(V[:2], U[:2])
        err = float((Mo[:2] - ref).abs().max() / ref.abs().max())
        t_tn = timeit(tn)
    except RuntimeError as e:
        t_tn, err = float("nan"), str(e)[:60]
    t2, t1 = timeit(nn2), timeit(nn1)
    print(f"[{M} x {K}].[{K} x {N}] x {P}: NN two-level {t2:.0f} us ({flop / t2 / 1e6:.0f} TF)  NN one-level {t1:.0f} us ({flop / t1 / 1e6:.0f} TF)  "
          f"TN (wgrad kernel, one level, split-K slabs if any) {t_tn:.0f} us ({flop / t_tn / 1e6:.0f} TF; max err vs torch {err})", flush=True)
