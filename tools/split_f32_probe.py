"""Feasibility probe (torch only, not on the product path): an fp32-accurate GEMM from three bf16 terms per
operand and six bf16 MFMA products, on the Winograd-domain shape of the dominant conv
(64 planes x [3872 x 512] . [512 x 512]).  Accuracy against fp64 beside a plain fp32 GEMM, and the time of
six library bf16 batched GEMMs beside this repo's fp32 MFMA kernel (1.05 ms)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
P, M, K, N = 64, 3872, 512, 512
A = torch.randn(P, M, K, device=dev)
B = torch.randn(P, K, N, device=dev) * 0.05
def split3(x):
    x0 = x.bfloat16(); r = x - x0.float()
    x1 = r.bfloat16(); r = r - x1.float()
    x2 = r.bfloat16()
    return x0, x1, x2
a0, a1, a2 = split3(A); b0, b1, b2 = split3(B)
def mm(x, y):       # bf16 operands, fp32 accumulate and output
    return torch.bmm(x, y, out_dtype=torch.float32) if "out_dtype" in torch.bmm.__doc__ else torch.bmm(x, y).float()
def split_gemm():
    hi = torch.bmm(a0, b0).float()
    lo = (torch.bmm(a0, b1).float() + torch.bmm(a1, b0).float()) + (torch.bmm(a1, b1).float() + torch.bmm(a0, b2).float() + torch.bmm(a2, b0).float())
    return hi + lo
ref = torch.bmm(A[:4].double(), B[:4].double())
f32 = torch.bmm(A[:4], B[:4])
def rel(x): return float((x.double() - ref).abs().sum() / ref.abs().sum())
print("fp32 GEMM (library) rel-L1 vs fp64:", rel(f32))
# bf16 library GEMMs round their OUTPUT to bf16; emulate fp32 accumulation of the six products on 4 planes in fp32 math
def exact_split(i):
    t = lambda x: x[i].float()
    hi = t(a0) @ t(b0)
    lo = t(a0) @ t(b1) + t(a1) @ t(b0) + t(a1) @ t(b1) + t(a0) @ t(b2) + t(a2) @ t(b0)
    return hi + lo
sp = torch.stack([exact_split(i) for i in range(4)])
print("three-term split, six products, fp32 accumulate rel-L1 vs fp64:", rel(sp))
sp1 = torch.stack([(a0[i].float() @ b0[i].float()) for i in range(4)])
print("one bf16 term (plain bf16 operands) rel-L1 vs fp64:", rel(sp1))
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
t1 = timeit(lambda: torch.bmm(a0, b0))
t32 = timeit(lambda: torch.bmm(A, B))
flop = 2.0 * P * M * K * N
print(f"library bf16 bmm: {t1*1e3:.0f} us = {flop/t1*1e-9:.0f} TF;  x6 = {6*t1*1e3:.0f} us = {flop/(6*t1)*1e-9:.0f} TF fp32-equivalent")
print(f"library fp32 bmm: {t32*1e3:.0f} us = {flop/t32*1e-9:.0f} TF;  this repo's wino_gemm_kernel<128,2>: 1054 us = 123 TF")
