"""VERDICT r4 #4, second build: tools/probes/split6_fast2.hip (256 x 256 block, 16-deep stages, three of them, MFMA 32x32x16)
against the first build (split6_fast.hip: 256 x 128, two 32-deep stages) and the product's native fp32 MFMA GEMM
(mmh_wino_gemm_levels, two levels) on the Winograd-domain problems of the stack: 64 planes x [3872 x K] . [K x N].

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/split6_fast.hip -o tools/probes/build/split6_fast.so
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/split6_fast2.hip -o tools/probes/build/split6_fast2.so
    python tools/probes/split6_fast2_probe.py [B]"""
import ctypes as C
import os
import statistics
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
f1 = C.CDLL(os.path.join(HERE, "build", "split6_fast.so"))
f1.split3.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
f1.split6_gemm_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
f2 = C.CDLL(os.path.join(HERE, "build", "split6_fast2.so"))
f2.split6_gemm_fast2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=10, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


for (Cin, Cout) in ((512, 512), (256, 256), (512, 256), (256, 512)):
    H = 64; P = 64
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.02
    tiles = B * (-(-H // 6)) ** 2
    V = torch.empty((P, tiles, Cin), device=dev)
    L.call("mmh_wino_input", x.data_ptr(), B, H, H, Cin, 1, 6, L.F32, V.data_ptr(), None)
    U = ops.wino_weights(w, 6, False, False)                       # [P][K][N]
    Ut = U.transpose(1, 2).contiguous()                            # [P][N][K]
    A3p = torch.empty((3, P, tiles, Cin), dtype=torch.bfloat16, device=dev)
    B3p = torch.empty((3, P, Cout, Cin), dtype=torch.bfloat16, device=dev)
    assert f1.split3(Ut.data_ptr(), B3p.data_ptr(), Ut.numel(), st) == 0
    assert f1.split3(V.data_ptr(), A3p.data_ptr(), V.numel(), st) == 0
    kb = lambda t, rows, k: t.view(3, P, rows, Cin // k, k).permute(0, 1, 3, 2, 4).contiguous()
    A16, B16, A32, B32 = kb(A3p, tiles, 16), kb(B3p, Cout, 16), kb(A3p, tiles, 32), kb(B3p, Cout, 32)
    M2 = torch.zeros((P, tiles, Cout), device=dev); M1 = torch.zeros_like(M2); Mn = torch.empty_like(M2)

    def g2(dbg=0):
        rc = f2.split6_gemm_fast2(A16.data_ptr(), B16.data_ptr(), M2.data_ptr(), tiles, Cin, Cout, P, dbg, st)
        assert rc == 0, rc

    def g1():
        rc = f1.split6_gemm_fast(A32.data_ptr(), B32.data_ptr(), M1.data_ptr(), tiles, Cin, Cout, P, 0, st)
        assert rc == 0, rc
    native = lambda: L.call("mmh_wino_gemm_levels", V.data_ptr(), U.data_ptr(), Mn.data_ptr(), tiles, Cin, Cout, P, 2, st)
    g2(); g1(); native(); torch.cuda.synchronize()
    sel = [0, 21, 42, 63]
    ref = torch.bmm(V[sel].double(), U[sel].double())
    rel = lambda a: float((a.double() - ref).abs().sum() / ref.abs().sum())
    print(f"{Cin}->{Cout}: relative L1 against fp64 (4 planes): second build {rel(M2[sel]):.3e}   first build {rel(M1[sel]):.3e}   "
          f"native two-level {rel(Mn[sel]):.3e};  max |second - first| / max|ref| = {float((M2 - M1).abs().max() / ref.abs().max()):.2e}")
    runs = []
    for _ in range(3):
        M2.zero_(); g2(); torch.cuda.synchronize(); runs.append(M2.clone())
    print(f"    reproducible run to run: {all(torch.equal(runs[0], r) for r in runs[1:])}")
    flop = 2.0 * P * tiles * Cin * Cout
    t2, t1, tn = timeit(g2), timeit(g1), timeit(native)
    print(f"    second build {t2:.0f} us = {flop / t2 / 1e6:.0f} TF fp32-equivalent ({6 * flop / t2 / 1e6:.0f} TF of bf16 MFMA)   first build {t1:.0f} us   "
          f"native two-level {tn:.0f} us = {flop / tn / 1e6:.0f} TF")
    print(f"    speed-up over the native GEMM: {tn / t2:.2f}x (first build {tn / t1:.2f}x)", flush=True)
    for dbg, what in ((1, "no DMA after the prologue"), (2, "no fragment reads"), (4, "no MFMAs"), (8, "no stores"), (11, "MFMAs + loop only")):
        print(f"        [{what}]: {timeit(lambda: g2(dbg)):.0f} us", flush=True)
