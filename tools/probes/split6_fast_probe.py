"""VERDICT r4 #4, the speed half: tools/probes/split6_fast.hip (the six-bf16-product fp32 GEMM as a real LDS-DMA / MFMA kernel)
against the product's native fp32 MFMA GEMM (mmh_wino_gemm_levels, two levels) on the Winograd-domain problem of the dominant
conv: 64 planes x [3872 x 512] . [512 x 512] (B = 32, 64x64, 512 -> 512).  Reports: bit-identity with the accuracy probe's
variant 6 (tools/probes/split6_gemm.hip), error against fp64 on 4 planes, and microseconds of (a) the operand split pass
(fp32 V -> three bf16 term planes), (b) the split GEMM, (c) the native GEMM - and the timing-only ablations of (b).

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/split6_fast.hip -o tools/probes/build/split6_fast.so
    python tools/probes/split6_fast_probe.py [B]"""
import ctypes as C
import os
import statistics
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
fast = C.CDLL(os.path.join(HERE, "build", "split6_fast.so"))
fast.split3.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
fast.split6_gemm_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
slow = C.CDLL(os.path.join(HERE, "build", "split6_gemm.so"))
slow.split6_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=10, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)
for (Cin, Cout) in ((512, 512), (256, 256), (512, 256)):
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
    assert fast.split3(Ut.data_ptr(), B3p.data_ptr(), Ut.numel(), st) == 0
    split_a = lambda: fast.split3(V.data_ptr(), A3p.data_ptr(), V.numel(), st)
    assert split_a() == 0
    # the split is exact to 2^-24 or so: x == t0 + t1 + t2 in fp32 arithmetic
    rec = A3p[0].float() + (A3p[1].float() + A3p[2].float())
    print(f"{Cin}->{Cout}: split3 reconstruction max |x - (t0+t1+t2)| / max|x| = {float((rec - V).abs().max() / V.abs().max()):.2e}")
    # k-blocked panels [3][P][K/32][rows][32] (what a producing transform would write directly)
    A3 = A3p.view(3, P, tiles, Cin // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
    B3 = B3p.view(3, P, Cout, Cin // 32, 32).permute(0, 1, 3, 2, 4).contiguous()
    del rec
    Mo = torch.zeros((P, tiles, Cout), device=dev)
    def gemm(dbg=0):
        rc = fast.split6_gemm_fast(A3.data_ptr(), B3.data_ptr(), Mo.data_ptr(), tiles, Cin, Cout, P, dbg, st)
        assert rc == 0, rc
    gemm(); torch.cuda.synchronize()
    Mn = torch.empty((P, tiles, Cout), device=dev)
    native = lambda lv=2: L.call("mmh_wino_gemm_levels", V.data_ptr(), U.data_ptr(), Mn.data_ptr(), tiles, Cin, Cout, P, lv, st)
    native(); torch.cuda.synchronize()
    # accuracy on 4 planes against fp64, and bit-identity with the accuracy probe's variant 6 (needs M % 16 == 0)
    sel = [0, 21, 42, 63]
    ref = torch.bmm(V[sel].double(), U[sel].double())
    rel = lambda a: float((a.double() - ref).abs().sum() / ref.abs().sum())
    print(f"    relative L1 against fp64 (4 planes): split6 kernel {rel(Mo[sel]):.3e}   native two-level {rel(Mn[sel]):.3e}")
    if tiles % 16 == 0:
        Ms = torch.empty((4, tiles, Cout), device=dev)
        Vs, Us = V[sel].contiguous(), U[sel].contiguous()
        assert slow.split6_gemm(Vs.data_ptr(), Us.data_ptr(), Ms.data_ptr(), tiles, Cin, Cout, 4, 6, st) == 0
        torch.cuda.synchronize()
        print(f"    bit-identical to the accuracy probe's variant 6: {torch.equal(Ms, Mo[sel])}  (max diff {float((Ms - Mo[sel]).abs().max()):.3e})")
    runs = []
    for _ in range(3):
        Mo.zero_(); gemm(); torch.cuda.synchronize(); runs.append(Mo.clone())
    print(f"    reproducible run to run: {all(torch.equal(runs[0], r) for r in runs[1:])}")
    flop = 2.0 * P * tiles * Cin * Cout
    t_split, t_gemm, t_nat, t_nat1 = timeit(split_a), timeit(gemm), timeit(native), timeit(lambda: native(1))
    print(f"    split pass (A) {t_split:.0f} us ({V.numel() * 10 / t_split / 1e6:.2f} TB/s)   split6 GEMM {t_gemm:.0f} us = {flop / t_gemm / 1e6:.0f} TF fp32-equivalent "
          f"({6 * flop / t_gemm / 1e6:.0f} TF of bf16 MFMA)   native two-level {t_nat:.0f} us = {flop / t_nat / 1e6:.0f} TF   one-level {t_nat1:.0f} us")
    print(f"    speed-up over the native two-level GEMM: GEMM alone {t_nat / t_gemm:.2f}x, with the standalone split pass {t_nat / (t_gemm + t_split):.2f}x", flush=True)
    for dbg, what in ((16, "every wave issues its DMA right behind the barrier"), (1, "no DMA after the prologue"), (2, "no fragment reads"), (4, "no MFMAs"), (8, "no stores"), (3, "no DMA, no fragment reads"),
                      (11, "MFMAs + loop only")):
        print(f"        [{what}]: {timeit(lambda: gemm(dbg)):.0f} us", flush=True)
