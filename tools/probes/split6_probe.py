"""VERDICT r4 #4, the accuracy half: the Winograd-domain GEMMs of the dominant conv (F(6x6,3x3), 512 -> 512 @64x64: 64 planes x
[tiles x 512] . [512 x 512]) from SIX bf16 MFMA products per fp32 product (tools/probes/split6_gemm.hip) beside the native fp32
MFMA kernel (mmh_wino_gemm_levels: one summation level, and the product's two), all against fp64 - at the GEMM level (M planes)
and at the conv level (behind the product's own output transform, which amplifies GEMM rounding by up to 32^2).
Operands are REAL transformed tensors: V = mmh_wino_input(x), U = mmh_wino_weights(w), x ~ N(0,1), w ~ N(0, 0.02).

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/split6_gemm.hip -o tools/probes/build/split6_gemm.so
    python tools/probes/split6_probe.py [B]          (default B = 8: 968 tiles; B = 32 is the training shape)"""
import ctypes as C
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from mmhand_amd import lib as L, ops
dev = torch.device("cuda:0")
so = C.CDLL(os.path.join(HERE, "build", "split6_gemm.so"))
so.split6_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = 64; Cin = Cout = 512; P = 64
torch.manual_seed(0)
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.02
tiles = B * (-(-H // 6)) ** 2
Mpad = (tiles + 15) // 16 * 16
V = torch.zeros((P, Mpad, Cin), device=dev)
Vt = torch.empty((P, tiles, Cin), device=dev)
L.call("mmh_wino_input", x.data_ptr(), B, H, H, Cin, 1, 6, L.F32, Vt.data_ptr(), None)
V[:, :tiles] = Vt
U = ops.wino_weights(w, 6, False, False)
torch.cuda.synchronize()
ref = torch.bmm(V.double(), U.double())                                # fp64 planes
def rel(a, r): return float((a.double() - r).abs().sum() / r.abs().sum())
def relmax(a, r): return float((a.double() - r).abs().max() / r.abs().max())
st = torch.cuda.current_stream().cuda_stream
def split(variant):
    Mo = torch.empty((P, Mpad, Cout), device=dev)
    rc = so.split6_gemm(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Mpad, Cin, Cout, P, variant, st)
    assert rc == 0, rc
    return Mo
def native(levels):
    Mo = torch.empty((P, Mpad, Cout), device=dev)
    L.call("mmh_wino_gemm_levels", V.data_ptr(), U.data_ptr(), Mo.data_ptr(), Mpad, Cin, Cout, P, levels, st)
    return Mo
# conv level: the product's output transform on each set of planes, against the fp64 convolution
import torch.nn.functional as F
xr = F.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
yref = F.conv2d(xr, w.double().permute(3, 2, 0, 1)).permute(0, 2, 3, 1).contiguous()
def conv_of(Mo):
    y = torch.empty((B, H, H, Cout), device=dev)
    Mt = Mo[:, :tiles].contiguous()
    L.call("mmh_wino_output", Mt.data_ptr(), y.data_ptr(), None, B, H, H, Cout, 0, 6, L.F32, None, 0, st)
    return y
rows = [("native fp32 MFMA, one level (mmh_wino_gemm_levels 1)", native(1)),
        ("native fp32 MFMA, two levels (the product's forward GEMM)", native(2)),
        ("fp32 fmaf chain in k order (probe kernel; = one level)", split(10)),
        ("fp32 two-level chain, 32-deep (probe kernel)", split(11)),
        ("SIX bf16 products, one accumulator, small terms first", split(0)),
        ("SIX bf16 products, one accumulator, large term first", split(4)),
        ("SIX bf16 products, hi / lo accumulators", split(1)),
        ("SIX bf16 products, hi / lo, folded into totals every 128 k", split(5)),
        ("SIX bf16 products, ONE set: chain from 0 per k32, fp32 fold", split(6)),
        ("SIX bf16 products, ONE set: chain from 0 per k64, fp32 fold", split(7)),
        ("three bf16 products (a0b0 + a0b1 + a1b0)", split(2)),
        ("one bf16 product (plain bf16 operands)", split(3))]
print(f"B={B}: {P} planes x [{tiles} x {Cin}] . [{Cin} x {Cout}], real F(6x6,3x3) operands; relative L1 / max-norm error against fp64")
print(f"{'':62s} {'GEMM L1':>10s} {'GEMM max':>10s} {'conv L1':>10s} {'conv max':>10s}")
for name, Mo in rows:
    y = conv_of(Mo)
    print(f"{name:62s} {rel(Mo, ref):10.3e} {relmax(Mo, ref):10.3e} {rel(y, yref):10.3e} {relmax(y, yref):10.3e}", flush=True)
# the direct fp32 kernel's distance, for scale
yd = ops.raw_conv_fprop(x, w, None, 1, 1, True, 0) if hasattr(ops, "raw_conv_fprop") else None
if yd is not None:
    ops.USE_WINOGRAD, old = False, ops.USE_WINOGRAD
    ops.bump_weights_epoch()
    yd = ops.raw_conv_fprop(x, w, None, 1, 1, True, 0)
    ops.USE_WINOGRAD = old
    print(f"{'direct fp32 implicit GEMM (no Winograd)':62s} {'':>10s} {'':>10s} {rel(yd, yref):10.3e} {relmax(yd, yref):10.3e}")
