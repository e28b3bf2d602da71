"""VERDICT r5 #8 (probe first): does running an F(6x6,3x3) conv in BATCH CHUNKS - so that a chunk's transformed input V and
products M (64 planes each) stay in the 256 MiB Infinity Cache between input transform -> GEMMs -> output transform - cut the
time the transforms spend on HBM traffic (44.6 ms of the 262 ms fp32 step)?  One conv as ONE op (B = 32) against the same conv
as 2 / 4 / 8 / 16 chunks that reuse one V and one M buffer, per stage and in total, at the three channel pairs of the stack.
    python tools/probes/wino_chunk_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                              # noqa: E402
from mmhand_amd import lib as L, ops                      # noqa: E402

dev = torch.device("cuda:0")
B, H = 32, 64
TPI = (-(-H // 6)) ** 2        # tiles per image (ragged)


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(Cin, Cout, chunk, reps=12):
    g = torch.Generator(device=dev).manual_seed(Cin * 7 + Cout)        # the same operands for every chunk size
    x = torch.randn(B, H, H, Cin, device=dev, generator=g)
    w = torch.randn(3, 3, Cin, Cout, device=dev, generator=g) * 0.05
    ops.bump_weights_epoch()
    U = ops.wino_weights(w, 6, False, False)
    y = torch.empty(B, H, H, Cout, device=dev)
    V = torch.empty(64, chunk * TPI, Cin, device=dev)
    M = torch.empty(64, chunk * TPI, Cout, device=dev)
    st = ops._stream()
    tot = {"in": 0.0, "gemm": 0.0, "out": 0.0}
    wall = 0.0
    for r in range(reps + 2):
        marks = []
        w0, w1 = ev(), ev()
        w0.record()
        for b0 in range(0, B, chunk):
            xs, ys = x[b0:b0 + chunk], y[b0:b0 + chunk]
            e = [ev() for _ in range(4)]
            e[0].record()
            L.call("mmh_wino_input", ops._ptr(xs), chunk, H, H, Cin, 1, 6, L.F32, ops._ptr(V), st)
            e[1].record()
            L.call("mmh_wino_gemm_levels", ops._ptr(V), ops._ptr(U), ops._ptr(M), chunk * TPI, Cin, Cout, 64, 2, st)
            e[2].record()
            L.call("mmh_wino_output", ops._ptr(M), ops._ptr(ys), None, chunk, H, H, Cout, 0, 6, L.F32, None, 0, st)
            e[3].record()
            marks.append(e)
        w1.record()
        torch.cuda.synchronize()
        if r >= 2:
            wall += w0.elapsed_time(w1)
            for e in marks:
                tot["in"] += e[0].elapsed_time(e[1]); tot["gemm"] += e[1].elapsed_time(e[2]); tot["out"] += e[2].elapsed_time(e[3])
    k = 1e3 / reps
    return wall * k, tot["in"] * k, tot["gemm"] * k, tot["out"] * k, y


for Cin, Cout in ((512, 512), (256, 256), (512, 256)):
    ref = None
    print(f"-- {Cin} -> {Cout} @ {H}x{H}, B = {B}: V {64 * B * TPI * Cin * 4 / 2**20:.0f} MiB + M {64 * B * TPI * Cout * 4 / 2**20:.0f} MiB as one op", flush=True)
    for chunk in (32, 16, 8, 4, 2):
        wall, ti, tg, to, y = run(Cin, Cout, chunk)
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(ref, y))
        print(f"   chunks of {chunk:2d} images ({64 * chunk * TPI * (Cin + Cout) * 4 / 2**20:4.0f} MiB of V + M): whole conv {wall:7.1f} us = input transform "
              f"{ti:6.1f} + GEMMs {tg:7.1f} + output transform {to:6.1f}   (events add {wall - ti - tg - to:5.1f})   bit-identical to one op: {same}", flush=True)
