"""HBM-bound kernels at the shapes of the step: time and achieved GB/s (algorithmic bytes), first
generation (pw_v2=0) against second generation (pw_v2=1).

    python tools/bench_pointwise.py [--lp 1]      # lp: 0 fp32 tensors, 1 bf16 conv-facing tensors
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib as L, ops          # noqa: E402

SHAPES = [(32, 64, 64, 256), (32, 64, 64, 512), (32, 128, 128, 128), (32, 256, 256, 64)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lp", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    td = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[a.lp]
    es = 4 if a.lp == 0 else 2
    lib = L.load()
    print("%-22s %-18s %10s %10s %8s %8s" % ("kernel", "shape", "v1 us", "v2 us", "v1 GB/s", "v2 GB/s"))
    for shp in SHAPES:
        B, H, W, C = shp
        n = B * H * W * C
        x = torch.randn(shp, device=dev).to(td)
        g = torch.randn(shp, device=dev).to(td)
        groups, rows = B, H * W
        mean, m2, _ = ops.raw_norm_stats(x, groups)
        scale, shift, invstd = ops.raw_norm_finalize(mean, m2, rows, None, None, None, None)
        _, kb = ops.raw_scale_shift_act(x, scale, shift, None, True, 0.5, 7, None, keep_bits=True, out_lp=a.lp)
        s1 = torch.empty((groups, C), device=dev); s2 = torch.empty((groups, C), device=dev)
        ws = torch.empty(lib.mmh_norm_bwd_ws_bytes(groups, rows, C) // 4 + 4, device=dev)
        dx = torch.empty_like(x)
        st = ops._stream
        P = ops._ptr
        cases = [
            ("norm_stats", lambda: ops.raw_norm_stats(x, groups), n * es),
            ("scale_shift_act", lambda: ops.raw_scale_shift_act(x, scale, shift, None, True, 0.0, 0, None, keep_bits=True,
                                                                out_lp=a.lp), n * es * 2 + n // 8),
            ("scale_shift_act+drop", lambda: ops.raw_scale_shift_act(x, scale, shift, None, True, 0.5, 7, None, keep_bits=True,
                                                                     out_lp=a.lp), n * es * 2 + n // 8),
            ("norm_bwd_reduce", lambda: L.call("mmh_norm_bwd_reduce", P(g), P(kb), P(x), P(mean), P(invstd), groups, rows, C,
                                               2, 0.5, P(s1), P(s2), P(ws), ws.numel() * 4, ops._tdt(g), ops._tdt(x), st()),
             n * es * 2 + n // 8),
            ("norm_bwd_apply", lambda: L.call("mmh_norm_bwd_apply", P(g), P(kb), P(x), P(mean), P(invstd), None, P(s1), P(s2),
                                              float(rows), groups, rows, C, 2, 0.5, P(dx), ops._tdt(g), ops._tdt(x),
                                              ops._tdt(dx), st()), n * es * 3 + n // 8),
            ("colsum", lambda: ops.raw_colsum(B * H * W, C, g), n * es),
        ]
        if C == 256:
            x1 = torch.randn(shp, device=dev); s1g = torch.randn(shp, device=dev)
            cases.append(("gate_fwd", lambda: ops.GateFn.forward(_Ctx(), x1, s1g, None, None, True, a.lp, x, g)
                          if a.lp else ops.GateFn.forward(_Ctx(), x1, s1g, x, g, True, 0), n * (8 + 2 * es + 4 + 4 * es)))
        for name, fn, nbytes in cases:
            t = []
            for v in (0, 1):
                lib.mmh_set_option(b"pw_v2", v)
                t.append(timeit(fn))
            lib.mmh_set_option(b"pw_v2", 1)
            print("%-22s %-18s %10.1f %10.1f %8.0f %8.0f" % (name, "x".join(map(str, shp)), t[0], t[1],
                                                             nbytes / t[0] * 1e-3, nbytes / t[1] * 1e-3))


class _Ctx:
    def save_for_backward(self, *a):
        pass

    def mark_non_differentiable(self, *a):
        pass


if __name__ == "__main__":
    main()
