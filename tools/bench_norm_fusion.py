"""Input transforms of the F(6x6,3x3) stack with and without the norm arithmetic inside (us, B=32 @64x64).

    python tools/bench_norm_fusion.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import lib as L, ops      # noqa: E402

dev = torch.device("cuda:0")
P, st = ops._ptr, ops._stream


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
    return e0.elapsed_time(e1) / n * 1e3


for C in (256, 512):
    B, H, W = 32, 64, 64
    shape = (B, H, W, C)
    x = torch.randn(shape, device=dev); g = torch.randn(shape, device=dev)
    groups, rows = B, H * W
    mean, m2, _ = ops.raw_norm_stats(x, groups)
    scale, shift, invstd = ops.raw_norm_finalize(mean, m2, rows, None, None, None, None)
    dbits, drows = ops.raw_dropout_bits(shape, 0.5, 7, None, dev, rows=True)
    tiles = B * 11 * 11
    V = torch.empty((64, tiles, C), device=dev); Y = torch.empty_like(V)
    s1 = torch.randn((groups, C), device=dev); s2 = torch.randn((groups, C), device=dev)
    out = torch.empty_like(x)
    kb = ops.raw_scale_shift_act(x, scale, shift, None, True, 0.5, 7, None, keep_bits=True)[1]
    t = {}
    t["dropout_bits(+rows)"] = timeit(lambda: ops.raw_dropout_bits(shape, 0.5, 7, None, dev, rows=True))
    t["scale_shift_act"] = timeit(lambda: ops.raw_scale_shift_act(x, scale, shift, None, True, 0.5, 7, None, keep_bits=True))
    t["wino_input"] = timeit(lambda: L.call("mmh_wino_input", P(x), B, H, W, C, 1, 6, L.F32, P(V), st()))
    t["wino_input_normact(relu)"] = timeit(lambda: L.call("mmh_wino_input_normact", P(x), B, H, W, C, 1, P(V), P(scale), P(shift), groups, 1, 0.0, None, st()))
    t["wino_input_normact(relu+drop)"] = timeit(lambda: L.call("mmh_wino_input_normact", P(x), B, H, W, C, 1, P(V), P(scale), P(shift), groups, 1, 0.5, P(drows), st()))
    t["norm_bwd_apply"] = timeit(lambda: L.call("mmh_norm_bwd_apply", P(g), P(kb), P(x), P(mean), P(invstd), None, P(s1), P(s2), float(rows), groups, rows, C, 2, 0.5, P(out), L.F32, L.F32, L.F32, st()))
    t["norm_bwd_apply_rc"] = timeit(lambda: L.call("mmh_norm_bwd_apply_rc", P(g), P(x), P(mean), P(invstd), None, P(s1), P(s2), P(scale), P(shift), P(dbits), float(rows), groups, rows, C, 1, 0.5, P(out), st()))
    t["wino_input_dy"] = timeit(lambda: L.call("mmh_wino_input_dy", P(g), B, H, W, C, 6, L.F32, P(V), P(Y), 1, st()))
    t["wino_input_dy_normbwd(relu)"] = timeit(lambda: L.call("mmh_wino_input_dy_normbwd", P(g), P(x), B, H, W, C, P(V), P(Y), 1, P(mean), P(invstd), None, P(s1), P(s2), float(rows), P(scale), P(shift), None, groups, 1, 0.0, st()))
    t["wino_input_dy_normbwd(relu+drop)"] = timeit(lambda: L.call("mmh_wino_input_dy_normbwd", P(g), P(x), B, H, W, C, P(V), P(Y), 1, P(mean), P(invstd), None, P(s1), P(s2), float(rows), P(scale), P(shift), P(drows), groups, 1, 0.5, st()))
    print(f"C={C}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()), flush=True)
