"""Probe: the F(6x6,3x3) filter transform and its back-transform per launch at the step's three channel pairs."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mmhand_amd import ops, lib as L
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return statistics.median(ts)
for cin, cout in ((256, 256), (512, 256), (512, 512)):
    w = torch.randn(3, 3, cin, cout, device=dev); U = torch.empty(64, cin, cout, device=dev); dw = torch.empty_like(w)
    for flip in (0, 1):
        us = t(lambda: L.call("mmh_wino_weights", ops._ptr(w), cin, cout, flip, 6, L.F32, ops._ptr(U), ops._stream()))
        print(f"wino_weights {cin}x{cout} flip {flip}: {us:.1f} us ({64 * cin * cout * 4 / us / 1e6:.2f} TB/s written)")
    us = t(lambda: L.call("mmh_wino_dw", ops._ptr(U), cin, cout, 6, ops._ptr(dw), 0, ops._stream()))
    print(f"wino_dw      {cin}x{cout}: {us:.1f} us ({64 * cin * cout * 4 / us / 1e6:.2f} TB/s read)")
