import os, sys
sys.path.insert(0, "/root/repo")
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
B, H, Cin, Cout = 1, 16, 64, 256
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05; bias = torch.randn(Cout, device=dev)
xb = ops.lp16_twin(x, True)
outs = {}
for dbg in (128, 0):
    lib.check(L.mmh_set_option(b"lp16_dbg", dbg), "s")
    outs[dbg] = ops.raw_conv3x3_lp16(xb, w, bias, True, 0, True, 0, out16=True).float().clone()
lib.check(L.mmh_set_option(b"lp16_dbg", 0), "s")
d = (outs[128] - outs[0]).abs()
print("max diff", float(d.max()), "frac differing", float((d > 0).float().mean()))
bad = (d > 0).nonzero()
print(bad[:20].tolist())
chs = sorted(set(bad[:, 3].tolist())); print("channels differing:", chs[:64], len(chs))
a, b = outs[128][0, 0, 0], outs[0][0, 0, 0]
print("pixel 0,0 old:", [round(v, 3) for v in a[:40].tolist()])
print("pixel 0,0 new:", [round(v, 3) for v in b[:40].tolist()])
