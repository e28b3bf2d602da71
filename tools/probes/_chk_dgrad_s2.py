import sys; sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B, H = 4, 64
w = torch.randn(3, 3, 64, 128, device=dev) * 0.05
dy = torch.randn(B, H // 2, H // 2, 128, device=dev)
outs = {}
for v in (1, 0):
    lib.call("mmh_set_option", b"dgrad_s2_halo", v)
    t = torch.full((B, H, H, 64), float("nan"), device=dev); del t
    outs[v] = ops.raw_conv_dgrad(dy, w, (B, H, H, 64), 2, 1, False).clone(); torch.cuda.synchronize()
    print(v, "nan count", int(torch.isnan(outs[v]).sum()))
d = (outs[1].double() - outs[0].double()).abs()
print("rel", float(d.sum() / outs[0].double().abs().sum()), "max", float(d.max()), "equal", torch.equal(outs[0], outs[1]))
