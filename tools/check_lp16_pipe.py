"""conv_lp16 kernel variants against each other (lp16_shape 16 = reference) on ragged and full-size
shapes, both modes, repeated runs (races show up as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0")
L = lib.load()
g = torch.Generator().manual_seed(0)
CASES = ((1, 16, 16, 256, 512, False), (2, 20, 24, 64, 256, True), (1, 16, 16, 64, 512, True), (3, 17, 33, 128, 256, False),
         (4, 32, 32, 128, 512, True), (32, 64, 64, 256, 512, True), (32, 64, 64, 512, 512, True), (32, 64, 64, 512, 256, True),
         (4, 128, 128, 256, 256, True))
for variant in (17, 18):
    for (B, H, W, Cin, Cout, refl) in CASES:
        for mode in (0, 1):
            if mode == 1 and Cin % 256:
                continue
            x = torch.rand(B, H, W, Cin if mode == 0 else Cout, generator=g) * 2 - 1
            w = (torch.rand(3, 3, Cin, Cout, generator=g) * 2 - 1) * 0.1
            ops.bump_weights_epoch()
            xb = ops.lp16_twin(x.to(dev), True)
            wd = w.to(dev)
            L.mmh_set_option(b"lp16_shape", 16)
            ref = ops.raw_conv3x3_lp16(xb, wd, None, refl and mode == 0, 0, True, mode)
            L.mmh_set_option(b"lp16_shape", variant)
            outs = [ops.raw_conv3x3_lp16(xb, wd, None, refl and mode == 0, 0, True, mode) for _ in range(3)]
            d = max((o - ref).abs().max().item() for o in outs)
            print(f"variant {variant} B{B} {H}x{W} {Cin}->{Cout} refl={refl} mode{mode}: max |diff| vs 16 = {d:.3e}" + ("   <-- MISMATCH" if d > 1e-3 else ""), flush=True)
L.mmh_set_option(b"lp16_shape", 19)
