"""fp32 stride-2 dgrad 64 -> 128 @256x256 (B=32): halo-resident kernel (dgrad_s2.hip) against the four parity-class GEMMs,
and the ConvTranspose2d fprop 128 -> 64 @128x128 that shares the arithmetic."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 32))
def timeit(fn, iters=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for H, Cin, Cout in ((256, 64, 128), (128, 128, 256)):
  w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
  dy = torch.randn(B, H // 2, H // 2, Cout, device=dev)
  bias = torch.randn(Cin, device=dev)
  flop = 2.0 * B * (H // 2) ** 2 * Cout * Cin * 9
  fns = {f"s2 dgrad {Cin}->{Cout} @{H}": lambda: ops.raw_conv_dgrad(dy, w, (B, H, H, Cin), 2, 1, False),
         f"convT fprop {Cout}->{Cin} @{H // 2}": lambda: ops.raw_convT_fprop(dy, w, bias)}
  for name, fn in fns.items():
      res, outs = {0: [], 1: []}, {}
      for v in (0, 1):
          lib.call("mmh_set_option", b"dgrad_s2_halo", v); outs[v] = fn().clone(); torch.cuda.synchronize()
      rel = float((outs[1].double() - outs[0].double()).abs().sum() / outs[0].double().abs().sum())
      for _ in range(5):
          for v in (0, 1):
              lib.call("mmh_set_option", b"dgrad_s2_halo", v); res[v].append(timeit(fn))
      m = {v: statistics.median(res[v]) for v in res}
      print(f"{name}: parity-class GEMMs {m[0] * 1e3:.0f} us = {flop / m[0] / 1e9:.1f} TF ({flop / m[0] / 1e9 / 157.3:.2f}) | "
            f"halo-resident {m[1] * 1e3:.0f} us = {flop / m[1] / 1e9:.1f} TF ({flop / m[1] / 1e9 / 157.3:.2f}) | rel diff {rel:.1e}", flush=True)
lib.call("mmh_set_option", b"dgrad_s2_halo", 1)
