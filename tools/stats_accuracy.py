"""InstanceNorm statistics of a stem output: per-tile partials from the direct conv's epilogue against the separate
statistics pass, both against float64 statistics of the same fp32 y (stream-3 stem: replicated depth planes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmhand_amd import ops      # noqa: E402
from oracle import mmhand_ref as O      # noqa: E402
dev = torch.device("cuda:0")
b = O.synthetic_batch(4, 64, 64, seed=11)
x = torch.cat((b["D1"], b["D2"]), 1).permute(0, 2, 3, 1).contiguous()          # [B,H,W,6]
x = torch.nn.functional.pad(x, (0, 2)).to(dev)                                  # pad to 8 channels
torch.manual_seed(0)
w = (torch.randn(7, 7, 8, 64) * 0.02).to(dev)
bias = torch.zeros(64, device=dev)
B = x.shape[0]
y = ops.raw_conv_fprop(x, w, bias, 1, 3, True, want_stats=True)
fast = ops.raw_norm_stats_finalize_pending(y, B)
ops._pending_stats.clear()
mean_b, m2_b, rows = ops.raw_norm_stats(y, B)
sc_b, sf_b, is_b = ops.raw_norm_finalize(mean_b, m2_b, rows, None, None, None, None)
yd = y.double().reshape(B, -1, 64)
mean64 = yd.mean(1); is64 = 1.0 / torch.sqrt(yd.var(1, unbiased=False) + ops.EPS)
for name, m, i in (("epilogue partials", fast[0], fast[3]), ("statistics pass", mean_b, is_b)):
    em = float(((m.double() - mean64).abs() / (mean64.abs() + 1e-12)).max())
    ei = float(((i.double() - is64).abs() / is64).max())
    print(f"{name:18s}: max rel err mean {em:.2e}  invstd {ei:.2e}")
print("invstd range", float(is64.min()), float(is64.max()))
