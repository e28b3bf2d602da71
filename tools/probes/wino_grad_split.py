"""Probe: which Winograd pass carries the gradient noise of the F(6x6,3x3) path?

Full-size Generator (ngf 64, 9 PATBlocks, 256x256, B=2, instance norm), parameter gradients against the all-direct run,
with the Winograd kernels enabled for a subset of {fprop, dgrad, wgrad} only.  Also: the all-direct run against itself with
the input scaled by (1 + 2^-22), i.e. what one rounding's worth of input change does to the same gradients.
"""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from bench import synthetic_batch_gpu
from mmhand_amd import ops
from mmhand_amd.networks import Generator

dev = torch.device("cuda:0")
B, S = 2, 256
b = synthetic_batch_gpu(B, S, S, 49, dev)
g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
probe = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(3)).to(dev)
orig = ops._wino_tile
allowed = set()


def gated(B_, H, W_, Cin, Cout, k, stride, pad, bf16, op="fprop"):
    return orig(B_, H, W_, Cin, Cout, k, stride, pad, bf16, op) if op in allowed else 0


ops._wino_tile = gated


def run(which, scale=1.0):
    allowed.clear(); allowed.update(which)
    ops.bump_weights_epoch()
    net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
    net.flatten_parameters()
    out = net([t * scale for t in g_in])
    (out * probe).sum().backward()
    return out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}


rel = lambda a, c: float((a.double() - c.double()).abs().sum() / c.double().abs().sum().clamp_min(1e-30))   # noqa: E731
ref = run(())
for label, which, scale in (("direct, input * (1 + 2^-22)", (), 1.0 + 2.0 ** -22),
                            ("winograd fprop only", ("fprop",), 1.0),
                            ("winograd dgrad only", ("dgrad",), 1.0),
                            ("winograd wgrad only", ("wgrad",), 1.0),
                            ("winograd dgrad + wgrad", ("dgrad", "wgrad"), 1.0),
                            ("winograd everywhere", ("fprop", "dgrad", "wgrad"), 1.0)):
    o, g = run(which, scale)
    errs = sorted(rel(g[n], r) for n, r in ref[1].items() if float(r.abs().sum()) > 0)
    print(f"{label:30s} output {rel(o, ref[0]):.2e}   gradients: median {statistics.median(errs):.2e}  "
          f"p90 {errs[len(errs) * 9 // 10]:.2e}  max {errs[-1]:.2e}", flush=True)
