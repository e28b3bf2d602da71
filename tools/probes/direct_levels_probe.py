"""Does a more accurate FORWARD bring the gradient-exact hybrid (--fp32_exact_grads: direct fprop, Winograd dgrad / wgrad) closer to
float64 truth at full size?  mmh_set_option("conv_levels", 2) runs the direct fp32 fprop with two-level summation (a fresh MFMA
chain per 32-deep k-step, folded by vector adds).  Against tests/golden/fullsize_grad_sketch.npz (the reference's Generator in
float64): output and per-tensor gradient distance, and the forward + backward time of the Generator, one level against two."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
import bench                                              # noqa: E402
from mmhand_amd import lib, ops                           # noqa: E402
from mmhand_amd.networks import Generator, logical_grads  # noqa: E402

dev = torch.device("cuda:0")
L = lib.load()
fix = np.load(os.path.join(os.path.dirname(bench.__file__), "tests", "golden", "fullsize_grad_sketch.npz"))
b = {k: v.to(dev) for k, v in bench.sketch_inputs(2, 256, 256, 49).items()}
g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
probe = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
for mode in ("bwd", "off", "all"):
    for levels in (1, 2):
        ops.set_winograd_mode(mode, direct_levels=levels)
        net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
        net.flatten_parameters()
        for rep in range(2):
            net.zero_grad()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = net(g_in)
            (out * probe).sum().backward()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        errs, oerr = bench.fp64_sketch_distance(fix, logical_grads(net), out.detach().contiguous())
        v = sorted(errs.values())
        print(f"mode {mode} conv_levels {levels}: output {oerr:.2e}; gradients vs fp64 median {statistics.median(v):.2e} max {v[-1]:.2e}, "
              f"{sum(e > 1e-3 for e in v)} of {len(v)} above 1e-3; Generator fwd+bwd B=2 {ms:.1f} ms", flush=True)
        del net, out
# the all-Winograd path on F(4x4,3x3) (MMH_WINOGRAD_TILE=4: 4x instead of 5.06x fewer multiplications, better-conditioned transforms)
for tile in (6, 4):
    ops.WINOGRAD_TILE = tile
    ops.set_winograd_mode("all", direct_levels=1)
    net = Generator([3, 42, 6], 3, 64, "instance", False, 9).init_weights("normal", 49).to(dev).train()
    net.flatten_parameters()
    for rep in range(2):
        net.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = net(g_in)
        (out * probe).sum().backward()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    errs, oerr = bench.fp64_sketch_distance(fix, logical_grads(net), out.detach().contiguous())
    v = sorted(errs.values())
    print(f"mode all, Winograd tile {tile}: output {oerr:.2e}; gradients vs fp64 median {statistics.median(v):.2e} max {v[-1]:.2e}; "
          f"Generator fwd+bwd B=2 {ms:.1f} ms", flush=True)
    del net, out
ops.WINOGRAD_TILE = 6
lib.check(L.mmh_set_option(b"conv_levels", 1), "opt")
ops.set_winograd_mode("all")
