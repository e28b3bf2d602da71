"""How far the Winograd paths move the Generator's parameter gradients from the direct kernels' at the
real size (ngf 64, 9 PATBlocks, 256x256): same weights, inputs and loss probe, per-tensor relative L1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_batch_gpu
from mmhand_amd import ops
from mmhand_amd.networks import Generator
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
norm = sys.argv[2] if len(sys.argv) > 2 else "instance"
b = synthetic_batch_gpu(B, 256, 256, 49, dev)
g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
probe = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
res = {}
for tile in (0, 4, 6):
    ops.USE_WINOGRAD = tile > 0; ops.WINOGRAD_TILE = tile or 6
    net = Generator([3, 42, 6], 3, 64, norm, False, 9).init_weights("normal", 49).to(dev).train()
    net.flatten_parameters()
    out = net(g_in)
    (out * probe).sum().backward()
    res[tile] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
def rel(a, b): return float((a.double() - b.double()).abs().sum() / b.double().abs().sum().clamp_min(1e-30))
for tile in (4, 6):
    errs = sorted(((rel(res[tile][1][n], g), n) for n, g in res[0][1].items() if g.abs().sum() > 0), reverse=True)
    import statistics
    print(f"{norm} B={B} F({tile}x{tile},3x3) vs direct: out {rel(res[tile][0], res[0][0]):.1e}; gradients: median {statistics.median(e for e, _ in errs):.1e}, "
          f"90% below {errs[len(errs) // 10][0]:.1e}, worst {[(f'{e:.1e}', n) for e, n in errs[:3]]}")
