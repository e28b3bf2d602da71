"""Where along the backward pass the Winograd path's gradients leave the direct kernels': relative L1
between the two paths of every intermediate gradient of the wide-channel test Generator (and of the
forward activations), in backward order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from mmhand_amd import ops
from mmhand_amd.networks import Generator
from oracle import mmhand_ref as O
dev = torch.device("cuda:0")
norm = sys.argv[1] if len(sys.argv) > 1 else "instance"
NGF, SIZE, NB, B = 32, 64, 2, 4
sd = Generator([3, 42, 6], 3, NGF, norm, False, NB).init_weights("normal", 49).state_dict()
b = O.synthetic_batch(B, SIZE, SIZE, seed=11)
g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
probe = torch.randn(B, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(3))
rec = {}
for wino in (False, True):
    ops.USE_WINOGRAD = wino; ops.bump_weights_epoch()
    net = Generator([3, 42, 6], 3, NGF, norm, False, NB); net.load_state_dict(sd); net.to(dev).train(); net.flatten_parameters()
    acts, grads = {}, {}
    # wrap the functional layers to tap every intermediate
    orig_conv, orig_norm, orig_convT = net.conv, net.normact, net.convT
    cnt = [0]
    def tap(name, t):
        acts[name] = t.detach().clone()
        if t.requires_grad:
            t.register_hook(lambda g, n=name: grads.__setitem__(n, g.detach().clone()))
        return t
    def conv(cp, x, *a, **k):
        cnt[0] += 1; return tap(f"{cnt[0]:02d}.conv{tuple(cp.weight.shape)}", orig_conv(cp, x, *a, **k))
    def normact(bag, idx, x, *a, **k):
        cnt[0] += 1; return tap(f"{cnt[0]:02d}.norm", orig_norm(bag, idx, x, *a, **k))
    def convT(cp, x):
        cnt[0] += 1; return tap(f"{cnt[0]:02d}.convT", orig_convT(cp, x))
    net.conv, net.normact, net.convT = conv, normact, convT
    out = net([t.to(dev) for t in g_in])
    (out * probe.to(dev)).sum().backward()
    rec[wino] = (acts, grads)
rel = lambda a, b_: float((a.double() - b_.double()).abs().sum() / b_.double().abs().sum().clamp_min(1e-30))
names = sorted(rec[False][0])
print("forward activations (wino vs direct):")
for n in names:
    print(f"  {n:40s} {rel(rec[True][0][n], rec[False][0][n]):.1e}")
print("gradients, backward order:")
for n in reversed(names):
    if n in rec[False][1]:
        print(f"  {n:40s} {rel(rec[True][1][n], rec[False][1][n]):.1e}")
