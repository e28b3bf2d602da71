"""Seeded random shapes through the conv dispatch (Winograd F(6x6,3x3) / F(4x4,3x3) / direct / thin
kernels, whichever ops.py selects) against the fp64 oracle: all three passes via autograd."""
import random

import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _mk(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dev)


def _cases():
    rng = random.Random(20261002)
    out = []
    for i in range(14):         # 3x3 stride-1 pad-1: the Winograd-eligible family
        B = rng.choice([1, 2, 3])
        H, W = rng.randint(12, 45), rng.randint(12, 45)
        Cin, Cout = rng.choice([32, 64, 96, 128, 160, 256]), rng.choice([32, 64, 128, 192, 256])
        out.append((B, H, W, Cin, Cout, 3, 1, 1, rng.random() < 0.6))
    for i in range(4):          # 7x7 stems / head
        B = rng.choice([1, 2])
        H, W = rng.randint(8, 40), rng.randint(8, 70)
        Cin, Cout = rng.choice([(4, 64), (8, 64), (24, 64), (44, 64), (64, 4), (128, 4)])
        out.append((B, H, W, Cin, Cout, 7, 1, 3, True))
    for i in range(3):          # stride 2
        B = rng.choice([1, 2])
        H, W = 2 * rng.randint(4, 16), 2 * rng.randint(4, 16)
        out.append((B, H, W, rng.choice([16, 64, 128]), rng.choice([32, 128, 256]), 3, 2, 1, False))
    return out


@pytest.mark.parametrize("case", _cases())
def test_conv_random_shape_all_passes(case, dev):
    from mmhand_amd import ops
    B, H, W, Cin, Cout, k, s, p, refl = case
    x = _mk((B, H, W, Cin), 1, dev).requires_grad_(True)
    w = _mk((k, k, Cin, Cout), 2, dev, 0.1).requires_grad_(True)
    b = _mk((Cout,), 3, dev).requires_grad_(True)
    y = ops.Conv2dFn.apply(x, w, b, s, p, refl, 1)
    dy = _mk(tuple(y.shape), 4, dev)
    y.backward(dy)
    yr = R.conv2d(x.detach().cpu(), w.detach().cpu(), b.detach().cpu(), s, p, refl, 1)
    # ReLU backward is discontinuous: one pre-activation within rounding of 0 flips a mask bit and
    # moves dx by 5e-4 relative at these sizes.  Take the mask the kernels used (y > 0) and compare
    # the linear gradients of the masked dy.
    dym = dy * (y.detach() > 0)
    _, dxr, dwr, dbr = R.conv2d_grads(x.detach().cpu(), w.detach().cpu(), b.detach().cpu(), dym.cpu(), s, p, refl, 0)
    assert R.rel_l1(y, yr) < TOL, ("fprop", R.rel_l1(y, yr))
    assert R.rel_l1(x.grad, dxr) < TOL, ("dgrad", R.rel_l1(x.grad, dxr))
    assert R.rel_l1(w.grad, dwr) < TOL, ("wgrad", R.rel_l1(w.grad, dwr))
    assert R.rel_l1(b.grad, dbr) < 5e-5, ("bias grad", R.rel_l1(b.grad, dbr))
