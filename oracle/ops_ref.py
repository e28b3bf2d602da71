"""Per-operator CPU oracle (plain PyTorch fp32/fp64 on the host).  TEST INFRASTRUCTURE ONLY.

Each function restates one torch.nn operator the reference's hot path uses, on NHWC tensors and
with the build's physical weight layout [kh, kw, Cin, Cout], so the GPU parity tests can compare
buffers directly.  Reference call sites: models/Generator.py:40-113,158-259,
models/Discriminator.py:14-99 (Conv2d / ReflectionPad2d / ConvTranspose2d / norm / ReLU),
models/Generator.py:115-130 (gate), models/network_utils.py:129-163 (GANLoss),
losses/L1_plus_perceptualLoss.py:32-75.
"""
import torch
import torch.nn.functional as F


def to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def w_to_oihw(w):
    """[kh, kw, Cin, Cout] -> [Cout, Cin, kh, kw]"""
    return w.permute(3, 2, 0, 1).contiguous()


def oihw_to_w(w):
    return w.permute(2, 3, 1, 0).contiguous()


def conv2d(x, w, bias, stride, pad, reflect, act=0, dtype=torch.float64):
    """x NHWC, w [kh,kw,Cin,Cout] -> y NHWC (act: 0 none, 1 relu, 2 tanh)."""
    xn = to_nchw(x.to(dtype))
    if reflect and pad > 0:
        xn = F.pad(xn, (pad, pad, pad, pad), mode="reflect")
        p = 0
    else:
        p = pad
    y = F.conv2d(xn, w_to_oihw(w.to(dtype)), None if bias is None else bias.to(dtype), stride, p)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.tanh(y)
    return to_nhwc(y)


def conv2d_grads(x, w, bias, dy, stride, pad, reflect, act=0, dtype=torch.float64):
    """returns y, dx, dw, db for upstream gradient dy (all NHWC / [kh,kw,Cin,Cout])."""
    x = x.to(dtype).requires_grad_(True)
    w = w.to(dtype).requires_grad_(True)
    b = None if bias is None else bias.to(dtype).requires_grad_(True)
    y = conv2d(x, w, b, stride, pad, reflect, act, dtype)
    ins = [x, w] + ([b] if b is not None else [])
    gs = torch.autograd.grad(y, ins, dy.to(dtype))
    return y.detach(), gs[0], gs[1], (gs[2] if b is not None else None)


def convT2d(x, w, bias, dtype=torch.float64):
    """ConvTranspose2d(k3,s2,p1,op1).  x NHWC [B,h,w,CinT]; w physical [kh,kw,CoutT,CinT]."""
    wt = w.to(dtype).permute(3, 2, 0, 1).contiguous()       # logical [CinT, CoutT, kh, kw]
    y = F.conv_transpose2d(to_nchw(x.to(dtype)), wt, None if bias is None else bias.to(dtype),
                           stride=2, padding=1, output_padding=1)
    return to_nhwc(y)


def convT2d_grads(x, w, bias, dy, dtype=torch.float64):
    x = x.to(dtype).requires_grad_(True)
    w = w.to(dtype).requires_grad_(True)
    b = None if bias is None else bias.to(dtype).requires_grad_(True)
    y = convT2d(x, w, b, dtype)
    ins = [x, w] + ([b] if b is not None else [])
    gs = torch.autograd.grad(y, ins, dy.to(dtype))
    return y.detach(), gs[0], gs[1], (gs[2] if b is not None else None)


def norm_act(x, gamma, beta, mode, relu, mask=None, drop_p=0.0, residual=None, eps=1e-5,
             dtype=torch.float64):
    """[Batch|Instance]Norm2d in training mode -> ReLU -> dropout(mask) (+ residual). x NHWC."""
    xn = to_nchw(x.to(dtype))
    if mode == "instance":
        y = F.instance_norm(xn, eps=eps)
    else:
        y = F.batch_norm(xn, None, None, None if gamma is None else gamma.to(dtype),
                         None if beta is None else beta.to(dtype), True, 0.1, eps)
    if relu:
        y = torch.relu(y)
    y = to_nhwc(y)
    if mask is not None:
        y = y * mask.to(dtype) / (1.0 - drop_p)
    if residual is not None:
        y = y + residual.to(dtype)
    return y


def gate(x1, s1, s2, s3):
    out = x1 + s1 * torch.sigmoid(s2) * torch.sigmoid(s3)
    return out, torch.cat([s3, out], -1), torch.cat([s2, out], -1)


def bce_const(x, target, weight):
    t = torch.full_like(x, target)
    return weight * F.binary_cross_entropy_with_logits(x, t)


def rel_l1(a, b):
    """relative L1 error |a-b|_1 / |b|_1 — the parity metric of BASELINE.json (1e-3)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.abs().sum().item()
    return (a - b).abs().sum().item() / (den if den > 0 else 1.0)
