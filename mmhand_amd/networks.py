"""Generator / Discriminator of MM-HAND on the HIP kernels — drop-in for the reference classes.

Same constructor signatures, same ``state_dict`` key names and logical shapes, same forward
semantics as models/Generator.py:286-313 and models/Discriminator.py:58-154, but nothing here is
an ``nn.Conv2d``: the modules only hold parameters, and ``forward`` strings together the
autograd shims of mmhand_amd.ops (implicit-GEMM convs, fused norm+ReLU+dropout, fused gate).

Storage: every parameter lives in one flat fp32 buffer per network (and every gradient in a
second one), so Adam, zero_grad and the data-parallel all-reduce are each a single launch.
Conv weights are stored in the kernels' layout [kh, kw, Cin_pad, Cout_pad]; ``state_dict()`` /
``load_state_dict()`` convert to and from the reference's OIHW (Conv2d) / IOHW (ConvTranspose2d).
"""
import functools
import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .ops import pad4


# ----------------------------------------------------------------------------- parameter holders
class ConvParam(nn.Module):
    """Parameters of one nn.Conv2d / nn.ConvTranspose2d in kernel layout."""

    def __init__(self, cin, cout, k, bias, transposed=False):
        super().__init__()
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        # Conv2d: [k,k,Cin_p,Cout_p]; ConvTranspose2d(CinT=cin, CoutT=cout): [k,k,CoutT_p,CinT_p]
        a, b = (pad4(cout), pad4(cin)) if transposed else (pad4(cin), pad4(cout))
        self.weight = nn.Parameter(torch.zeros(k, k, a, b))
        self.bias = nn.Parameter(torch.zeros(pad4(cout))) if bias else None

    # logical <-> physical --------------------------------------------------
    def logical_weight(self):
        w = self.weight.detach()
        if self.transposed:   # physical [k,k,CoutT,CinT] -> logical [CinT, CoutT, k, k]
            return w.permute(3, 2, 0, 1)[: self.cin, : self.cout]
        return w.permute(3, 2, 0, 1)[: self.cout, : self.cin]     # [Cout, Cin, k, k]

    def set_logical(self, weight=None, bias=None):
        with torch.no_grad():
            if weight is not None:
                self.weight.zero_()
                self.logical_weight().copy_(weight)
            if bias is not None and self.bias is not None:
                self.bias.zero_()
                self.bias[: self.cout].copy_(bias)
        ops.bump_weights_epoch()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.logical_weight().contiguous().clone()
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias.detach()[: self.cout].clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        for name, want in (("weight", True), ("bias", self.bias is not None)):
            key = prefix + name
            if key in state_dict:
                if not want:
                    unexpected_keys.append(key)
                    continue
                t = state_dict[key]
                exp = tuple(self.logical_weight().shape) if name == "weight" else (self.cout,)
                if tuple(t.shape) != exp:
                    error_msgs.append(f"size mismatch for {key}: checkpoint {tuple(t.shape)} vs {exp}")
                    continue
                self.set_logical(**{name: t})
            elif want and strict:
                missing_keys.append(key)


class NormParam(nn.Module):
    """State of nn.BatchNorm2d(affine=True): weight, bias, running stats."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0

    # nn.BatchNorm2d adds 1 to num_batches_tracked in every training forward; with momentum 0.1 (the reference never passes
    # None) nothing reads it but state_dict().  A device-side add per norm site is 101 one-element kernels per iteration:
    # the forwards are counted on the host and folded into the buffer when it is read out.
    def count_batch(self):
        self._pending_batches += 1

    def flush_batches(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0
        return self.num_batches_tracked

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)


class Bag(nn.Module):
    """Named container reproducing the reference's nn.Sequential index keys."""

    def put(self, name, mod):
        self.add_module(str(name), mod)
        return mod

    def __getitem__(self, name):
        return self._modules[str(name)]


def logical_grads(net, device=None):
    """{reference state_dict key: gradient in the reference's logical layout} (Conv2d OIHW, ConvTranspose2d IOHW, norm affine
    [C]) from the physical-layout parameter gradients ([kh][kw][Cin_pad][Cout_pad] views of the flat gradient buffer):
    what the reference's `named_parameters()` gradients are compared with (models/Generator.py, models/Discriminator.py)."""
    out = {}
    mv = (lambda t: t if device is None else t.to(device))
    for name, m in net.named_modules():
        if isinstance(m, ConvParam):
            if m.weight.grad is not None:
                g = m.weight.grad.permute(3, 2, 0, 1)
                out[name + ".weight"] = mv(g[: m.cin, : m.cout] if m.transposed else g[: m.cout, : m.cin])
            if m.bias is not None and m.bias.grad is not None:
                out[name + ".bias"] = mv(m.bias.grad[: m.cout])
        elif isinstance(m, NormParam):
            if m.weight.grad is not None:
                out[name + ".weight"] = mv(m.weight.grad)
                out[name + ".bias"] = mv(m.bias.grad)
    return out


def _norm_kind(norm_layer):
    """'batch' | 'instance' from the reference's norm_layer argument (a functools.partial)."""
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    if isinstance(f, str):
        return f
    if f is nn.InstanceNorm2d:
        return "instance"
    if f is nn.BatchNorm2d:
        return "batch"
    raise NotImplementedError(f"normalization layer {norm_layer} is not supported")


# SyncBN packing: norm sites that do not depend on each other - the three generator streams at
# one depth, the passes of a discriminator (or of both) that run side by side - share ONE all-gather forward and ONE
# all-reduce backward (ops.NormActMultiFn) instead of one tiny collective per site: 202 -> 100 per iteration at full
# size.  Only taken where SyncBN is on (--norm batch under data parallelism); MMH_PACK_SYNCBN=0: one collective per site.
import os as _os
PACK_SYNCBN = _os.environ.get("MMH_PACK_SYNCBN", "1") != "0"


def packing(*nets):
    """the given networks run SyncBN and may pack its collectives"""
    return bool(PACK_SYNCBN and nets and all(n.norm == "batch" and n.training and n.sync_group is not None for n in nets)
                and len({id(n.sync_group) for n in nets}) == 1)


def normact_multi(entries):
    """entries: dicts with the arguments of _Net.normact (net, bag, idx, x, relu, and optionally drop, site, residual,
    out_lp, defer) of norm sites that do not depend on each other -> list of what normact returns for each.  Under
    SyncBN the sites become one autograd node with packed collectives; otherwise this is a plain loop."""
    nets = [e["net"] for e in entries]
    if len(entries) < 2 or not packing(*nets):
        return [e["net"].normact(e["bag"], e["idx"], e["x"], e["relu"], e.get("drop", False), e.get("site"),
                                 e.get("residual"), e.get("out_lp", 0), e.get("defer", 0)) for e in entries]
    flat, meta = [], []
    for e in entries:
        net, x = e["net"], e["x"]
        x16 = None
        if isinstance(x, tuple):
            x, x16 = x
        relu, drop, defer, out_lp = e["relu"], e.get("drop", False), e.get("defer", 0), e.get("out_lp", 0)
        drop_p = 0.5 if drop else 0.0
        mask, seed = None, 0
        if drop_p > 0:
            if net._mask_src is not None:
                mask = net._mask_src[e.get("site")]
            else:
                seed = ops.next_dropout_seed()
        np_ = e["bag"][e["idx"]]
        np_.count_batch()
        if defer:
            assert x16 is None and not out_lp and (defer == 3 or e.get("residual") is None)
        flat += [x, np_.weight, np_.bias, e.get("residual"), np_.running_mean, np_.running_var, relu, drop_p, seed, mask,
                 out_lp, x16, defer, None]
        meta.append((x, relu, drop_p, out_lp, defer))
    outs = ops.NormActMultiFn.apply(nets[0].sync_group, len(entries), *flat)
    res = []
    for i, (x, relu, drop_p, out_lp, defer) in enumerate(meta):
        o = outs[i * ops.NORM_SITE_OUTS:(i + 1) * ops.NORM_SITE_OUTS]
        if defer in (1, 2):
            res.append((o[0], ops.NormDefer(x.detach(), o[1], o[2], 1, relu, drop_p, o[3])))
        elif out_lp:
            res.append((o[0], o[1]))
        else:
            res.append(o[0])
    return res


class _Net(nn.Module):
    """Shared machinery: flat buffers, init, norm/conv helpers."""

    def __init__(self, norm_layer, use_dropout):
        super().__init__()
        self.norm = _norm_kind(norm_layer)
        self.use_bias = self.norm == "instance"
        self.use_dropout = use_dropout
        self.sync_group = None        # set by MMHandModel for SyncBN under data parallel
        self.bf16 = False             # bf16 MFMA compute for the convs (apex O1/O2 analogue)
        self.flat_param = None
        self.flat_grad = None
        self._mask_src = None         # test hook: dict site -> uint8 NHWC keep mask

    # -- construction helpers
    def _conv(self, bag, idx, cin, cout, k, transposed=False, bias=None):
        return bag.put(idx, ConvParam(cin, cout, k, self.use_bias if bias is None else bias,
                                      transposed))

    def _normp(self, bag, idx, c):
        if self.norm == "batch":
            bag.put(idx, NormParam(c))

    # -- flat storage
    def flatten_parameters(self):
        """Move all parameters into one flat buffer, and give them persistent flat gradients."""
        ps = list(self.parameters())
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                flat[off:off + k].copy_(p.reshape(-1))
                p.data = flat[off:off + k].view(p.shape)
                p.grad = gflat[off:off + k].view(p.shape)
                off += k
        self.flat_param, self.flat_grad = flat, gflat
        ops.bump_weights_epoch()
        return flat, gflat

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        if self.flat_param is not None and any(p.device != self.flat_param.device
                                               for p in self.parameters()):
            self.flat_param = None
        return out

    def zero_grad(self, set_to_none=False):
        if self.flat_grad is not None:
            self.flat_grad.zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def init_weights(self, init_type="normal", seed=None):
        """init_weights (models/network_utils.py:12-71).  'normal': conv W ~ N(0,0.02); 'xavier':
        xavier_normal(gain 0.02); 'kaiming': kaiming_normal(fan_in); 'orthogonal': gain 1.  In all
        four BN gamma ~ N(1,0.02), beta = 0, and conv biases keep nn.Conv2d's default
        U(+-1/sqrt(fan_in)).  Draws come from a seeded CPU generator (the reference never seeds)."""
        if init_type not in ("normal", "xavier", "kaiming", "orthogonal"):
            raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        for m in self.modules():
            if isinstance(m, ConvParam):
                shape = tuple(m.logical_weight().shape)          # OIHW, or IOHW when transposed
                rf = m.k * m.k
                fan_in, fan_out = shape[1] * rf, shape[0] * rf   # torch's convention on dims 1 / 0
                if init_type == "normal":
                    w = torch.randn(shape, generator=g) * 0.02
                elif init_type == "xavier":
                    w = torch.randn(shape, generator=g) * (0.02 * math.sqrt(2.0 / (fan_in + fan_out)))
                elif init_type == "kaiming":
                    w = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
                else:
                    flat = torch.randn((shape[0], fan_in), generator=g)
                    q, r = torch.linalg.qr(flat.t() if shape[0] < fan_in else flat)
                    q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
                    w = (q.t() if shape[0] < fan_in else q).reshape(shape)
                b = None
                if m.bias is not None:
                    bound = 1.0 / math.sqrt((m.cout if m.transposed else m.cin) * rf)
                    b = (torch.rand(m.cout, generator=g) * 2 - 1) * bound
                m.set_logical(w, b)
            elif isinstance(m, NormParam):
                with torch.no_grad():
                    m.weight.copy_(1.0 + torch.randn(m.weight.shape, generator=g) * 0.02)
                    m.bias.zero_()
        return self

    # -- functional layers
    def conv(self, cp, x, stride=1, pad=0, reflect=False, act=L.ACT_NONE, dx_channels=0, y_lp=False, to_norm=False,
             g_defer=False, res_tok=None, x_twin=None):
        """x: an fp32 NHWC tensor, a (proxy, x16) pair from a producer that wrote it in 16 bits, or a
        (proxy, NormDefer) pair from a norm whose apply pass runs inside this conv (normact(defer)).
        y_lp (see _lp_edge): the consumer (normact / the PATBlock gate) takes the output in 16 bits ->
        returns a (proxy, y16) pair.  g_defer: the norm behind this conv defers its backward apply
        pass to this conv's backward (ops.USE_NORM_FUSION)."""
        x16 = None
        if isinstance(x, tuple):
            x, x16 = x
        if isinstance(x16, ops.NormDefer):
            return ops.Conv2dFn.apply(x, cp.weight, cp.bias, stride, pad, reflect, act, self.bf16, dx_channels, None,
                                      False, bool(to_norm and self.norm == "instance" and self.training), x16, g_defer)
        # to_norm: the output goes straight into this net's norm layer; under InstanceNorm the conv bias then
        # has an identically zero gradient (ops.EXACT_NULL_BIAS_GRAD)
        # ... and the conv's epilogue leaves partial statistics of its output for that norm (ops.FUSE_NORM_STATS); a
        # bias-free conv in front of a BatchNorm gets them too (its statistics are merged over the whole batch)
        nb = bool(to_norm and self.training and (self.norm == "instance" or cp.bias is None))
        if x16 is not None:
            res_tok = None      # tokens belong to fp32 block inputs (ops.ResidualToken)
        if y_lp:
            p, y16 = ops.Conv2dFn.apply(x, cp.weight, cp.bias, stride, pad, reflect, act, self.bf16, dx_channels, x16, True, nb,
                                        None, False, res_tok, x_twin)
            return p, y16
        return ops.Conv2dFn.apply(x, cp.weight, cp.bias, stride, pad, reflect, act, self.bf16, dx_channels, x16, False, nb,
                                  None, g_defer, res_tok, x_twin)

    def _lp_edge(self, cp, stride=1, reflect=True):
        """16-bit hand-over to the 3x3 / pad 1 conv (or ConvTranspose2d) `cp` (training, 16-bit mode, all
        three passes of the conv on kernels that read and write 16-bit tensors): returns the operand
        type, else 0."""
        if not (self.bf16 and self.training and cp is not None and cp.k == 3):
            return 0
        ws = cp.weight.shape        # physical: conv [k,k,Cin,Cout]; transposed conv [k,k,CoutT,CinT]
        if cp.transposed:
            return self.bf16 if ops.convT_lp16_ok(ws[3], ws[2], self.bf16) else 0
        return self.bf16 if ops.lp16_chain_ok(ws[2], ws[3], 3, stride, 1, reflect, self.bf16) else 0

    def _lp_edge_head(self, cp, x):
        """16-bit hand-over to the Generator's 7x7 head `cp` (x: its input, a tensor or a (proxy, x16) pair of that shape)"""
        if not (self.bf16 and self.training and ops.USE_LP16_EDGES and cp is not None and cp.k == 7):
            return 0
        B, H, W, _ = (x[0] if isinstance(x, tuple) else x).shape
        ws = cp.weight.shape
        return self.bf16 if ops.head16_ok(B, H, W, ws[2], ws[3], 7, 1, 3, True, self.bf16) else 0

    def _lp_out(self, cp, stride=1, reflect=True):
        """the conv `cp` hands its output (and takes its gradient) in 16 bits"""
        return bool(ops.USE_LP16_EDGES and self._lp_edge(cp, stride, reflect))

    def _lp_out_stem(self, cp, B, H, W, input_grad, dx_channels=0):
        """the 7x7 stem `cp` hands its output over in 16 bits (flat-K fprop; an input gradient, if any,
        only for the first <= 4 channels)"""
        if not (self.bf16 and self.training and ops.USE_LP16_EDGES) or (input_grad and not 0 < dx_channels <= 4):
            return False
        ws = cp.weight.shape
        d = ops.conv_desc(B, H, W, ws[2], ws[3], cp.k, 1, cp.k // 2, True)
        return ops.stem_lp16_ok(d, self.bf16, dx_channels)

    def convT(self, cp, x, y_lp=False, to_norm=False):
        """x: fp32 NHWC or a (proxy, x16) pair; y_lp: returns a (proxy, y16) pair; to_norm as in conv()"""
        x16 = None
        if isinstance(x, tuple):
            x, x16 = x
        nb = bool(to_norm and self.norm == "instance" and self.training)
        if y_lp:
            p, y16 = ops.ConvT2dFn.apply(x, cp.weight, cp.bias, self.bf16, x16, True, nb)
            return p, y16
        return ops.ConvT2dFn.apply(x, cp.weight, cp.bias, self.bf16, x16, False, nb)

    def normact(self, bag, idx, x, relu, drop=False, site=None, residual=None, out_lp=0, defer=0, res_tok=None, want_twin=0):
        """out_lp: hand the result to the next conv in 16 bits -> returns a (proxy, x16) pair.
        x may itself be a (proxy, x16) pair from a 16-bit convolution (conv(y_lp=True)).
        defer (1 | 2, training fp32): only the statistics are finalised here; returns a (proxy, NormDefer)
        pair for the conv that consumes it (2: the backward apply pass goes to the producing conv)."""
        x16 = None
        if isinstance(x, tuple):
            x, x16 = x
        drop_p = 0.5 if (drop and self.training) else 0.0
        mask = None
        seed = 0
        if drop_p > 0:
            if self._mask_src is not None:
                mask = self._mask_src[site]
            else:
                seed = ops.next_dropout_seed()
        if defer == 3:      # written as usual; only its backward apply pass goes to the producing conv
            assert self.training and x16 is None and not out_lp and not relu and not drop
            if self.norm == "instance":
                return ops.NormActFn.apply(x, None, None, residual, None, None, "instance", False, 0.0, 0, None, None,
                                           0, None, 3)
            np_ = bag[idx]
            np_.count_batch()
            return ops.NormActFn.apply(x, np_.weight, np_.bias, residual, np_.running_mean, np_.running_var, "batch",
                                       False, 0.0, 0, None, self.sync_group, 0, None, 3)
        if defer:
            assert self.training and x16 is None and residual is None and not out_lp
            if self.norm == "instance":
                args, groups = (None, None, None, None, None, "instance"), x.shape[0]
                sync = None
            else:
                np_ = bag[idx]
                np_.count_batch()
                args, groups = (np_.weight, np_.bias, None, np_.running_mean, np_.running_var, "batch"), 1
                sync = self.sync_group
            p, scale, shift, drows = ops.NormActFn.apply(x, *args, relu, drop_p, seed, mask, sync, 0, None, defer)
            return p, ops.NormDefer(x.detach(), scale, shift, groups, relu, drop_p, drows)
        # want_twin: the fp32 output has a second, 16-bit consumer (the next 3x3 conv) -> returns (out, twin16 | None)
        want_twin = want_twin if (self.training and not out_lp) else 0
        if self.norm == "instance":
            return ops.NormActFn.apply(x, None, None, residual, None, None, "instance", relu,
                                       drop_p, seed, mask, None, out_lp, x16, 0, res_tok, want_twin)
        np_ = bag[idx]
        if self.training:
            np_.count_batch()
            return ops.NormActFn.apply(x, np_.weight, np_.bias, residual, np_.running_mean,
                                       np_.running_var, "batch", relu, drop_p, seed, mask,
                                       self.sync_group, out_lp, x16, 0, res_tok, want_twin)
        scale = np_.weight / torch.sqrt(np_.running_var + ops.EPS)
        shift = np_.bias - np_.running_mean * scale
        y = ops.AffineActFn.apply(x, scale, shift, relu)
        return y if residual is None else y + residual

    def _norm_fusion(self, c1, c2, x):
        """fp32 training, both convs of a two-conv block on Winograd F(6x6,3x3): 1 = the apply pass of the norm
        between them runs inside c2's input transform; 2 = its backward apply pass inside c1's backward
        transform too (needs c1's bias gradient to be null or absent: the fused transform leaves no dy to sum)."""
        if not (ops.USE_NORM_FUSION and self.training and not self.bf16 and torch.is_tensor(x)
                and ops.KEEP_WINOGRAD_INPUT):
            return 0
        B, H, W, Cin = x.shape
        C1, C2 = c1.weight.shape[3], c2.weight.shape[3]
        if not (ops.norm_fusion_ok(C1) and B * H * W * C1 < 2 ** 31 and ops._wino_tile(B, H, W, C1, C2, 3, 1, 1, False) == 6):
            return 0
        null_db = c1.bias is None or (self.norm == "instance" and ops.EXACT_NULL_BIAS_GRAD)
        if (null_db and ops.FUSE_WINO6_BWD and ops._wino_tile(B, H, W, Cin, C1, 3, 1, 1, False) == 6
                and ops._wino_tile(B, H, W, Cin, C1, 3, 1, 1, False, "dgrad") == 6):
            return 2
        return 1

    def _norm_bwd_fusion(self, c2, x):
        """the norm behind conv c2 (3x3, input of x's spatial size and c2's input channels) can hand the apply pass
        of its backward to c2's fused F(6x6,3x3) backward transform"""
        if not (ops.USE_NORM_FUSION and self.training and not self.bf16 and ops.KEEP_WINOGRAD_INPUT and ops.FUSE_WINO6_BWD):
            return False
        B, H, W, _ = x.shape
        Cin, Cout = c2.weight.shape[2], c2.weight.shape[3]
        null_db = c2.bias is None or (self.norm == "instance" and ops.EXACT_NULL_BIAS_GRAD)
        return (null_db and ops.norm_fusion_ok(Cout) and B * H * W * Cout < 2 ** 31
                and ops._wino_tile(B, H, W, Cin, Cout, 3, 1, 1, False) == 6
                and ops._wino_tile(B, H, W, Cin, Cout, 3, 1, 1, False, "dgrad") == 6)

    def _res_token(self, x):
        """a token for a block input with two consumers (ops.ResidualToken): 16-bit training mode, fp32 tensor"""
        return ops.ResidualToken() if (ops.USE_RESIDUAL_TOKENS and self.bf16 and self.training and torch.is_tensor(x)
                                       and x.requires_grad) else None

    def _twin_for(self, cp):
        """the operand type of the 16-bit twin that the 3x3 / stride 1 conv `cp` would read in place of its fp32 input, else 0"""
        if not (self.bf16 and self.training and cp is not None and cp.k == 3 and not cp.transposed):
            return 0
        ws = cp.weight.shape
        return self.bf16 if (ops.lp16_v2_ok(ws[2], ws[3], 3, 1, 1, 0) and ops.norm_twin_ok(ws[2])) else 0

    def two_conv_block(self, blk, x, site, last_norm, residual=None, res_tok=None, x_twin=None, want_twin=0):
        """RP1-conv-norm-ReLU-(Dropout)-RP1-conv-(norm) (build_conv_block in both reference nets).
        res_tok: x is also read by the caller's residual add (the PATBlock gate); with residual is x (ResnetBlock)
        the token is made here."""
        if residual is not None and residual is x and last_norm:
            res_tok = self._res_token(x)
        i2 = 6 if self.use_dropout else 5
        # 16-bit mode: every tensor that faces one of these convolutions (input, output, both gradients)
        # lives in HBM in 16 bits only, as under apex O1; without a last norm the caller (the PATBlock
        # gate) receives the (proxy, y16) pair
        fuse = self._norm_fusion(blk[1], blk[i2], x)
        # the block's last norm (no ReLU / dropout; feeds the gate or the residual add): its backward apply pass
        # inside conv 2's backward transform, as for the first norm
        fuse_last = 3 if (last_norm and last_norm != "gate" and torch.is_tensor(x) and self._norm_bwd_fusion(blk[i2], x)) else 0
        y = self.conv(blk[1], x, 1, 1, True, y_lp=self._lp_out(blk[1]), to_norm=True, g_defer=fuse == 2, res_tok=res_tok,
                      x_twin=x_twin if torch.is_tensor(x) else None)
        y = self.normact(blk, 2, y, True, self.use_dropout, site, out_lp=self._lp_edge(blk[i2]), defer=fuse)
        y = self.conv(blk[i2], y, 1, 1, True, y_lp=self._lp_out(blk[i2]), to_norm=bool(last_norm), g_defer=fuse_last == 3)
        if last_norm == "gate":     # the caller's gate applies this norm itself (ops.GateNormFn): the (proxy, y16) pair
            return y
        if last_norm:
            # want_twin: the block's output also feeds the next block's first conv -> (out, twin16 | None)
            y = self.normact(blk, i2 + 1, y, False, residual=residual, defer=fuse_last,
                             res_tok=res_tok if residual is not None else None, want_twin=want_twin if not fuse_last else 0)
        return y


def two_conv_blocks_lockstep(entries):
    """_Net.two_conv_block for several independent blocks side by side (entries: dicts net, blk, x, site, last_norm,
    residual): conv 1 of every block, their norms as one packed node, conv 2 of every block, the last norms as one
    packed node.  Same arithmetic per block as two_conv_block."""
    pre = []
    for e in entries:
        net, blk, x = e["net"], e["blk"], e["x"]
        i2 = 6 if net.use_dropout else 5
        fuse = net._norm_fusion(blk[1], blk[i2], x)
        fuse_last = 3 if (e["last_norm"] and torch.is_tensor(x) and net._norm_bwd_fusion(blk[i2], x)) else 0
        y = net.conv(blk[1], x, 1, 1, True, y_lp=net._lp_out(blk[1]), to_norm=True, g_defer=fuse == 2)
        pre.append((i2, fuse, fuse_last, y))
    ys = normact_multi([dict(net=e["net"], bag=e["blk"], idx=2, x=y, relu=True, drop=e["net"].use_dropout and e["net"].training,
                             site=e["site"], out_lp=e["net"]._lp_edge(e["blk"][i2]), defer=fuse)
                        for e, (i2, fuse, fuse_last, y) in zip(entries, pre)])
    zs = []
    for e, (i2, fuse, fuse_last, _), y in zip(entries, pre, ys):
        net, blk = e["net"], e["blk"]
        zs.append(net.conv(blk[i2], y, 1, 1, True, y_lp=net._lp_out(blk[i2]), to_norm=e["last_norm"], g_defer=fuse_last == 3))
    last = [i for i, e in enumerate(entries) if e["last_norm"]]
    if last:
        outs = normact_multi([dict(net=entries[i]["net"], bag=entries[i]["blk"], idx=pre[i][0] + 1, x=zs[i], relu=False,
                                   residual=entries[i].get("residual"), defer=pre[i][2]) for i in last])
        for i, o in zip(last, outs):
            zs[i] = o
    return zs


# ----------------------------------------------------------------------------- Generator
class Generator(_Net):
    """Three-stream PATN generator (models/Generator.py:133-313)."""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 n_blocks=6, gpu_ids=[], padding_type="reflect", n_downsampling=2):
        assert type(input_nc) == list and len(input_nc) == 3, \
            "The AttModule take input_nc in format of list only!!"
        assert n_blocks >= 0
        if padding_type != "reflect":
            raise NotImplementedError("padding [%s] is not implemented" % padding_type)
        super().__init__(norm_layer, use_dropout)
        self.input_nc, self.output_nc, self.ngf = list(input_nc), output_nc, ngf
        self.n_blocks, self.n_down = n_blocks, n_downsampling
        self.gpu_ids = gpu_ids
        m = self.model = Bag()
        for s, nc in zip((1, 2, 3), input_nc):
            d = m.put(f"stream{s}_down", Bag())
            self._conv(d, 1, nc, ngf, 7)
            self._normp(d, 2, ngf)
            for i in range(n_downsampling):
                c = ngf * 2 ** i
                self._conv(d, 4 + 3 * i, c, 2 * c, 3)
                self._normp(d, 5 + 3 * i, 2 * c)
        dim = ngf * 2 ** n_downsampling
        att = m.put("att", Bag())
        i2 = 6 if use_dropout else 5
        for b in range(n_blocks):
            blk = att.put(b, Bag())
            for s in (1, 2, 3):
                cb = blk.put(f"conv_block_stream{s}", Bag())
                wide = (s != 1) and b > 0          # cated_stream2: 2*dim in, 2*dim mid, dim out
                cin = 2 * dim if wide else dim
                self._conv(cb, 1, cin, cin, 3)
                self._normp(cb, 2, cin)
                self._conv(cb, i2, cin, dim, 3)
                if s == 1:
                    self._normp(cb, i2 + 1, dim)
        up = m.put("stream1_up", Bag())
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            self._conv(up, 3 * i, c, c // 2, 3, transposed=True)
            self._normp(up, 3 * i + 1, c // 2)
        self._conv(up, 3 * n_downsampling + 1, ngf, output_nc, 7, bias=True)

    def _down_lockstep(self, x1, x2, x3):
        """the three down-sampling streams (stem + n_down stride-2 convs each) depth by depth: same layers and
        hand-over types as the per-stream loop of forward_nhwc, the three norms of a depth as one packed node"""
        m = self.model
        ds = [m[f"stream{s}_down"] for s in (1, 2, 3)]
        xs = [x1, x2, x3]
        first = [d[4] if self.n_down > 0 else None for d in ds]
        ys = []
        for d, x in zip(ds, xs):
            Bx, Hx, Wx, _ = x.shape
            ys.append(self.conv(d[1], x, 1, 3, True, y_lp=self._lp_out_stem(d[1], Bx, Hx, Wx, x.requires_grad), to_norm=True))
        xs = normact_multi([dict(net=self, bag=d, idx=2, x=y, relu=True, out_lp=self._lp_edge(f, 2, False))
                            for d, y, f in zip(ds, ys, first)])
        for i in range(self.n_down):
            ys, lps = [], []
            for s, d, x in zip((1, 2, 3), ds, xs):
                cp = d[4 + 3 * i]
                if i + 1 < self.n_down:
                    out_lp = self._lp_edge(d[4 + 3 * (i + 1)], 2, False)
                elif s != 1 and self.n_blocks > 0:
                    out_lp = self._lp_edge(m["att"][0][f"conv_block_stream{s}"][1])
                else:
                    out_lp = 0
                lps.append(out_lp)
                ys.append(self.conv(cp, x, 2, 1, False, y_lp=self._lp_out(cp, 2, False), to_norm=True))
            xs = normact_multi([dict(net=self, bag=d, idx=5 + 3 * i, x=y, relu=True, out_lp=lp)
                                for d, y, lp in zip(ds, ys, lps)])
        return xs

    def forward_nhwc(self, x1, x2, x3):
        """x1,x2,x3: NHWC (channels zero-padded to 4) -> NHWC [B,H,W,pad4(output_nc)]."""
        m = self.model
        xs = []
        x1_twin0 = None
        if packing(self):       # SyncBN: the three streams side by side, one packed collective per depth
            x1, x2, x3 = self._down_lockstep(x1, x2, x3)
        # 16-bit mode: the tensors between the 3x3 convs and their norms travel in 16 bits (see two_conv_block);
        # the 7x7 stems read fp32 inputs and write fp32
        for s, x in (() if packing(self) else zip((1, 2, 3), (x1, x2, x3))):
            d = m[f"stream{s}_down"]
            first = d[4] if self.n_down > 0 else None
            Bx, Hx, Wx, _ = x.shape
            x = self.normact(d, 2, self.conv(d[1], x, 1, 3, True, y_lp=self._lp_out_stem(d[1], Bx, Hx, Wx, x.requires_grad), to_norm=True),
                             True, out_lp=self._lp_edge(first, 2, False))
            for i in range(self.n_down):
                cp = d[4 + 3 * i]
                if i + 1 < self.n_down:
                    out_lp = self._lp_edge(d[4 + 3 * (i + 1)], 2, False)
                elif s != 1 and self.n_blocks > 0:      # streams 2 / 3 feed only block 0's first conv
                    out_lp = self._lp_edge(m["att"][0][f"conv_block_stream{s}"][1])
                else:
                    out_lp = 0
                # stream 1 stays fp32 (it is the gate's residual input too): its 16-bit twin for block 0's first conv comes
                # out of the same norm pass (ops.USE_NORM_TWIN)
                wt = (self._twin_for(m["att"][0]["conv_block_stream1"][1])
                      if (s == 1 and i + 1 == self.n_down and self.n_blocks > 0 and not out_lp) else 0)
                x = self.normact(d, 5 + 3 * i, self.conv(cp, x, 2, 1, False, y_lp=self._lp_out(cp, 2, False), to_norm=True), True,
                                 out_lp=out_lp, want_twin=wt)
                if wt:
                    x, x1_twin0 = x
            xs.append(x)
        if not packing(self):
            x1, x2, x3 = xs
        for b in range(self.n_blocks):
            blk = m["att"][b]
            p = f"model.att.{b}.conv_block_stream"
            tok = None
            # x1's 16-bit twin: the second half of cat(s3, out) the previous gate wrote in 16 bits (ops.USE_LP16_CAT_TWIN)
            x1_twin = x2[1][..., x1.shape[3]:] if (b > 0 and isinstance(x2, tuple) and torch.is_tensor(x1)) else None
            if b == 0 and torch.is_tensor(x1):
                x1_twin = x1_twin0
            if packing(self):
                s1, s2, s3 = two_conv_blocks_lockstep(
                    [dict(net=self, blk=blk[f"conv_block_stream{s}"], x=x, site=p + str(s), last_norm=s == 1)
                     for s, x in zip((1, 2, 3), (x1, x2, x3))])
            else:
                tok = self._res_token(x1)       # x1 feeds the stream-1 conv AND the gate's residual add
                # 16-bit mode, InstanceNorm: the block's last norm runs inside the gate (ops.GateNormFn)
                i2 = 6 if self.use_dropout else 5
                gate_norm = bool(self.norm == "instance" and self.training and self.bf16 and torch.is_tensor(x1)
                                 and self._lp_out(blk["conv_block_stream1"][i2])
                                 and ops.gate_norm_ok(x1.shape[0], x1.shape[1] * x1.shape[2], x1.shape[3]))
                s1 = self.two_conv_block(blk["conv_block_stream1"], x1, p + "1", "gate" if gate_norm else True, res_tok=tok,
                                         x_twin=x1_twin)
                s2 = self.two_conv_block(blk["conv_block_stream2"], x2, p + "2", False)
                s3 = self.two_conv_block(blk["conv_block_stream3"], x3, p + "3", False)
            # (out, cat(s3,out), cat(s2,out)): the reference's stream swap (Generator.py:130 vs :278)
            more = b + 1 < self.n_blocks
            cat_lp = self._lp_edge(m["att"][b + 1]["conv_block_stream2"][1]) if more else 0
            s16 = (None, None)
            if isinstance(s2, tuple):       # stream 2 / 3 end in a 16-bit convolution
                (s2, a), (s3, b_) = s2, s3
                s16 = (a, b_)
            gate_norm = isinstance(s1, tuple)       # (proxy, y16) of the stream-1 conv: its norm runs inside the gate
            if cat_lp:      # the cats feed only the next block's 16-bit convs: written in 16 bits
                if gate_norm:
                    x1, p2, p3, c2, c3 = ops.GateNormFn.apply(x1, s1[0], s2, s3, True, cat_lp, s1[1], *s16, tok)
                else:
                    x1, p2, p3, c2, c3 = ops.GateFn.apply(x1, s1, s2, s3, True, cat_lp, *s16, tok)
                x2, x3 = (p2, c2), (p3, c3)
            elif gate_norm:
                x1, x2, x3 = ops.GateNormFn.apply(x1, s1[0], s2, s3, more, 0, s1[1], *s16, tok)
            else:
                x1, x2, x3 = ops.GateFn.apply(x1, s1, s2, s3, more, 0, *s16, tok)
        up = m["stream1_up"]
        y = x1
        for i in range(self.n_down):
            nxt = up[3 * (i + 1)] if i + 1 < self.n_down else None
            yc = self.convT(up[3 * i], y, y_lp=self._lp_out(up[3 * i]), to_norm=True)
            # the 7x7 head takes its input in 16 bits where all three of its passes have 16-bit kernels (ops.head16_ok)
            out_lp = self._lp_edge(nxt) if nxt is not None else self._lp_edge_head(up[3 * self.n_down + 1], yc)
            y = self.normact(up, 3 * i + 1, yc, True, out_lp=out_lp)
        return self.conv(up[3 * self.n_down + 1], y, 1, 3, True, L.ACT_TANH)

    def forward(self, input):
        """input: list of three NCHW tensors (any strides) -> logical NCHW [B,output_nc,H,W]."""
        xs = []
        for x, nc in zip(input, self.input_nc):
            B, Cc, H, W = x.shape
            assert Cc == nc
            xs.append(ops.PackFn.apply(pad4(nc), x.float(), True, nc))
        y = self.forward_nhwc(*xs)
        return ops.nhwc_to_nchw_view(y, self.output_nc)


# ----------------------------------------------------------------------------- Discriminator
class Discriminator(_Net):
    """ResNet-style feature discriminator without a 1-channel head
    (models/Discriminator.py:58-154)."""

    def __init__(self, input_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False, n_blocks=6,
                 gpu_ids=[], padding_type="reflect", use_sigmoid=False, n_downsampling=2):
        assert n_blocks >= 0
        if padding_type != "reflect":
            raise NotImplementedError("padding [%s] is not implemented" % padding_type)
        if use_sigmoid or n_downsampling > 2:
            raise NotImplementedError("use_sigmoid / n_downsampling>2 are not on the MM-HAND path")
        super().__init__(norm_layer, use_dropout)
        self.input_nc, self.ngf, self.n_blocks, self.n_down = input_nc, ngf, n_blocks, n_downsampling
        self.gpu_ids = gpu_ids
        m = self.model = Bag()
        self._conv(m, 1, input_nc, ngf, 7)
        self._normp(m, 2, ngf)
        for i in range(n_downsampling):
            c = ngf * 2 ** i
            self._conv(m, 4 + 3 * i, c, 2 * c, 3)
            self._normp(m, 5 + 3 * i, 2 * c)
        dim = ngf * 2 ** n_downsampling
        i2 = 6 if use_dropout else 5
        base = 4 + 3 * n_downsampling
        for b in range(n_blocks):
            cb = m.put(base + b, Bag()).put("conv_block", Bag())
            self._conv(cb, 1, dim, dim, 3)
            self._normp(cb, 2, dim)
            self._conv(cb, i2, dim, dim, 3)
            self._normp(cb, i2 + 1, dim)

    def forward_nhwc(self, x, dx_channels=0):
        """dx_channels > 0: the caller needs the gradient of only the first dx_channels input
        channels (the generated image inside cat(fake, P2) / cat(fake, H1))."""
        m = self.model
        first = m[4] if self.n_down > 0 else None
        Bx, Hx, Wx, _ = x.shape
        y = self.normact(m, 2, self.conv(m[1], x, 1, 3, True, dx_channels=dx_channels, to_norm=True,
                                         y_lp=self._lp_out_stem(m[1], Bx, Hx, Wx, x.requires_grad, dx_channels)), True,
                         out_lp=self._lp_edge(first, 2, False))
        base = 4 + 3 * self.n_down
        twin = None         # 16-bit twin of the residual stream, written by the norm that produced it (ops.USE_NORM_TWIN)
        for i in range(self.n_down):
            cp = m[4 + 3 * i]
            out_lp = self._lp_edge(m[4 + 3 * (i + 1)], 2, False) if i + 1 < self.n_down else 0
            wt = self._twin_for(m[base]["conv_block"][1]) if (i + 1 == self.n_down and self.n_blocks > 0) else 0
            y = self.normact(m, 5 + 3 * i, self.conv(cp, y, 2, 1, False, y_lp=self._lp_out(cp, 2, False), to_norm=True), True,
                             out_lp=out_lp, want_twin=wt)
            if wt and not out_lp:
                y, twin = y
        for b in range(self.n_blocks):
            wt = self._twin_for(m[base + b + 1]["conv_block"][1]) if b + 1 < self.n_blocks else 0
            y = self.two_conv_block(m[base + b]["conv_block"], y, f"model.{base + b}.conv_block",
                                    True, residual=y, x_twin=twin, want_twin=wt)
            twin = None
            if wt:
                y, twin = y
        return y

    def forward(self, input):
        B, Cc, H, W = input.shape
        assert Cc == self.input_nc
        x = ops.PackFn.apply(pad4(Cc), input.float(), True, Cc)
        return ops.nhwc_to_nchw_view(self.forward_nhwc(x))


def discriminators_lockstep(entries):
    """Several Discriminator passes side by side - entries: [(net, x NHWC, dx_channels)], the same network twice (its
    real and its fake batch: models/MMHandModel.py:263-274) or two networks on their own inputs (D_PB and D_PP inside
    the generator step, :238-243) -> [logits].  Every pass keeps its OWN batch statistics, exactly as separate calls
    of forward_nhwc do (one after the other wherever SyncBN packing is off); what they share under SyncBN is the
    collective: one all-gather / all-reduce per depth for all of them (packing())."""
    nets = [e[0] for e in entries]
    if len(entries) < 2 or not packing(*nets) or len({(n.n_down, n.n_blocks) for n in nets}) != 1:
        return [net.forward_nhwc(x, dx_channels=dxc) for net, x, dxc in entries]
    n0 = nets[0]
    ys = []
    for net, x, dxc in entries:
        Bx, Hx, Wx, _ = x.shape
        ys.append(net.conv(net.model[1], x, 1, 3, True, dx_channels=dxc, to_norm=True,
                           y_lp=net._lp_out_stem(net.model[1], Bx, Hx, Wx, x.requires_grad, dxc)))
    ys = normact_multi([dict(net=net, bag=net.model, idx=2, x=y, relu=True,
                             out_lp=net._lp_edge(net.model[4] if net.n_down > 0 else None, 2, False))
                        for net, y in zip(nets, ys)])
    for i in range(n0.n_down):
        zs = []
        for net, y in zip(nets, ys):
            cp = net.model[4 + 3 * i]
            zs.append(net.conv(cp, y, 2, 1, False, y_lp=net._lp_out(cp, 2, False), to_norm=True))
        ys = normact_multi([dict(net=net, bag=net.model, idx=5 + 3 * i, x=z, relu=True,
                                 out_lp=net._lp_edge(net.model[4 + 3 * (i + 1)], 2, False) if i + 1 < net.n_down else 0)
                            for net, z in zip(nets, zs)])
    base = 4 + 3 * n0.n_down
    for b in range(n0.n_blocks):
        ys = two_conv_blocks_lockstep([dict(net=net, blk=net.model[base + b]["conv_block"], x=y,
                                            site=f"model.{base + b}.conv_block", last_norm=True, residual=y)
                                       for net, y in zip(nets, ys)])
    return ys


# ----------------------------------------------------------------------------- VGG19[:4]
# torchvision vgg19 "E" configuration: features[i] = conv3x3 (channels below) / ReLU / MaxPool2d(2, 2) ("M")
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def vgg19_layers(upto):
    """[(index, kind, cin, cout)] of vgg19.features[0 : upto + 1] - the slice losses/L1_plus_perceptualLoss.py:22-27 takes
    (`for i, layer in enumerate(vgg): add; if i == perceptual_layers: break`); kind in conv | relu | pool"""
    out, i, cin = [], 0, 3
    for v in VGG19_CFG:
        if v == "M":
            out.append((i, "pool", cin, cin)); i += 1
        else:
            out.append((i, "conv", cin, v)); out.append((i + 1, "relu", v, v)); i += 2; cin = v
    if not 0 <= upto < len(out):
        raise ValueError("--perceptual_layers %r: vgg19.features has indices 0..%d" % (upto, len(out) - 1))
    return out[:upto + 1]


class VGGHead(nn.Module):
    """vgg19.features[0 : perceptual_layers + 1], frozen (losses/L1_plus_perceptualLoss.py:22-27; the shipped value 3 is
    conv3x3 3->64 + ReLU, conv3x3 64->64 + ReLU).  torchvision weights are not available offline: load them with
    load_state_dict({'0.weight','0.bias','2.weight','2.bias', ...}) or a torchvision vgg19 state_dict."""

    def __init__(self, perceptual_layers=3):
        super().__init__()
        self.bf16 = False
        self.layers = vgg19_layers(perceptual_layers)
        self.net = Bag()
        for i, kind, cin, cout in self.layers:
            if kind == "conv":
                self.net.put(i, ConvParam(cin, cout, 3, True))
        self.conv_ids = [i for i, kind, _, _ in self.layers if kind == "conv"]
        self.out_channels = self.layers[-1][3]
        for p in self.parameters():
            p.requires_grad_(False)

    def init_random(self, seed=1234):
        g = torch.Generator().manual_seed(seed)
        for i in self.conv_ids:
            m = self.net[i]
            fan_in = m.cin * 9
            m.set_logical(torch.randn(m.cout, m.cin, 3, 3, generator=g) * math.sqrt(2.0 / fan_in),
                          torch.randn(m.cout, generator=g) * 0.05)
        return self

    def state_dict(self, *a, **kw):
        sd = super().state_dict(*a, **kw)
        return type(sd)((k.replace("net.", "", 1), v) for k, v in sd.items())

    def load_state_dict(self, sd, strict=True):
        """Accepts {'0.weight','0.bias','2.weight','2.bias', ...} or a torchvision vgg19 state_dict
        ('features.0.weight', ...; layers beyond the slice and the classifier are ignored)."""
        if any(k.startswith("features.") for k in sd):
            keep = {str(i) for i in self.conv_ids}
            sd = {k[len("features."):]: v for k, v in sd.items()
                  if k.startswith("features.") and k.split(".")[1] in keep}
        return super().load_state_dict({"net." + k: v for k, v in sd.items()}, strict)

    def _pair(self, x):
        """(conv1_1, conv1_2) when this is the shipped slice and x takes the 16-bit pair kernels, else None"""
        if [k for _, k, _, _ in self.layers] == ["conv", "relu", "conv", "relu"]:
            m1, m2 = self.net[self.layers[0][0]], self.net[self.layers[2][0]]
            frozen = not any(t.requires_grad for t in (m1.weight, m1.bias, m2.weight, m2.bias))    # the pair nodes return no weight gradients
            if frozen and ops.vgg_pair_ok(x, m1.weight, m2.weight, self.bf16):
                return m1, m2
        return None

    def l1_pair(self, x):
        return self._pair(x) if ops.USE_VGG_L1_LP16 else None

    def forward_nhwc(self, x):
        n = len(self.layers)
        pair = self._pair(x)
        if pair is not None:            # 16-bit mode: one node, 16-bit edge inside (ops.VggPairFn)
            m1, m2 = pair
            return ops.VggPairFn.apply(x, m1.weight, m1.bias, m2.weight, m2.bias, self.bf16)
        for pos, (i, kind, _, _) in enumerate(self.layers):
            if kind == "conv":
                m = self.net[i]
                relu = pos + 1 < n and self.layers[pos + 1][1] == "relu"        # the ReLU rides in the conv epilogue
                x = ops.Conv2dFn.apply(x, m.weight, m.bias, 1, 1, False, L.ACT_RELU if relu else L.ACT_NONE, self.bf16)
            elif kind == "pool":
                x = ops.MaxPool2x2Fn.apply(x)
        return x
