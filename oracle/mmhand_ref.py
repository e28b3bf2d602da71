"""CPU oracle of the MM-HAND training step.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A functional pure-PyTorch restatement of the reference's hot path, driven by state_dicts in the
reference's own checkpoint format (same key names and shapes), so that reference weights load
unchanged.  It follows, function by function:

  generator_forward      models/Generator.py:115-130 (PATBlock.forward, incl. the stream swap
                         of :130 vs :278), :269-283 (PATNModel.forward), :152-259 (layer list)
  discriminator_forward  models/Discriminator.py:53-55, :79-151
  gan_loss               models/network_utils.py:129-163 (always BCEWithLogits, :141)
  l1_plus_perceptual     losses/L1_plus_perceptualLoss.py:32-75
  ImagePoolRef           util/image_pool.py:14-34
  StepOracle             models/MMHandModel.py:215-221 (forward), :236-261 (backward_G),
                         :263-292 (backward_D_*), :310-330 (optimize_parameters),
                         :90-98 (three Adams)
  pose maps              data/generic_dataset.py:191-217,239-242; util/util.py:94-114

Pinning: the reference ships no tests for this path.  This oracle is pinned against
the reference's own leaf modules: tests/golden/make_golden.py imports them from /root/reference in
the build container, asserts that this file reproduces their outputs, gradients and 3-iteration
step traces, and writes what they produced to tests/golden/*.npz; tests/test_oracle_cpu.py
re-checks this file against those vectors wherever the tests run.  VGG19 pretrained weights and NVIDIA
apex are third-party and absent: the perceptual term is checked as an operator on seeded
VGG-shaped weights, and DDP/SyncBN semantics are pinned to "mean of rank gradients / statistics
over the global batch" (parity unpinned by the reference for those two).
"""
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5
MOMENTUM = 0.1


# ----------------------------------------------------------------------------- building blocks
class _Net:
    """A state_dict wrapper: tensors that require grad are the parameters."""

    def __init__(self, sd, norm, use_dropout):
        self.sd = OrderedDict()
        for k, v in sd.items():
            t = v.detach().clone()
            if t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                t.requires_grad_(True)
            self.sd[k] = t
        self.norm = norm
        self.use_dropout = use_dropout
        self.training = True

    def parameters(self):
        return [t for t in self.sd.values() if t.requires_grad]

    def named_parameters(self):
        return [(k, t) for k, t in self.sd.items() if t.requires_grad]

    def get(self, key):
        return self.sd.get(key)

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.sd.items())


def _norm(net, x, prefix):
    """norm_layer(C) at `prefix` (models/network_utils.py:74-84)."""
    if net.norm == "instance":
        return F.instance_norm(x, eps=EPS)
    if net.norm == "batch":
        w, b = net.sd[prefix + ".weight"], net.sd[prefix + ".bias"]
        rm, rv = net.sd[prefix + ".running_mean"], net.sd[prefix + ".running_var"]
        y = F.batch_norm(x, rm, rv, w, b, net.training, MOMENTUM, EPS)
        if net.training:
            net.sd[prefix + ".num_batches_tracked"] += 1
        return y
    raise ValueError(net.norm)


def _conv(net, x, prefix, stride=1, pad=0, reflect=0):
    if reflect:
        x = F.pad(x, (reflect,) * 4, mode="reflect")
    return F.conv2d(x, net.sd[prefix + ".weight"], net.get(prefix + ".bias"), stride, pad)


def _drop(net, x, site, masks):
    """Dropout(0.5) after ReLU.  masks: None -> torch RNG; dict -> injected 0/1 masks by site;
    'off' -> identity (reference with --no_dropout has no such module at all)."""
    if not net.use_dropout or not net.training or masks == "off":
        return x
    if masks is None:
        return F.dropout(x, 0.5, True)
    return x * masks[site].to(x.dtype) * 2.0


def _two_conv_block(net, x, prefix, masks, last_norm):
    """build_conv_block: RP1-conv-norm-ReLU-(Drop)-RP1-conv-(norm)."""
    i2 = 6 if net.use_dropout else 5
    y = _conv(net, x, f"{prefix}.1", reflect=1)
    y = torch.relu(_norm(net, y, f"{prefix}.2"))
    y = _drop(net, y, prefix, masks)
    y = _conv(net, y, f"{prefix}.{i2}", reflect=1)
    if last_norm:
        y = _norm(net, y, f"{prefix}.{i2 + 1}")
    return y


def generator_forward(net, inputs, n_blocks=9, n_down=2, masks=None):
    """inputs: [H1 (B,3,H,W), cat(P1,P2) (B,42,H,W), cat(D1,D2) (B,6,H,W)] -> (B,3,H,W)."""
    xs = []
    for s, x in zip((1, 2, 3), inputs):
        p = f"model.stream{s}_down"
        x = torch.relu(_norm(net, _conv(net, x, f"{p}.1", reflect=3), f"{p}.2"))
        for i in range(n_down):
            x = torch.relu(_norm(net, _conv(net, x, f"{p}.{4 + 3 * i}", 2, 1), f"{p}.{5 + 3 * i}"))
        xs.append(x)
    x1, x2, x3 = xs
    for b in range(n_blocks):
        p = f"model.att.{b}"
        s1 = _two_conv_block(net, x1, f"{p}.conv_block_stream1", masks, True)
        s2 = _two_conv_block(net, x2, f"{p}.conv_block_stream2", masks, False)
        s3 = _two_conv_block(net, x3, f"{p}.conv_block_stream3", masks, False)
        out = x1 + s1 * torch.sigmoid(s2) * torch.sigmoid(s3)
        # the block returns (out, cat(s3,out), cat(s2,out), .) and the caller unpacks x1,x2,x3:
        # the pose and depth streams swap every block.
        x1, x2, x3 = out, torch.cat((s3, out), 1), torch.cat((s2, out), 1)
    p = "model.stream1_up"
    y = x1
    for i in range(n_down):
        y = F.conv_transpose2d(y, net.sd[f"{p}.{3 * i}.weight"], net.get(f"{p}.{3 * i}.bias"),
                               stride=2, padding=1, output_padding=1)
        y = torch.relu(_norm(net, y, f"{p}.{3 * i + 1}"))
    k = 3 * n_down + 1
    return torch.tanh(_conv(net, y, f"{p}.{k}", reflect=3))


def discriminator_forward(net, x, n_blocks=3, n_down=2, masks=None):
    """(B,C,H,W) -> raw (B,4*ndf,H/4,W/4) feature logits (no head, no sigmoid)."""
    p = "model"
    y = torch.relu(_norm(net, _conv(net, x, f"{p}.1", reflect=3), f"{p}.2"))
    for i in range(n_down):
        y = torch.relu(_norm(net, _conv(net, y, f"{p}.{4 + 3 * i}", 2, 1), f"{p}.{5 + 3 * i}"))
    base = 4 + 3 * n_down
    for b in range(n_blocks):
        y = y + _two_conv_block(net, y, f"{p}.{base + b}.conv_block", masks, True)
    return y


def gan_loss(pred, target_is_real):
    t = torch.tensor(1.0 if target_is_real else 0.0, dtype=pred.dtype).expand_as(pred)
    return F.binary_cross_entropy_with_logits(pred, t)


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")


def vgg_features(vgg, x, perceptual_layers=3):
    """vgg19.features[0 : perceptual_layers + 1] as losses/L1_plus_perceptualLoss.py:22-27 slices it (default 3: conv3x3
    3->64 + ReLU, conv3x3 64->64 + ReLU); torchvision's "E" configuration: conv3x3 pad 1 / ReLU / MaxPool2d(2, 2)."""
    i = 0
    for v in VGG19_CFG:
        if i > perceptual_layers:
            break
        if v == "M":
            x = F.max_pool2d(x, 2, 2); i += 1
        else:
            x = F.conv2d(x, vgg["%d.weight" % i], vgg["%d.bias" % i], 1, 1); i += 1
            if i <= perceptual_layers:
                x = torch.relu(x)
            i += 1
    return x


def l1_plus_perceptual(vgg, fake, real, lambda_l1, lambda_p, percep_is_l1=1, perceptual_layers=3):
    loss_l1 = F.l1_loss(fake, real) * lambda_l1
    mean = torch.tensor(IMAGENET_MEAN, dtype=fake.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=fake.dtype).view(1, 3, 1, 1)
    f = vgg_features(vgg, ((fake + 1) / 2 - mean) / std, perceptual_layers)
    r = vgg_features(vgg, ((real + 1) / 2 - mean) / std, perceptual_layers).detach()
    lp = (F.l1_loss(f, r) if percep_is_l1 == 1 else F.mse_loss(f, r)) * lambda_p
    return loss_l1 + lp, loss_l1, lp


class ImagePoolRef:
    """50-image history buffer driven by Python's `random` (util/image_pool.py)."""

    def __init__(self, pool_size, rng=random):
        self.pool_size = pool_size
        self.rng = rng
        self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for img in images:
            img = img.unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(img)
                out.append(img)
            elif self.rng.uniform(0, 1) > 0.5:
                j = self.rng.randint(0, self.pool_size - 1)
                out.append(self.images[j].clone())
                self.images[j] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


class StepOracle:
    """One MMHandModel.optimize_parameters() per call to step()."""

    def __init__(self, sd_G, sd_DPB, sd_DPP, vgg, norm="batch", use_dropout=False,
                 use_dropout_D=False, n_blocks=9, n_layers_D=3, lr=2e-4, beta1=0.5,
                 lambda_A=10.0, lambda_B=10.0, lambda_GAN=5.0, pool_size=50, DG_ratio=1,
                 masks=None, rng=random, percep_is_l1=1, perceptual_layers=3):
        self.G = _Net(sd_G, norm, use_dropout)
        self.DPB = _Net(sd_DPB, norm, use_dropout_D)
        self.DPP = _Net(sd_DPP, norm, use_dropout_D)
        self.vgg = {k: v.detach().clone() for k, v in vgg.items()}
        self.n_blocks, self.n_layers_D = n_blocks, n_layers_D
        self.lA, self.lB, self.lG = lambda_A, lambda_B, lambda_GAN
        self.DG_ratio = DG_ratio
        self.percep_is_l1 = percep_is_l1
        self.perceptual_layers = perceptual_layers
        self.masks = masks
        self.opt_G = torch.optim.Adam(self.G.parameters(), lr=lr, betas=(beta1, 0.999))
        self.opt_DPB = torch.optim.Adam(self.DPB.parameters(), lr=lr, betas=(beta1, 0.999))
        self.opt_DPP = torch.optim.Adam(self.DPP.parameters(), lr=lr, betas=(beta1, 0.999))
        self.pool_PP = ImagePoolRef(pool_size, rng)
        self.pool_PB = ImagePoolRef(pool_size, rng)
        self.losses = OrderedDict()

    def forward(self, batch):
        g_in = [batch["H1"], torch.cat((batch["P1"], batch["P2"]), 1),
                torch.cat((batch["D1"], batch["D2"]), 1)]
        return generator_forward(self.G, g_in, self.n_blocks, masks=self.masks)

    def _d(self, net, x):
        return discriminator_forward(net, x, self.n_layers_D, masks=self.masks)

    def _d_step(self, net, opt, pool, real, fake_now):
        opt.zero_grad()
        fake = pool.query(fake_now.detach())
        loss = (gan_loss(self._d(net, real), True) * self.lG +
                gan_loss(self._d(net, fake.detach()), False) * self.lG) * 0.5
        loss.backward()
        opt.step()
        return loss.detach()

    def step(self, batch):
        fake = self.forward(batch)
        self.fake_p2 = fake
        self.opt_G.zero_grad()
        g_pb = gan_loss(self._d(self.DPB, torch.cat((fake, batch["P2"]), 1)), True)
        g_pp = gan_loss(self._d(self.DPP, torch.cat((fake, batch["H1"]), 1)), True)
        l_tot, l_l1, l_p = l1_plus_perceptual(self.vgg, fake, batch["H2"], self.lA, self.lB,
                                               self.percep_is_l1, self.perceptual_layers)
        pair_gan = (g_pb * self.lG + g_pp * self.lG) / 2
        (l_tot + pair_gan).backward()
        self.opt_G.step()
        for _ in range(self.DG_ratio):
            d_pp = self._d_step(self.DPP, self.opt_DPP, self.pool_PP,
                                torch.cat((batch["H2"], batch["H1"]), 1),
                                torch.cat((fake, batch["H1"]), 1))
        for _ in range(self.DG_ratio):
            d_pb = self._d_step(self.DPB, self.opt_DPB, self.pool_PB,
                                torch.cat((batch["H2"], batch["P2"]), 1),
                                torch.cat((fake, batch["P2"]), 1))
        self.losses = OrderedDict([
            ("pair_L1loss", float(l_tot.detach())), ("D_PP", float(d_pp)), ("D_PB", float(d_pb)),
            ("pair_GANloss", float(pair_gan.detach())), ("origin_L1", float(l_l1.detach())),
            ("perceptual", float(l_p.detach()))])
        return self.losses


# ----------------------------------------------------------------------------- pose maps
def pose_heatmap(x, y, H, W, sigma=6.0):
    """One Gaussian pose map, float64 maths then fp32 (generic_dataset.py:212-216,239-242)."""
    gridy, gridx = np.mgrid[0:H, 0:W]
    d2 = (gridx - x) ** 2 + (gridy - y) ** 2
    m = np.exp(-d2 / 2.0 / sigma / sigma)
    m[m > 1] = 1
    m[m < 0.0099] = 0
    return m.astype(np.float32)


def pose_heatmaps(uv, H, W, sigma=6.0):
    return np.stack([pose_heatmap(x, y, H, W, sigma) for x, y in uv])


def map_to_cord(pose_map, threshold=0.1):
    """pose_map [H,W,K] -> int [K,2] (y,x) of the first arg-max above threshold, else -1."""
    K = pose_map.shape[-1]
    y, x, z = np.where(np.logical_and(pose_map == pose_map.max(axis=(0, 1)), pose_map > threshold))
    out = -np.ones((K, 2), dtype=np.int64)
    seen = set()
    for yi, xi, zi in zip(y, x, z):
        if zi not in seen:
            seen.add(zi)
            out[zi] = (yi, xi)
    return out


def decode_sample(img_bgr, depth_bgr):
    """uint8 HxWx3 BGR image and depth PNG -> (H [3,H,W], D [3,H,W]) fp32, float64 maths as in
    data/generic_dataset.py:140-158 (normalize, depth = 256*G + R, /700, (.-0.5)/0.5)."""
    rgb = img_bgr[:, :, ::-1].astype(np.float64)
    h = ((rgb / 255.0) - 0.5) / 0.5
    h = torch.tensor(np.ascontiguousarray(h)).permute(2, 0, 1).float()
    d = torch.tensor(256.0 * depth_bgr[:, :, 1] + depth_bgr[:, :, 2])
    d = ((torch.stack([d, d, d]) / 700.0) - 0.5) / 0.5
    return h, d.float()


# ----------------------------------------------------------------------------- synthetic data
def synthetic_batch(B, H, W, seed):
    """SURVEY.md §8(d) synthetic batch: H ~ U(-1,1); P = 21 Gaussian maps at uv ~ U(20,H-20);
    D = one U(-1,1) plane replicated to 3 channels."""
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    out = {}
    for s in ("1", "2"):
        out["H" + s] = torch.rand((B, 3, H, W), generator=g) * 2 - 1
        lo, hi = min(20, H // 4), max(H - 20, 3 * H // 4)
        uv = rs.uniform(lo, hi, size=(B, 21, 2))
        out["P" + s] = torch.from_numpy(np.stack([pose_heatmaps(u, H, W) for u in uv]))
        d = torch.rand((B, 1, H, W), generator=g) * 2 - 1
        out["D" + s] = d.expand(B, 3, H, W).contiguous()
    return out
