"""BASELINE.json configs that no other test runs at their real shapes:

  configs[3]  STB 256x256, batch 64, inference-only Generator (BN folded, hipGraph-captured)
  configs[4]  RHD 512x512, per-GPU batch 4, bf16 (128x128 feature maps, channels 256 / 512)

The CPU oracle cannot run B=64 / 512^2 at full width in seconds, so the full shapes are checked
through properties of the HIP path itself (folded == unfolded, graph replay == eager, adjoint
identities) plus an oracle spot check on ONE sample."""
import pytest
import torch

from oracle import mmhand_ref as O
from oracle import ops_ref as R

pytestmark = pytest.mark.gpu


def _eval_sd(net, seed=1):
    """init_weights + non-trivial BN running statistics (as a trained checkpoint would have)."""
    sd = net.state_dict()
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


def test_config3_inference_b64_full_shape(dev):
    """Full-size Generator (ngf 64, 9 PATBlocks, BatchNorm, use_dropout=True as aug.py:31-39 builds
    it), B=64, 256x256: hipGraph replay of the BN-folded forward == the unfolded eval-mode forward
    on the HIP path (5e-4: folding moves gamma/sqrt(var) into the weights, a different rounding of
    the same arithmetic), a second replay on new inputs is consistent, and sample 0 matches the
    eval-mode CPU oracle at 1e-3."""
    from bench import synthetic_batch_gpu
    from mmhand_amd.inference import InferenceGenerator
    from mmhand_amd.networks import Generator
    B = 64
    net = Generator([3, 42, 6], 3, 64, "batch", True, 9).init_weights("normal", 49)
    sd = _eval_sd(net)
    net.load_state_dict(sd)
    net.to(dev).eval()
    gen = InferenceGenerator(net, use_graph=True)
    for seed in (49, 50):
        b = synthetic_batch_gpu(B, 256, 256, seed, dev)
        g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
        out = gen(g_in).clone()
        assert tuple(out.shape) == (B, 3, 256, 256) and bool(torch.isfinite(out).all())
        with torch.no_grad():
            ref_hip = net(g_in)                     # unfolded: conv -> affine(running stats) -> ReLU
        assert R.rel_l1(out, ref_hip) < 5e-4, R.rel_l1(out, ref_hip)
    onet = O._Net(sd, "batch", True)
    onet.training = False
    ref = O.generator_forward(onet, [t[:1].cpu() for t in g_in], 9)
    assert R.rel_l1(out[:1], ref) < 1e-3, R.rel_l1(out[:1], ref)


@pytest.mark.parametrize("lp,tol", [(True, 1.5e-2), (2, 2.5e-3)], ids=["bf16", "fp16"])
def test_config5_generator_512_bf16_vs_oracle(lp, tol, dev):
    """configs[4] shapes on the 16-bit path: ngf 64 at 512x512 -> PATBlocks at 128x128 with 256 / 512
    channels (conv_lp16.hip for every 3x3 conv and the stems' fprop).  B=1, 2 PATBlocks (the oracle runs
    on the CPU).  Every tensor that faces a convolution is stored in 16 bits, as under apex O1, so each
    conv output carries one rounding to the storage type on top of the 16-bit operands: the output
    stays within 1.5e-2 rel-L1 of the fp64 oracle in bf16 (8 significand bits; measured 1.04e-2, and
    8e-3 with fp32 storage between the convs) and within 2.5e-3 in fp16 (11 bits - apex's own type).  The
    backward runs on the 16-bit dgrad / wgrad kernels and its weight gradients point the oracle's way
    (cosine > 0.97 on every conv weight; worst on the pose stem, whose input is sparse)."""
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator
    NB = 2
    net = Generator([3, 42, 6], 3, 64, "instance", False, NB).init_weights("normal", 49)
    sd = net.state_dict()
    net.to(dev).train()
    net.flatten_parameters()
    net.bf16 = lp
    assert ops.lp16_v2_ok(512, 512, 3, 1, 1, 0) and ops._wino_tile(1, 128, 128, 512, 512, 3, 1, 1, True, "fprop") == 0
    b = O.synthetic_batch(1, 512, 512, seed=3)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(3))
    out = net([t.to(dev) for t in g_in])
    (out * probe.to(dev)).sum().backward()
    onet = O._Net(sd, "instance", False)
    ref = O.generator_forward(onet, g_in, NB)
    (ref * probe).sum().backward()
    err = R.rel_l1(out, ref.detach())
    assert 1e-6 < err < tol, err                      # > 1e-6: the 16-bit kernels really ran
    from tests.test_model_gpu import logical_grads
    og = dict((k, t.grad) for k, t in onet.named_parameters())
    worst = 1.0
    for k, g in logical_grads(net).items():
        if k.endswith(".weight") and g.dim() == 4:
            c = torch.nn.functional.cosine_similarity(g.flatten().double(), og[k].flatten().double(), dim=0).item()
            worst = min(worst, c)
            assert c > 0.97, (k, c)
    print(f"\n512x512 {'fp16' if lp == 2 else 'bf16'} generator: out rel-L1 {err:.2e}, worst weight-gradient cosine {worst:.4f}")


CFG5 = [
    # B, H, W, Cin, Cout, k, stride, pad, reflect      (512x512 inputs, per-GPU batch 4)
    (4, 128, 128, 512, 512, 3, 1, 1, True),
    (4, 128, 128, 512, 256, 3, 1, 1, True),
    (4, 128, 128, 256, 256, 3, 1, 1, True),
    (4, 256, 256, 128, 256, 3, 2, 1, False),
    (4, 512, 512, 44, 64, 7, 1, 3, True),
]


@pytest.mark.parametrize("case", CFG5)
@pytest.mark.parametrize("bf16", [False, True])
def test_config5_conv_adjoint_identities(case, bf16, dev):
    """<conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)> at the configs[4] tensor shapes."""
    from mmhand_amd import ops
    from tests.test_fullsize_gpu import _dot, _rand
    B, H, W, Cin, Cout, k, s, p, refl = case
    x = _rand((B, H, W, Cin), 1, dev)
    w = _rand((k, k, Cin, Cout), 2, dev, 0.05)
    ops.bump_weights_epoch()
    y = ops.raw_conv_fprop(x, w, None, s, p, refl, 0, bf16=bf16)
    dy = _rand(tuple(y.shape), 3, dev)
    dx = ops.raw_conv_dgrad(dy, w, x.shape, s, p, refl, bf16=bf16)
    dw = ops.raw_conv_wgrad(x, dy, k, s, p, refl, bf16=bf16)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    # bf16: each pass rounds its own copy of the operands (the fprop and dgrad filters are rounded in
    # their own Winograd domains), so the three inner products differ by the bf16 rounding of w
    tol = 5e-3 if bf16 else 2e-5
    scale = max(abs(a), (y.double().abs() * dy.double().abs()).sum().item() * 1e-3)
    assert abs(a - b) / scale < tol and abs(a - c) / scale < tol, (a, b, c)


def test_vgg_weights_policy(dev, tmp_path):
    """the default --L1_type l1_plus_perL1 must not silently train against random VGG
    weights; torchvision-style 'features.N.*' keys load."""
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    from tests.golden import recipe as RC
    kw = dict(batchSize=1, ngf=8, ndf=8, n_layers_D=1, G_n_blocks=1, norm="instance", name="vggpol",
              checkpoints_dir=str(tmp_path), local_rank=0)
    with pytest.raises(RuntimeError, match="vgg_weights"):
        MMHandModel(default_train_opt(vgg_random_init=False, **kw))
    with pytest.raises(ValueError, match="perceptual_layers"):
        MMHandModel(default_train_opt(perceptual_layers=37, **kw))         # vgg19.features has indices 0..36
    tv = {"features." + k: v for k, v in RC.vgg_recipe().items()}
    tv["features.5.weight"] = torch.zeros(128, 64, 3, 3)           # later layers are ignored
    tv["classifier.0.weight"] = torch.zeros(8, 8)
    path = str(tmp_path / "vgg19.pth")
    torch.save(tv, path)
    m = MMHandModel(default_train_opt(vgg_random_init=False, vgg_weights=path, **kw))
    assert m.vgg_source.startswith("file:")
    for k, v in RC.vgg_recipe().items():
        assert torch.equal(m.vgg.state_dict()[k].cpu(), v), k
    m2 = MMHandModel(default_train_opt(**kw))
    assert "random" in m2.vgg_source


def test_optimize_parameters_mse_perceptual_vs_oracle(dev):
    """--percep_is_l1 0: 2 iterations against the StepOracle with the MSE perceptual branch."""
    import random
    from collections import OrderedDict

    import numpy as np

    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    opt = default_train_opt(batchSize=2, ngf=8, ndf=8, n_layers_D=2, G_n_blocks=2, norm="instance",
                            no_dropout=True, no_dropout_D=True, pool_size=2, name="mse",
                            checkpoints_dir="/tmp/mmh_pytest_ckpt", local_rank=0, percep_is_l1=0)
    model = MMHandModel(opt)
    sds = [OrderedDict((k, v.cpu()) for k, v in n.state_dict().items())
           for n in (model.netG, model.netD_PB, model.netD_PP)]
    vgg = OrderedDict((k, v.cpu()) for k, v in model.vgg.state_dict().items())
    orc = O.StepOracle(sds[0], sds[1], sds[2], vgg, "instance", False, False, 2, 2, pool_size=2,
                       rng=random.Random(1), percep_is_l1=0)
    random.seed(1)
    for it in range(2):
        batch = O.synthetic_batch(2, 32, 32, seed=60 + it)
        model.set_input(batch)
        model.optimize_parameters()
        got = [float(v) for v in model.get_current_errors().values()]
        want = list(orc.step(batch).values())
        assert np.allclose(got, want, rtol=1e-3), (it, got, want)


def test_bench_bf16_side_run_with_stack_meter(dev):
    """bench.py's `bf16_path` side key (the driver-visible 16-bit numbers): one bracketed step of the full-width model at
    128x128 (32x32 feature maps, so the halo kernel with the reflect fold, its statistics epilogue and the nine-tap wgrad
    all run) goes through StackMeter without an error - it once indexed ("fprop", "dgrad") with the dgrad's mode 2 - and
    reports every pass of the 256- and 512-channel convs, the dgrad WITHOUT border launches."""
    import bench
    out = bench.side_train_run(dev, 2, 128, 1, warmup=1, stack=True, stack_steps=1, opt_level="O1")
    assert "error" not in out and out["losses_finite"] and out["stack_frac"] > 0
    per = out["per_pass"]
    for k in ("fprop_256x256", "dgrad_256x256", "wgrad_256x256", "fprop_512x512", "dgrad_512x512", "wgrad_512x512"):
        assert per[k]["launches"] > 0 and per[k]["tflops"] > 0, (k, per)
    assert not any(k.startswith("border") for k in per), per
    # every 256/512-channel 3x3 conv is counted once per pass whichever entry point ran it (plain, with the statistics
    # epilogue, with the residual gradient added): fprop launches == dgrad launches == wgrad launches + frozen-D passes
    assert per["fprop_256x256"]["launches"] == per["dgrad_256x256"]["launches"], per
    assert out["roofline"]["frac"] > 0


def test_bench_norm_batch_o1_side_run_reports_host_enqueue(dev):
    """bench.py's `norm_batch_o1` side key (VERDICT r4 #5a: the reference's shipped --norm batch + --opt_level O1,
    scripts/mm-train-ratio.sh:7-40) and the host-side figures every side region now carries (#5d): the wall time of the
    optimize_parameters() CALL with the GPU idle when it starts, and the C-ABI calls it makes."""
    import bench
    out = bench.side_train_run(dev, 2, 64, 1, warmup=1, norm="batch", opt_level="O1")
    assert "error" not in out and out["losses_finite"] and out["images_per_s"] > 0
    assert out["host_enqueue_ms"] > 0 and out["c_abi_calls_per_step"] > 100
    assert abs(out["host_enqueue_over_step"] - out["host_enqueue_ms"] / out["ms_per_step"]) < 2e-3
    # a kernel-selection switch applied for one region only is restored behind it
    from mmhand_amd import lib as L
    out2 = bench.side_train_run(dev, 2, 64, 1, warmup=1, opt_level="O1", lib_options={"lp16_persist": (0, 1)})
    assert out2["losses_finite"]
    L.call("mmh_set_option", b"lp16_persist", 1)
