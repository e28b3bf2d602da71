"""The overflow-skip rule of MMHandModel.optimize_parameters (models/MMHandModel.py:294-330): a
non-finite gradient makes that optimizer step and every later step of the same iteration a no-op,
decided on the device (mmh_grad_nonfinite -> mmh_adam_step(skip_flag)); skipped steps do not
advance the Adam step count; and the optimizer state survives save / --continue_train."""
import os
import random

import pytest
import torch

from oracle import mmhand_ref as O
from tests.golden import recipe as RC
from tests.test_model_gpu import _small_opt

pytestmark = pytest.mark.gpu
S = RC.SMALL


def _batch(seed):
    return O.synthetic_batch(S["B"], S["H"], S["W"], seed=seed)


def test_grad_nonfinite_flag_and_guarded_adam(dev):
    from mmhand_amd import ops
    n = 100003                      # not a multiple of 4: exercises the scalar tail
    g = torch.randn(n + 1, device=dev)[:n]      # keep 16-byte alignment of the base
    flags = torch.zeros(3, dtype=torch.int32, device=dev)
    ops.grad_nonfinite(g, flags[0:1])
    assert flags.tolist() == [0, 0, 0]
    for pos, val in ((n - 1, float("inf")), (0, float("nan")), (n // 2, float("-inf"))):
        gg = g.clone()
        gg[pos] = val
        ops.grad_nonfinite(gg, flags[1:2], flags[0:1])
        assert flags.tolist()[:2] == [0, 1], (pos, val)
    ops.grad_nonfinite(g, flags[2:3], flags[1:2])       # sticky: finite gradient, flag carried over
    assert flags.tolist() == [0, 1, 1]
    p = torch.randn(n, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
    p0 = p.clone()
    ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 1, 1.0, flags[1:2])
    assert torch.equal(p, p0) and not m.any() and not v.any()
    ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 1, 1.0, flags[0:1])
    assert not torch.equal(p, p0) and m.any() and v.any()


@pytest.mark.parametrize("where", ["G", "D_PP"])
def test_overflow_skips_this_and_later_steps_of_the_iteration(where, dev):
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(49)
    model = MMHandModel(_small_opt("instance"))
    model.set_input(_batch(100))
    model.optimize_parameters()                       # a clean iteration first
    snap = {n: getattr(model, n).flat_param.clone() for n in ("netG", "netD_PP", "netD_PB")}
    steps = [o.step_count for o in model.optimizers]
    assert steps == [1, 1, 1]

    target = "backward_G" if where == "G" else "backward_D_PP"
    net = model.netG if where == "G" else model.netD_PP
    orig = getattr(model, target)

    def poisoned():
        orig()
        net.flat_grad[7] = float("inf")
    setattr(model, target, poisoned)
    model.set_input(_batch(101))
    model.optimize_parameters()
    setattr(model, target, orig)
    changed = {n: not torch.equal(getattr(model, n).flat_param, snap[n]) for n in snap}
    # order of the reference's step: G, then D_PP, then D_PB; the flag is sticky
    assert changed == ({"netG": False, "netD_PP": False, "netD_PB": False} if where == "G" else
                       {"netG": True, "netD_PP": False, "netD_PB": False})
    for n in snap:
        assert torch.isfinite(getattr(model, n).flat_param).all()

    snap2 = {n: getattr(model, n).flat_param.clone() for n in snap}
    model.set_input(_batch(102))
    model.optimize_parameters()
    # flags are settled one iteration late (the host may run ahead of the GPU); drain them now
    model._settle_overflow(drain=True)
    assert model.skipped_steps == (3 if where == "G" else 2)
    # skipped steps were taken back: G, D_PB, D_PP in model.optimizers order
    want = [2, 2, 2] if where == "G" else [3, 2, 2]
    assert [o.step_count for o in model.optimizers] == want
    for n in snap2:
        assert not torch.equal(getattr(model, n).flat_param, snap2[n])     # training goes on
    model.set_input(_batch(103))
    model.optimize_parameters()
    model._settle_overflow(drain=True)
    assert not model.last_overflow and model.skipped_steps == (3 if where == "G" else 2)


def test_optimizer_state_checkpoint_roundtrip(dev, tmp_path):
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(49)
    opt = _small_opt("instance", checkpoints_dir=str(tmp_path))
    m = MMHandModel(opt)
    for it in range(2):
        m.set_input(_batch(200 + it))
        m.optimize_parameters()
    m.save("latest")
    assert os.path.exists(os.path.join(str(tmp_path), opt.name, "latest_net_amp.pth"))
    m2 = MMHandModel(_small_opt("instance", checkpoints_dir=str(tmp_path), continue_train=True))
    for a, b in zip(m.optimizers, m2.optimizers):
        assert b.step_count == a.step_count == 2
        assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    # the resumed run continues exactly like the original one (same inputs, same pool draws)
    state = random.getstate()
    m2.fake_PP_pool.images = [t.clone() for t in m.fake_PP_pool.images]
    m2.fake_PB_pool.images = [t.clone() for t in m.fake_PB_pool.images]
    m.set_input(_batch(300)); m.optimize_parameters()
    random.setstate(state)
    m2.set_input(_batch(300)); m2.optimize_parameters()
    for n in ("netG", "netD_PP", "netD_PB"):
        assert torch.equal(getattr(m, n).flat_param, getattr(m2, n).flat_param), n


def test_loss_scale_update_kernel(dev):
    """apex LossScaler.update_scale on the device: overflow -> scale * 0.5, clean count reset;
    `window` clean steps -> scale * 2; bounded by [1, 2^24]."""
    from mmhand_amd import ops
    st = torch.tensor([65536.0, 0.0], device=dev)
    yes = torch.ones(1, dtype=torch.int32, device=dev); no = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.loss_scale_update(st, no, window=3); assert st.tolist() == [65536.0, 1.0]
    ops.loss_scale_update(st, yes, window=3); assert st.tolist() == [32768.0, 0.0]
    for _ in range(3):
        ops.loss_scale_update(st, no, window=3)
    assert st.tolist() == [65536.0, 0.0]
    st[0] = 1.0
    ops.loss_scale_update(st, yes, window=3); assert st.tolist() == [1.0, 0.0]       # floor
    st[0] = 2.0 ** 24; st[1] = 2.0
    ops.loss_scale_update(st, no, window=3); assert st.tolist() == [2.0 ** 24, 0.0]  # ceiling
    # Adam unscales with the device-resident scale: same update as an unscaled gradient
    g = torch.randn(4096, device=dev)
    p1 = torch.randn(4096, device=dev); p2 = p1.clone()
    m1 = torch.zeros_like(p1); v1 = torch.zeros_like(p1); m2 = torch.zeros_like(p1); v2 = torch.zeros_like(p1)
    sc = torch.tensor([1024.0, 0.0], device=dev)
    ops.adam_step(p1, g * 1024.0, m1, v1, 2e-4, 0.5, 0.999, 1e-8, 1, 1.0, None, sc)
    ops.adam_step(p2, g, m2, v2, 2e-4, 0.5, 0.999, 1e-8, 1)
    assert torch.allclose(p1, p2, rtol=0, atol=1e-9) and torch.allclose(v1, v2, rtol=1e-6, atol=0)


def test_opt_level_O1_dynamic_loss_scaling(dev):
    """--opt_level O1 (the reference's shipped default, scripts/mm-train-ratio.sh:7): apex's dynamic
    loss scaling around the three backward passes.  A clean iteration leaves the scales at 2^16 and
    counts one clean step each; a forced overflow in the generator's gradient halves the
    generator's scale, skips ALL THREE optimizer steps of that iteration (models/MMHandModel.py:
    316-328: `self.overflow` is sticky) while the two discriminator scalers - whose own backward
    passes were clean - count a clean step, exactly as apex updates each loss's scaler when its
    amp.scale_loss context exits; the scales grow again after `window` clean steps; the scaler state
    travels in <label>_net_amp.pth, the file apex's amp.state_dict() fills in the reference."""
    from mmhand_amd.mmhand_model import MMHandModel
    random.seed(49)
    opt = _small_opt("instance", opt_level="O1", checkpoints_dir="/tmp/mmh_pytest_amp", name="amp")
    model = MMHandModel(opt)
    assert model.loss_scaling and model.bf16
    ref = MMHandModel(_small_opt("instance", opt_level="BF16"))
    for n in ("netG", "netD_PB", "netD_PP"):
        getattr(ref, n).load_state_dict(getattr(model, n).state_dict())
    assert not ref.loss_scaling
    st = random.getstate()
    model.set_input(_batch(100)); model.optimize_parameters()
    random.setstate(st)
    ref.set_input(_batch(100)); ref.optimize_parameters()
    assert model._scaler.tolist() == [[65536.0, 1.0]] * 3
    # scaling by 2^16 and unscaling inside Adam is exact in fp32 (a power of two): same weights
    for n in ("netG", "netD_PB", "netD_PP"):
        assert torch.allclose(getattr(model, n).flat_param, getattr(ref, n).flat_param, rtol=0, atol=2e-6), n
    a = [float(v) for v in model.get_current_errors().values()]
    b = [float(v) for v in ref.get_current_errors().values()]
    assert a == pytest.approx(b, rel=1e-5)                      # reported losses are unscaled

    snap = {n: getattr(model, n).flat_param.clone() for n in ("netG", "netD_PP", "netD_PB")}
    orig = model.backward_G

    def poisoned():
        orig()
        model.netG.flat_grad[11] = float("inf")
    model.backward_G = poisoned
    model.set_input(_batch(101)); model.optimize_parameters()
    model.backward_G = orig
    assert all(torch.equal(getattr(model, n).flat_param, snap[n]) for n in snap)       # all three skipped
    assert model._scaler.tolist() == [[32768.0, 0.0], [65536.0, 2.0], [65536.0, 2.0]]
    assert model.loss_scale(0) == 32768.0

    model.loss_scale_window = 3                                  # grow after 3 clean steps
    model.set_input(_batch(102)); model.optimize_parameters()
    assert model._scaler.tolist() == [[32768.0, 1.0], [131072.0, 0.0], [131072.0, 0.0]]
    model._settle_overflow(drain=True)
    assert model.skipped_steps == 3
    model.save("latest")
    opt2 = _small_opt("instance", opt_level="O1", checkpoints_dir="/tmp/mmh_pytest_amp", name="amp",
                      continue_train=True)
    m2 = MMHandModel(opt2)
    assert m2._scaler.tolist() == model._scaler.tolist() and m2.skipped_steps == 3
    with pytest.raises(ValueError):
        MMHandModel(_small_opt("instance", opt_level="O3"))


def test_opt_level_O1_FP16_natural_overflow_and_recovery(dev):
    """--opt_level O1_FP16: IEEE fp16 MFMA operands, i.e. apex O1's own numerics.  (1) three
    iterations track the fp32 run within the mixed-precision tolerance (2 %); (2) with a loss scale
    far too large the scaled fp16 gradients overflow BY THEMSELVES (no injected inf): every step of
    the iteration is skipped and each overflowing loss halves its scale - iterating drives the
    scale down until steps land again, which is how apex finds its operating point."""
    import numpy as np

    from mmhand_amd.mmhand_model import MMHandModel
    rows = {}
    models = {}
    for level in ("O0", "O1_FP16"):
        random.seed(49)
        m = MMHandModel(_small_opt("instance", opt_level=level, ngf=16, ndf=16))
        if models:
            for n in ("netG", "netD_PB", "netD_PP"):
                getattr(m, n).load_state_dict(getattr(models["O0"], n).state_dict())
        else:
            init = {n: getattr(m, n).state_dict() for n in ("netG", "netD_PB", "netD_PP")}
        models[level] = m
    for n in ("netG", "netD_PB", "netD_PP"):
        models["O0"].__getattr__(n).load_state_dict(init[n])
        models["O1_FP16"].__getattr__(n).load_state_dict(init[n])
    assert models["O1_FP16"].bf16 == 2 and models["O1_FP16"].loss_scaling
    for level, m in models.items():
        random.seed(49)
        out = []
        for it in range(3):
            m.set_input(_batch(100 + it))
            m.optimize_parameters()
            out.append([float(v) for v in m.get_current_errors().values()])
        rows[level] = np.array(out)
    assert np.allclose(rows["O1_FP16"], rows["O0"], rtol=2e-2), (rows["O1_FP16"], rows["O0"])
    assert not np.array_equal(rows["O1_FP16"], rows["O0"])
    m = models["O1_FP16"]
    m._settle_overflow(drain=True)
    skipped0 = m.skipped_steps
    m._scaler[:, 0] = 2.0 ** 40                      # scaled fp16 gradients overflow on their own
    snap = m.netG.flat_param.clone()
    m.set_input(_batch(200)); m.optimize_parameters()
    assert torch.equal(m.netG.flat_param, snap) and torch.isfinite(m.netG.flat_param).all()
    assert m._scaler[0, 0].item() == 2.0 ** 39
    for it in range(40):                             # back off until the steps land again
        m.set_input(_batch(201 + it)); m.optimize_parameters()
        if not torch.equal(m.netG.flat_param, snap):
            break
    assert not torch.equal(m.netG.flat_param, snap) and torch.isfinite(m.netG.flat_param).all()
    m._settle_overflow(drain=True)
    assert m.skipped_steps > skipped0 and 2.0 ** 8 <= m.loss_scale(0) < 2.0 ** 40
