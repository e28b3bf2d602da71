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
