"""Two shortcuts of the backward pass that must not change what the reference computes:
 * ops.ACCUM_PARAM_GRADS - weight / bias gradients added straight into the parameters' .grad views of
   the flat gradient buffer by the wgrad kernels (what autograd's AccumulateGrad does in a separate add
   kernel; MMHandModel.optimize_parameters turns it on for single-process training);
 * ops.EXACT_NULL_BIAS_GRAD - conv biases that feed an InstanceNorm get their exact gradient, zero,
   instead of the rounding noise a reduction pass produces (the reference's autograd included)."""
import pytest
import torch

from tests.test_model_gpu import _small_opt
from oracle import mmhand_ref as O

pytestmark = pytest.mark.gpu


def _batch(seed):
    from tests.golden import recipe as RC
    S = RC.SMALL
    return O.synthetic_batch(S["B"], S["H"], S["W"], seed=seed)


def _grads_of_one_backward(model, accum, monkeypatch):
    from mmhand_amd import ops
    import random
    monkeypatch.setattr(ops, "ACCUM_PARAM_GRADS", accum)
    ops.set_dropout_seed(1234)
    random.seed(7)                                  # ImagePool draws from Python's RNG
    model.fake_PB_pool.images.clear()
    model.fake_PP_pool.images.clear()
    model.forward()
    out = []
    for opt_, fn in ((model.optimizer_G, model.backward_G), (model.optimizer_D_PP, model.backward_D_PP),
                     (model.optimizer_D_PB, model.backward_D_PB)):
        opt_.zero_grad()
        fn()
        out.append(opt_.net.flat_grad.clone())
    return out


@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("level", ["O0", "O1"])
def test_in_place_parameter_gradients_equal_accumulate_grad(norm, level, dev, monkeypatch):
    from mmhand_amd.mmhand_model import MMHandModel
    torch.manual_seed(0)
    model = MMHandModel(_small_opt(norm, opt_level=level))
    model.set_input(_batch(5))
    ref = _grads_of_one_backward(model, False, monkeypatch)
    got = _grads_of_one_backward(model, True, monkeypatch)
    for a, b in zip(ref, got):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b)


def test_null_bias_gradients_are_noise_and_returned_as_exact_zero(dev, monkeypatch):
    """With MMH_NULL_BIAS_GRAD=compute the gradient of a conv bias in front of an InstanceNorm is a sum
    that cancels to rounding noise (|db| <= 1e-5 x the sum of |g| it is made of); the default returns
    the exact zero; no other gradient, output or loss moves."""
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import MMHandModel
    from tests.golden import recipe as RC
    from tests.test_model_gpu import logical_grads
    torch.manual_seed(0)
    model = MMHandModel(_small_opt("instance"))
    model.set_input(_batch(5))
    res = {}
    for exact in (False, True):
        monkeypatch.setattr(ops, "EXACT_NULL_BIAS_GRAD", exact)
        ops.set_dropout_seed(77)
        model.forward()
        model.optimizer_G.zero_grad()
        model.backward_G()
        res[exact] = ({k: v.clone() for k, v in logical_grads(model.netG).items()}, model.fake_p2.clone(),
                      float(model.pair_L1loss), float(model.pair_GANloss))
    (g0, out0, l0, a0), (g1, out1, l1, a1) = res[False], res[True]
    assert torch.equal(out0, out1) and l0 == l1 and a0 == a1
    wmax = max(float(v.abs().max()) for k, v in g0.items() if k.endswith(".weight"))
    n_null = 0
    for k in g0:
        if RC.is_null_grad_bias("G", k, "instance"):
            n_null += 1
            assert float(g1[k].abs().max()) == 0.0, k                  # exact
            assert float(g0[k].abs().max()) < 1e-4 * wmax, (k, float(g0[k].abs().max()), wmax)     # noise
        else:
            assert torch.equal(g0[k], g1[k]), k
    assert n_null > 10
