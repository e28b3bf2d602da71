"""What 16-bit mixed precision does to the Generator's gradients on the problem of tests/test_lp16_step_gpu.py, measured
with PyTorch itself: the CPU oracle (oracle/mmhand_ref.py StepOracle, pinned against the reference modules by
make_golden.py) under torch.autocast(bfloat16 | float16) - convolutions in 16 bits, norms / losses in fp32, which is
what apex O1 does to the reference (models/MMHandModel.py:99-108) - against the same oracle in float64.

    python tests/golden/make_lp16_cond.py        (CPU, about a minute: fp16 convolutions are slow on the host)

Writes tests/golden/lp16_cond.npz: per Generator parameter the relative L1 distance of its iteration-1 gradient from
float64 (`bf16/<key>`, `fp16/<key>`, and `fp32/<key>` = the oracle in plain float32), the generated image's
(`bf16/image`, ...) and the six fp64 losses.  The GPU test
holds the HIP 16-bit path to these figures (x 1.2): no 16-bit implementation of this network can be closer to float64
than the arithmetic allows - bf16: median 1.7e-1 per tensor, fp16: 5e-2."""
import os
import random
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mmhand_ref as O          # noqa: E402
from oracle import ops_ref as R             # noqa: E402
from tests.golden import recipe as RC       # noqa: E402

NGF, SIZE, NB, NLD, SEED = 64, 64, 2, 3, 300


def nets():
    """the weights both sides start from: the build's own seeded init on the CPU (weights_init_normal, seeds 49-51)"""
    from mmhand_amd.networks import Discriminator, Generator, VGGHead
    g = Generator([3, 42, 6], 3, NGF, "instance", False, NB).init_weights("normal", 49)
    dpb = Discriminator(24, NGF, "instance", False, NLD).init_weights("normal", 50)
    dpp = Discriminator(6, NGF, "instance", False, NLD).init_weights("normal", 51)
    vgg = VGGHead().init_random()
    return [OrderedDict((k, v.detach().clone()) for k, v in n.state_dict().items()) for n in (g, dpb, dpp, vgg)]


def f64(sd):
    return OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items())


def main():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    sds = nets()
    batch = O.synthetic_batch(2, SIZE, SIZE, seed=SEED)

    def run(dt, ac):
        cv = f64 if dt == torch.float64 else (lambda sd: OrderedDict((k, v.clone()) for k, v in sd.items()))
        o = O.StepOracle(cv(sds[0]), cv(sds[1]), cv(sds[2]), cv(sds[3]), "instance", False, False, NB, NLD, pool_size=2,
                         rng=random.Random(49))
        b = {k: v.to(dt) for k, v in batch.items()}
        if ac is None:
            o.step(b)
        else:
            with torch.autocast("cpu", dtype=ac):
                o.step(b)
        return o

    o64 = run(torch.float64, None)
    ref = dict((k, t.grad) for k, t in o64.G.named_parameters())
    out = {"losses64": np.array(list(o64.losses.values()))}
    # "fp32": PyTorch's own fp32 run of the same step (no autocast) - the conditioning yardstick of the fp32 HIP paths
    for tag, ac in (("bf16", torch.bfloat16), ("fp16", torch.float16), ("fp32", None)):
        o = run(torch.float32, ac)
        out[f"{tag}/image"] = np.float64(R.rel_l1(o.fake_p2.detach().double(), o64.fake_p2.detach()))
        errs = []
        for k, t in o.G.named_parameters():
            if RC.is_null_grad_bias("G", k, "instance") or ref.get(k) is None or t.grad is None:
                continue
            e = R.rel_l1(t.grad.double(), ref[k])
            out[f"{tag}/{k}"] = np.float64(e)
            errs.append(e)
        print(f"{tag}: image {out[tag + '/image']:.2e}, gradients median {np.median(errs):.2e} max {max(errs):.2e}")
    np.savez(os.path.join(ROOT, "tests", "golden", "lp16_cond.npz"), **out)


if __name__ == "__main__":
    main()
