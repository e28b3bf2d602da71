"""Generator gradients of the 16-bit mode against the fp64 oracle (the problem of tests/test_lp16_step_gpu.py) with parts
of the 16-bit path switched off: median / max relative L1 per tensor, beside PyTorch's autocast figures of the fixture.

    python tools/lp16_grad_probe.py [O1_FP16|O1]"""
import os, sys, random, statistics
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(16)
from oracle import mmhand_ref as O
from oracle import ops_ref as R
from tests.golden import recipe as RC
from tests.golden.make_lp16_cond import SEED, nets, NGF, SIZE, NB, NLD
from tests.test_model_gpu import logical_grads
from mmhand_amd import ops
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
cond = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lp16_cond.npz")))
sds = nets()
f64 = lambda sd: OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items())
o64 = O.StepOracle(f64(sds[0]), f64(sds[1]), f64(sds[2]), f64(sds[3]), "instance", False, False, NB, NLD, pool_size=2, rng=random.Random(49))
batch = O.synthetic_batch(2, SIZE, SIZE, seed=SEED)
o64.step({k: v.double() for k, v in batch.items()})
og = dict((k, t.grad) for k, t in o64.G.named_parameters())
LEVEL = sys.argv[1] if len(sys.argv) > 1 else "O1_FP16"
tag = "fp16" if LEVEL.endswith("FP16") else "bf16"


def run(vname, level, scale):
    opt = default_train_opt(batchSize=2, ngf=NGF, ndf=NGF, n_layers_D=NLD, G_n_blocks=NB, norm="instance", no_dropout=True,
                            no_dropout_D=True, pool_size=2, name="probe", checkpoints_dir="/tmp/mmh_probe", local_rank=0,
                            fineSize=SIZE, opt_level=level)
    model = MMHandModel(opt)
    for net, sd in zip((model.netG, model.netD_PB, model.netD_PP, model.vgg), sds):
        net.load_state_dict(sd)
    if level != "O0":
        model._scaler[:, 0] = scale
    model.set_input(batch)
    model.forward()
    model.optimizer_G.zero_grad()
    model.backward_G()
    e_img = R.rel_l1(model.fake_p2, o64.fake_p2.detach())
    gg = logical_grads(model.netG)
    errs = [(k, R.rel_l1(g.double() / scale, og[k])) for k, g in gg.items()
            if not RC.is_null_grad_bias("G", k, "instance") and og.get(k) is not None]
    med = statistics.median(e for _, e in errs)
    d = dict(errs)
    print(f"[{vname:16s}] {level:8s} image {e_img:.1e} grads median {med:.2e} max {max(e for _, e in errs):.2e} | head bias "
          f"{d['model.stream1_up.7.bias']:.1e} head w {d['model.stream1_up.7.weight']:.1e} up.3 {d['model.stream1_up.3.weight']:.1e} "
          f"up.0 {d['model.stream1_up.0.weight']:.1e}  (autocast median "
          f"{statistics.median(float(cond[tag + '/' + k]) for k, _ in errs):.2e})", flush=True)


for vname, toggles in [("default", {}), ("edges off", {"USE_LP16_EDGES": False}), ("v2 kernels off", {"USE_LP16_V2": False}),
                       ("thin off", {"USE_THIN": False}), ("edges+v2 off", {"USE_LP16_EDGES": False, "USE_LP16_V2": False})]:
    for k, v in toggles.items():
        setattr(ops, k, v)
    run(vname, LEVEL, 1024.0)
    for k in toggles:
        setattr(ops, k, True)
run("fp32", "O0", 1.0)
