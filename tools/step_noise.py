"""How far 3 optimize_parameters() iterations drift from the fp64 oracle at the Winograd test size
(ngf = ndf = 32, 64x64, B=2): HIP path with F(6x6,3x3), HIP path with the direct kernels, and the
oracle's own fp32 run.  Per iteration: rel-L1 of the generated image and the worst loss deviation."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import OrderedDict
import numpy as np
import torch
torch.set_num_threads(16)
from mmhand_amd import ops
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
from oracle import mmhand_ref as O
from oracle import ops_ref as R
NGF, SIZE, NB, NLD = int(os.environ.get("NGF", 32)), int(os.environ.get("SIZE", 64)), 2, 2
B = int(os.environ.get("B", 2))
init = os.environ.get("INIT", "normal")


def run(norm):
    opt = default_train_opt(batchSize=B, ngf=NGF, ndf=NGF, n_layers_D=NLD, G_n_blocks=NB, norm=norm, no_dropout=True,
                            no_dropout_D=True, pool_size=2, name="noise", checkpoints_dir="/tmp/mmh_noise", local_rank=0)
    models = {}
    for tag, wino in (("wino6", True), ("direct", False)):
        ops.USE_WINOGRAD = wino
        ops.bump_weights_epoch()
        models[tag] = MMHandModel(opt)
    sds = [OrderedDict((k, v.cpu()) for k, v in n.state_dict().items())
           for n in (models["wino6"].netG, models["wino6"].netD_PB, models["wino6"].netD_PP)]
    vgg = OrderedDict((k, v.cpu()) for k, v in models["wino6"].vgg.state_dict().items())
    f64 = lambda sd: OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items())
    o64 = O.StepOracle(f64(sds[0]), f64(sds[1]), f64(sds[2]), f64(vgg), norm, False, False, NB, NLD, pool_size=2, rng=random.Random(49))
    o32 = O.StepOracle(sds[0], sds[1], sds[2], vgg, norm, False, False, NB, NLD, pool_size=2, rng=random.Random(49))
    rngs = {t: random.Random(49) for t in models}
    for it in range(3):
        batch = O.synthetic_batch(B, SIZE, SIZE, seed=200 + it)
        w64 = np.array(list(o64.step({k: v.double() for k, v in batch.items()}).values()))
        w32 = np.array(list(o32.step(batch).values()))
        row = [f"{norm} it{it}: oracle32 fake {R.rel_l1(o32.fake_p2.detach(), o64.fake_p2.detach()):.1e} loss {np.abs(w32 / w64 - 1).max():.1e}"]
        for tag, m in models.items():
            ops.USE_WINOGRAD = tag == "wino6"
            ops.bump_weights_epoch()
            random.setstate(rngs[tag].getstate())
            m.set_input(batch)
            m.optimize_parameters()
            rngs[tag].setstate(random.getstate())
            got = np.array([float(v) for v in m.get_current_errors().values()])
            row.append(f"{tag} fake {R.rel_l1(m.fake_p2, o64.fake_p2.detach()):.1e} loss {np.abs(got / w64 - 1).max():.1e}")
        print(" | ".join(row), flush=True)
    # weight drift
    for tag, m in models.items():
        osd = o64.G.state_dict()
        worst = max(((v.cpu().double() - osd[k]).abs().max().item(), k) for k, v in m.netG.state_dict().items() if v.is_floating_point())
        print(f"   {tag}: worst |dw| vs fp64 oracle {worst[0]:.2e} at {worst[1]}")


for norm in ("instance", "batch"):
    run(norm)
