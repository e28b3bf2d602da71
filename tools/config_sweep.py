"""Robustness sweep: two optimize_parameters() of small-image models over option combinations that move the
16-bit edges around (full-width channels so that the conv_lp16 paths are taken at 16x16 / 8x8 feature maps)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
from bench import synthetic_batch_gpu
dev = torch.device("cuda:0")
n = 0
for level, norm, size, nblk, drop, ratio in itertools.product(("O0", "O1", "O1_FP16"), ("instance", "batch"), (64, 48), (0, 1, 2),
                                                              (False, True), (1, 2)):
    if (size, nblk, drop, ratio) not in ((64, 2, False, 1), (64, 1, True, 2), (48, 2, True, 1), (64, 0, False, 1)):
        continue
    opt = default_train_opt(batchSize=2, norm=norm, name="sweep", checkpoints_dir="/tmp/mmh_sweep", opt_level=level,
                            G_n_blocks=nblk, no_dropout=drop, DG_ratio=ratio, pool_size=3)
    m = MMHandModel(opt)
    m.set_input(synthetic_batch_gpu(2, size, size, 7, dev))
    for _ in range(2):
        m.optimize_parameters()
    errs = {k: float(v) for k, v in m.get_current_errors().items()}
    assert all(e == e and abs(e) < 1e6 for e in errs.values()), (level, norm, size, nblk, drop, ratio, errs)
    n += 1
    print(level, norm, size, nblk, drop, ratio, "ok", {k: round(v, 3) for k, v in errs.items()}, flush=True)
    del m
    torch.cuda.empty_cache()
print("configs run:", n)
