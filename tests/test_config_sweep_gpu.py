"""Option combinations that move the 16-bit edges around - full-width channels (ngf 64) on small
images, so the conv_lp16 kernels run at 16x16 (halo tiles) and 12x12 (row-tile fallback) feature maps;
0 / 1 / 2 PATBlocks, dropout on / off, DG_ratio 1 / 2, both norms: every combination trains two
iterations in fp32, bf16 and fp16 mode, and the six losses of the 16-bit modes stay next to fp32's."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
CFGS = [(64, 2, False, 1), (64, 1, True, 2), (48, 2, True, 1), (64, 0, False, 1)]


@pytest.mark.parametrize("norm", ["instance", "batch"])
@pytest.mark.parametrize("cfg", CFGS, ids=lambda c: "s%d_b%d_%s_r%d" % (c[0], c[1], "nodrop" if c[2] else "drop", c[3]))
def test_option_combinations_train_in_every_precision(cfg, norm, dev):
    from bench import synthetic_batch_gpu
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    size, nblk, no_drop, ratio = cfg
    res = {}
    for level in ("O0", "O1", "O1_FP16"):
        torch.manual_seed(49)
        random.seed(49)
        opt = default_train_opt(batchSize=2, norm=norm, name="sweep", checkpoints_dir="/tmp/mmh_sweep", opt_level=level,
                                G_n_blocks=nblk, no_dropout=no_drop, DG_ratio=ratio, pool_size=3)
        m = MMHandModel(opt)
        ops.set_dropout_seed(4949)
        m.set_input(synthetic_batch_gpu(2, size, size, 7, dev))
        m.optimize_parameters()
        first = {k: float(v) for k, v in m.get_current_errors().items()}
        m.optimize_parameters()
        second = {k: float(v) for k, v in m.get_current_errors().items()}
        assert all(v == v and abs(v) < 1e4 for v in list(first.values()) + list(second.values())), (level, first, second)
        assert not ops._lp_grads                      # no 16-bit gradient left parked
        res[level] = first
        del m
    for level, tol in (("O1", 2e-2), ("O1_FP16", 5e-3)):
        for k, v in res["O0"].items():
            assert abs(res[level][k] - v) <= tol * max(1.0, abs(v)), (level, k, res[level][k], v)
