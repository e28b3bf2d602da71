"""Soak of the 16-bit training modes with every fusion on: 400 iterations each of --opt_level O1 (bf16), O1_FP16 and O1 with
--norm batch, a new synthetic batch every 50 iterations (so the packs, their 16-bit twins and the image pools turn over);
prints the six losses every 100 iterations and the number of optimizer steps skipped for overflow.  Round 5: all finite, pair L1
11.5 -> 10.9-11.2, no skipped step.      python tools/soak_16bit.py"""
import os, sys, random
sys.path.insert(0, os.getcwd())
import torch
from bench import synthetic_batch_gpu
from mmhand_amd.mmhand_model import MMHandModel
from mmhand_amd.options import default_train_opt
dev = torch.device("cuda:0")
for level, norm in (("O1", "instance"), ("O1_FP16", "instance"), ("O1", "batch")):
    random.seed(0); torch.manual_seed(0)
    model = MMHandModel(default_train_opt(batchSize=8, norm=norm, name="soak", checkpoints_dir="/tmp/mmh_soak", opt_level=level))
    bad = 0
    for it in range(400):
        if it % 50 == 0:
            model.set_input(synthetic_batch_gpu(8, 256, 256, 49 + it, dev))      # a new batch every 50 iterations
        model.optimize_parameters()
        if it % 100 == 99:
            e = {k: round(float(v), 4) for k, v in model.get_current_errors().items()}
            fin = all(v == v and abs(v) < 1e4 for v in e.values())
            bad += not fin
            print(level, norm, it, e, "skipped", model.skipped_steps, flush=True)
    print(level, norm, "OK" if not bad else "NON-FINITE", "skipped steps:", model.skipped_steps, flush=True)
