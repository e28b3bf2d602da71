"""Training driver — counterpart of the reference's train.py:12-65 on the HIP path.

    python -m mmhand_amd.train --name run --batchSize 32 --norm instance --niter 1 --niter_decay 0
    python -m torch.distributed.run --nproc-per-node 8 -m mmhand_amd.train ... --distributed

Same loop shape (epochs x batches, set_input + optimize_parameters, print/save cadence in
*samples*, update_learning_rate per epoch) and the same loss_log.txt line format
(util/visualizer.py:116-123).  Data: with `--dataroot DIR --dataset rhd|stb` the reference's prepared directory
(annotation.pickle + colour / depth PNGs; data.HandFolderLoader, decoded on the device), else synthetic RHD/STB-shaped
batches (`--synthetic_samples N` sets the epoch length)."""
import os
import sys
import time

import torch

from .data import make_loader
from .mmhand_model import MMHandModel
from .options import TrainOptions


def main(argv=None):
    o = TrainOptions()
    o.initialize()
    o.parser.add_argument("--synthetic_samples", type=int, default=256)
    opt = o.parse(argv)
    if opt.batchSize is None:
        opt.batchSize = 1
    torch.cuda.set_device(opt.local_rank)
    loader = make_loader(opt, opt.synthetic_samples)
    model = MMHandModel(opt)
    model.pprint("#training images = %d" % len(loader))
    model.pprint("model [%s] was created" % model.name())
    log_name = os.path.join(opt.checkpoints_dir, opt.name, "loss_log.txt")
    total_steps = 0
    # the interpreter's generation-2 collections walk everything alive; after the first iteration the
    # long-lived objects (modules, parameters, pools) are frozen into the permanent generation so a
    # collection inside an iteration only walks that iteration's autograd graph, and full collections
    # are run at the points that synchronise with the device anyway (loss printing, end of epoch)
    import gc
    frozen = False
    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        epoch_start = time.time()
        epoch_iter = 0
        for data in loader:
            iter_start = time.time()
            total_steps += opt.batchSize
            epoch_iter += opt.batchSize
            model.set_input(data)
            model.optimize_parameters()
            if not frozen:
                gc.collect()
                gc.freeze()
                frozen = True
            if total_steps % opt.print_freq == 0 and model.master:
                errors = model.get_current_errors()          # float() -> the only D2H sync
                t = (time.time() - iter_start) / opt.batchSize
                msg = "(epoch: %d, iters: %d, time: %.3f) " % (epoch, epoch_iter, t)
                msg += "".join("%s: %.3f " % (k, float(v)) for k, v in errors.items())
                print(msg)
                with open(log_name, "a") as f:
                    f.write("%s\n" % msg)
                gc.collect()
            if total_steps % opt.save_latest_freq == 0:      # every rank: save() settles the overflow
                model.pprint("saving the latest model (epoch %d, total_steps %d)" % (epoch, total_steps))
                model.save("latest")                          # flags; only the master writes files
        loader.set_epoch(epoch)
        gc.collect()
        if epoch % opt.save_epoch_freq == 0:
            model.pprint("saving the model at the end of epoch %d, iters %d" % (epoch, total_steps))
            model.save("latest")
            model.save(epoch)
        model.pprint("End of epoch %d / %d \t Time Taken: %d sec" %
                     (epoch, opt.niter + opt.niter_decay, time.time() - epoch_start))
        model.update_learning_rate()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
