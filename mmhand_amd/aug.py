"""Generation driver — counterpart of the reference's aug.py:14-71.

    python -m mmhand_amd.aug <checkpoint_name> <dataroot> <dst_dir> <rhd|stb> <ratio> <device>     (the reference's argv, aug.py:16)
    python -m mmhand_amd.aug <checkpoint_name> <dst_dir> [n_batches] [batch] [device]              (synthetic batches)

The first form reads the reference's prepared directory (data.HandFolderLoader: annotation.pickle + colour / depth PNGs,
the generation side of the augmentation_ratio split, batch size 1, decoded on the device) and writes each generated image
to <dst>/<folder of the TARGET image>/<its file name> (aug.py:66-71).

Loads checkpoints/<name>/latest_net_netG.pth (reference format), builds
Generator([3,42,6],3,64,BatchNorm,use_dropout=True,n_blocks=9).eval(), folds BN into the convs,
captures the forward in a hipGraph and writes the generated images ((x*0.5+0.5)*255, RGB->BGR
then cv2.imwrite) as PNG files via PIL (cv2 is absent here; cv2.imwrite of a BGR float array
rounds to nearest uint8 and stores RGB-ordered PNG pixels, which is what PIL is given) or, without
PIL, as .npy arrays."""
import os
import sys

import numpy as np
import torch

from .data import HandFolderLoader, SyntheticHandLoader
from .inference import InferenceGenerator
from .networks import Generator
from .options import default_train_opt


def main(argv, ngf=64, n_blocks=9, size=None):
    """argv as the reference's aug.py; ngf / n_blocks / size are the reference's hard-coded 64 / 9 / 256 (aug.py:31-39),
    keyword-overridable so that a test can drive the whole path on a small checkpoint."""
    ckp = argv[0]
    real = len(argv) == 6 and argv[3] in ("rhd", "stb")        # _, ckp, dataroot, DST, dataset, ratio, device = sys.argv
    if real:
        dataroot, dst, dataset, ratio, device = argv[1], argv[2], argv[3], float(argv[4]), int(argv[5])
        n_batches, batch = None, 1
    else:
        dst = argv[1]
        n_batches = int(argv[2]) if len(argv) > 2 else 4
        batch = int(argv[3]) if len(argv) > 3 else 1
        device = int(argv[4]) if len(argv) > 4 else 0
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    weights = torch.load(os.path.join("checkpoints", ckp, "latest_net_netG.pth"), map_location="cpu")
    model = Generator(input_nc=[3, 42, 6], output_nc=3, ngf=ngf, norm_layer="batch", use_dropout=True,
                      n_blocks=n_blocks)
    model.load_state_dict(weights)
    gen = InferenceGenerator(model.to(dev).eval(), use_graph=True)
    opt = default_train_opt(batchSize=batch, local_rank=device, isTrain=False)
    if real:
        # aug.py:18-26: isTrain False, batchSize 1, not distributed; the loader hands decoded NCHW views
        opt.dataroot, opt.dataset, opt.augmentation_ratio, opt.distributed = dataroot, dataset, ratio, False
        loader = HandFolderLoader(opt, device=dev, decoded=True)
    else:
        loader = SyntheticHandLoader(opt, n_batches * batch, size=size)
    os.makedirs(dst, exist_ok=True)
    written = []
    for i, sample in enumerate(loader):
        fake = gen([sample["H1"], torch.cat((sample["P1"], sample["P2"]), 1),
                    torch.cat((sample["D1"], sample["D2"]), 1)])
        img = ((fake.permute(0, 2, 3, 1) * 0.5 + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8)
        arr = img.cpu().numpy()                                     # RGB, what the PNG stores
        for j in range(arr.shape[0]):
            # aug.py:66-71: <dst>/<folder of the target image>/<its file name>
            *_, folder, name = sample["H2_path"][j].split("/")
            os.makedirs(os.path.join(dst, folder), exist_ok=True)
            path = os.path.join(dst, folder, name)
            try:
                from PIL import Image
                Image.fromarray(arr[j]).save(path)
            except ImportError:
                path = os.path.splitext(path)[0] + ".npy"
                np.save(path, arr[j])
            written.append(path)
    return written


if __name__ == "__main__":
    main(sys.argv[1:])
