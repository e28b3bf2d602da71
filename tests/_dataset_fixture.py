"""A tiny prepared dataset directory in the reference's layout, written by the tests (no reference data needed):
RHD-style `annotation.pickle` {folder: {file name: {"uv_coord": [21][2], "depth": [21]}}} with folders color / depth / mask
and 8-bit RGB PNGs `<root>/color/<n>.png`, `<root>/depth/<n>.png` (data/rhd_dataset.py:25-37), or STB-style folders
`B1Counting` ... with `SK_color_<n>.png` / `SK_depth_<n>.png` / `BB_*` files (data/stb_dataset.py:24-41)."""
import os
import pickle

import numpy as np


def write_rhd(root, n=7, size=32, seed=3, names=None):
    from PIL import Image
    rs = np.random.RandomState(seed)
    names = names or [f"{i:05d}.png" for i in rs.permutation(n * 3)[:n]]         # unsorted on purpose
    ann = {"color": {}, "depth": {}, "mask": {}}
    for folder in ann:
        os.makedirs(os.path.join(root, folder), exist_ok=True)
    for name in names:
        lab = {"uv_coord": rs.uniform(-4, size + 4, size=(21, 2)).tolist(), "depth": rs.uniform(100, 690, size=21).tolist()}
        for folder in ann:
            ann[folder][name] = lab
            Image.fromarray(rs.randint(0, 256, size=(size, size, 3), dtype=np.uint8)).save(os.path.join(root, folder, name))
    with open(os.path.join(root, "annotation.pickle"), "wb") as fh:
        pickle.dump(ann, fh)
    return names


def write_stb(root, n=4, size=32, seed=5):
    from PIL import Image
    rs = np.random.RandomState(seed)
    ann = {}
    for folder in ("B2Random", "B1Counting"):
        ann[folder] = {}
        os.makedirs(os.path.join(root, folder), exist_ok=True)
        for i in range(n):
            lab = {"uv_coord": rs.uniform(2, size - 2, size=(21, 2)).tolist(), "depth": rs.uniform(100, 690, size=21).tolist()}
            for cam in ("SK", "BB"):
                for spec in ("color", "depth"):
                    name = f"{cam}_{spec}_{i}.png"
                    ann[folder][name] = lab
                    Image.fromarray(rs.randint(0, 256, size=(size, size, 3), dtype=np.uint8)).save(os.path.join(root, folder, name))
    with open(os.path.join(root, "annotation.pickle"), "wb") as fh:
        pickle.dump(ann, fh)
