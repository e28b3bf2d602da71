"""Batches for the step, with the reference loader's batch-dict contract (data/generic_dataset.py:169-180: keys
H1,H2,P1,P2,D1,D2,C1,C2,H1_path,H2_path).

`SyntheticHandLoader`: RHD/STB-shaped synthetic batches: H ~ U(-1,1) [B,3,H,W]; P = 21 Gaussian pose
maps (sigma 6, threshold 0.0099, clamp 1) at joints uv ~ U(20,H-20); D = one U(-1,1) plane replicated to 3 channels (RHD
and STB share this live code path, generic_dataset.py:151-159).  Pose maps are synthesised on the device by
mmh_pose_heatmaps.

`HandFolderLoader`: the reference's PREPARED dataset directory -
`annotation.pickle` + `<folder>/<name>.png` colour images + the depth PNG at the same path with "color" replaced by
"depth" (data/generic_dataset.py:86-180, data/rhd_dataset.py:16-45, data/stb_dataset.py:16-45) - read with PIL (cv2 is
absent from this image) into pinned uint8 batches that travel to the device as they are; everything arithmetic the
reference's loader workers do per sample on the CPU (normalise, depth = 256 G + R, / 700, 21 full-image Gaussians x 2)
happens in ONE device kernel behind `MMHandModel.set_input` (mmh_decode_inputs).  The legacy pair-list format of
data/mmhand_dataset.py (CSV pairs + .npy pose arrays) is not read."""
import os
import pickle
import random

import numpy as np
import torch

from . import ops


class SyntheticHandLoader:
    """Iterable of device-resident batches; per-rank shard like DistributedSampler
    (data/mmhand_dataset_data_loader.py:22-24): rank r sees samples r, r+world, ..."""

    def __init__(self, opt, n_samples=256, size=None, device=None):
        self.opt = opt
        self.n = int(n_samples)
        self.size = size or opt.fineSize
        self.device = device or torch.device("cuda", opt.local_rank)
        self.world = getattr(opt, "world_size", 1) or 1
        self.rank = torch.distributed.get_rank() if (self.world > 1 and torch.distributed.is_initialized()) else 0
        self.epoch = 0
        self.name = "synthetic-" + str(getattr(opt, "dataset", None) or "rhd")

    def __len__(self):
        return self.n // self.world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def make_batch(self, B, seed):
        H = W = self.size
        dev = self.device
        g = torch.Generator(device=dev).manual_seed(int(seed))
        out = {}
        for s in ("1", "2"):
            out["H" + s] = torch.rand((B, 3, H, W), generator=g, device=dev) * 2 - 1
            uv = torch.rand((B * 21, 2), generator=g, device=dev, dtype=torch.float64) * (H - 40) + 20
            out["C" + s] = uv.view(B, 21, 2)
            out["P" + s] = ops.pose_heatmaps(uv.contiguous(), H, W).view(B, 21, H, W)
            d = torch.rand((B, 1, H, W), generator=g, device=dev) * 2 - 1
            out["D" + s] = d.expand(B, 3, H, W).contiguous()
        out["H1_path"] = [f"{self.name}/{seed:08d}_{i}_a.png" for i in range(B)]
        out["H2_path"] = [f"{self.name}/{seed:08d}_{i}_b.png" for i in range(B)]
        return out

    def __iter__(self):
        B = self.opt.batchSize
        per_rank = len(self)
        for it in range(per_rank // B):
            seed = (getattr(self.opt, "seed", 49) * 1000003 + self.epoch * 7919 + it) * self.world + self.rank
            yield self.make_batch(B, seed)


# ----------------------------------------------------------------------------- the reference's prepared directories
def _read_bgr(path):
    """what cv2.imread(path) returns for the 8-bit PNGs of the prepared datasets: uint8 [H,W,3] in B,G,R order"""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


class HandFolderLoader:
    """Iterable of RAW device batches over a prepared RHD / STB directory; `MMHandModel.set_input` decodes them on the
    device (keys img1, img2, dep1, dep2: uint8 [B,H,W,3] BGR as cv2.imread delivers them; uv1, uv2: float64 [B,21,2];
    C1, C2: float64 [B,21,3] = (u, v, depth / 700 * 255); H1_path, H2_path).  `decoded=True` yields the reference's own
    keys instead (H1, P1, D1, ... as NCHW fp32 views of the decoded buffers) for callers that want tensors.

    Mirrors, by file:line of the reference:
      * image lists: RHD - every image of annotation folder "color", sorted by the integer file stem
        (rhd_dataset.py:25-37); STB - the "*_color_*" images of the non-"BB" cameras, sorted by (folder digit, folder
        letter, frame number) (stb_dataset.py:24-41);
      * the split (generic_dataset.py:100-131): sep = int((1 - ratio) * n); a root with "test" in its path serves all of
        it (and refuses training); training takes [sep:], generation [:sep]; targets in sorted order, sources = the
        same list shuffled with Python's `random` (the generator one image is drawn FROM, the other TO);
      * labels (generic_dataset.py:204-210), the depth file = the colour path with "color" -> "depth" (:147-148);
      * batching (mmhand_dataset_data_loader.py:22-48): in order, no shuffle; under --distributed a
        DistributedSampler with its defaults (shuffle with seed 0, the epoch never advanced, padded to a multiple of the
        world size, rank r takes r, r + world, ...); the last batch may be short; iteration stops after
        `max_dataset_size` BATCHES (the reference compares the batch index with it)."""

    def __init__(self, opt, device=None, decoded=False, threads=None):
        self.opt = opt
        self.root_dir = opt.dataroot
        if not self.root_dir or not os.path.isfile(os.path.join(self.root_dir, "annotation.pickle")):
            raise FileNotFoundError(f"--dataroot {self.root_dir!r}: no annotation.pickle (the directory create_RHD_DB.py / "
                                    "create_STB_DB.py of the reference prepare)")
        with open(os.path.join(self.root_dir, "annotation.pickle"), "rb") as fh:
            self.annotations = pickle.load(fh)
        kind = (getattr(opt, "dataset", None) or "rhd").lower()
        if kind == "rhd":
            data = [os.path.join(self.root_dir, folder, image) for folder in self.annotations
                    for image in self.annotations[folder] if folder == "color"]
            key = lambda x: int(x.split("/")[-1][0:-4])                                        # noqa: E731
        elif kind == "stb":
            data = []
            for folder in self.annotations:
                for image in self.annotations[folder]:
                    camera, spec, _ = image.split("_")
                    if camera != "BB" and spec == "color":
                        data.append(os.path.join(self.root_dir, folder, image))

            def key(x):
                *_, folder, name = x.split("/")
                return int(folder[1]), folder[2], int(name[0:-4].split("_")[-1])
        else:
            raise NotImplementedError(f"--dataset {kind}: HandFolderLoader reads the prepared 'rhd' and 'stb' directories "
                                      "(the pair-list format of data/mmhand_dataset.py is not supported)")
        ratio = getattr(opt, "augmentation_ratio", None)
        self.image_source, self.image_target = self._get_src_tgt(0.0 if ratio is None else float(ratio), data, key)
        self.device = device or torch.device("cuda", getattr(opt, "local_rank", 0) or 0)
        self.decoded = decoded
        self.world = (getattr(opt, "world_size", 1) or 1) if getattr(opt, "distributed", False) else 1
        self.rank = torch.distributed.get_rank() if (self.world > 1 and torch.distributed.is_initialized()) else 0
        self.threads = int(threads if threads is not None else (getattr(opt, "nThreads", 4) or 1))
        self.epoch = 0          # the reference never calls sampler.set_epoch: the permutation is the same every epoch
        self.name = type(self).__name__

    def _get_src_tgt(self, ratio, data, sort_fn):
        assert len(data) > 0, "no images listed in annotation.pickle for this --dataset"
        data.sort(key=sort_fn)
        sep_pnt = int((1 - ratio) * len(data))
        if "test" in self.root_dir:
            assert not self.opt.isTrain, "a 'test' directory serves generation only (generic_dataset.py:116-118)"
            tgt = data
        else:
            tgt = data[sep_pnt:] if self.opt.isTrain else data[:sep_pnt]
        src = tgt.copy()
        random.shuffle(src)
        return src, tgt

    def get_labels(self, image_path):
        *_, folder, name = image_path.split("/")
        if "joints" in name:
            parts = name.split("_")
            name = parts[0] + "_" + parts[1] + "_" + parts[-1]
        return self.annotations[folder][name]

    def set_epoch(self, epoch):
        """kept for the training loop's interface; the reference's sampler stays at epoch 0 (see the class docstring)"""

    def indices(self):
        n = len(self.image_source)
        if self.world <= 1:
            return list(range(n))
        g = torch.Generator().manual_seed(0 + self.epoch)
        idx = torch.randperm(n, generator=g).tolist()
        total = -(-n // self.world) * self.world
        pad = total - n
        idx += idx[:pad] if pad <= len(idx) else (idx * -(-pad // len(idx)))[:pad]      # DistributedSampler's wrap-around padding
        return idx[self.rank:total:self.world]

    def n_batches(self):
        B = self.opt.batchSize
        return -(-len(self.indices()) // B)

    def __len__(self):
        return min(len(self.image_source), self.opt.max_dataset_size)

    def load_sample(self, item):
        """host side of generic_dataset.py:133-180 for ONE pair: file reads only - the arithmetic is the device's"""
        h_1, h_2 = self.image_source[item], self.image_target[item]
        a1, a2 = self.get_labels(h_1), self.get_labels(h_2)
        uv1 = np.asarray(a1["uv_coord"], dtype=np.float64).reshape(21, 2)
        uv2 = np.asarray(a2["uv_coord"], dtype=np.float64).reshape(21, 2)
        z1 = np.expand_dims(np.asarray(a1["depth"], dtype=np.float64), -1) / 700.0 * 255
        z2 = np.expand_dims(np.asarray(a2["depth"], dtype=np.float64), -1) / 700.0 * 255
        return dict(img1=_read_bgr(h_1), img2=_read_bgr(h_2), dep1=_read_bgr(h_1.replace("color", "depth")),
                    dep2=_read_bgr(h_2.replace("color", "depth")), uv1=uv1, uv2=uv2,
                    C1=np.concatenate([uv1, z1], axis=-1), C2=np.concatenate([uv2, z2], axis=-1), H1_path=h_1, H2_path=h_2)

    def _collate(self, samples):
        out = {}
        for k in ("img1", "img2", "dep1", "dep2", "uv1", "uv2", "C1", "C2"):
            t = torch.from_numpy(np.stack([s[k] for s in samples]))
            out[k] = t.pin_memory() if torch.cuda.is_available() else t
        out["H1_path"] = [s["H1_path"] for s in samples]
        out["H2_path"] = [s["H2_path"] for s in samples]
        return out

    def host_batches(self):
        """pinned host batches in loader order, file reads on a thread pool, one batch prepared ahead"""
        from concurrent.futures import ThreadPoolExecutor
        idx, B = self.indices(), self.opt.batchSize
        groups = [idx[i:i + B] for i in range(0, len(idx), B)]
        groups = groups[: int(min(len(groups), self.opt.max_dataset_size))]
        if not groups:
            return
        with ThreadPoolExecutor(max_workers=max(1, self.threads)) as pool:
            nxt = [pool.submit(self.load_sample, i) for i in groups[0]]
            for gi in range(len(groups)):
                cur = nxt
                nxt = [pool.submit(self.load_sample, i) for i in groups[gi + 1]] if gi + 1 < len(groups) else []
                yield self._collate([f.result() for f in cur])

    def to_device(self, hb):
        out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in hb.items()}
        if not self.decoded:
            return out
        xh1, xh2, xp, xd = ops.decode_inputs(out["img1"], out["img2"], out["dep1"], out["dep2"], out["uv1"], out["uv2"])
        v = ops.nhwc_to_nchw_view
        return {"H1": v(xh1, 3), "H2": v(xh2, 3), "P1": v(xp)[:, :21], "P2": v(xp)[:, 21:42], "D1": v(xd)[:, :3],
                "D2": v(xd)[:, 3:6], "C1": out["C1"], "C2": out["C2"], "H1_path": out["H1_path"], "H2_path": out["H2_path"]}

    def __iter__(self):
        for hb in self.host_batches():
            yield self.to_device(hb)


def make_loader(opt, synthetic_samples=256, device=None):
    """--dataroot given: the prepared directory; else synthetic batches (train.py:15-17's CreateDataLoader)"""
    if getattr(opt, "dataroot", None):
        return HandFolderLoader(opt, device=device)
    return SyntheticHandLoader(opt, synthetic_samples, device=device)
