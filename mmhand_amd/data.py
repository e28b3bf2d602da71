"""Synthetic RHD/STB-shaped batches with the reference loader's batch-dict contract
(data/generic_dataset.py:169-180: keys H1,H2,P1,P2,D1,D2,C1,C2,H1_path,H2_path).

The real readers need the datasets, cv2 and pickled annotations and are out of scope
(SURVEY.md §2 row 10); what the step needs is the *shape and statistics* of a batch
(SURVEY.md §8(d)): H ~ U(-1,1) [B,3,H,W]; P = 21 Gaussian pose maps (sigma 6, threshold 0.0099,
clamp 1) at joints uv ~ U(20,H-20); D = one U(-1,1) plane replicated to 3 channels (RHD and STB
share this live code path, generic_dataset.py:151-159).  Pose maps are synthesised on the device
by mmh_pose_heatmaps (the next-row input pipeline of SURVEY.md §8(f)-1)."""
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
