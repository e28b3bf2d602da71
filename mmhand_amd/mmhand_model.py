"""MMHandModel — the training-step object of MM-HAND on MI355X.

Drop-in for models/MMHandModel.py + models/base_model.py: same constructor argument (``opt``
namespace, field list in SURVEY.md §5.6), same methods called by train.py:15-65
(set_input / optimize_parameters / get_current_errors / get_current_visuals / save /
update_learning_rate / pprint / name) and the same checkpoint files
(<label>_net_<netG|netD_PB|netD_PP>.pth holding reference-format state_dicts).

What differs is underneath: no apex, no cuDNN.  Every conv / norm / gate / loss / Adam launch is
a hand-written gfx950 kernel reached through libmmhand_hip.so; data parallelism is one process
per GPU with one RCCL all-reduce per network per backward on a side stream; the generator's
all-reduce + Adam overlap the two discriminator steps (which only need the *old* fake image,
models/MMHandModel.py:279-289).
"""
import os
import random
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import lib as L
from . import ops
from .networks import Discriminator, Generator, VGGHead, discriminators_lockstep
from .ops import pad4


MERGE_D_PASSES = os.environ.get("MMH_MERGE_D", "1") != "0"
_LAST_BUCKET_LOG = None      # MMH_DP_LOG=1: the event log of the last data-parallel model (tests)


# ----------------------------------------------------------------------------- helpers
def get_norm_layer(norm_type="instance"):
    """models/network_utils.py:74-84 — returns the tag the networks understand."""
    if norm_type in ("batch", "instance"):
        return norm_type
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def get_scheduler(optimizer, opt):
    """models/network_utils.py:87-109."""
    from torch.optim import lr_scheduler
    if opt.lr_policy == "lambda":
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + 1 + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == "plateau":
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, threshold=0.01,
                                              patience=5)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % opt.lr_policy)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr, betas, eps=1e-8, no weight decay) over a network's flat
    parameter buffer: one fused kernel launch per step (mmh_adam_step)."""

    def __init__(self, net, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        if net.flat_param is None:
            net.flatten_parameters()
        self.net = net
        super().__init__([net.flat_param], dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = torch.zeros_like(net.flat_param)
        self.exp_avg_sq = torch.zeros_like(net.flat_param)
        self.step_count = 0
        self.grad_scale = 1.0
        # --graph_step: the step count and the learning rate live on the device (mmh_adam_step_dev), so that the launch holds
        # nothing that changes from one iteration to the next and can be replayed from a captured graph
        self.dev_state = None       # (step int32[1], lr fp32[1], coef fp32[2]) once device_state() was called
        self.external_count = False  # the step's owner counts step_count itself (a replayed graph never runs step())

    def device_state(self):
        if self.dev_state is None:
            dev = self.net.flat_param.device
            self.dev_state = (torch.full((1,), int(self.step_count), dtype=torch.int32, device=dev),
                              torch.full((1,), float(self.param_groups[0]["lr"]), dtype=torch.float32, device=dev),
                              torch.zeros(2, dtype=torch.float32, device=dev))
            self._lr_on_device = float(self.param_groups[0]["lr"])
        return self.dev_state

    def sync_lr(self):
        """after a scheduler step: the device copy of lr follows param_groups (a fill kernel, no host copy)"""
        if self.dev_state is not None and float(self.param_groups[0]["lr"]) != self._lr_on_device:
            self._lr_on_device = float(self.param_groups[0]["lr"])
            self.dev_state[1].fill_(self._lr_on_device)

    def zero_grad(self, set_to_none=False):
        self.net.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None, skip_flag=None, loss_scale=None):
        """skip_flag: int32 device scalar; when it is non-zero the kernel leaves p, m, v untouched.
        The caller takes the step back from step_count once the flag has reached the host
        (MMHandModel._settle_overflow), as apex does not count a skipped step.
        loss_scale: fp32 device scalar the gradient is divided by (dynamic loss scaling)."""
        g = self.param_groups[0]
        if not self.external_count:
            self.step_count += 1
        if self.dev_state is not None:
            step_dev, lr_dev, coef = self.dev_state
            L.call("mmh_adam_step_dev", ops._ptr(self.net.flat_param), ops._ptr(self.net.flat_grad), ops._ptr(self.exp_avg),
                   ops._ptr(self.exp_avg_sq), self.net.flat_param.numel(), ops._ptr(lr_dev), float(g["betas"][0]),
                   float(g["betas"][1]), float(g["eps"]), ops._ptr(step_dev), float(self.grad_scale), ops._ptr(skip_flag),
                   ops._ptr(loss_scale), ops._ptr(coef), ops._stream())
        else:
            ops.adam_step(self.net.flat_param, self.net.flat_grad, self.exp_avg, self.exp_avg_sq,
                          g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.step_count,
                          self.grad_scale, skip_flag, loss_scale)
        ops.bump_weights_epoch(within=self.net.flat_param)     # this network's derived weight copies only

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        if self.dev_state is not None:
            self.dev_state[0].fill_(self.step_count)


class ImagePool:
    """util/image_pool.py:14-34 on device tensors; host RNG is Python's ``random`` as there."""

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        # Pass 1 walks the batch with the reference's host RNG sequence and only records decisions; pass 2 makes ONE
        # gather of the images that are actually stored into a compact buffer the pool keeps views of (the reference
        # clones image by image - 64 device copies per iteration at B = 32).  A stored view pins only the images
        # stored by the same query, never the whole batch: the pool holds at most pool_size images plus those that
        # left it while a sibling of the same query is still inside.
        out = []                    # per output slot: a tensor, or an int k = "the k-th image stored by this query"
        stored = []                 # batch indices stored by this query, in order
        # decisions go into a COPY of the pool list, written back only once `compact` exists: an allocation failure in
        # pass 2 (the realistic one: OOM at a step boundary) leaves self.images as it was, never holding ints
        pool = list(self.images)    # entries: tensors (earlier queries) or ints (this query)
        for i in range(images.shape[0]):
            if len(pool) < self.pool_size:
                pool.append(len(stored))
                stored.append(i)
                out.append(images[i:i + 1])
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.pool_size - 1)
                out.append(pool[j])
                pool[j] = len(stored)
                stored.append(i)
            else:
                out.append(images[i:i + 1])
        if stored:
            # torch.cat of slices: one device copy and no host-to-device index transfer (which would stall the host
            # that runs an iteration ahead of the GPU)
            compact = images.clone() if len(stored) == images.shape[0] else torch.cat([images[i:i + 1] for i in stored], 0)
            for j, e in enumerate(pool):
                if isinstance(e, int):
                    pool[j] = compact[e:e + 1]
            out = [compact[e:e + 1] if isinstance(e, int) else e for e in out]
        result = torch.cat(out, 0)
        self.images = pool
        return result


class DevicePool:
    """ImagePool (util/image_pool.py:14-34) with the pool resident in ONE device buffer and every query the same two
    launches (mmh_pool_exchange) driven by device index arrays - what a captured training step needs: the host draws the
    reference's decisions (same `random.uniform` / `random.randint` sequence) BEFORE the step is launched (`plan`), uploads
    them as indices, and the launches inside the step never change.  Same values as ImagePool query by query
    (tests/test_graph_step_gpu.py)."""

    def __init__(self, pool_size, queries_per_iteration=1):
        self.pool_size = pool_size
        self.count = 0                  # filled slots
        self.buf = None                 # [pool_size, ...] fp32, allocated at the first query
        self.nq = queries_per_iteration
        self.idx = None                 # device int32 [nq, 2, B]: (src, dst) of each query of the iteration
        self._pinned, self._events, self._slot = [], [], 0
        self._by_batch = {}
        self._plans = []                # decisions drawn by plan() and not yet consumed by query()
        self._q = 0                     # query index within the iteration

    def decide(self, B):
        """one query's decisions, drawn exactly as ImagePool.query walks the batch: src[i] (>= 0: slot as it was before the
        query; < 0: image -1 - src of the batch), dst[i] (slot image i is stored in, -1: none; one writer per slot)"""
        src, dst = [0] * B, [-1] * B
        holder = {}                     # slot -> batch image that holds it by now (stored earlier in this query)
        for i in range(B):
            if self.count < self.pool_size:
                holder[self.count] = i
                self.count += 1
                src[i] = -1 - i
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.pool_size - 1)
                src[i] = (-1 - holder[j]) if j in holder else j
                holder[j] = i
            else:
                src[i] = -1 - i
        for j, i in holder.items():
            dst[i] = j
        return src, dst

    def begin_iteration(self, B, device, decide=True):
        """draw (decide) and upload the index arrays of all nq queries of the coming iteration; the upload is queued on the
        current stream from a ring of pinned buffers, in front of the step's launches"""
        if self.pool_size == 0:
            return
        if self.idx is None or self.idx.shape[2] != B:
            # one index array (and pinned ring) per batch size, never re-allocated: a captured iteration holds its address
            if B not in self._by_batch:
                self._by_batch[B] = (torch.zeros((self.nq, 2, B), dtype=torch.int32, device=device),
                                     [torch.zeros((self.nq, 2, B), dtype=torch.int32).pin_memory() for _ in range(4)], [None] * 4)
            self.idx, self._pinned, self._events = self._by_batch[B]
        if decide:
            self._plans = [self.decide(B) for _ in range(self.nq)]
        k = self._slot
        self._slot = (k + 1) % len(self._pinned)
        if self._events[k] is not None:
            self._events[k].synchronize()       # four iterations old: long complete
        self._pinned[k].copy_(torch.tensor(self._plans, dtype=torch.int32))
        self.idx.copy_(self._pinned[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[k] = ev
        self._q = 0

    def query(self, images):
        if self.pool_size == 0:
            return images
        assert images.dtype == torch.float32 and images.is_contiguous() and self.idx is not None and self._q < self.nq, \
            "DevicePool.query without begin_iteration()"
        if self.buf is None:
            self.buf = torch.zeros((self.pool_size,) + tuple(images.shape[1:]), dtype=torch.float32, device=images.device)
        out = torch.empty_like(images)
        q = self._q
        self._q += 1
        L.call("mmh_pool_exchange", ops._ptr(self.buf), ops._ptr(images), ops._ptr(out), ops._ptr(self.idx[q, 0]),
               ops._ptr(self.idx[q, 1]), images.shape[0], images[0].numel(), ops._stream())
        return out


class GANLoss:
    """models/network_utils.py:129-163: BCEWithLogits against a constant label, mean."""

    def __call__(self, logits_nhwc, target_is_real):
        return ops.BCEWithLogitsConstFn.apply(logits_nhwc, 1.0 if target_is_real else 0.0, 1.0)


class L1PlusPerceptualLoss:
    """losses/L1_plus_perceptualLoss.py:32-75 on NHWC tensors (3 channels padded to 4)."""
    MEAN = (0.485, 0.456, 0.406)
    STD = (0.229, 0.224, 0.225)

    def __init__(self, lambda_L1, lambda_perceptual, vgg, percep_is_l1=1):
        self.lambda_L1, self.lambda_perceptual, self.vgg = lambda_L1, lambda_perceptual, vgg
        # L1_plus_perceptualLoss.py:64-71: L1 on the VGG features when percep_is_l1 == 1, else MSE
        self.percep_fn = ops.L1MeanFn if percep_is_l1 == 1 else ops.MSEMeanFn
        dev = next(vgg.parameters()).device
        sc = [0.5 / s for s in self.STD] + [0.0]
        sh = [(0.5 - m) / s for m, s in zip(self.MEAN, self.STD)] + [0.0]
        self.scale = torch.tensor(sc, dtype=torch.float32, device=dev)
        self.shift = torch.tensor(sh, dtype=torch.float32, device=dev)

    def features(self, x):
        return self.vgg.forward_nhwc(ops.AffineActFn.apply(x, self.scale, self.shift, False))

    def __call__(self, fake, real):
        B, H, W, _ = fake.shape
        loss_l1 = ops.L1MeanFn.apply(fake, real, self.lambda_L1, float(B * 3 * H * W))
        xf = ops.AffineActFn.apply(fake, self.scale, self.shift, False)
        with torch.no_grad():
            xr = ops.AffineActFn.apply(real, self.scale, self.shift, False)
        pair = self.vgg.l1_pair(xf) if self.percep_fn is ops.L1MeanFn else None
        if pair is not None:        # 16-bit mode, the shipped slice: features and their L1 as one node (ops.VggL1Fn)
            m1, m2 = pair
            loss_p = ops.VggL1Fn.apply(xf, xr, m1.weight, m1.bias, m2.weight, m2.bias, self.lambda_perceptual, self.vgg.bf16)
        else:
            f = self.vgg.forward_nhwc(xf)
            with torch.no_grad():
                r = self.vgg.forward_nhwc(xr)
            loss_p = self.percep_fn.apply(f, r, self.lambda_perceptual, float(f.numel()))
        return loss_l1 + loss_p, loss_l1, loss_p


class _frozen:
    """Run a network with requires_grad off on its parameters (the G step back-propagates
    *through* both discriminators but their weight gradients are never used:
    models/MMHandModel.py:238-243 followed by zero_grad at :320,326)."""

    def __init__(self, *nets):
        self.ps = [p for n in nets for p in n.parameters()]

    def __enter__(self):
        self.old = [p.requires_grad for p in self.ps]
        for p in self.ps:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p, o in zip(self.ps, self.old):
            p.requires_grad_(o)


# ----------------------------------------------------------------------------- the model
class MMHandModel(torch.nn.Module):
    def name(self):
        return "MMHandModel"

    def __init__(self, opt):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("MMHandModel needs an MI355X: the HIP path has no CPU fallback")
        L.load()
        torch.cuda.set_device(opt.local_rank)      # train.py:14; kernels go to this device's stream
        self.opt = opt
        self.gpu_ids = [opt.local_rank]
        self.isTrain = opt.isTrain
        self.device = torch.device("cuda", opt.local_rank)
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.master = opt.local_rank == 0 and (not dist.is_initialized() or dist.get_rank() == 0)
        self.overflow = False
        self.world = dist.get_world_size() if (getattr(opt, "distributed", False)
                                               and dist.is_initialized()) else 1
        if getattr(opt, "fp32_exact_grads", False):
            ops.set_winograd_mode("bwd")     # process-wide, like MMH_WINOGRAD=bwd (one model family per process)
        elif os.environ.get("MMH_WINOGRAD") in ("bwd", "0"):
            # the environment form of the same choice: the direct fprop's two-level summation comes with it
            ops.set_winograd_mode("bwd" if os.environ["MMH_WINOGRAD"] == "bwd" else "off")
        seed = getattr(opt, "seed", 49)
        # dropout masks: an independent stream per rank, as each reference rank has its own RNG
        # (the weights are seeded identically on every rank and broadcast from rank 0).  ImagePool
        # draws from Python's global `random`, exactly as util/image_pool.py does; seed it from the
        # caller (random.seed) when a reproducible pool history is wanted.
        rank = dist.get_rank() if dist.is_initialized() else 0
        ops.set_dropout_seed((int(seed) * 0x9E3779B97F4A7C15 + 0x5EED0001 + rank * 0xD1B54A32D192ED03) % (1 << 64))
        norm = get_norm_layer(opt.norm)
        input_nc = [opt.H_input_nc, opt.P_input_nc * 2, opt.D_input_nc * 2]
        self.netG = Generator(input_nc, opt.output_nc, opt.ngf, norm, not opt.no_dropout,
                              n_blocks=getattr(opt, "G_n_blocks", 9),
                              n_downsampling=opt.G_n_downsampling)
        self.netG.to(self.device).init_weights(opt.init_type, seed)
        nets = [self.netG]
        if self.isTrain:
            self.netD_PB = Discriminator(opt.H_input_nc + opt.P_input_nc, opt.ndf, norm,
                                         not opt.no_dropout_D, opt.n_layers_D,
                                         n_downsampling=opt.D_n_downsampling)
            self.netD_PP = Discriminator(opt.H_input_nc + opt.H_input_nc, opt.ndf, norm,
                                         not opt.no_dropout_D, opt.n_layers_D,
                                         n_downsampling=opt.D_n_downsampling)
            self.netD_PB.to(self.device).init_weights(opt.init_type, seed + 1)
            self.netD_PP.to(self.device).init_weights(opt.init_type, seed + 2)
            nets += [self.netD_PB, self.netD_PP]
        if not self.isTrain or opt.continue_train:
            self.load_network()
        for n in nets:
            n.flatten_parameters()
        # --opt_level (apex AMP in the reference, scripts/mm-train-ratio.sh:7-11):
        #   O0          fp32
        #   O1 / O2     mixed precision as apex runs it: 16-bit MFMA compute (bf16 here) with fp32
        #               master weights / accumulation / statistics AND apex's dynamic loss scaling
        #               (one scaler per loss, num_losses=3): scale -> backward -> overflow check on
        #               the scaled gradients -> unscale inside Adam -> skip / back off / grow
        #   O1_FP16 / O2_FP16   the same with IEEE fp16 MFMA operands - apex's own numerics: the
        #               dynamic loss scaler is then REQUIRED (fp16 gradients under- and overflow)
        #   BF16        the same bf16 compute without a loss scaler (bf16 has fp32's exponent range)
        level = str(getattr(opt, "opt_level", "O0")).upper()
        if level not in ("O0", "O1", "O2", "O1_FP16", "O2_FP16", "BF16"):
            raise ValueError("--opt_level %r: expected O0 | O1 | O2 | O1_FP16 | O2_FP16 | BF16 (apex's O3 = "
                             "pure fp16 weights is 'not recommended' by the reference and not built)"
                             % (opt.opt_level,))
        # the networks' `bf16` attribute carries the operand type: False fp32, True bf16, 2 fp16
        self.bf16 = 2 if level.endswith("_FP16") else level in ("O1", "O2", "BF16")
        self.loss_scaling = self.isTrain and level != "O0" and level != "BF16"
        for n in nets:
            n.bf16 = self.bf16

        if self.isTrain:
            self.old_lr = opt.lr
            # --graph_step (MMH_GRAPH_STEP=1): the whole iteration replayed from a captured hipGraph (single process).
            # Everything that changes between iterations moves behind device pointers first - Adam's step count and lr,
            # the dropout salt, the image pools' decisions - and that form also runs eagerly (warm-up, fallback), so the
            # replayed step is the eager one bit for bit.
            self.graph_step = bool(getattr(opt, "graph_step", False) or os.environ.get("MMH_GRAPH_STEP") == "1")
            if self.graph_step and getattr(opt, "distributed", False) and dist.is_initialized():
                self.pprint("--graph_step: single-process only (the data-parallel step interleaves collectives); off")
                self.graph_step = False
            Pool = (lambda n: DevicePool(n, opt.DG_ratio)) if self.graph_step else ImagePool
            self.fake_PP_pool = Pool(opt.pool_size)
            self.fake_PB_pool = Pool(opt.pool_size)
            self.criterionGAN = GANLoss()
            if opt.L1_type == "l1_plus_perL1":
                # the reference slices vgg19.features up to (and including) this index (L1_plus_perceptualLoss.py:24-27)
                self.vgg = VGGHead(int(getattr(opt, "perceptual_layers", 3))).to(self.device)
                self.vgg.bf16 = self.bf16
                vgg_path = getattr(opt, "vgg_weights", None)
                if vgg_path:
                    self.vgg.load_state_dict(torch.load(vgg_path, map_location="cpu"))
                    self.vgg_source = "file:" + os.path.abspath(vgg_path)
                elif getattr(opt, "vgg_random_init", False):
                    self.vgg.init_random()
                    self.vgg_source = "random-init (seed 1234): NOT the reference's pretrained objective"
                    self.pprint("WARNING: perceptual loss on RANDOM vgg19.features[0:4] weights "
                                "(--vgg_random_init); the reference uses torchvision's pretrained VGG19")
                else:
                    raise RuntimeError(
                        "--L1_type l1_plus_perL1 needs torchvision's pretrained vgg19.features "
                        "(losses/L1_plus_perceptualLoss.py:22), which cannot be downloaded here: pass "
                        "--vgg_weights <state_dict file with 0.weight/0.bias/2.weight/2.bias, or the "
                        "torchvision 'features.' keys>, or opt in to random weights with --vgg_random_init")
                self.criterionL1 = L1PlusPerceptualLoss(opt.lambda_A, opt.lambda_B, self.vgg,
                                                        opt.percep_is_l1)
            elif opt.L1_type == "origin":
                # The reference builds torch.nn.L1Loss() here (models/MMHandModel.py:81-82) and backward_G then indexes
                # its 0-dim result (`losses[0]`, :247-250): IndexError on the first iteration - the branch cannot train
                # there.  Intended behaviour, implemented in backward_G below and tested (test_l1_type_origin):
                # pair_L1loss = origin_L1 = mean|fake - H2| with weight 1, perceptual = 0, no VGG.
                self.vgg = None
                self.criterionL1 = None
            else:
                raise Exception("Unsurportted type of L1!")
            betas = (opt.beta1, 0.999)
            self.optimizer_G = FlatAdam(self.netG, opt.lr, betas)
            self.optimizer_D_PB = FlatAdam(self.netD_PB, opt.lr, betas)
            self.optimizer_D_PP = FlatAdam(self.netD_PP, opt.lr, betas)
            self.optimizers = [self.optimizer_G, self.optimizer_D_PB, self.optimizer_D_PP]
            self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
            # MMH_FORCE_DP=1: take the data-parallel code path even with one rank (RCCL smoke test)
            self.dp = self.world > 1 or (os.environ.get("MMH_FORCE_DP") == "1" and dist.is_initialized())
            if self.dp:
                self._init_data_parallel()
            self.skipped_steps = 0
            if opt.continue_train:
                self.load_train_state()
        if not getattr(self, "dp", False):
            self.comm_stream = None
        if self.isTrain:
            # overflow skip (models/MMHandModel.py:294-330) decided on the device: one sticky int32
            # flag per optimizer step of an iteration, order G, D_PP x DG_ratio, D_PB x DG_ratio
            self._nflags = 1 + 2 * opt.DG_ratio
            self._flags = torch.zeros(self._nflags, dtype=torch.int32, device=self.device)
            # dynamic loss scaling: {scale, clean steps} per loss id (0 = G, 1 = D_PB, 2 = D_PP as
            # MMHandModel.py:261,281,291 number them) and each step's OWN overflow flag, all on the device
            self._own = torch.zeros(self._nflags, dtype=torch.int32, device=self.device)
            if getattr(self, "_scaler", None) is None:
                self._scaler = torch.tensor([[ops.LOSS_SCALE_INIT, 0.0]] * 3, dtype=torch.float32,
                                            device=self.device)
            self.loss_scale_window = ops.LOSS_SCALE_WINDOW
            self._flags_free = [torch.zeros(self._nflags, dtype=torch.int32).pin_memory() for _ in range(3)]
            self._flags_pending = []        # [(event, pinned host copy)] of iterations not settled yet
            self.skipped_steps = getattr(self, "skipped_steps", 0)   # optimizer steps skipped so far
            self.last_overflow = False      # did the last settled iteration skip anything
            self._graph = None              # the captured iteration (torch.cuda.CUDAGraph) once --graph_step has one
            self._graph_state = "off"
            if self.graph_step:
                self._graph_state = "warmup"
                self._graph_warm = max(2, int(os.environ.get("MMH_GRAPH_WARMUP", "3")))
                self._graph_iters = 0
                self.graph_replays = 0
                self.graph_error = None
                self._salt = torch.zeros(1, dtype=torch.int64, device=self.device)
                self._seed_base = ops._seed_counter[0]
                for o in self.optimizers:
                    o.device_state()
                    o.external_count = True

    # ------------------------------------------------------------------ data parallel
    def _init_data_parallel(self):
        """apex DDP(delay_allreduce=True)+convert_syncbn_model (models/MMHandModel.py:99-116):
        broadcast rank-0 parameters once; gradients are averaged per backward; batch-norm
        statistics are taken over the global batch."""
        for net in (self.netG, self.netD_PB, self.netD_PP):
            dist.broadcast(net.flat_param, 0)
            for b in net.buffers():
                dist.broadcast(b, 0)
            if net.norm == "batch":
                net.sync_group = dist.group.WORLD
        ops.bump_weights_epoch()        # the broadcast rewrote every weight in place
        for o in self.optimizers:
            o.grad_scale = 1.0 / self.world
        # gradient all-reduce: reverse-layer-order buckets of each flat gradient buffer, launched
        # from autograd hooks on a side stream while the backward pass is still running (dp.py)
        from .dp import GradBuckets
        self.comm_stream = torch.cuda.Stream(self.device)
        self._bucket_log = [] if os.environ.get("MMH_DP_LOG") == "1" else None
        global _LAST_BUCKET_LOG
        _LAST_BUCKET_LOG = self._bucket_log
        # SyncBN's small all-gathers / all-reduces are issued from inside the forward and backward passes; on the WORLD
        # communicator they queue behind whatever 32 MB gradient bucket is in flight.  MMH_DP_BUCKET_GROUP=1 gives the
        # buckets a communicator of their own.  OFF by default: two RCCL communicators running concurrently from one
        # process can deadlock unless both kernels are co-resident on every rank, and that has never run on more than
        # one rank of this hardware - one communicator, one issue order, is correct by construction.
        self._bucket_group = None
        if (any(n.norm == "batch" for n in (self.netG, self.netD_PB, self.netD_PP))
                and os.environ.get("MMH_DP_BUCKET_GROUP", "0") == "1"):
            self._bucket_group = dist.new_group()
        # on RCCL the buckets' collectives are issued through the C-ABI (mmh_allreduce_bucket) on torch.distributed's own
        # communicator; any other backend (gloo in the tests), MMH_DP_NATIVE=0 or a separate bucket group: dist.all_reduce
        from .dp import native_comm
        self.dp_native, self.dp_native_why = (None, "separate bucket communicator") if self._bucket_group is not None \
            else native_comm(None)
        self._buckets = {o: GradBuckets(list(o.net.parameters()), o.net.flat_grad, comm_stream=self.comm_stream,
                                        log=self._bucket_log, name=n, group=self._bucket_group,
                                        flat_param=o.net.flat_param, native=self.dp_native)
                         for n, o in (("G", self.optimizer_G), ("D_PB", self.optimizer_D_PB),
                                      ("D_PP", self.optimizer_D_PP))}
        # in-place parameter gradients under data parallelism too (the shims report to the buckets: dp.param_use / _done)
        self.dp_accum = os.environ.get("MMH_DP_ACCUM", "1") != "0"

    # ------------------------------------------------------------------ input
    def set_input(self, input):
        """models/MMHandModel.py:200-221.  Host tensors travel on a copy stream of their own: the host
        runs an iteration ahead of the GPU, so the H2D copies of batch i+1 (453 MB at B=32, 256^2) overlap
        the kernels of iteration i instead of queueing behind them; the compute stream only waits for
        the copies' event.  (Pinned source tensors make the copies asynchronous; the caller must not
        overwrite them before the next set_input, as with any non_blocking copy.)"""
        dev = self.device
        if "img1" in input:
            # a RAW batch of data.HandFolderLoader (uint8 images / depth PNGs, float64 joints): decoded on the device
            raw = ("img1", "img2", "dep1", "dep2", "uv1", "uv2")
            if any(not input[k].is_cuda for k in raw):
                if getattr(self, "_copy_stream", None) is None:
                    self._copy_stream = torch.cuda.Stream(dev)
                cur = torch.cuda.current_stream(dev)
                with torch.cuda.stream(self._copy_stream):
                    t = {k: input[k].to(dev, non_blocking=True).contiguous() for k in raw}
                cur.wait_stream(self._copy_stream)
                for v in t.values():
                    v.record_stream(cur)
            else:
                t = {k: input[k].contiguous() for k in raw}
            return self.set_input_raw(*[t[k] for k in raw], paths=(input["H1_path"], input["H2_path"]) if "H1_path" in input else None)
        keys = ("H1", "P1", "D1", "H2", "P2", "D2")
        if dev.type == "cuda" and any(not input[k].is_cuda for k in keys):
            if getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(dev)
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._copy_stream):
                t = {k: input[k].to(dev, non_blocking=True).float() for k in keys}
            cur.wait_stream(self._copy_stream)
            for v in t.values():
                v.record_stream(cur)
        else:
            t = {k: input[k].to(dev, non_blocking=True).float() for k in keys}
        o = self.opt
        B, _, H, W = t["H1"].shape
        hc, pc, dc = o.H_input_nc, o.P_input_nc, o.D_input_nc
        if getattr(self, "_graph_state", "off") == "replay":
            st = self._static_inputs
            self._graph_odd = tuple(st["input_H1"].shape) != tuple(t["H1"].shape)
            if not self._graph_odd:
                # the captured iteration reads these buffers by address: new batches are written INTO them
                for k, v in st.items():
                    setattr(self, k, v)         # (an odd-shaped batch before this one had re-pointed the attributes)
                for k in keys:
                    st["input_" + k].copy_(t[k])
                tw = self.bf16 if ops.USE_LP16_EDGES else 0
                tws = self._static_twins
                ops.raw_pack([(st["input_H1"], True, hc)], B, H, W, pad4(hc), dev, out=st["x_H1"],
                             twin=tw if tws["x_H1"] is not None else 0, twin_out=tws["x_H1"])
                ops.raw_pack([(st["input_P1"], True, pc), (st["input_P2"], True, pc)], B, H, W, pad4(2 * pc), dev, out=st["x_P"],
                             twin=tw if tws["x_P"] is not None else 0, twin_out=tws["x_P"])
                ops.raw_pack([(st["input_D1"], True, dc), (st["input_D2"], True, dc)], B, H, W, pad4(2 * dc), dev, out=st["x_D"],
                             twin=tw if tws["x_D"] is not None else 0, twin_out=tws["x_D"])
                ops.raw_pack([(st["input_H2"], True, hc)], B, H, W, pad4(hc), dev, out=st["x_H2"])
                if "H1_path" in input:
                    self.image_paths = input["H1_path"][0] + "___" + input["H2_path"][0]
                return
            # a batch of another shape (the short last batch of an epoch): this iteration runs in the eager form on buffers of
            # its own; the captured buffers stay untouched for the next full batch
        self.input_H1, self.input_P1, self.input_D1 = t["H1"], t["P1"], t["D1"]
        self.input_H2, self.input_P2, self.input_D2 = t["H2"], t["P2"], t["D2"]
        # NHWC packs: concat + zero-pad to multiples of 4 in one kernel each.  16-bit training: the same pass leaves the
        # generator stems' padded 16-bit inputs (ops.raw_pack twin; they stay valid for every step on this batch)
        tw = self.bf16 if (self.isTrain and ops.USE_LP16_EDGES) else 0
        for old in (getattr(self, "x_H1", None), getattr(self, "x_P", None), getattr(self, "x_D", None)):
            ops.pack_twin_drop(old)
        self.x_H1 = ops.raw_pack([(t["H1"], True, hc)], B, H, W, pad4(hc), dev, twin=tw, keep_twin=True)
        self.x_P = ops.raw_pack([(t["P1"], True, pc), (t["P2"], True, pc)], B, H, W, pad4(2 * pc), dev, twin=tw, keep_twin=True)
        self.x_D = ops.raw_pack([(t["D1"], True, dc), (t["D2"], True, dc)], B, H, W, pad4(2 * dc), dev, twin=tw, keep_twin=True)
        self.x_H2 = ops.raw_pack([(t["H2"], True, hc)], B, H, W, pad4(hc), dev)
        if "H1_path" in input:
            self.image_paths = input["H1_path"][0] + "___" + input["H2_path"][0]

    def set_input_raw(self, img1, img2, dep1, dep2, uv1, uv2, paths=None):
        """Input straight from decoded files: uint8 BGR images / depth PNGs [B,H,W,3] and float64
        joints [B,21,2] on the device.  One kernel (mmh_decode_inputs) does what the reference's
        loader workers do per sample on the CPU (data/generic_dataset.py:133-180) and writes the
        stems' NHWC buffers directly; the NCHW tensors the rest of the API exposes are views."""
        if getattr(self, "_graph_state", "off") == "replay":
            xh1, xh2, xp, xd = ops.decode_inputs(img1, img2, dep1, dep2, uv1, uv2)
            v, o = ops.nhwc_to_nchw_view, self.opt
            d = {"H1": v(xh1, o.H_input_nc), "H2": v(xh2, o.H_input_nc), "P1": v(xp)[:, : o.P_input_nc],
                 "P2": v(xp)[:, o.P_input_nc: 2 * o.P_input_nc], "D1": v(xd)[:, : o.D_input_nc],
                 "D2": v(xd)[:, o.D_input_nc: 2 * o.D_input_nc]}
            if paths is not None:
                d["H1_path"], d["H2_path"] = paths
            return self.set_input(d)
        for old in (getattr(self, "x_H1", None), getattr(self, "x_P", None), getattr(self, "x_D", None)):
            ops.pack_twin_drop(old)         # 16-bit copies a set_input() parked for the previous batch
        self.x_H1, self.x_H2, self.x_P, self.x_D = ops.decode_inputs(img1, img2, dep1, dep2, uv1, uv2)
        v = ops.nhwc_to_nchw_view
        o = self.opt
        self.input_H1, self.input_H2 = v(self.x_H1, o.H_input_nc), v(self.x_H2, o.H_input_nc)
        self.input_P1 = v(self.x_P)[:, : o.P_input_nc]
        self.input_P2 = v(self.x_P)[:, o.P_input_nc: 2 * o.P_input_nc]
        self.input_D1 = v(self.x_D)[:, : o.D_input_nc]
        self.input_D2 = v(self.x_D)[:, o.D_input_nc: 2 * o.D_input_nc]
        if paths is not None:
            self.image_paths = paths[0][0] + "___" + paths[1][0]

    def forward(self):
        self.fake_nhwc = self.netG.forward_nhwc(self.x_H1, self.x_P, self.x_D)
        self.fake_p2 = ops.nhwc_to_nchw_view(self.fake_nhwc, self.opt.output_nc)

    def test(self):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def _cat_PB(self, img_nhwc, img_is_fake):
        """cat(image, P2) -> NHWC [B,H,W,24]."""
        o = self.opt
        B, H, W, _ = img_nhwc.shape
        return ops.PackFn.apply((pad4(o.H_input_nc + o.P_input_nc), self._stem_twin()), img_nhwc, False, o.H_input_nc,
                                self.input_P2, True, o.P_input_nc)

    def _stem_twin(self):
        """operand type of the padded 16-bit copy a pack leaves for the discriminator stem that reads it (16-bit training)"""
        return self.bf16 if (self.isTrain and ops.USE_LP16_EDGES and torch.is_grad_enabled()) else 0

    def _cat_PP(self, img_nhwc):
        """cat(image, H1) -> NHWC [B,H,W,8] (6 real channels)."""
        o = self.opt
        return ops.PackFn.apply((pad4(2 * o.H_input_nc), self._stem_twin()), img_nhwc, False, o.H_input_nc,
                                self.input_H1, True, o.H_input_nc)

    # ------------------------------------------------------------------ G
    def backward_G(self):
        o = self.opt
        with _frozen(self.netD_PB, self.netD_PP):
            nc = o.H_input_nc     # only the generated image inside the concat carries a gradient
            # both discriminators on the generated image: one after the other, or - under SyncBN - depth by depth
            # side by side with their norm collectives packed (networks.discriminators_lockstep)
            cat_PB, cat_PP = self._cat_PB(self.fake_nhwc, True), self._cat_PP(self.fake_nhwc)
            # the discriminator steps of this iteration feed the SAME concatenations (of the detached image) to their pools:
            # kept instead of packed a second time (two pack launches, 0.25 ms per step)
            self._fake_cats = (self.fake_nhwc, cat_PB.detach(), cat_PP.detach())
            pred_fake_PB, pred_fake_PP = discriminators_lockstep([(self.netD_PB, cat_PB, nc), (self.netD_PP, cat_PP, nc)])
            self.loss_G_GAN_PB = self.criterionGAN(pred_fake_PB, True)
            self.loss_G_GAN_PP = self.criterionGAN(pred_fake_PP, True)
            if self.criterionL1 is not None:
                losses = self.criterionL1(self.fake_nhwc, self.x_H2)
            else:
                B, H, W, _ = self.fake_nhwc.shape
                l1 = ops.L1MeanFn.apply(self.fake_nhwc, self.x_H2, 1.0, float(B * 3 * H * W))
                losses = (l1, l1.detach(), torch.zeros_like(l1))
            self.loss_G_L1 = losses[0]
            self.loss_originL1 = losses[1].detach()
            self.loss_perceptual = losses[2].detach()
            pair_L1loss = self.loss_G_L1
            pair_GANloss = (self.loss_G_GAN_PB * o.lambda_GAN + self.loss_G_GAN_PP * o.lambda_GAN) / 2
            pair_loss = pair_L1loss + pair_GANloss
            self.loss_backward(pair_loss, 0)
        self.pair_L1loss = pair_L1loss.detach()
        self.pair_GANloss = pair_GANloss.detach()

    # ------------------------------------------------------------------ D
    def loss_backward(self, loss, loss_id=0):
        """MMHandModel.loss_backward (models/MMHandModel.py:294-308): with mixed precision the
        backward runs on loss * scale (amp.scale_loss); the overflow check, the unscale and the
        scaler update happen at the optimizer step (_guarded_step), all on the device."""
        if self.loss_scaling:
            (loss * self._scaler[loss_id, 0]).backward()
        else:
            loss.backward()

    def loss_scale(self, loss_id=0):
        """Current loss scale of loss `loss_id` (synchronises; for logging / tests)."""
        return float(self._scaler[loss_id, 0])

    def backward_D_basic(self, netD, real, fake, loss_id=0):
        """real: a [2B,H,W,C] buffer whose first half holds the real batch (backward_D_PB / _PP pack it there) when the two
        passes run as one - or _pack_real's (proxy, x16) pair in 16-bit training - else the [B,H,W,C] real batch."""
        o = self.opt
        merged16 = isinstance(real, tuple)
        if merged16 or real.shape[0] == 2 * fake.shape[0]:
            # --norm instance: every sample is normalised on its own, so netD(real) and netD(fake) (two passes,
            # models/MMHandModel.py:263-274) ARE netD(cat(real, fake)) sample by sample - one pass over 2B images: half the
            # launches of a discriminator step and fuller tail rounds in every kernel.  (BatchNorm keeps the two passes: each
            # has its own batch statistics.)  MMH_MERGE_D=0: two passes.
            B = fake.shape[0]
            if merged16:
                # 16-bit training: the stem reads nothing but the padded 16-bit copy of its input, so only that is built -
                # the real half by the pack kernel (_pack_real), the fake half by one conversion pass over the pool's
                # images - and the fp32 [2B] buffer, its pack, the copy of the fake half and the conversion of the real half
                # are gone; the stem finds the copy under the proxy's address (ops.pack_twin_put)
                real, x16 = real
                ops.lp16_pad8(fake.detach(), self.bf16, out=x16[B:])
                ops.pack_twin_put(real, x16)
            else:
                real[B:].copy_(fake.detach())
            pred = netD.forward_nhwc(real)
            loss_D_real, loss_D_fake = ops.BCEWithLogitsHalvesFn.apply(pred, 1.0)
            loss_D = (loss_D_real * o.lambda_GAN + loss_D_fake * o.lambda_GAN) * 0.5
            self.loss_backward(loss_D, loss_id)
            return loss_D
        # the real and the fake batch keep their own batch statistics (two passes, models/MMHandModel.py:263-274);
        # under SyncBN they run side by side and share each depth's collective
        pred_real, pred_fake = discriminators_lockstep([(netD, real, 0), (netD, fake.detach(), 0)])
        loss_D_real = self.criterionGAN(pred_real, True) * o.lambda_GAN
        loss_D_fake = self.criterionGAN(pred_fake, False) * o.lambda_GAN
        loss_D = (loss_D_real + loss_D_fake) * 0.5
        self.loss_backward(loss_D, loss_id)
        return loss_D

    def _real_buffer(self, netD, B, H, W, Cd):
        """[2B,H,W,Cd] when the discriminator step runs its real and fake batch as ONE pass (InstanceNorm; see
        backward_D_basic), else [B,H,W,Cd]; the real batch is packed into the first B images"""
        merged = netD.norm == "instance" and MERGE_D_PASSES
        return torch.empty(((2 if merged else 1) * B, H, W, Cd), dtype=torch.float32, device=self.device)

    def _pack_real(self, netD, srcs, B, H, W, Cd):
        """the real batch of a discriminator step, packed: an fp32 buffer (see _real_buffer), or - one merged pass in 16-bit
        training, where the stem reads only the padded 16-bit copy of its input - (proxy, x16 [2B,H,W,C8]) with the first
        half written and the second left for backward_D_basic"""
        merged = netD.norm == "instance" and MERGE_D_PASSES
        if (merged and self.bf16 and ops.USE_PACK_TWIN and Cd <= 56
                and netD._lp_out_stem(netD.model[1], 2 * B, H, W, False)):
            x16 = torch.empty((2 * B, H, W, (Cd + 7) // 8 * 8), dtype=ops._wd(self.bf16), device=self.device)
            ops.raw_pack(srcs, B, H, W, Cd, self.device, twin=self.bf16, twin_out=x16[:B], only16=True)
            return ops.lp_proxy((2 * B, H, W, Cd), self.device), x16
        real = self._real_buffer(netD, B, H, W, Cd)
        ops.raw_pack(srcs, B, H, W, Cd, self.device, out=real[:B])
        return real

    def backward_D_PB(self):
        o = self.opt
        B, _, H, W = self.input_H2.shape
        real_PB = self._pack_real(self.netD_PB, [(self.input_H2, True, o.H_input_nc), (self.input_P2, True, o.P_input_nc)],
                                  B, H, W, pad4(o.H_input_nc + o.P_input_nc))
        kept = getattr(self, "_fake_cats", None)
        if kept is not None and kept[0] is self.fake_nhwc:
            fake_now = kept[1]
        else:
            with torch.no_grad():
                fake_now = self._cat_PB(self.fake_nhwc.detach(), True)
        fake_PB = self.fake_PB_pool.query(fake_now)
        self.loss_D_PB = self.backward_D_basic(self.netD_PB, real_PB, fake_PB, 1).detach()

    def backward_D_PP(self):
        o = self.opt
        B, _, H, W = self.input_H2.shape
        real_PP = self._pack_real(self.netD_PP, [(self.input_H2, True, o.H_input_nc), (self.input_H1, True, o.H_input_nc)],
                                  B, H, W, pad4(2 * o.H_input_nc))
        kept = getattr(self, "_fake_cats", None)
        if kept is not None and kept[0] is self.fake_nhwc:
            fake_now = kept[2]
        else:
            with torch.no_grad():
                fake_now = self._cat_PP(self.fake_nhwc.detach())
        fake_PP = self.fake_PP_pool.query(fake_now)
        self.loss_D_PP = self.backward_D_basic(self.netD_PP, real_PP, fake_PP, 2).detach()

    # ------------------------------------------------------------------ overflow skip
    def _guarded_step(self, optimizer, k, loss_id=0):
        """`if not self.overflow: optimizer.step()` (models/MMHandModel.py:316-328) without a host
        round trip: flag k = flag k-1 | any(!isfinite(grad)), and the Adam launch is a no-op when
        it is set.  Called after the gradient all-reduce, so every rank sees the same flag (a
        non-finite term makes the sum non-finite everywhere): reduce_tensor (:381-384) for free.
        With loss scaling the same launches also unscale (Adam divides by the device-resident
        scale) and update the loss's scaler from this gradient's OWN flag, as apex does when the
        amp.scale_loss context of that loss exits."""
        own = self._own[k:k + 1] if self.loss_scaling else None
        ops.grad_nonfinite(optimizer.net.flat_grad, self._flags[k:k + 1],
                           self._flags[k - 1:k] if k > 0 else None, own)
        if self.loss_scaling:
            optimizer.step(skip_flag=self._flags[k:k + 1], loss_scale=self._scaler[loss_id])
            ops.loss_scale_update(self._scaler[loss_id], own, self.loss_scale_window)
        else:
            optimizer.step(skip_flag=self._flags[k:k + 1])

    def _settle_overflow(self, drain=False):
        """Fetch the overflow flags of finished iterations (copied to pinned memory when each ended)
        and take skipped steps back from the Adam step counts.  The newest iteration is left
        pending unless `drain`, so the host may run a full iteration ahead of the GPU: the flags it
        waits for here are two iterations old and long complete.  Consequence: after an overflow
        the Adam step counts are corrected one iteration late, i.e. the single iteration that
        follows a skipped step uses t+1 in its bias correction (apex would use t)."""
        while len(self._flags_pending) > (0 if drain else 1):
            ev, host = self._flags_pending.pop(0)
            ev.synchronize()
            f = host.tolist()
            self._flags_free.append(host)
            r = self.opt.DG_ratio
            per_opt = ((self.optimizer_G, f[0:1]), (self.optimizer_D_PP, f[1:1 + r]),
                       (self.optimizer_D_PB, f[1 + r:1 + 2 * r]))
            for o, fl in per_opt:
                o.step_count -= sum(1 for x in fl if x)
            self.skipped_steps += sum(1 for x in f if x)
            self.last_overflow = any(f)
            if self.last_overflow:
                self.pprint("non-finite gradient: skipped %d optimizer step(s)" % sum(1 for x in f if x))

    # ------------------------------------------------------------------ the step
    def optimize_parameters(self):
        """models/MMHandModel.py:310-330: one G step, DG_ratio D_PP steps, DG_ratio D_PB steps, each
        optimizer step skipped once a gradient of the iteration was not finite."""
        self._settle_overflow()
        ops.lp_grads_reset()
        if self.dp:
            self._optimize_parameters_dp()
        elif self.graph_step:
            self._optimize_parameters_graph()
        else:
            self._step_body()
        host = self._flags_free.pop()
        host.copy_(self._flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._flags_pending.append((ev, host))
        self.overflow = False

    def _step_body(self):
        """the single-process iteration: every launch of it, nothing else (what --graph_step captures)"""
        r = self.opt.DG_ratio
        # single process: the wgrad kernels add straight into the parameters' .grad views of the flat
        # gradient buffer (no AccumulateGrad add kernels; ops.ACCUM_PARAM_GRADS)
        prev, ops.ACCUM_PARAM_GRADS = ops.ACCUM_PARAM_GRADS, True
        try:
            self.forward()
            self.optimizer_G.zero_grad()
            self.backward_G()
            self._guarded_step(self.optimizer_G, 0, 0)
            for i in range(r):
                self.optimizer_D_PP.zero_grad()
                self.backward_D_PP()
                self._guarded_step(self.optimizer_D_PP, 1 + i, 2)
            for i in range(r):
                self.optimizer_D_PB.zero_grad()
                self.backward_D_PB()
                self._guarded_step(self.optimizer_D_PB, 1 + r + i, 1)
        finally:
            ops.ACCUM_PARAM_GRADS = prev

    # ------------------------------------------------------------------ the step as a captured graph
    def _graph_body(self):
        """the iteration in its replayable form: host dropout seeds restart from the model's base (every site draws the same
        by-value seed each iteration), the device salt - advanced here, on the stream - makes the masks fresh"""
        ops.set_dropout_seed(self._seed_base)
        L.call("mmh_set_dropout_salt", ops._ptr(self._salt))
        try:
            L.call("mmh_u64_add", ops._ptr(self._salt), 0x9E3779B97F4A7C15, ops._stream())
            self._step_body()
        finally:
            L.call("mmh_set_dropout_salt", None)

    def _optimize_parameters_graph(self):
        """models/MMHandModel.py:310-330 replayed: the first iterations run the replayable form eagerly (derived-weight
        batches and workspaces settle), then ONE iteration is captured into a hipGraph (torch.cuda.CUDAGraph: own stream,
        own memory pool) and every later call is: upload the image pools' decisions, replay.  A capture that fails leaves
        the model in the eager replayable form for good (`graph_error` says why) - never a process re-exec."""
        r = self.opt.DG_ratio
        B = self.input_H1.shape[0]
        for pool in (self.fake_PP_pool, self.fake_PB_pool):       # the reference's order of draws: D_PP's queries, then D_PB's
            pool.begin_iteration(B, self.device)
        self.optimizer_G.step_count += 1
        self.optimizer_D_PP.step_count += r
        self.optimizer_D_PB.step_count += r
        self._graph_iters += 1
        if self._graph_state == "replay" and not getattr(self, "_graph_odd", False):
            self._graph.replay()
            self.graph_replays += 1
            self.__dict__.update(self._graph_outputs)      # (an odd-shaped batch in between had re-pointed them)
            return
        if self._graph_state == "warmup" and self._graph_iters > self._graph_warm and os.environ.get("MMH_GRAPH_CAPTURE", "1") != "0":
            try:
                self._capture_step(B)
                self._graph.replay()        # capture executes nothing: this replay IS the iteration
                self.graph_replays += 1
                return
            except Exception as e:          # noqa: BLE001 - any capture failure: stay eager, say why
                self._graph, self._graph_state = None, "eager"
                self.graph_error = f"{type(e).__name__}: {e}"[:500]
                self.pprint("--graph_step: capture failed, staying eager (%s)" % self.graph_error)
                torch.cuda.synchronize()
                for pool in (self.fake_PP_pool, self.fake_PB_pool):
                    pool.begin_iteration(B, self.device, decide=False)     # the decisions already drawn for this iteration
                ops.lp_grads_reset()
        self._graph_body()

    def _capture_step(self, B):
        torch.cuda.synchronize()
        import gc
        # The previous iteration's autograd graph must be GONE before the capture builds its own: a leaf's AccumulateGrad
        # node is bound to the stream it was created on and is re-used while anything keeps the old graph alive - the
        # captured backward would then run the norm scales' / shifts' accumulation (the only gradients that still travel
        # through AccumulateGrad: the conv shims add theirs in place) on the eager iterations' stream, a fork out of the
        # capture that hipStreamEndCapture answers with a segfault (--norm batch; tools/probes/graph_bisect.py).
        self.fake_nhwc = self.fake_p2 = None
        self.loss_G_L1 = self.loss_G_GAN_PB = self.loss_G_GAN_PP = None
        self._fake_cats = None
        ops.lp_grads_reset()
        gc.collect()
        self._static_inputs = {k: getattr(self, k) for k in ("input_H1", "input_P1", "input_D1", "input_H2", "input_P2",
                                                             "input_D2", "x_H1", "x_P", "x_D", "x_H2")}
        self._static_twins = {k: ops.pack_twin_get(self._static_inputs[k], self.bf16, pop=False) if self.bf16 else None
                              for k in ("x_H1", "x_P", "x_D")}
        g = torch.cuda.CUDAGraph()
        for pool in (self.fake_PP_pool, self.fake_PB_pool):
            pool._q = 0
        with torch.cuda.graph(g):
            self._graph_body()
        # derived-weight caches filled before the capture are written by the replays: the graph keeps them alive
        self._graph_keep = (ops.derived_weights_snapshot(), ops.derived_batches_snapshot())
        # what the captured iteration leaves behind (static tensors the replays refill): losses, the generated image
        self._graph_outputs = {k: self.__dict__[k] for k in (
            "fake_nhwc", "fake_p2", "pair_L1loss", "pair_GANloss", "loss_D_PP", "loss_D_PB", "loss_originL1", "loss_perceptual",
            "loss_G_L1", "loss_G_GAN_PB", "loss_G_GAN_PP", "_fake_cats") if k in self.__dict__}
        self._graph, self._graph_state, self._graph_batch = g, "replay", B

    def _optimize_parameters_dp(self):
        """The same iteration under data parallelism.  Effects land in the reference's order (the
        flag chain and the optimizer steps run G -> D_PP -> D_PB), but the three backward passes are
        ENQUEUED back to back: each network's gradient buckets are all-reduced on the side stream
        from autograd hooks as they complete (dp.GradBuckets), so the Generator's 285 MB travel
        beneath the rest of its own backward and both discriminator passes, and D_PP's 16 MB beneath
        D_PB's forward + backward.  That is legal because the discriminator steps read only the
        already generated, detached image (models/MMHandModel.py:279-289) and the two discriminators
        share nothing.  A network's step is flushed before that network is used again (DG_ratio > 1)."""
        r = self.opt.DG_ratio
        pending = []

        def flush():
            for bk, o, k, lid in pending:
                bk.wait()                       # current stream waits for that network's collectives
                self._guarded_step(o, k, lid)
            pending.clear()

        def backward_of(o, fn, k, lid, pre=None):
            if any(q[1] is o for q in pending):
                flush()                         # its previous step must land before its next forward
            bk = self._buckets[o]
            o.zero_grad()
            bk.begin()                          # armed BEFORE the forward pass: the conv shims report their parameter uses
            if pre is not None:
                pre()
            fn()
            bk.launch_remaining()
            pending.append((bk, o, k, lid))

        prev, ops.ACCUM_PARAM_GRADS = ops.ACCUM_PARAM_GRADS, self.dp_accum
        try:
            backward_of(self.optimizer_G, self.backward_G, 0, 0, pre=self.forward)
            for i in range(r):
                backward_of(self.optimizer_D_PP, self.backward_D_PP, 1 + i, 2)
            for i in range(r):
                backward_of(self.optimizer_D_PB, self.backward_D_PB, 1 + r + i, 1)
            flush()
        finally:
            ops.ACCUM_PARAM_GRADS = prev

    # ------------------------------------------------------------------ reporting / io
    def get_current_errors(self):
        return OrderedDict([("pair_L1loss", self.pair_L1loss), ("D_PP", self.loss_D_PP),
                            ("D_PB", self.loss_D_PB), ("pair_GANloss", self.pair_GANloss),
                            ("origin_L1", self.loss_originL1), ("perceptual", self.loss_perceptual)])

    def get_current_visuals(self):
        """models/MMHandModel.py:343-369: H1 | P1 | D1 | H2 | P2 | D2 | fake strip of sample 0 as one
        uint8 [H, 7W, 3] array; the pose panels are draw_pose_from_map skeleton drawings
        (mmhand_amd/visuals.py)."""
        from . import visuals
        vis = visuals.visual_strip(self.input_H1, self.input_P1, self.input_D1, self.input_H2, self.input_P2,
                                   self.input_D2, self.fake_p2)
        return OrderedDict([("vis", vis)])

    def save_network(self, network, network_label, epoch_label, gpu_ids=None):
        if self.master:
            os.makedirs(self.save_dir, exist_ok=True)
            path = os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch_label, network_label))
            torch.save(OrderedDict((k, v.cpu()) for k, v in network.state_dict().items()), path)

    def save(self, label):
        """Call on EVERY rank (mmhand_amd/train.py does): the overflow flags are settled on the same
        schedule everywhere, so the Adam step counts stay identical across ranks; only the master
        writes files."""
        self.save_network(self.netG, "netG", label, self.gpu_ids)
        self.save_network(self.netD_PB, "netD_PB", label, self.gpu_ids)
        self.save_network(self.netD_PP, "netD_PP", label, self.gpu_ids)
        self.save_train_state(label)

    def save_train_state(self, label):
        """<label>_net_amp.pth — the file the reference's distributed runs fill with apex's
        loss-scaler state (models/base_model.py:54-56).  Here it carries what a resumed run needs: the
        three dynamic loss scalers ({scale, clean steps} per loss, what apex's amp.state_dict() holds), the
        three Adam states (step count and the flat exp_avg / exp_avg_sq buffers, in
        flatten_parameters() order) and the skipped-step count.  The learning-rate schedule is positioned by --epoch_count as in the reference
        (network_utils.py:92-95), so scheduler state is not stored.  The reference's loader feeds
        any *amp* file to amp.load_state_dict inside try/except, so it ignores this one."""
        self._settle_overflow(drain=True)
        if not self.master:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        state = {"format": "mmhand_amd.train_state.v1", "skipped_steps": self.skipped_steps,
                 "optimizers": {},
                 # {scale, clean steps} of the three dynamic loss scalers: what apex's amp.state_dict()
                 # holds in this file in the reference (loss_scaler0..2: loss_scale, unskipped)
                 "loss_scalers": self._scaler.cpu()}
        for name in ("optimizer_G", "optimizer_D_PB", "optimizer_D_PP"):
            sd = getattr(self, name).state_dict()
            state["optimizers"][name] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}
        torch.save(state, os.path.join(self.save_dir, "%s_net_amp.pth" % label))

    def load_train_state(self):
        path = os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_net_amp.pth" % self.opt.which_epoch)
        if not os.path.exists(path):
            return False
        state = torch.load(path, map_location="cpu")
        if not isinstance(state, dict) or state.get("format") != "mmhand_amd.train_state.v1":
            return False        # a real apex amp file: nothing to restore without a loss scaler
        for name, sd in state["optimizers"].items():
            o = getattr(self, name)
            if sd["exp_avg"].numel() != o.exp_avg.numel():
                raise RuntimeError(f"{path}: {name} state has {sd['exp_avg'].numel()} elements, "
                                   f"the network has {o.exp_avg.numel()}")
            o.load_state_dict(sd)
        self.skipped_steps = int(state.get("skipped_steps", 0))
        if "loss_scalers" in state:
            self._scaler = state["loss_scalers"].to(self.device, torch.float32).contiguous()
        self.pprint("restored optimizer state (Adam step %d)" % self.optimizer_G.step_count)
        return True

    def load_network(self):
        """models/base_model.py:60-80: load every <which_epoch>_net_<name>.pth in the run dir."""
        opt = self.opt
        d = os.path.join(opt.checkpoints_dir, opt.name)
        for fn in sorted(os.listdir(d)):
            if opt.which_epoch not in fn or not fn.endswith(".pth") or "amp" in fn:
                continue
            name = fn[:-4].replace(f"{opt.which_epoch}_net_", "")
            sub = getattr(self, name, None)
            if sub is None:
                continue
            sub.load_state_dict(torch.load(os.path.join(d, fn), map_location="cpu"))
            self.pprint(f"loading weights for {name}")

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()
        for o in self.optimizers:
            o.sync_lr()         # --graph_step: the device copy the replayed Adam launches read
        lr = self.optimizers[0].param_groups[0]["lr"]
        self.pprint("learning rate = %.7f" % lr)

    def pprint(self, msg):
        if self.master:
            print(msg)
