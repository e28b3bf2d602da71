"""Command-line surface of the reference (options/base_options.py:15-163,
options/train_options.py:7-40, options/test_options.py:4-14) — same flag names, types and
defaults, so scripts/mm-train-ratio.sh-style invocations parse unchanged.

Differences, all deliberate:
  * the non-distributed branch works (the reference raises AttributeError at
    base_options.py:191 by calling .split on a list);
  * --distributed reads RANK/LOCAL_RANK/WORLD_SIZE from the environment (torchrun) and
    initialises the 'nccl' backend, which is RCCL on ROCm;
  * --opt_level O0 = fp32; O1/O2 (apex AMP in the reference) = bf16 MFMA compute with fp32 master
    weights, accumulation and statistics plus apex's dynamic loss scaling (three device-resident
    scalers, skip / back off / grow); O1_FP16/O2_FP16 = the same with IEEE fp16 operands (apex's own
    numerics); BF16 = the bf16 compute without a scaler;
  * conv biases that feed an InstanceNorm get their exact (zero) gradient by default instead of the reference's
    rounding noise (ops.EXACT_NULL_BIAS_GRAD; MMH_NULL_BIAS_GRAD=compute restores it);
  * --fp32_exact_grads (addition): the gradient-exact fp32 hybrid, see ops.set_winograd_mode;
  * --graph_step (addition): single-process training replays the whole iteration from a captured hipGraph
    (MMHandModel._optimize_parameters_graph);
  * three additions: --G_n_blocks (the reference hard-codes 9), --vgg_weights (file with
    torchvision vgg19.features[0:4] weights; there is no download path offline) and
    --vgg_random_init (explicit opt-in to seeded random VGG weights; without either of the two
    the default --L1_type l1_plus_perL1 refuses to start).
"""
import argparse
import os

# (flag, kwargs) tables — base, train, test
_BASE = [
    ("--imageroot", dict(type=str, help="path to images")),
    ("--poseroot", dict(type=str, help="path to poses")),
    ("--batchSize", dict(type=int, help="input batch size")),
    ("--fineSize", dict(type=int, default=256, help="then crop to this size")),
    ("--output_nc", dict(type=int, default=3, help="# of output image channels")),
    ("--ngf", dict(type=int, default=64, help="# of generator filters in first conv layer")),
    ("--ndf", dict(type=int, default=64, help="# of discrimator filters in first conv layer")),
    ("--n_layers_D", dict(type=int, default=3, help="blocks used in D")),
    ("--gpu_ids", dict(type=str, default="0", help="gpu ids: e.g. 0  0,1,2, 0,2. use -1 for CPU")),
    ("--name", dict(type=str, default="experiment_name", help="name of the experiment")),
    ("--nThreads", dict(type=int, default=8, help="# threads for loading data")),
    ("--checkpoints_dir", dict(type=str, default="./checkpoints", help="models are saved here")),
    ("--norm", dict(type=str, default="batch", help="instance normalization or batch normalization")),
    ("--serial_batches", dict(action="store_true", help="take images in order to make batches")),
    ("--display_winsize", dict(type=int, default=256, help="display window size")),
    ("--display_id", dict(type=int, default=0, help="window id of the web display")),
    ("--display_port", dict(type=int, default=8097, help="visdom port of the web display")),
    ("--no_dropout", dict(action="store_true", help="no dropout for the generator")),
    ("--max_dataset_size", dict(type=int, default=float("inf"), help="max samples per dataset")),
    ("--no_flip", dict(action="store_true", help="do not flip the images for augmentation")),
    ("--init_type", dict(type=str, default="normal", help="network initialization")),
    ("--H_input_nc", dict(type=int, default=3, help="# of input image channels")),
    ("--P_input_nc", dict(type=int, default=21, help="# of pose-map channels")),
    ("--D_input_nc", dict(type=int, default=3, help="# of depth channels")),
    ("--padding_type", dict(type=str, default="reflect", help="padding type (always reflect)")),
    ("--pairLst", dict(type=str, help="market pairs")),
    ("--use_flip", dict(type=int, default=0, help="flip or not")),
    ("--G_n_downsampling", dict(type=int, default=2, help="down-sampling blocks for generator")),
    ("--D_n_downsampling", dict(type=int, default=2, help="down-sampling blocks for discriminator")),
    ("--augmentation_ratio", dict(type=float)),
    ("--augmentation_method", dict(type=str)),
    ("--dataset_mode", dict(type=str)),
    ("--dataset", dict(type=str)),
    ("--dataroot", dict(type=str)),
    ("--local_rank", dict(type=int, default=0, help="determine which is the master process")),
    ("--distributed", dict(action="store_true", help="one process per GPU, RCCL all-reduce")),
    ("--seed", dict(type=int, default=49, help="manual seed for weight init")),
    ("--opt_level", dict(type=str, default="O0",
                         help="O0 fp32 | O1/O2 bf16 MFMA compute + dynamic loss scaling | O1_FP16/O2_FP16 the same in "
                              "IEEE fp16 | BF16 (no scaler)")),
    ("--G_n_blocks", dict(type=int, default=9, help="PATBlocks in the generator")),
    ("--vgg_weights", dict(type=str, default=None, help="vgg19.features[0:4] state_dict file")),
    ("--vgg_random_init", dict(action="store_true",
                               help="perceptual loss on seeded RANDOM VGG weights (benchmarks / tests; "
                                    "not the reference's objective)")),
    ("--fp32_exact_grads", dict(action="store_true",
                                help="fp32 only: forward 3x3 convs on the direct implicit-GEMM kernels with two-level summation, "
                                     "dgrad / wgrad on Winograd F(6x6,3x3) - at 256x256 the parameter gradients a median 9.5e-4 "
                                     "from float64 (PyTorch's own fp32: 7.6e-4; the all-Winograd default: 3e-3 on this network's "
                                     "ill-conditioned gradients); = MMH_WINOGRAD=bwd")),
    ("--graph_step", dict(action="store_true",
                          help="single process: capture one optimize_parameters() - forward, three backward passes, three Adam "
                               "steps - into a hipGraph after a few eager iterations and replay it (Adam step count / lr, dropout "
                               "salt and image-pool decisions live behind device pointers); = MMH_GRAPH_STEP=1")),
]
_TRAIN = [
    ("--display_freq", dict(type=int, default=100)),
    ("--display_single_pane_ncols", dict(type=int, default=0)),
    ("--update_html_freq", dict(type=int, default=1000)),
    ("--print_freq", dict(type=int, default=100)),
    ("--save_latest_freq", dict(type=int, default=5000)),
    ("--save_epoch_freq", dict(type=int, default=1)),
    ("--continue_train", dict(action="store_true")),
    ("--epoch_count", dict(type=int, default=1)),
    ("--phase", dict(type=str, default="train")),
    ("--which_epoch", dict(type=str, default="latest")),
    ("--niter", dict(type=int, default=500)),
    ("--niter_decay", dict(type=int, default=200)),
    ("--beta1", dict(type=float, default=0.5)),
    ("--lr", dict(type=float, default=0.0002)),
    ("--no_lsgan", dict(action="store_true")),
    ("--lambda_A", dict(type=float, default=10.0)),
    ("--lambda_B", dict(type=float, default=10.0)),
    ("--lambda_GAN", dict(type=float, default=5.0)),
    ("--pool_size", dict(type=int, default=50)),
    ("--no_html", dict(action="store_true")),
    ("--lr_policy", dict(type=str, default="lambda")),
    ("--lr_decay_iters", dict(type=int, default=50)),
    ("--L1_type", dict(type=str, default="l1_plus_perL1")),
    ("--perceptual_layers", dict(type=int, default=3)),
    ("--percep_is_l1", dict(type=int, default=1)),
    ("--no_dropout_D", dict(action="store_true")),
    ("--DG_ratio", dict(type=int, default=1)),
]
_TEST = [
    ("--ntest", dict(type=int, default=float("inf"))),
    ("--results_dir", dict(type=str, default="./results/")),
    ("--aspect_ratio", dict(type=float, default=1.0)),
    ("--phase", dict(type=str, default="test")),
    ("--which_epoch", dict(type=str, default="latest")),
    ("--how_many", dict(type=int, default=200)),
]


class BaseOptions:
    isTrain = None
    _extra = []

    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        self.initialized = False

    def initialize(self):
        for flag, kw in _BASE + self._extra:
            self.parser.add_argument(flag, **kw)
        self.initialized = True

    def parse(self, args=None, init_dist=True, save=True):
        if not self.initialized:
            self.initialize()
        opt = self.parser.parse_args(args)
        opt.isTrain = self.isTrain
        import torch
        if opt.distributed:
            opt.local_rank = int(os.environ.get("LOCAL_RANK", opt.local_rank))
            opt.gpu = opt.local_rank
            if torch.cuda.is_available():
                torch.cuda.set_device(opt.gpu)
            if init_dist and not torch.distributed.is_initialized():
                backend = "nccl" if torch.cuda.is_available() else "gloo"
                torch.distributed.init_process_group(backend=backend, init_method="env://")
            opt.world_size = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
            if opt.batchSize is not None:
                opt.batchSize = opt.batchSize // opt.world_size   # base_options.py:178
            opt.gpu_ids = [opt.local_rank]
        else:
            opt.gpu_ids = [int(s) for s in str(opt.gpu_ids).split(",") if int(s) >= 0]
            opt.gpu = opt.gpu_ids[0] if opt.gpu_ids else -1
            if opt.gpu_ids and torch.cuda.is_available():
                torch.cuda.set_device(opt.gpu_ids[0])
            opt.world_size = 1
        self.opt = opt
        if save:
            self._dump(opt)
        return opt

    @staticmethod
    def _dump(opt):
        lines = ["------------ Options -------------"]
        lines += ["%s: %s" % (k, v) for k, v in sorted(vars(opt).items())]
        lines += ["-------------- End ----------------"]
        if opt.local_rank == 0:
            print("\n".join(lines))
            expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
            os.makedirs(expr_dir, exist_ok=True)
            with open(os.path.join(expr_dir, "opt.txt"), "wt") as f:
                f.write("\n".join(lines) + "\n")


class TrainOptions(BaseOptions):
    isTrain = True
    _extra = _TRAIN


class TestOptions(BaseOptions):
    isTrain = False
    _extra = _TEST


def default_train_opt(**overrides):
    """Programmatic TrainOptions namespace (defaults of the tables above) for bench/tests.  Never
    touches the current CUDA device or the process group: the caller owns both.  Unlike the command
    line it opts in to --vgg_random_init (bench and tests have no pretrained VGG file)."""
    o = TrainOptions()
    o.initialize()
    opt = o.parser.parse_args([])
    opt.isTrain = True
    lr = int(overrides.get("local_rank", 0))
    opt.gpu_ids = [lr]
    opt.gpu = lr
    opt.world_size = 1
    opt.vgg_random_init = True
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt
