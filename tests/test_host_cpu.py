"""Host-side logic that needs no GPU: the C-ABI library exports, the options surface, the
checkpoint (state_dict) contract of the drop-in networks."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def built_lib():
    from mmhand_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mmhand_amd", "csrc")])
    return lib


def test_abi_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "mmhand_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mmh_[a-z0-9_]+)\s*\(", hdr, flags=re.I))
    declared = {d for d in declared if not d.endswith("_t")}
    assert len(declared) >= 30
    assert declared == set(built_lib.SIGNATURES), declared ^ set(built_lib.SIGNATURES)
    cdll = ctypes.CDLL(built_lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), name
    assert built_lib.load().mmh_version() >= 100
    # ... and nothing else: compiled with -fvisibility=hidden and linked through csrc/exports.map (global: mmh_*), the header
    # is the boundary - not even hipcc's per-translation-unit `__hip_cuid_*` markers are in the dynamic symbol table
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_abi_rejects_bad_arguments_without_gpu(built_lib):
    """Argument validation happens before any launch, so it is testable on the CPU."""
    l = built_lib.load()
    d = built_lib.ConvDesc(1, 8, 8, 3, 8, 3, 3, 1, 1, 0, 8, 8, 3, 8, 0)   # Cin=3 not %4
    assert l.mmh_conv2d_fprop(ctypes.byref(d), None, None, None, None, 0, None) != 0
    assert b"multiples of 4" in l.mmh_last_error()
    assert l.mmh_adam_step(None, None, None, None, 10, 1e-3, 0.5, 0.999, 1e-8, 1, 1.0, None, None, None) != 0
    assert l.mmh_grad_nonfinite(None, 10, None, None, None, None) != 0
    assert l.mmh_loss_scale_update(None, None, 2.0, 0.5, 2000, 1.0, 2.0 ** 24, None) != 0


def test_round5_entry_points_plan_and_validate_without_gpu(built_lib):
    """The planning / validation half of the round-5 entry points runs on the CPU: which shapes the Generator head's 16-bit
    weight gradient, the dgrad with the norm-backward sums, the stride-2 nine-tap weight gradient, the 3x3 form of the stem
    kernel and the LDS-staged pack accept, their workspace sizes, and that bad arguments are refused before any launch."""
    L = built_lib
    l = L.load()
    mk = lambda B, H, W, Cin, Cout, k, s, p, refl, dt=L.BF16: L.ConvDesc(B, H, W, Cin, Cout, k, k, s, p, L.PAD_REFLECT if refl else L.PAD_ZERO,
                                                                       (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1, Cin, Cout, dt)
    by = ctypes.byref
    # the head: 7x7 / reflect 3 / 64 -> 4 only, 16-bit only
    head = mk(2, 32, 48, 64, 4, 7, 1, 3, True)
    assert l.mmh_conv7_head_wgrad_lp16_supported(by(head)) == 1
    assert l.mmh_conv7_head_wgrad_lp16_supported(by(mk(2, 32, 48, 64, 4, 7, 1, 3, True, L.F32))) == 0
    assert l.mmh_conv7_head_wgrad_lp16_supported(by(mk(2, 32, 48, 64, 4, 7, 1, 3, False))) == 0
    assert l.mmh_conv7_head_wgrad_lp16_supported(by(mk(2, 32, 48, 32, 4, 7, 1, 3, True))) == 0
    need = l.mmh_conv7_head_wgrad_lp16_ws_bytes(by(head))
    assert need >= 2 * 38 * 54 * 16 + 49 * 64 * 8 * 4            # the embedded dy + at least one slab
    assert l.mmh_conv7_head_wgrad_lp16(by(head), None, None, None, None, 0, 0, None, None) != 0
    # the dgrad whose epilogue takes the norm's backward sums: full 16 x 16 tiles, 256-column tiles, the halo kernel
    assert l.mmh_conv3x3_lp16_dgrad_nbr_chunks(by(mk(2, 32, 48, 256, 256, 3, 1, 1, True)), 2) == 2 * 2 * 3
    assert l.mmh_conv3x3_lp16_dgrad_nbr_chunks(by(mk(2, 32, 48, 256, 256, 3, 1, 1, False)), 1) == 2 * 2 * 3
    assert l.mmh_conv3x3_lp16_dgrad_nbr_chunks(by(mk(2, 32, 40, 256, 256, 3, 1, 1, False)), 1) == 0       # ragged tiles
    assert l.mmh_conv3x3_lp16_dgrad_nbr_chunks(by(mk(2, 16, 16, 256, 256, 3, 1, 1, True)), 2) == 0        # fold needs two tiles
    assert l.mmh_conv3x3_lp16_dgrad_nbr_chunks(by(mk(2, 32, 48, 128, 256, 3, 1, 1, False)), 1) == 0       # 128 dx columns
    assert l.mmh_conv3x3_lp16_dgrad_nbr(by(mk(2, 32, 40, 256, 256, 3, 1, 1, False)), 1, *([None] * 7), 2, 0.0, None, None, None, 0, None, None) != 0
    assert b"dgrad_nbr" in l.mmh_last_error()
    assert l.mmh_norm_bwd_sums_final(None, 2, 256, 12, None, None, None) != 0
    # stride-2 weight gradients: the nine-tap halo kernel's slabs decide the workspace (S x 9 x Cin x Cout floats)
    s2 = mk(4, 64, 64, 64, 128, 3, 2, 1, False)
    assert l.mmh_wgrad_lp16_flat_supported(by(s2), 64) == 1
    ws = l.mmh_wgrad_lp16_flat_ws_bytes(by(s2), 64)
    assert ws == 32 * (9 * 64 * 128 * 4)        # 4 x 16 x 2 blocks of 2 x 16 output pixels, at least four per split: 32 slabs
    assert l.mmh_set_option(b"lp16_wgrad_s2", 0) == 0
    try:
        assert l.mmh_wgrad_lp16_flat_ws_bytes(by(s2), 64) < ws                      # the flat-row kernel's own, smaller, slabs
    finally:
        assert l.mmh_set_option(b"lp16_wgrad_s2", 1) == 0
    # the stem kernel: 7x7 at C8 in 8..48, 3x3 at C8 == 8 only
    assert l.mmh_conv_stem16_supported(by(mk(2, 32, 32, 4, 64, 3, 1, 1, False)), 8) == 1
    assert l.mmh_conv_stem16_supported(by(mk(2, 32, 32, 12, 64, 3, 1, 1, False)), 16) == 0
    assert l.mmh_conv_stem16_supported(by(mk(2, 32, 32, 24, 64, 7, 1, 3, True)), 24) == 1
    assert l.mmh_conv_stem16_weights_bytes_k(8, 3) == 3 * 64 * (32 + 8) * 2 and l.mmh_conv_stem16_weights_bytes_k(16, 3) == 0
    assert l.mmh_conv_stem16_weights_bytes_k(24, 7) == l.mmh_conv_stem16_weights_bytes(24) > 0
    assert l.mmh_prep_weights_stem16_k(None, 3, 8, 3, L.BF16, None, None) != 0
    # the LDS-staged pack: Cd % 4, Cd <= 56, the 16-bit copy padded to a multiple of 8
    src = (L.PlaneSrc * 1)(L.PlaneSrc(None, 3, 0, 0, 0, 0))
    assert l.mmh_pack_nhwc_lp16(src, 1, None, None, 1, 4, 4, 4, 8, L.BF16, None) != 0            # neither output
    assert l.mmh_pack_nhwc_lp16(src, 1, ctypes.c_void_p(64), None, 1, 4, 4, 60, 0, L.BF16, None) != 0  # Cd > 56
    assert l.mmh_pack_nhwc_lp16(src, 1, None, ctypes.c_void_p(64), 1, 4, 4, 4, 12, L.BF16, None) != 0  # C8 % 8
    assert b"mmh_pack_nhwc_lp16" in l.mmh_last_error()


def test_mmh_options_environment(built_lib):
    """MMH_OPTIONS=key=value,... reaches mmh_set_option when the library loads; an unknown key stops the run"""
    code = "from mmhand_amd import lib; lib.load(); print('loaded')"
    ok = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                        env=dict(os.environ, MMH_OPTIONS="lp16_persist=0, col_chunks=1024"))
    assert ok.returncode == 0 and "loaded" in ok.stdout, ok.stderr[-400:]
    for bad in ("no_such_key=1", "lp16_persist", "lp16_persist=x"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, MMH_OPTIONS=bad))
        assert r.returncode != 0 and "MMH_OPTIONS" in r.stderr, (bad, r.stderr[-300:])


def test_product_path_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mmhand_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_options_surface():
    from mmhand_amd.options import TrainOptions, TestOptions
    args = ("--dataroot ./datasets/stb_dataset/train --name x --lambda_GAN 5 --lambda_A 10 "
            "--lambda_B 10 --no_lsgan --n_layers 3 --batchSize 3 --no_flip --nThreads 4 "
            "--checkpoints_dir /tmp/mmh_ckpt --opt_level O1 --augmentation_ratio 1 "
            "--augmentation_method GEN --dataset stb --save_latest_freq 400 --niter 100 "
            "--niter_decay 0").split()     # scripts/mm-train-ratio.sh:19-40 without --distributed
    opt = TrainOptions().parse(args, save=False)
    assert opt.isTrain and opt.n_layers_D == 3 and opt.batchSize == 3 and opt.norm == "batch"
    assert opt.lr == 2e-4 and opt.beta1 == 0.5 and opt.pool_size == 50 and opt.DG_ratio == 1
    assert opt.H_input_nc == 3 and opt.P_input_nc == 21 and opt.D_input_nc == 3 and opt.seed == 49
    assert opt.gpu_ids == [0] and not opt.no_dropout and not opt.no_dropout_D
    t = TestOptions().parse(["--name", "x", "--checkpoints_dir", "/tmp/mmh_ckpt"], save=False)
    assert not t.isTrain and t.how_many == 200


def test_lambda_lr_rule():
    from mmhand_amd.options import default_train_opt
    from mmhand_amd.mmhand_model import get_scheduler
    opt = default_train_opt(niter=2, niter_decay=2, epoch_count=1)
    p = torch.zeros(1, requires_grad=True)
    o = torch.optim.SGD([p], lr=1.0)
    s = get_scheduler(o, opt)
    lrs = []
    for _ in range(4):
        lrs.append(o.param_groups[0]["lr"])
        o.step(); s.step()
    assert np.allclose(lrs, [1.0, 1.0 - 1 / 3, 1.0 - 2 / 3, 0.0])


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_state_dict_contract_matches_reference(norm):
    """Key names and logical shapes of the true-size networks == the reference's (keys.json
    was dumped from the reference classes)."""
    from mmhand_amd.networks import Generator, Discriminator
    keys = json.load(open(os.path.join(G, "keys.json")))[norm]
    g = Generator([3, 42, 6], 3, 64, norm, True, 9)
    sd = g.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == keys["G"]
    g2 = Generator([3, 42, 6], 3, 64, norm, False, 9)
    assert {k: list(v.shape) for k, v in g2.state_dict().items()} == keys["G_nodrop"]
    for cin, tag in ((24, "D_PB"), (6, "D_PP")):
        d = Discriminator(cin, 64, norm, True, 3)
        assert {k: list(v.shape) for k, v in d.state_dict().items()} == keys[tag]
    # logical parameter count equals the reference's
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == keys["G_params"]


def test_state_dict_roundtrip_and_layout():
    from mmhand_amd.networks import Generator
    from tests.golden import recipe as RC
    g = Generator([3, 42, 6], 3, 8, "batch", True, 2)
    shapes = {k: tuple(v.shape) for k, v in g.state_dict().items()}
    sd = RC.recipe_state_dict(shapes)
    g.load_state_dict(sd)
    back = g.state_dict()
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    # physical layout: [kh,kw,Cin_pad,Cout]; pad lanes are zero
    cp = g.model["stream1_down"][1]
    assert tuple(cp.weight.shape) == (7, 7, 4, 8)
    assert torch.equal(cp.weight[:, :, :3, :].permute(3, 2, 0, 1), sd["model.stream1_down.1.weight"])
    assert float(cp.weight[:, :, 3, :].abs().sum()) == 0.0
    up = g.model["stream1_up"][0]                      # ConvTranspose2d(32 -> 16): logical [32,16,3,3]
    assert tuple(up.logical_weight().shape) == (32, 16, 3, 3) and tuple(up.weight.shape) == (3, 3, 16, 32)
    with pytest.raises(RuntimeError):
        bad = dict(sd); bad["model.stream1_down.1.weight"] = torch.zeros(8, 4, 7, 7)
        g.load_state_dict(bad)


def test_num_batches_tracked_is_counted_on_the_host_and_read_out_by_state_dict():
    """nn.BatchNorm2d bumps num_batches_tracked in every training forward (a checkpoint key of the reference's --norm batch
    nets); here the forwards are counted on the host and folded into the buffer when state_dict() reads it."""
    from mmhand_amd.networks import NormParam
    np_ = NormParam(8)
    for _ in range(3):
        np_.count_batch()
    assert int(np_.state_dict()["num_batches_tracked"]) == 3
    np_.count_batch()
    assert int(np_.state_dict()["num_batches_tracked"]) == 4 and int(np_.flush_batches()) == 4
    other = NormParam(8)
    other.count_batch()
    other.load_state_dict(np_.state_dict())         # a loaded count replaces what was pending
    assert int(other.state_dict()["num_batches_tracked"]) == 4


def test_flat_parameter_views():
    from mmhand_amd.networks import Discriminator
    d = Discriminator(6, 8, "batch", False, 1).init_weights("normal", seed=1)
    before = {k: v.clone() for k, v in d.state_dict().items()}
    flat, gflat = d.flatten_parameters()
    assert flat.numel() == sum(p.numel() for p in d.parameters())
    for k, v in d.state_dict().items():
        assert torch.equal(v, before[k]), k
    flat.mul_(2.0)                                     # parameters are views of the flat buffer
    assert torch.allclose(d.state_dict()["model.1.weight"], before["model.1.weight"] * 2)
    p = next(d.parameters())
    assert p.grad.data_ptr() == gflat.data_ptr()


@pytest.mark.parametrize("kind", ["normal", "xavier", "kaiming", "orthogonal"])
def test_init_types(kind):
    """models/network_utils.py:12-71: the four --init_type choices."""
    from mmhand_amd.networks import Discriminator
    d = Discriminator(6, 16, "batch", False, 1).init_weights(kind, seed=3)
    sd = d.state_dict()
    w = sd["model.4.weight"]                                   # Conv2d(16 -> 32, 3x3)
    if kind == "normal":
        assert abs(w.std().item() - 0.02) < 0.004
    elif kind == "kaiming":
        assert abs(w.std().item() - (2.0 / (16 * 9)) ** 0.5) < 0.02
    elif kind == "xavier":
        assert abs(w.std().item() - 0.02 * (2.0 / (16 * 9 + 32 * 9)) ** 0.5) < 1e-3
    else:
        m = w.reshape(32, -1)
        assert torch.allclose(m @ m.t(), torch.eye(32), atol=1e-4)
    assert torch.allclose(sd["model.5.bias"], torch.zeros(32)) and abs(sd["model.5.weight"].mean().item() - 1) < 0.02
    with pytest.raises(NotImplementedError):
        d.init_weights("bogus")


def test_visuals_colormap_and_tensor2im_match_reference_fixture():
    """util/util.py labelcolormap(22) / tensor2im outputs stored by tests/golden/make_golden.py."""
    from mmhand_amd import visuals as V
    fix = dict(np.load(os.path.join(G, "visuals.npz")))
    assert np.array_equal(V.labelcolormap(22), fix["cmap"])
    assert np.array_equal(V.tensor2im(torch.from_numpy(fix["img"])), fix["img_u8"])
    assert np.array_equal(V.tensor2im(torch.from_numpy(fix["one"])), fix["one_u8"])


def test_draw_pose_from_cords_rasterisation():
    """util/util.py:164-191 restated without cv2 (parity unpinned, see mmhand_amd/visuals.py): the
    drawing's structure is checked - palm polygon in label 1's colour, each finger bone an ellipse of
    its own label over it, half axes (length/2, 8) along the bone - plus the two OpenCV primitives on
    cases with known answers."""
    from mmhand_amd import visuals as V
    sq = np.zeros((12, 12), np.uint8)
    V.fill_convex_poly(sq, [(2, 3), (8, 3), (8, 9), (2, 9)], 7)           # axis-aligned square, inclusive
    assert sq.sum() == 7 * 7 * 7 and sq[3:10, 2:9].min() == 7
    tri = np.zeros((10, 10), np.uint8)
    V.fill_convex_poly(tri, [(0, 0), (8, 0), (0, 8)], 1)
    assert tri[0, :9].all() and tri[:9, 0].all() and tri[8, 8] == 0 and tri[4, 4] == 1 and tri[5, 5] == 0
    e = V.ellipse2poly((50, 40), (20, 8), 0)
    assert tuple(e[0]) == (70, 40) and e[:, 0].max() == 70 and e[:, 0].min() == 30
    assert e[:, 1].max() == 48 and e[:, 1].min() == 32 and len(e) > 100
    e90 = V.ellipse2poly((50, 40), (20, 8), 90)
    assert e90[:, 1].max() == 60 and e90[:, 0].max() == 58
    # a hand: wrist at the bottom, five fingers fanning up
    J = np.zeros((21, 2), np.int64)
    J[0] = (110, 64)
    for f, x in enumerate((24, 44, 64, 84, 104)):
        for k in range(4):
            J[1 + 4 * f + k] = (90 - 20 * k, x)
    img = V.draw_pose_from_cords(J, (128, 128))
    cmap = V.labelcolormap(22)
    assert img.shape == (128, 128, 3) and img.dtype == np.uint8
    assert tuple(img[100, 64]) == tuple(cmap[1])                         # inside the palm polygon
    for f, x in enumerate((24, 44, 64, 84, 104)):
        for k in range(3):                                               # bone k of finger f: label 2 + 3f + k
            assert tuple(img[80 - 20 * k, x]) == tuple(cmap[2 + 3 * f + k]), (f, k)
    assert tuple(img[5, 5]) == (0, 0, 0)
    labels = {tuple(c) for c in img.reshape(-1, 3)}
    assert labels == {tuple(cmap[i]) for i in range(17)}
    # vertical bone (length 20): ellipse half axes 10 along y... the reference passes (length/2, 8) with
    # the bone's angle, so the long axis follows the bone: 8 px to each side across it
    col = np.all(img[60] == cmap[3], axis=1)                             # row through the middle of finger 0, bone 1
    assert col[24 - 8] and col[24 + 8] and not col[24 - 10]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE around it (how the driver invokes it) starts N child ranks
    itself and relays rank 0's JSON line (scripts/mm-train-ratio.sh:19-21 is the reference's launcher).  The
    hidden --selftest-ranks makes each rank rendezvous over gloo instead of touching a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                            "MASTER_PORT")}
    env["GLOO_SOCKET_IFNAME"] = "lo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-ranks"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # stdout carries exactly rank 0's line
    assert json.loads(lines[0]) == {"ranks": 2, "world": 2, "local_rank": 0}


def test_bench_refuses_a_mismatched_launch():
    """--gpus N inside a launch of another size is an error, not a silent single-rank run"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "1-rank launch" in out.stderr


def test_cpu_baseline_thread_sweep_reports_the_fastest(monkeypatch):
    """bench.cpu_baseline_with_sweep (VERDICT r4 #5b): 16 / 32 / 64 threads, the fastest is `value`, all three are kept."""
    import bench
    seen = []

    def fake(H, W, norm, budget_s=60.0, hard_timeout_s=300, threads=bench.CPU_THREADS, max_steps=10):
        seen.append((threads, max_steps, hard_timeout_s))
        v = {16: 0.6, 32: 0.9, 64: None}[threads]
        return {"value": v, "unit": "images/s", "cores": threads, "kind": "port", "sample": f"{threads} threads"}
    monkeypatch.setattr(bench, "cpu_baseline", fake)
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 256)
    out = bench.cpu_baseline_with_sweep(256, 256, "instance")
    assert [t for t, _, _ in seen] == [16, 32, 64] and seen[0][1] == 10 and seen[1][1] == 2 and seen[2][2] <= 90
    assert out["value"] == 0.9 and out["cores"] == 32 and out["kind"] == "port"
    assert set(out["thread_sweep"]) == {"16", "32", "64"} and out["thread_sweep"]["64"]["value"] is None
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 8)
    out = bench.cpu_baseline_with_sweep(256, 256, "instance")
    assert out["value"] == 0.6 and out["thread_sweep"]["32"]["value"] is None


def test_pack_twin_and_norm_site_registries_hand_over_once_and_notice_stale_tensors():
    """The two hand-over registries of the 16-bit step (host logic, no GPU): ops.pack_twin_put / _get - the padded 16-bit copy a
    pack leaves for the stem that reads the fp32 tensor - is handed out once (or every time when kept), never for a tensor that
    changed since (its version counter) or for another tensor at the same address and shape class; it is bounded.
    ops.nbr_site_put / _take - the norm-backward site a norm leaves under its 16-bit output - goes to the conv that consumes
    exactly that tensor object, once."""
    import torch
    from mmhand_amd import ops
    ops._pack_twins.clear(); ops._nbr_sites.clear()
    x = torch.zeros(2, 4, 4, 12)
    t = torch.zeros(2, 4, 4, 16, dtype=torch.bfloat16)
    ops.pack_twin_put(x, t)
    assert ops.pack_twin_get(x, True) is t and ops.pack_twin_get(x, True) is None          # handed out once
    ops.pack_twin_put(x, t, keep=True)
    assert ops.pack_twin_get(x, True) is t and ops.pack_twin_get(x, True) is t              # kept
    assert ops.pack_twin_get(x, 2) is None                                                  # fp16 asked, bf16 parked: dropped
    assert not ops._pack_twins
    ops.pack_twin_put(x, t, keep=True)
    x.add_(1.0)                                                                             # the fp32 tensor changed
    assert ops.pack_twin_get(x, True) is None and not ops._pack_twins
    ops.pack_twin_put(x, t, keep=True)
    ops.pack_twin_drop(x)
    assert ops.pack_twin_get(x, True) is None
    ops.pack_twin_put(x, t)
    assert ops.pack_twin_get(x[:1], True) is None                                           # same address, another shape
    held = [torch.zeros(1, 2, 2, 4) for _ in range(40)]
    for h in held:
        ops.pack_twin_put(h, torch.zeros(1, 2, 2, 8, dtype=torch.bfloat16))
    assert len(ops._pack_twins) <= 16
    a16 = torch.zeros(2, 4, 4, 256, dtype=torch.bfloat16)
    site = ops.NormBwdSite(a16, None, torch.zeros(2, 256), torch.ones(2, 256), 2, 0.0)
    ops.nbr_site_put(a16, site)
    assert ops.nbr_site_take(a16.view_as(a16)) is None                                      # a view is another object: not it
    ops.nbr_site_put(a16, site)
    assert ops.nbr_site_take(a16) is site and ops.nbr_site_take(a16) is None
    for h in held[:20]:
        ops.nbr_site_put(h, site)
    assert len(ops._nbr_sites) <= 8
    ops.lp_grads_reset()
    assert not ops._nbr_sites
    ops._pack_twins.clear()


def test_stride2_dgrad_kernel_has_no_register_spills():
    """conv_s2d_kernel must not touch scratch: a spilled DMA address or weight fragment is reloaded inside the tile loop by
    scratch_load + s_waitcnt vmcnt(0), which drains the halo DMA of the next tile and the stores of the last one - the kernel
    ran 205 us instead of 130 until round 5 found it (tools/scratch_audit.sh lists every kernel with scratch).
    hipcc cross-compiles without a GPU: this is a build-time property."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "mmhand_amd", "csrc")
    out = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-Wno-unused-function",
                          "-Wno-inline-asm", "-Rpass-analysis=kernel-resource-usage", "-c", "conv_s2_lp16.hip", "-o", os.devnull],
                         cwd=src, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    name, seen = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    s2d = {k: v for k, v in seen.items() if "conv_s2d_kernel" in k}
    assert len(s2d) == 2 and all(v == 0 for v in s2d.values()), s2d
    # the default instantiations of the fprop kernel (64 -> 128 stride 2; the 64 -> 64 stride-1 form) are spill-free as well
    dflt = {k: v for k, v in seen.items() if "conv_s2f_kernel" in k and ("ELi1ELi2ELi2ELb0ELi2ELi128" in k or "ELi1ELi4ELi2ELb0ELi1ELi64" in k)}
    assert len(dflt) == 4 and all(v == 0 for v in dflt.values()), dflt


def test_winograd_forward_gemm_keeps_scratch_out_of_its_k_loop():
    """wino_gemm_kernel<128,2> - the dominant kernel of the fp32 headline (bench.py roofline) - carries 48 bytes of scratch
    per lane: tile-constant addresses reloaded once per TILE of the persistent loop.  That is harmless; the same reload inside
    the k-loop would put a scratch_load + s_waitcnt vmcnt(0) - a drain of every prefetched operand - into each of its 16
    k-steps (what cost conv_s2d_kernel a third of its time).  Build-time property, read from the ISA: no
    scratch access in any basic block of loop depth >= 2, and no more scratch than round 5 measured."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "mmhand_amd", "csrc")
    asm = "/tmp/_mmh_conv_igemm_gate.s"
    out = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=gfx950", "-Wno-unused-function",
                          "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", asm, "conv_igemm.hip"],
                         cwd=src, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = open(asm).read().splitlines()
    os.remove(asm)
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*wino_gemm_kernelILi128ELi2E\w*:", l)]
    assert len(start) == 1, start
    name = re.match(r"^(_Z\w+):", lines[start[0]]).group(1)
    end = next(i for i in range(start[0], len(lines)) if lines[i].strip() == f".amdhsa_kernel {name}")
    scratch_bytes = next(int(l.split()[-1]) for l in lines[end:end + 60] if ".amdhsa_private_segment_fixed_size" in l)
    depth, by_depth, mfma_by_depth = 0, {}, {}
    for l in lines[start[0] + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            m = re.search(r"Depth=(\d+)", l)
            depth = int(m.group(1)) if m else 0
        elif re.match(r"^\s*;\s+(Parent Loop|=>|Child Loop)", l):      # continuation comments of a block label
            m = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", l)
            if m:
                depth = int(m.group(1))
        elif "scratch_" in l:
            by_depth[depth] = by_depth.get(depth, 0) + 1
        elif "v_mfma" in l:
            mfma_by_depth[depth] = mfma_by_depth.get(depth, 0) + 1
    assert mfma_by_depth.get(2, 0) >= 32, mfma_by_depth              # the k-loop IS the depth-2 loop (the parse found it)
    assert not any(d >= 2 for d in by_depth), by_depth                # ... and it touches no scratch
    assert scratch_bytes <= 48, scratch_bytes


@pytest.fixture
def data_dir():
    """a scratch directory whose PATH does not contain "test" (pytest's tmp_path does; generic_dataset.py:116 keys on it)"""
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="mmh_ds_")
    assert "test" not in d
    yield d
    shutil.rmtree(d, ignore_errors=True)


def test_hand_folder_loader_host_logic(data_dir):
    """data.HandFolderLoader without a GPU: image lists, sort keys, the augmentation_ratio split, the shuffled sources, labels,
    the depth path rule, DistributedSampler's default order and the pinned-host raw batches, against the rules of
    data/generic_dataset.py:100-131,204-210, data/rhd_dataset.py:25-45, data/stb_dataset.py:24-45 and
    data/mmhand_dataset_data_loader.py:22-48 restated here on a directory the test writes."""
    import random
    from PIL import Image
    from tests._dataset_fixture import write_rhd, write_stb
    from mmhand_amd.data import HandFolderLoader
    from mmhand_amd.options import default_train_opt
    tmp_path = data_dir
    root = os.path.join(tmp_path, "rhd_train")
    names = write_rhd(root, n=10, size=16)
    ordered = sorted(names, key=lambda x: int(x[:-4]))
    opt = default_train_opt(batchSize=3, dataroot=root, dataset="rhd", augmentation_ratio=0.3, nThreads=2)
    random.seed(11)
    ld = HandFolderLoader(opt, device=torch.device("cpu"))
    sep = int((1 - 0.3) * 10)
    want_tgt = [os.path.join(root, "color", n) for n in ordered[sep:]]            # training: the LAST ratio share
    assert ld.image_target == want_tgt
    random.seed(11)
    src = want_tgt.copy(); random.shuffle(src)
    assert ld.image_source == src
    opt_g = default_train_opt(batchSize=1, dataroot=root, dataset="rhd", augmentation_ratio=0.3, isTrain=False)
    lg = HandFolderLoader(opt_g, device=torch.device("cpu"))
    assert lg.image_target == [os.path.join(root, "color", n) for n in ordered[:sep]]   # generation: the first share
    # one raw host batch: what cv2.imread would hand over (BGR), labels by file name, depth file by path replacement
    hb = next(iter(ld.host_batches()))
    assert hb["img1"].dtype == torch.uint8 and tuple(hb["img1"].shape) == (3, 16, 16, 3) and hb["uv1"].dtype == torch.float64
    p1, p2 = hb["H1_path"][0], hb["H2_path"][0]
    assert p1 == ld.image_source[0] and p2 == ld.image_target[0]
    rgb = np.asarray(Image.open(p1).convert("RGB"))
    assert np.array_equal(hb["img1"][0].numpy(), rgb[:, :, ::-1])
    dep = np.asarray(Image.open(p2.replace("color", "depth")).convert("RGB"))
    assert np.array_equal(hb["dep2"][0].numpy(), dep[:, :, ::-1])
    lab = ld.annotations["color"][os.path.basename(p1)]
    assert np.array_equal(hb["uv1"][0].numpy(), np.asarray(lab["uv_coord"]))
    assert np.allclose(hb["C1"][0, :, 2].numpy(), np.asarray(lab["depth"]) / 700.0 * 255)
    assert ld.n_batches() == 1 and len(ld) == 3 and len(list(ld.host_batches())) == 1
    # max_dataset_size caps BATCHES (the reference compares the batch index with it) and __len__
    opt.batchSize, opt.max_dataset_size = 1, 2
    assert len(list(ld.host_batches())) == 2 and len(ld) == 2
    # DistributedSampler defaults: permutation of seed 0, padded by wrap-around, rank r takes r, r + world, ...
    ld.world, ld.rank = 2, 1
    perm = torch.randperm(3, generator=torch.Generator().manual_seed(0)).tolist()
    assert ld.indices() == (perm + perm[:1])[1:4:2]
    # a 'test' directory: everything, generation only
    troot = os.path.join(tmp_path, "rhd_test")
    tn = write_rhd(troot, n=4, size=16)
    lt = HandFolderLoader(default_train_opt(batchSize=1, dataroot=troot, dataset="rhd", augmentation_ratio=0.5, isTrain=False),
                          device=torch.device("cpu"))
    assert len(lt.image_target) == 4
    with pytest.raises(AssertionError):
        HandFolderLoader(default_train_opt(batchSize=1, dataroot=troot, dataset="rhd", augmentation_ratio=0.5), device=torch.device("cpu"))
    # STB: BB cameras and depth files are not sources; order = (folder digit, folder letter, frame)
    sroot = os.path.join(tmp_path, "stb")
    write_stb(sroot, n=3, size=16)
    ls = HandFolderLoader(default_train_opt(batchSize=1, dataroot=sroot, dataset="stb", augmentation_ratio=1.0), device=torch.device("cpu"))
    assert [p.split("/")[-2:] for p in ls.image_target] == [[f, f"SK_color_{i}.png"] for f in ("B1Counting", "B2Random") for i in range(3)]
    with pytest.raises(FileNotFoundError):
        HandFolderLoader(default_train_opt(batchSize=1, dataroot=os.path.join(tmp_path, "nope"), dataset="rhd"), device=torch.device("cpu"))


def test_device_pool_decisions_equal_image_pool_on_the_host():
    """DevicePool.decide (the host half of --graph_step's image pool: indices for mmh_pool_exchange) against ImagePool.query
    (util/image_pool.py:14-34 restated) on the same `random` sequence, the exchange itself emulated with index arithmetic:
    the same images come back, the same images stay in the same slots - through the fill phase, repeated swaps with one slot
    inside a query and images that enter and leave within one query.  (The kernel is held to this in tests/test_graph_step_gpu.py.)"""
    import random
    from mmhand_amd.mmhand_model import DevicePool, ImagePool
    for pool_size, B in ((3, 4), (1, 5), (2, 6), (5, 2), (50, 3)):
        a, b = ImagePool(pool_size), DevicePool(pool_size, 1)
        slots = [None] * pool_size                      # the emulated device buffer: image ids
        next_id = 0
        for it in range(40):
            ids = list(range(next_id, next_id + B))
            next_id += B
            x = torch.tensor(ids, dtype=torch.float32).view(B, 1, 1, 1)
            random.seed(500 + it)
            want = a.query(x).view(-1).tolist()
            random.seed(500 + it)
            src, dst = b.decide(B)
            got = [slots[s] if s >= 0 else ids[-1 - s] for s in src]       # every read of the pool before any write
            assert all(s < 0 or slots[s] is not None for s in src)
            writers = [d for d in dst if d >= 0]
            assert len(writers) == len(set(writers))                       # one writer per slot
            for i, d in enumerate(dst):
                if d >= 0:
                    slots[d] = ids[i]
            assert got == [int(v) for v in want], (pool_size, B, it)
            assert [s for s in slots[: b.count]] == [int(t.view(-1)[0]) for t in a.images]
