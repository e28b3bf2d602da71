"""Headline benchmark: 256x256 hand images/s for one full G+D step (optimize_parameters).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (no WORLD_SIZE in the environment) and with --gpus N > 1, bench.py starts
its own N ranks as CHILD processes - one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before
this process has touched the GPU - and relays rank 0's JSON line (the reference's launcher:
scripts/mm-train-ratio.sh:19-21 + options/base_options.py:171-178).

Workload (BASELINE.json configs[1]): RHD-shaped synthetic batch, 256x256, per-GPU batch 32, fp32,
Generator (9 PATBlocks, ngf 64) + D_PB + D_PP + L1/perceptual/GAN losses + 3 Adams, dropout on,
--norm instance (north_star; --norm batch is the reference's script default).  One process per
GPU; with N>1 the global batch is sharded (weak scaling) and G/D gradients are all-reduced over
RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE_STEP = 2444.4       # algorithmic conv FLOPs of one G+D step
PEAK_F32_MFMA_TF = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA_TF = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak


def synthetic_batch_gpu(B, H, W, seed, dev):
    """SURVEY.md §8(d) synthetic inputs generated on the device (pose maps by the HIP kernel)."""
    from mmhand_amd import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for s in ("1", "2"):
        out["H" + s] = torch.rand((B, 3, H, W), generator=g, device=dev) * 2 - 1
        uv = (torch.rand((B * 21, 2), generator=g, device=dev, dtype=torch.float64) * (H - 40) + 20)
        out["P" + s] = ops.pose_heatmaps(uv.contiguous(), H, W).view(B, 21, H, W)
        d = torch.rand((B, 1, H, W), generator=g, device=dev) * 2 - 1
        out["D" + s] = d.expand(B, 3, H, W).contiguous()
    return out


def sketch_inputs(B, H, W, seed=49):
    """The SURVEY 8(d) synthetic batch built on the HOST, bit for bit the one tests/golden/fullsize_grad_sketch.npz was made
    on (torch's CPU generator for images and depth, float64 host arithmetic for the 21 pose maps - the arithmetic of
    data/generic_dataset.py:212-216,239-242 - so that not even the last bit of an input differs from the fixture's: a 2^-22
    input change moves this network's gradients by 2e-3).  tests/test_fullsize_gpu.py holds it equal to the
    oracle's generator.  Only the fp64 comparison uses it; every timed region generates its inputs on the device."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    ys = np.arange(H, dtype=np.float64)[:, None]
    xs = np.arange(W, dtype=np.float64)[None, :]
    out = {}
    for s_ in ("1", "2"):
        out["H" + s_] = torch.rand((B, 3, H, W), generator=g) * 2 - 1
        lo, hi = min(20, H // 4), max(H - 20, 3 * H // 4)
        uv = rs.uniform(lo, hi, size=(B, 21, 2))
        maps = np.empty((B, 21, H, W), np.float32)
        for b in range(B):
            for j in range(21):
                m = np.exp(-((xs - uv[b, j, 0]) ** 2 + (ys - uv[b, j, 1]) ** 2) / 2.0 / 6.0 / 6.0)
                m[m > 1] = 1
                m[m < 0.0099] = 0
                maps[b, j] = m
        out["P" + s_] = torch.from_numpy(maps)
        d = torch.rand((B, 1, H, W), generator=g) * 2 - 1
        out["D" + s_] = d.expand(B, 3, H, W).contiguous()
    return out


def fp64_sketch_distance(fix, grads, out=None, n=1024):
    """per-tensor relative L1 of `grads` ({reference key: logical-layout gradient}) from the float64 gradients of the
    REFERENCE's Generator, through the positions tests/golden/fullsize_grad_sketch.npz keeps (tests/golden/make_golden.py
    make_fullsize).  Returns ({key: distance}, distance of the output or None)."""
    from tests.golden import recipe as RC
    errs = {}
    for k in fix.files:
        if not k.startswith("cond_sampled/"):
            continue
        key = k[len("cond_sampled/"):]
        g = grads[key].reshape(-1)
        idx = RC.sketch_indices(key, g.numel(), n).to(g.device)
        want = torch.from_numpy(fix["s/" + key]).to(g.device).double()
        errs[key] = float((g[idx].double() - want).abs().sum() / want.abs().sum().clamp_min(1e-300))
    oerr = None
    if out is not None:
        o = out.reshape(-1)
        idx = RC.sketch_indices("out", o.numel(), n).to(o.device)
        want = torch.from_numpy(fix["out_sample"]).to(o.device).double()
        oerr = float((o[idx].double() - want).abs().sum() / want.abs().sum().clamp_min(1e-300))
    return errs, oerr


class KernelTimer:
    """HIP-event brackets around every launch of ONE kernel shape during the timed region: either
    the direct implicit-GEMM fprop matching `match` (conv desc fields) or, with Winograd on, the
    batched Winograd-domain GEMM with (tiles, K, N) == `gemm`."""

    def __init__(self, match, gemm=None):
        self.match, self.gemm, self.pairs, self.op_pairs, self.enabled = match, gemm, [], [], False

    def want(self, d):
        return self.enabled and self.gemm is None and all(getattr(d, k) == v for k, v in self.match.items())

    def want_gemm(self, P, tiles, K, N):
        return self.enabled and self.gemm == (P, tiles, K, N)

    def bracket(self):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        self.pairs.append((e0, e1))
        return e0, e1

    def bracket_op(self):
        """(gemm start, gemm end, op start, op end): the Winograd GEMM launch alone and the whole
        convolution (input transform + GEMM + output transform)."""
        e0, e1 = self.bracket()
        o0 = torch.cuda.Event(enable_timing=True)
        o1 = torch.cuda.Event(enable_timing=True)
        self.op_pairs.append((o0, o1))
        return e0, e1, o0, o1

    def op_mean_ms(self):
        ts = [a.elapsed_time(b) for a, b in self.op_pairs]
        return (sum(ts) / len(ts)) if ts else None

    def mean_ms(self):
        ts = [a.elapsed_time(b) for a, b in self.pairs]
        if os.environ.get("MMH_BENCH_TIMER_DEBUG") == "1":
            st = sorted(ts)
            print(f"[timer] n={len(ts)} min={st[0]:.3f} med={st[len(st) // 2]:.3f} max={st[-1]:.3f} "
                  f"first8={[round(t, 3) for t in ts[:8]]}", file=sys.stderr, flush=True)
        return sum(ts) / len(ts), len(ts)


CPU_THREADS = 16   # measured on the GPU box (256 logical CPUs): oneDNN conv fwd+bwd is fastest at
                   # 16 threads (14 ms) and 14x slower at 128 (tools/cpu_probe.py)


def cpu_baseline_child(H, W, norm, budget_s, threads=CPU_THREADS, max_steps=10):
    """Runs in a child process that never touches the GPU.  Prints one JSON object.
    SURVEY.md §8(d) protocol: B=2, 256x256, fp32, 2 warm-up + 10 timed iterations (BASELINE.json
    configs[0]); the timed count shrinks only if the budget would be exceeded (stated in `sample`)."""
    import random
    from mmhand_amd.networks import Discriminator, Generator, VGGHead
    from oracle import mmhand_ref as O
    nthreads = max(1, min(threads, len(os.sched_getaffinity(0))))
    torch.set_num_threads(nthreads)
    g = Generator([3, 42, 6], 3, 64, norm, True, 9).init_weights("normal", 49)
    dpb = Discriminator(24, 64, norm, True, 3).init_weights("normal", 50)
    dpp = Discriminator(6, 64, norm, True, 3).init_weights("normal", 51)
    vgg = VGGHead().init_random()
    orc = O.StepOracle(g.state_dict(), dpb.state_dict(), dpp.state_dict(), vgg.state_dict(), norm,
                       True, True, 9, 3, rng=random.Random(0))
    B = 2
    batch = O.synthetic_batch(B, H, W, seed=49)
    t0 = time.time()
    nwarm = 0
    while nwarm < (1 if max_steps <= 2 else 2) and (nwarm < 1 or time.time() - t0 < budget_s / 2):
        orc.step(batch); nwarm += 1
    warm = time.time() - t0
    n, t1 = 0, time.time()
    while n < max_steps and (n < 1 or time.time() - t1 < budget_s):
        orc.step(batch); n += 1
    dt = (time.time() - t1) / n
    print(json.dumps({"value": round(B / dt, 4), "unit": "images/s", "cores": nthreads,
                      "kind": "port", "sample": f"{n} timed G+D steps at B={B}, {H}x{W}, fp32, --norm "
                      f"{norm}, dropout on, after {nwarm} warm-up step(s) ({warm:.1f}s), {nthreads} threads of "
                      f"{os.cpu_count()} logical CPUs; oracle/mmhand_ref.py StepOracle"}), file=_OUT, flush=True)


def cpu_baseline_start(H, W, norm, budget_s=60.0, hard_timeout_s=300, threads=CPU_THREADS, max_steps=10):
    """start the CPU-oracle child (never touches the GPU) -> handle for cpu_baseline_collect"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--size", str(H),
           "--norm", norm, "--cpu-budget", str(budget_s), "--cpu-threads", str(threads), "--cpu-max-steps", str(max_steps)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MMH_FORCE_DP", "OMP_NUM_THREADS"):
        env.pop(k, None)
    return (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env), time.time() + hard_timeout_s,
            threads, hard_timeout_s)


def cpu_baseline_collect(handle):
    import subprocess
    proc, deadline, threads, hard_timeout_s = handle
    try:
        out, err = proc.communicate(timeout=max(1.0, deadline - time.time()))
        for line in reversed(out.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port", "sample": "child failed: " + err[-300:]}
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"child exceeded the {hard_timeout_s}s hard timeout"}


def cpu_baseline(H, W, norm, budget_s=60.0, hard_timeout_s=300, threads=CPU_THREADS, max_steps=10):
    """The oracle (pure PyTorch restatement, pinned against the reference modules) timed on this
    host's cores on a bounded sample of the same workload, in a child process with a hard timeout."""
    return cpu_baseline_collect(cpu_baseline_start(H, W, norm, budget_s, hard_timeout_s, threads, max_steps))


CPU_SWEEP = (32, 64)    # beside CPU_THREADS: "is 16 the fastest this host does" measured on the line itself (VERDICT r4 #5b)


def cpu_baseline_with_sweep(H, W, norm):
    """cpu_baseline at CPU_THREADS (the SURVEY 8(d) protocol: 2 warm-up + 10 timed steps) plus the same oracle at 32 and 64
    threads - 1 warm-up + 2 timed steps each, one child per thread count, 25 s budget and a hard 75 s timeout each (oneDNN
    at 128+ threads did not finish a step in 240 s on this host: profiles/r04_cpu_all_cores.json) - run one after the
    other so that they do not share cores.  `value` / `cores` = the FASTEST of the three; `thread_sweep` keeps all."""
    base = cpu_baseline(H, W, norm)
    sweep = {str(base.get("cores", CPU_THREADS)): {"value": base.get("value"), "sample": base.get("sample")}}
    best = base
    for t in CPU_SWEEP:
        if (os.cpu_count() or 0) < t:
            sweep[str(t)] = {"value": None, "sample": f"host has {os.cpu_count()} logical CPUs"}
            continue
        r = cpu_baseline(H, W, norm, budget_s=25.0, hard_timeout_s=75, threads=t, max_steps=2)
        sweep[str(t)] = {"value": r.get("value"), "sample": r.get("sample")}
        if r.get("value") and (not best.get("value") or r["value"] > best["value"]):
            best = r
    out = dict(best)
    out["thread_sweep"] = sweep
    return out


def infer_main(a):
    """configs[3]: STB-shaped 256x256, batch 64, inference-only Generator throughput."""
    from mmhand_amd.inference import InferenceGenerator
    from mmhand_amd.networks import Generator
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B = a.batch if a.batch != 32 else 64
    net = Generator([3, 42, 6], 3, 64, "batch", True, 9).init_weights("normal", 49).to(dev).eval()
    gen = InferenceGenerator(net, use_graph=True, bf16=a.dtype == "bf16")
    b = synthetic_batch_gpu(B, a.size, a.size, 49, dev)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    for _ in range(max(a.warmup, 1)):
        gen(g_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        gen(g_in)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ips = B * a.steps / dt
    print(json.dumps({
        "metric": "256x256 hand images/sec (Generator inference)", "value": round(ips, 2),
        "unit": "images/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"STB-shaped {a.size}x{a.size}, batch {B}, "
                   f"{'bf16 MFMA / fp32 storage' if a.dtype == 'bf16' else 'fp32'}, Generator forward only "
                   "(eval BatchNorm folded into convs, hipGraph replay)"},
        # direct-convolution-equivalent rate: BASELINE.json's 611.68 GFLOP/image forward count x images/s,
        # over the dense MFMA peak.  Winograd executes 2.25-4x fewer multiplications on the 3x3
        # stack, so this can exceed 1.0; it is a throughput yardstick, not a utilisation.
        "direct_equiv_tflops": round(611.68 * (a.size * a.size / 65536.0) * ips / 1e3, 1),
        "direct_equiv_frac_of_peak": round(611.68 * (a.size * a.size / 65536.0) * ips / 1e3 /
                                           (PEAK_BF16_MFMA_TF if a.dtype == "bf16" else PEAK_F32_MFMA_TF), 4)}),
        file=_OUT, flush=True)


_OUT = sys.stdout       # the JSON line goes here; everything else the process prints goes to stderr


def _json_only_stdout():
    """stdout carries exactly ONE JSON line: file descriptor 1 is pointed at stderr for everything else the
    process prints - Python (model warnings) and native libraries (RCCL's version banner) alike."""
    global _OUT
    sys.stdout.flush()
    _OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr


class StackMeter:
    """HIP-event brackets around every C-ABI call of the 3x3 / stride-1 stack with 256 / 512 channels (SURVEY.md
    §2.3 K4: the PATBlocks' and the Discriminators' residual convs) in 16-bit mode - fprop (mmh_conv3x3_lp16,
    mmh_conv3x3_lp16_fprop_stats), dgrad (mmh_conv3x3_lp16 / mmh_conv3x3_lp16_dgrad_add / _dgrad_nbr mode 2, or mode 1 + the border terms of
    mmh_conv2d_dgrad_border where the fold does not apply) and the wgrad (mmh_wgrad3x3_lp16) - during a few steps.  stack fraction = sum(algorithmic FLOPs 2.B.H.W.Cin.Cout.9 of every
    pass) / sum(bracketed time) / peak: the north_star's ">= 40 % MFMA on the 3x3 generator conv stack" as measured."""
    NAMES = ("mmh_conv3x3_lp16", "mmh_conv3x3_lp16_fprop_stats", "mmh_conv3x3_lp16_dgrad_add", "mmh_conv3x3_lp16_dgrad_nbr",
             "mmh_wgrad3x3_lp16", "mmh_conv2d_dgrad_border")

    def __init__(self):
        self.rec = []

    def __enter__(self):
        from mmhand_amd import lib as L
        self.L, self.real = L, L.call

        def timed(name, *args):
            if name not in self.NAMES:
                return self.real(name, *args)
            d = next((getattr(x, "_obj", None) for x in args if isinstance(getattr(x, "_obj", None), L.ConvDesc)), None)
            if d is None or d.kh != 3 or d.stride != 1 or min(d.Cin, d.Cout) < 256:
                return self.real(name, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.real(name, *args)
            e1.record()
            if name.endswith("border"):
                kind = "border"
            elif "wgrad" in name:
                kind = "wgrad"
            elif name.endswith("fprop_stats"):
                kind = "fprop"
            elif name.endswith("dgrad_add") or name.endswith("dgrad_nbr"):
                # (dgrad_nbr: the bracket also holds the first norm's backward sums, taken in the epilogue, and their
                # finishing launch - work of a row kernel counted against the stack's time)
                kind = "dgrad"
            else:       # mmh_conv3x3_lp16 mode: 0 fprop, 1 zero-pad dgrad (+ border call), 2 the complete reflect dgrad
                kind = "fprop" if int(args[1]) == 0 else "dgrad"
            flop = 0.0 if kind == "border" else 2.0 * d.B * d.H * d.W * d.Cin * d.Cout * 9
            self.rec.append((kind, (d.Cin, d.Cout), flop, e0, e1))
            return r
        L.call = timed
        return self

    def __exit__(self, *exc):
        self.L.call = self.real

    def summary(self, peak_tf):
        torch.cuda.synchronize()
        by = {}
        for kind, shape, flop, e0, e1 in self.rec:
            c = by.setdefault((kind, shape), [0, 0.0, 0.0])
            c[0] += 1; c[1] += flop; c[2] += e0.elapsed_time(e1)
        tot_f = sum(c[1] for c in by.values()); tot_ms = sum(c[2] for c in by.values())
        per = {f"{k}_{ci}x{co}": {"launches": c[0], "ms": round(c[2], 3),
                                  "tflops": round(c[1] / (c[2] * 1e-3) / 1e12, 1) if c[1] else None}
               for (k, (ci, co)), c in sorted(by.items())}
        return {"stack_frac": round(tot_f / (tot_ms * 1e-3) / 1e12 / peak_tf, 4) if tot_ms else None,
                "stack_tflop": round(tot_f / 1e12, 3), "stack_ms": round(tot_ms, 2), "per_pass": per}, by


def _rank_ms(dt_s, steps, dev):
    """(max, min) over the ranks of this rank's seconds for `steps` steps, as ms per step"""
    ms = dt_s / steps * 1e3
    if not dist.is_initialized():
        return ms, ms
    t = torch.tensor([ms, -ms], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0]), float(-t[1])


def _sync(dp):
    if dp and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


class CallCounter:
    """C-ABI calls of the enclosed region (every launch goes through lib.call)"""

    def __enter__(self):
        from mmhand_amd import lib as L
        self.L, self.real, self.n = L, L.call, 0

        def counted(name, *args):
            self.n += 1
            return self.real(name, *args)
        L.call = counted
        return self

    def __exit__(self, *exc):
        self.L.call = self.real


def host_enqueue(model, dp, reps=2):
    """What the HOST spends on one optimize_parameters() (tools/host_overhead.py, VERDICT r4 #5d): the wall time of the
    call itself with the GPU idle when it starts - the call returns when everything is enqueued (overflow flags are read
    one iteration late) - and the C-ABI calls it makes.  enqueue / step near 1 = the configuration is host-bound."""
    best, calls = None, 0
    for _ in range(reps):
        _sync(dp)
        with CallCounter() as c:
            t0 = time.perf_counter()
            model.optimize_parameters()
            ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
        calls = c.n
    _sync(dp)
    return round(best, 2), calls


def side_train_run(dev, B, size, steps, warmup=2, stack=False, stack_steps=2, dp=False, seed=49, lib_options=None, **opt_kw):
    """A fresh MMHandModel with option overrides: images/s over `steps` optimize_parameters() calls (inputs resident),
    measured like the headline region; optionally the 16-bit stack fraction from a separate bracketed pass.
    dp: the model is built on the process group (distributed=True): barriers around the timed region, the slowest rank's
    time counts, images/s is the whole job's, and `comm_exposed_ms` = the step minus the same step with every
    collective stubbed out (dp.set_no_comm: gradients stay rank-local - measured LAST, the replicas drift apart)."""
    from mmhand_amd import lib as L
    world = dist.get_world_size() if (dp and dist.is_initialized()) else 1
    kw = dict(batchSize=B, norm="instance", name="bench_side", local_rank=dev.index, checkpoints_dir="/tmp/mmh_bench",
              distributed=bool(dp))
    kw.update(opt_kw)
    # lib_options: {key: (value for this region, value to restore)} of mmh_set_option (kernel-selection switches)
    for k, (v, _) in (lib_options or {}).items():
        L.call("mmh_set_option", k.encode(), int(v))
    try:
        return _side_train_run(dev, B, size, steps, warmup, stack, stack_steps, dp, seed, world, kw)
    finally:
        for k, (_, v0) in (lib_options or {}).items():
            L.call("mmh_set_option", k.encode(), int(v0))


def _side_train_run(dev, B, size, steps, warmup, stack, stack_steps, dp, seed, world, kw):
    import gc
    from mmhand_amd import ops
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    model = MMHandModel(default_train_opt(**kw))
    model.set_input(synthetic_batch_gpu(B, size, size, seed, dev))
    for _ in range(warmup):
        model.optimize_parameters()
    ops.collective_counter.clear()
    _sync(dp)
    t0 = time.perf_counter()
    for _ in range(steps):
        model.optimize_parameters()
    _sync(dp)
    ms, ms_min = _rank_ms(time.perf_counter() - t0, steps, dev)
    out = {"images_per_s": round(world * B / ms * 1e3, 3), "ms_per_step": round(ms, 2), "steps": steps, "warmup": warmup}
    if dp:
        cc = dict(ops.collective_counter)
        out.update({"n_gpus": world, "global_batch": world * B, "ms_per_step_fastest_rank": round(ms_min, 2),
                    "bucket_allreduce": "mmh_allreduce_bucket (C-ABI, torch.distributed's RCCL communicator)"
                    if getattr(model, "dp_native", None) else f"dist.all_reduce ({getattr(model, 'dp_native_why', '-')})",
                    "inplace_param_grads": bool(getattr(model, "dp_accum", False)),
                    "syncbn_collectives_per_step": {k: round(v / steps, 1) for k, v in cc.items()} if cc else None})
    if stack:
        # stack_frac describes the SHIPPED configuration: with ops.USE_NBR on - as the timed steps above ran - the
        # dgrad launches of the stack also take the first norm's backward sums in their epilogue (a row kernel's work inside
        # a conv launch).  The stack metered alone, with that epilogue off, goes beside it as
        # stack_frac_without_norm_sums_in_dgrad (the figure the rounds before the epilogue existed reported).
        from mmhand_amd import ops as _ops
        with StackMeter() as m:
            for _ in range(stack_steps):
                model.optimize_parameters()
        summ, by = m.summary(PEAK_BF16_MFMA_TF)
        out.update(summ)
        if _ops.USE_NBR:
            nbr_was = _ops.USE_NBR
            try:
                _ops.USE_NBR = False
                with StackMeter() as m0:
                    for _ in range(stack_steps):
                        model.optimize_parameters()
                out["stack_frac_without_norm_sums_in_dgrad"] = m0.summary(PEAK_BF16_MFMA_TF)[0]["stack_frac"]
            finally:
                _ops.USE_NBR = nbr_was
        c = by.get(("fprop", (512, 512)))
        if c and c[2] > 0:
            ach = c[1] / (c[2] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_MFMA_TF, 4),
                               "kernel": f"conv_lp16h2_kernel fprop 3x3 512->512 @{size // 4}x{size // 4} (B={B}): "
                                         f"{c[1] / c[0] / 1e9:.1f} GFLOP/launch, {c[2] / c[0]:.3f} ms avg over {c[0]} launches"}
    out["losses_finite"] = all(torch.isfinite(v).item() for v in model.get_current_errors().values())
    if getattr(model, "graph_step", False):
        # --graph_step: the timed steps above were replays of the captured iteration (or, had the capture failed, its eager form)
        out["graph_step"] = model._graph is not None
        out["graph_replays"] = model.graph_replays
        if model.graph_error:
            out["graph_error"] = model.graph_error
    enq, calls = host_enqueue(model, dp)
    out["host_enqueue_ms"] = enq
    out["c_abi_calls_per_step"] = calls
    out["host_enqueue_over_step"] = round(enq / ms, 3)
    if dp:
        out["comm_exposed_ms"] = comm_exposed_ms(model, dev, ms, max(2, min(3, steps)))
    del model
    gc.collect()
    torch.cuda.empty_cache()
    return out


def comm_exposed_ms(model, dev, ms_with_comm, steps):
    """ms per step that the collectives cost the data-parallel step: the step as measured minus the same step with every
    collective stubbed out (gradient buckets and SyncBN exchanges: dp.set_no_comm).  ~0 = the all-reduces are hidden
    beneath the backward passes.  Run it LAST on a model: without collectives the replicas drift apart."""
    from mmhand_amd import dp as DP

    def timed():
        model.optimize_parameters()
        _sync(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            model.optimize_parameters()
        _sync(True)
        return _rank_ms(time.perf_counter() - t0, steps, dev)[0]
    DP.set_no_comm(True)
    try:
        ms0 = timed()
    finally:
        DP.set_no_comm(False)
    # ... and WITH the collectives once more, back to back with the run without: on some boxes the first timed region of a
    # process runs 2-5 % slow (268 against 262 ms for the same fp32 step a minute later), which the difference against the
    # region's own earlier figure would book as 13 ms of "exposed communication" at world size 1
    ms1 = timed()
    return {"value": round(ms1 - ms0, 2), "ms_per_step_without_collectives": round(ms0, 2),
            "ms_per_step_with_collectives": round(ms1, 2), "ms_per_step_region": round(ms_with_comm, 2), "steps": steps}


def gradient_parity_run(dev, size, norm):
    """the trade the headline path makes, on the driver line.  Parameter gradients of the full-size Generator
    (ngf 64, 9 PATBlocks, B=2) through the Winograd F(6x6,3x3) path against the same gradients through the direct
    implicit-GEMM kernels: relative L1 per tensor.  (Against the fp64 oracle the direct kernels are at 1e-6 ... 1e-3 per
    tensor and Winograd at <= 5e-3, median <= 2e-3: tests/test_winograd_step_gpu.py; no oracle is imported here.)"""
    import gc
    import statistics
    from mmhand_amd import ops
    from mmhand_amd.networks import Generator
    import numpy as np
    from mmhand_amd.networks import logical_grads
    B = 2
    # the fixture's own inputs (host-generated, bit for bit): the same four runs then also stand against fp64 truth
    b = {k: v.to(dev) for k, v in sketch_inputs(B, size, size, 49).items()}
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    probe = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(3)).to(dev)
    sketch_path = os.path.join(ROOT, "tests", "golden", "fullsize_grad_sketch.npz")
    fix = np.load(sketch_path) if (size == 256 and norm == "instance" and os.path.exists(sketch_path)) else None
    vs64 = {}
    res, old_mode = {}, ("off" if not ops.USE_WINOGRAD else "all" if ops.WINOGRAD_FPROP else "bwd")
    # key -> (ops.set_winograd_mode, factor on the network input, summation levels of the direct fp32 fprop)
    runs = {"direct": ("off", 1.0, 1), "direct_two_level": ("off", 1.0, 2), "winograd": ("all", 1.0, 1),
            "winograd_bwd_only": ("bwd", 1.0, 2), "direct_input_2ulp": ("off", 1.0 + 2.0 ** -22, 1)}
    try:
        for key, (mode, scale, levels) in runs.items():
            ops.set_winograd_mode(mode, direct_levels=levels)
            net = Generator([3, 42, 6], 3, 64, norm, False, 9).init_weights("normal", 49).to(dev).train()
            net.flatten_parameters()
            out = net([t * scale for t in g_in])
            (out * probe).sum().backward()
            res[key] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
            if fix is not None and scale == 1.0:
                vs64[key] = fp64_sketch_distance(fix, logical_grads(net), out.detach().contiguous())   # forward() returns logical NCHW
            del net, out
    finally:
        ops.set_winograd_mode(old_mode)
    rel = lambda a, c: float((a.double() - c.double()).abs().sum() / c.double().abs().sum().clamp_min(1e-30))   # noqa: E731

    def against_direct(key, base="direct"):
        errs = sorted(rel(res[key][1][n], g) for n, g in res[base][1].items() if float(g.abs().sum()) > 0)
        return {"output_rel_l1": float(f"{rel(res[key][0], res[base][0]):.3e}"),
                "grad_rel_l1_median": float(f"{statistics.median(errs):.3e}"),
                "grad_rel_l1_p90": float(f"{errs[(len(errs) * 9) // 10]:.3e}"),
                "grad_rel_l1_max": float(f"{errs[-1]:.3e}"), "tensors": len(errs)}

    def against_fp64(key):
        errs, oerr = vs64[key]
        v = sorted(errs.values())
        cond = {k: float(fix["cond_sampled/" + k]) for k in errs}
        return {"output_rel_l1": float(f"{oerr:.3e}"), "grad_rel_l1_median": float(f"{statistics.median(v):.3e}"),
                "grad_rel_l1_p90": float(f"{v[(len(v) * 9) // 10]:.3e}"), "grad_rel_l1_max": float(f"{v[-1]:.3e}"),
                "tensors": len(v), "tensors_above_1e-3": sum(e > 1e-3 for e in v),
                "tensors_above_1e-3_and_1p5x_pytorch_fp32": sum(e > max(1e-3, 1.5 * cond[k]) for k, e in errs.items())}

    out = {"winograd_vs_direct": against_direct("winograd"),
           "winograd_dgrad_wgrad_only_vs_direct": against_direct("winograd_bwd_only", "direct_two_level"),
           "direct_input_times_1p2e-22_vs_direct": against_direct("direct_input_2ulp"),
           "note": f"full-size Generator (ngf 64, 9 PATBlocks, {size}x{size}, B={B}, --norm {norm}, dropout off), gradients of "
                   "sum(out * probe) per parameter tensor against direct_path's kernels (their own distance from float64 truth: "
                   "vs_fp64).  Second key: --fp32_exact_grads (ops.set_winograd_mode('bwd'), timed as hybrid_path): direct fprop "
                   "with two-level summation, Winograd dgrad and wgrad, against the all-direct run with the SAME forward "
                   "(direct_two_level), i.e. what the backward kernels themselves add.  Third key: the direct kernels against "
                   "themselves with the network input scaled by (1 + 2^-22) - the gradients' conditioning, which the "
                   "Winograd fprop's 7e-6 output difference excites (ReLU masks within rounding of zero flip)"}
    if fix is not None:
        cond = sorted(float(fix[k]) for k in fix.files if k.startswith("cond_sampled/"))
        out["vs_fp64"] = {
            "direct": against_fp64("direct"), "direct_two_level": against_fp64("direct_two_level"),
            "hybrid_winograd_bwd_only": against_fp64("winograd_bwd_only"), "winograd": against_fp64("winograd"),
            "pytorch_fp32_cpu": {"output_rel_l1": float(f"{float(fix['cond/out']):.3e}"),
                                 "grad_rel_l1_median": float(f"{statistics.median(cond):.3e}"),
                                 "grad_rel_l1_max": float(f"{cond[-1]:.3e}"), "tensors": len(cond),
                                 "tensors_above_1e-3": sum(c > 1e-3 for c in cond)},
            "note": "per parameter tensor, relative L1 against the float64 gradients of the REFERENCE's own Generator "
                    "(models/Generator.py in double precision, run in the build container: tests/golden/make_golden.py "
                    "make_fullsize) at 1024 seeded positions per tensor, identical inputs and weights bit for bit; "
                    "pytorch_fp32_cpu = the same reference module in float32 on the CPU through the same positions: no fp32 "
                    "implementation, PyTorch's included, holds 1e-3 on every tensor of this network at 256x256.  direct = the "
                    "direct kernels with one k-ordered MFMA chain per output (direct_path); direct_two_level / hybrid = the "
                    "same forward with two-level summation (ops.set_winograd_mode's default for 'off' / 'bwd': what "
                    "--fp32_exact_grads and hybrid_path run)"}
    del res
    gc.collect()
    torch.cuda.empty_cache()
    return out


def line_summary(line):
    """The figures of the side regions in one compact object, emitted as the LAST key of the JSON line (the
    driver keeps the tail of a long line; round 5's bf16_path.images_per_s and stack_frac fell off it)."""
    def g(key, field="images_per_s"):
        v = line.get(key)
        return v.get(field) if isinstance(v, dict) else None
    gp = line.get("gradient_parity") if isinstance(line.get("gradient_parity"), dict) else {}
    v64 = gp.get("vs_fp64", {}) if isinstance(gp.get("vs_fp64"), dict) else {}
    s = {"value": line.get("value"), "ms_per_step": line.get("ms_per_step"), "dtype": line.get("dtype"),
         "roofline_frac": (line.get("roofline") or {}).get("frac"),
         "bf16_path": g("bf16_path"), "bf16_ms_per_step": g("bf16_path", "ms_per_step"),
         "stack_frac": g("bf16_path", "stack_frac"),
         "stack_frac_without_norm_sums_in_dgrad": g("bf16_path", "stack_frac_without_norm_sums_in_dgrad"),
         "bf16_roofline_frac": (g("bf16_path", "roofline") or {}).get("frac"),
         "bf16_path_graph": g("bf16_path_graph"), "bf16_graph_host_enqueue_ms": g("bf16_path_graph", "host_enqueue_ms"),
         "bf16_graph_replayed": g("bf16_path_graph", "graph_step"),
         "size512_bf16_b4": g("size512_bf16_b4"), "size512_bf16_b4_graph": g("size512_bf16_b4_graph"),
         "size512_graph_host_enqueue_ms": g("size512_bf16_b4_graph", "host_enqueue_ms"),
         "graph_step": g("graph_step"), "graph_step_host_enqueue_ms": g("graph_step", "host_enqueue_ms"),
         "norm_batch": g("norm_batch"), "norm_batch_o1": g("norm_batch_o1"),
         "hybrid_path": g("hybrid_path"), "direct_path": g("direct_path"), "set_input_in_loop": g("set_input_in_loop"),
         "dp_rccl_world1": g("dp_rccl_world1"), "dp_bf16_path": g("dp_bf16_path"), "dp_norm_batch": g("dp_norm_batch"),
         "dp_norm_batch_o1": g("dp_norm_batch_o1"), "dp_size512_bf16_b4": g("dp_size512_bf16_b4"),
         "infer_b64_f32": g("infer_b64_f32"), "infer_b64_bf16": g("infer_b64_bf16"),
         "grad_median_winograd_vs_direct": (gp.get("winograd_vs_direct") or {}).get("grad_rel_l1_median"),
         "grad_vs_fp64_median": {k: (v64.get(k) or {}).get("grad_rel_l1_median")
                                 for k in ("direct", "direct_two_level", "hybrid_winograd_bwd_only", "winograd", "pytorch_fp32_cpu")} if v64 else None,
         "cpu_baseline": (line.get("cpu_baseline") or {}).get("value"), "rccl_ranks": line.get("rccl_ranks")}
    return {k: v for k, v in s.items() if v is not None}


def side_infer_run(dev, B, size, steps, bf16):
    """configs[3]: Generator forward only, eval BatchNorm folded into the convs, hipGraph replay"""
    import gc
    from mmhand_amd.inference import InferenceGenerator
    from mmhand_amd.networks import Generator
    net = Generator([3, 42, 6], 3, 64, "batch", True, 9).init_weights("normal", 49).to(dev).eval()
    gen = InferenceGenerator(net, use_graph=True, bf16=bf16)
    b = synthetic_batch_gpu(B, size, size, 49, dev)
    g_in = [b["H1"], torch.cat((b["P1"], b["P2"]), 1), torch.cat((b["D1"], b["D2"]), 1)]
    for _ in range(2):
        gen(g_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gen(g_in)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del gen, net
    gc.collect()
    torch.cuda.empty_cache()
    return {"images_per_s": round(B / ms * 1e3, 2), "ms_per_batch": round(ms, 2), "batch": B, "steps": steps}


def rccl_child_run(steps, warmup, batch, size, timeout_s=600):
    """The same step through the data-parallel code path on RCCL with ONE rank, in a child process with a hard timeout:
    init_process_group("nccl"), the parameter broadcast, the bucketed gradient all-reduces on the side stream and the
    deferred optimizer steps (mmhand_amd/dp.py) all run on the real communicator.  A 1-GPU box cannot show scaling; it
    can show that the path the N-GPU launch takes works on this hardware, and what it costs."""
    import subprocess
    env = dict(os.environ, MMH_FORCE_DP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
           "--batch", str(batch), "--size", str(size), "--no-cpu-baseline", "--dp-side-runs"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": f"child exceeded {timeout_s}s"}
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            j = json.loads(line)
            return {"images_per_s": j["value"], "ms_per_step": j["ms_per_step"], "steps": steps,
                    "rccl_ranks": j.get("rccl_ranks"), "backend": j.get("backend"),
                    **{k: j[k] for k in ("bucket_allreduce", "inplace_param_grads", "comm_exposed_ms", "dp_bf16_path",
                                         "dp_norm_batch", "dp_norm_batch_o1", "dp_bf16_path_nopersist", "dp_size512_bf16_b4",
                                         "dp_size512_bf16_b4_norm_batch") if k in j}}
    return {"error": f"rc {out.returncode}: " + out.stderr[-400:]}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv, timeout_s=2400.0, share_gpu=False):
    """python bench.py --gpus N without a launcher: start N ranks as child processes of this one (which
    never touches the GPU: torch.cuda.device_count() does not initialise it) and relay rank 0's JSON
    line.  A rank that fails ends the others (by their own PIDs) and its exit code becomes ours; ranks still
    running after timeout_s (a hung collective) are ended the same way and the exit code is 124."""
    import subprocess
    if "--selftest-ranks" not in argv and not share_gpu:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py --gpus {n}: this node exposes {have} GPU(s)", file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if share_gpu else str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // n))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=_OUT if r == 0 else sys.stderr, stderr=sys.stderr))
    rc = 0
    live = list(procs)
    deadline = time.time() + timeout_s
    while live:
        time.sleep(0.2)
        if time.time() > deadline:
            print(f"bench.py --gpus {n}: ranks still running after {timeout_s:.0f}s - ending them", file=sys.stderr, flush=True)
            for q in live:
                q.terminate()
            time.sleep(5)
            for q in live:
                if q.poll() is None:
                    q.kill()
            return 124
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:      # a dead rank would leave the others waiting in a collective
                    q.terminate()
    return rc


def init_rccl(dev, world=None, rank=None, port=None):
    """torch.distributed over RCCL (backend "nccl" on ROCm), one node: rendezvous on 127.0.0.1 (the container
    hostname may not resolve), RCCL's socket bootstrap on the loopback interface; the data path is xGMI / P2P."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if world is None:
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
    else:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port or _free_port()}", world_size=world,
                                rank=rank, device_id=dev)


def rccl_ranks(dev):
    """the number of ranks RCCL really connects: an all-reduce(SUM) of ones"""
    if not dist.is_initialized():
        return 0
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    return int(t.item())


def selftest_ranks():
    """hidden --selftest-ranks: what a rank started by self_launch sees, without a GPU (gloo) - the CPU test of
    the launcher (tests/test_host_cpu.py)"""
    dist.init_process_group("gloo", init_method="env://")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"ranks": int(t.item()), "world": dist.get_world_size(),
                          "local_rank": int(os.environ["LOCAL_RANK"])}), file=_OUT, flush=True)
    dist.destroy_process_group()


def main():
    _json_only_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--norm", default="instance", choices=["instance", "batch"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-winograd", action="store_true",
                    help="direct implicit-GEMM kernels for every conv (default: fp32 runs the 3x3 stride-1 convs "
                         "with >= 128x128 channels on Winograd F(6x6,3x3); 16-bit mode runs them direct on "
                         "conv_lp16.hip either way)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = BASELINE.json configs[1] (default); bf16 = the --opt_level O1 path (configs[2]/[4] "
                         "precision): bf16 MFMA operands, every conv-facing tensor 16-bit in HBM, fp32 master "
                         "weights / accumulation / statistics, dynamic loss scaling")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=60.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=CPU_THREADS, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-max-steps", type=int, default=10, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-all-cores", type=float, default=0.0, metavar="SECONDS",
                    help="also time the CPU oracle with os.cpu_count() threads inside this budget (off by default)")
    ap.add_argument("--selftest-ranks", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-timeout", type=float, default=2400.0,
                    help="seconds after which a self-started multi-rank run (a hung collective) is ended with exit code 124")
    ap.add_argument("--dp-side-runs", action="store_true",
                    help="with MMH_FORCE_DP=1 and one rank: also run the data-parallel side regions of an N-GPU launch")
    ap.add_argument("--dp-regions", default="all",
                    help="N-GPU launch: comma-separated side regions to run (dp_bf16_path, dp_norm_batch, dp_norm_batch_o1, "
                         "dp_bf16_path_nopersist, dp_size512_bf16_b4, dp_size512_bf16_b4_norm_batch); default all")
    ap.add_argument("--dp-512", action="store_true",
                    help="N-GPU launch: also time configs[4]'s shape (512x512, per-GPU batch 4, bf16); default only at N = 4")
    ap.add_argument("--share-gpu-gloo", action="store_true",
                    help="TEST AID (tests/test_dp_gpu.py): all N ranks on GPU 0 over the gloo backend - exercises every line of "
                         "the N-GPU bench on a one-GPU box; the throughput it prints means nothing")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the extra driver-visible measurements (direct-kernel steps, set_input in the loop, "
                         "16-bit mode, --norm batch, RCCL world-1, inference, 512x512); profiling runs use this to "
                         "keep the kernel trace to the headline path")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer = BASELINE.json configs[3]: Generator forward only, BN folded, hipGraph")
    a = ap.parse_args()
    if a.cpu_baseline_child:
        return cpu_baseline_child(a.size, a.size, a.norm, a.cpu_budget, a.cpu_threads, a.cpu_max_steps)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing has touched the GPU yet (and this process never will).
        sys.exit(self_launch(a.gpus, sys.argv[1:], a.launch_timeout, a.share_gpu_gloo))
    if a.selftest_ranks:
        return selftest_ranks()
    if a.mode == "infer":
        return infer_main(a)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} inside a {world}-rank launch: start it with "
                         f"--nproc-per-node {a.gpus} (or without a launcher: it starts its own ranks)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dp = os.environ.get("MMH_FORCE_DP") == "1" and "RANK" in os.environ
    if a.share_gpu_gloo:
        dist.init_process_group("gloo", init_method="env://")
    elif world > 1 or force_dp:
        init_rccl(dev)

    from mmhand_amd import ops
    if a.no_winograd:
        ops.USE_WINOGRAD = False
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    opt = default_train_opt(batchSize=a.batch, norm=a.norm, name="bench", local_rank=local,
                            checkpoints_dir="/tmp/mmh_bench", distributed=world > 1 or force_dp,
                            opt_level="O1" if a.dtype == "bf16" else "O0")
    peak = PEAK_BF16_MFMA_TF if a.dtype == "bf16" else PEAK_F32_MFMA_TF
    model = MMHandModel(opt)
    H = W = a.size
    batch = synthetic_batch_gpu(a.batch, H, W, 49 + rank, dev)
    model.set_input(batch)

    # dominant kernel: the 3x3 reflect-pad 512->512 conv fprop at 64x64 (PATBlock streams 2/3)
    hs = H // 4
    wtile = ops._wino_tile(a.batch, hs, hs, 512, 512, 3, 1, 1, a.dtype == "bf16")
    wino = wtile > 0
    planes = (wtile + 2) ** 2
    tiles = a.batch * (-(-hs // max(wtile, 1))) ** 2      # F(6x6,3x3) tiles are ragged: ceil
    timer = KernelTimer(dict(Cin=512, Cout=512, kh=3, stride=1, H=hs, W=hs),
                        gemm=(planes, tiles, 512, 512) if wino else None)
    ops.fprop_timer = timer

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    for _ in range(a.warmup):
        model.optimize_parameters()
    barrier()
    # The collector stays ENABLED in the timed region.  What is alive now (modules, parameters, the
    # option tables) is moved to the permanent generation first, so a generation-2 pass during the
    # timed steps walks one iteration's autograd graph instead of the whole heap; mmhand_amd/train.py
    # does the same after its first iteration.
    gc.collect()
    gc.freeze()
    timer.enabled = True
    ops.flop_meter = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.optimize_parameters()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    flops = dict(ops.flop_meter)
    ops.flop_meter = None
    dt_fastest = dt
    if dist.is_initialized():
        t = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_fastest = float(t[0]), float(-t[1])
    losses = {k: float(v) for k, v in model.get_current_errors().items()}
    peak_gib = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)

    def timed_steps(n, before_step=None):
        """n more optimize_parameters() calls bracketed like the main region; ms per step (max over ranks)."""
        barrier()
        t1 = time.perf_counter()
        for _ in range(n):
            if before_step is not None:
                before_step()
            model.optimize_parameters()
        barrier()
        d = time.perf_counter() - t1
        if dist.is_initialized():
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = tt.item()
        return d / n * 1e3

    n_rccl = rccl_ranks(dev)
    side = {}
    side_ok = not a.no_side_runs and world == 1 and not force_dp
    dp_side = (world > 1 and not a.no_side_runs) or (force_dp and a.dp_side_runs)
    dp_info = {}
    cpu_child = None
    if getattr(model, "dp", False):
        dp_info = {"bucket_allreduce": "mmh_allreduce_bucket (C-ABI, torch.distributed's RCCL communicator)"
                   if getattr(model, "dp_native", None) else f"dist.all_reduce ({getattr(model, 'dp_native_why', '-')})",
                   "inplace_param_grads": bool(getattr(model, "dp_accum", False)),
                   "ms_per_step_fastest_rank": round(dt_fastest / a.steps * 1e3, 2)}
    if world > 1 and rank == 0 and not a.no_cpu_baseline:
        # N-GPU launch: the CPU oracle runs on rank 0's host cores WHILE the side regions below run on the GPUs (16 of the
        # host's threads; every rank's own host work is one thread), collected before the line is printed
        cpu_child = cpu_baseline_start(H, W, a.norm)
    if dp_side:
        # the N-GPU launch measures every multi-GPU configuration of BASELINE.json in this one process group: the headline's
        # exposed communication, then - fresh models - configs[2] (bf16, per-GPU batch 32), the reference's default
        # --norm batch (SyncBN: packed collectives counted) and configs[4]'s shape (512x512, per-GPU batch 4; at N = 4)
        n_side = max(2, min(5, a.steps))
        dp_info["comm_exposed_ms"] = comm_exposed_ms(model, dev, dt / a.steps * 1e3, max(2, min(3, a.steps)))

        dp_regions = None if a.dp_regions == "all" else set(a.dp_regions.split(","))

        def guarded_dp(key, fn):
            if dp_regions is not None and key not in dp_regions:
                return
            try:
                side[key] = fn()
            except Exception as e:      # noqa: BLE001 - every rank raises alike (same code, same shapes): no rank is left behind
                side[key] = {"error": f"{type(e).__name__}: {e}"[:400]}

        ops.fprop_timer = None
        vgg_source = getattr(model, "vgg_source", "n/a")
        del model
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()
        model = None
        if a.dtype == "f32":
            guarded_dp("dp_bf16_path", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, dp=True, seed=49 + rank,
                                                                   opt_level="O1", norm=a.norm),
                                                    note="BASELINE.json configs[2]: --opt_level O1 (bf16), per-GPU batch "
                                                         f"{a.batch}, global batch {world * a.batch}, RCCL gradient all-reduce"))
        guarded_dp("dp_norm_batch", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, dp=True, seed=49 + rank, norm="batch",
                                                                opt_level="O1" if a.dtype == "bf16" else "O0"),
                                                 note="the reference's script default --norm batch under data parallelism = "
                                                      "SyncBN (apex convert_syncbn_model): statistics over the global batch, "
                                                      "collectives of independent norm sites packed"))
        if a.dtype == "f32":
            guarded_dp("dp_norm_batch_o1", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, dp=True, seed=49 + rank,
                                                                       norm="batch", opt_level="O1"),
                                                        note="the reference's SHIPPED combination (scripts/mm-train-ratio.sh:7-40): "
                                                             "--norm batch (SyncBN under DP) with --opt_level O1 (16-bit compute), dropout on"))
            if world > 1:
                # the 16-bit halo kernels launch ONE persistent workgroup per CU; an RCCL kernel holding CUs
                # while such a launch starts delays the workgroups that find no CU.  The same region with one workgroup per
                # tile (lp16_persist = 0; 3-8 % slower alone) answers on the first multi-GPU run which of the two wins there.
                guarded_dp("dp_bf16_path_nopersist",
                           lambda: dict(side_train_run(dev, a.batch, a.size, n_side, dp=True, seed=49 + rank, opt_level="O1",
                                                       norm=a.norm, lib_options={"lp16_persist": (0, 1)}),
                                        note="dp_bf16_path with mmh_set_option('lp16_persist', 0): one workgroup per tile instead "
                                             "of persistent tile lists in the 16-bit 3x3 kernels (A/B against RCCL's CU use)"))
        if world == 4 or a.dp_512:
            guarded_dp("dp_size512_bf16_b4", lambda: dict(side_train_run(dev, 4, 512, n_side, dp=True, seed=49 + rank, opt_level="O1"),
                                                          note="BASELINE.json configs[4]: 512x512, per-GPU batch 4 (global 16 at "
                                                               "4 GPUs), bf16"))
            guarded_dp("dp_size512_bf16_b4_norm_batch",
                       lambda: dict(side_train_run(dev, 4, 512, n_side, dp=True, seed=49 + rank, opt_level="O1", norm="batch"),
                                    note="configs[4]'s shape under SyncBN: the configuration closest to host-bound "
                                         "(host_enqueue_over_step; profiles/r04_host_overhead.txt: 0.71 at world 1)"))
    if side_ok:
        # (1) the reference's set_input inside the loop: the batch comes from pinned HOST memory every
        # step (H2D over PCIe + the NHWC pack), as train.py:35 does per iteration
        host = {k: v.cpu().pin_memory() for k, v in batch.items()}
        n_side = max(2, min(5, a.steps))
        model.set_input(host); model.optimize_parameters()
        ms = timed_steps(n_side, lambda: model.set_input(host))
        side["set_input_in_loop"] = {"images_per_s": round(world * a.batch / ms * 1e3, 3), "ms_per_step": round(ms, 2),
                                     "steps": n_side, "note": "PCIe-inclusive: 6 input tensors copied from pinned "
                                     "host memory and packed to NHWC inside every timed step"}
        model.set_input(batch)
        # (2) the same step on the direct implicit-GEMM kernels only (--no-winograd): the configuration
        # whose every gradient-level parity gate is 1e-3 (tests/test_model_gpu.py)
        if wino and not a.no_winograd:
            ops.USE_WINOGRAD = False
            ops.bump_weights_epoch()
            model.optimize_parameters()
            ms = timed_steps(n_side)
            side["direct_path"] = {"images_per_s": round(world * a.batch / ms * 1e3, 3), "ms_per_step": round(ms, 2),
                                   "steps": n_side, "note": "same workload with every conv on the direct "
                                   "implicit-GEMM MFMA kernels (bench.py --no-winograd)"}
            ops.USE_WINOGRAD = True
            ops.bump_weights_epoch()
    if model is not None:
        vgg_source = getattr(model, "vgg_source", "n/a")
    ops.fprop_timer = None
    if side_ok:
        # (3)... fresh models: the 16-bit mode, the reference's default --norm batch, RCCL with one rank, the other
        # BASELINE.json configs at their per-GPU shapes.  Each is guarded: a failure is reported under its key.
        del model
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()
        n_side = max(2, min(5, a.steps))

        def guarded(key, fn):
            try:
                side[key] = fn()
            except Exception as e:      # noqa: BLE001 - the headline line must survive a failing side run
                side[key] = {"error": f"{type(e).__name__}: {e}"[:400]}

        if a.dtype == "f32":
            guarded("bf16_path", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, stack=True, opt_level="O1",
                                                             norm=a.norm),
                                              note="configs[2]'s precision and per-GPU shape on one GPU: --opt_level O1 (bf16 "
                                                   "MFMA operands, 16-bit conv-facing tensors, fp32 master weights, dynamic "
                                                   "loss scaling); stack_frac = the 3x3 stride-1 256/512-channel stack, all "
                                                   "three passes incl. reflect border terms, algorithmic FLOPs / kernel time / 2500 TF"))
        gnote = ("--graph_step: the same iteration captured once into a hipGraph (forward, three backward passes, three Adam "
                 "steps; Adam step count / lr, dropout salt and image-pool indices behind device pointers) and replayed: "
                 "host_enqueue_ms is what the host spends per iteration")
        if a.dtype == "f32":
            guarded("bf16_path_graph", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, warmup=6, opt_level="O1",
                                                                   norm=a.norm, graph_step=True), note=gnote))
            guarded("size512_bf16_b4_graph", lambda: dict(side_train_run(dev, 4, 512, n_side, warmup=6, opt_level="O1",
                                                                         graph_step=True), note=gnote))
        guarded("graph_step", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, warmup=6, norm=a.norm, graph_step=True,
                                                          opt_level="O1" if a.dtype == "bf16" else "O0"),
                                           note="the headline configuration under " + gnote))
        if a.dtype == "f32" and not a.no_winograd:
            guarded("gradient_parity", lambda: gradient_parity_run(dev, a.size, a.norm))

            def hybrid():
                try:
                    r = side_train_run(dev, a.batch, a.size, n_side, norm=a.norm, fp32_exact_grads=True)
                finally:
                    ops.set_winograd_mode("all")
                gp = side.get("gradient_parity", {}).get("winograd_dgrad_wgrad_only_vs_direct", {})
                r.update({"grad_rel_l1_median_vs_direct": gp.get("grad_rel_l1_median"),
                          "grad_rel_l1_max_vs_direct": gp.get("grad_rel_l1_max"),
                          "direct_bound_images_per_s": round(PEAK_F32_MFMA_TF * 1e3 / (GFLOP_PER_IMAGE_STEP * (a.size * a.size / 65536.0)), 1),
                          "note": "--fp32_exact_grads: forward 3x3 convs on the direct implicit-GEMM kernels (identical activations "
                                  "and ReLU masks to direct_path), dgrad + wgrad on Winograd F(6x6,3x3) - every gradient tensor "
                                  "within 1e-3 (tests/test_lp16_step_gpu.py::..._fp32_full_width[bwd] vs fp64; this line: vs "
                                  "direct_path's kernels at full size), against BASELINE.md's <= 64 img/s bound for an all-direct "
                                  "fp32 step at 100 % MFMA"})
                return r
            guarded("hybrid_path", hybrid)
        guarded("norm_batch", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, norm="batch",
                                                          opt_level="O1" if a.dtype == "bf16" else "O0"),
                                           note="the reference's script default --norm batch (BatchNorm2d affine, conv bias off)"))
        if a.dtype == "f32":
            guarded("norm_batch_o1", lambda: dict(side_train_run(dev, a.batch, a.size, n_side, norm="batch", opt_level="O1"),
                                                  note="the reference's SHIPPED combination (scripts/mm-train-ratio.sh:7-40, "
                                                       "options/base_options.py:66-70): --norm batch with --opt_level O1 "
                                                       "(16-bit compute, dynamic loss scaling), dropout on"))
        guarded("dp_rccl_world1", lambda: dict(rccl_child_run(n_side, max(2, a.warmup), a.batch, a.size),
                                               note="same fp32 step through the data-parallel path (MMH_FORCE_DP=1) on RCCL, "
                                                    "world size 1, own process"))
        guarded("size512_bf16_b4", lambda: dict(side_train_run(dev, 4, 512, n_side, opt_level="O1"),
                                                note="configs[4] per-GPU shape: 512x512, batch 4 (16/4), bf16"))
        guarded("infer_b64_f32", lambda: dict(side_infer_run(dev, 64, 256, n_side, False),
                                              note="configs[3]: Generator forward, batch 64, BN folded, hipGraph replay"))
        guarded("infer_b64_bf16", lambda: side_infer_run(dev, 64, 256, n_side, True))

    if rank == 0:
        imgs_per_s = world * a.batch * a.steps / dt
        k_ms, k_n = timer.mean_ms()
        op_ms = timer.op_mean_ms()
        if wino:      # P x [tiles x 512] . [512 x 512]: the arithmetic this launch really does
            k_flop = planes * 2.0 * tiles * 512 * 512
            k_name = (f"{'wino_gemm_bf16_kernel' if a.dtype == 'bf16' else 'wino_gemm_kernel<128,2> (two-level accumulation)'}: the {planes} Winograd-domain GEMMs "
                      f"[{tiles}x512].[512x512] (F({wtile}x{wtile},3x3)) of the 3x3 512->512 fprop @{hs}x{hs}")
        else:
            k_flop = 2.0 * a.batch * hs * hs * 512 * 512 * 9
            k_name = (("conv_lp16h2_kernel<bf16> (16-bit operands by LDS-DMA, 16x16-pixel tile x 256 channels with its halo "
                       "resident in LDS for all nine taps, MFMA 16x16x32, fragment reads pipelined into the MFMA stream)"
                       if (a.dtype == "bf16" and ops.lp16_v2_ok(512, 512, 3, 1, 1, 0)) else
                       "conv_igemm_bf16_kernel<128,2,2>" if a.dtype == "bf16" else
                       "conv_igemm_kernel<256,2,2,false>") + " fprop 3x3 512->512 @64x64")
        traffic = None      # HBM bytes per launch of the roofline kernel, from the committed PMC run
        tname = "r06_traffic_bf16.json" if a.dtype == "bf16" else ("r06_traffic.json" if wino else "r01_traffic.json")
        tj = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tj):      # the previous round's counters until this round's are collected
            tj = os.path.join(ROOT, "profiles", tname.replace("r06_", "r05_"))
        if os.path.exists(tj) and a.batch == 32 and a.size == 256:
            tjd = json.load(open(tj))
            traffic = tjd.get("winograd_gemm" if wino else "direct", {}).get("hbm_bytes_per_launch")
        achieved = k_flop / (k_ms * 1e-3) / 1e12
        mfma_flop_step = flops.get("mfma", 0.0) / a.steps
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "kernel": f"{k_name} (B={a.batch}): {k_flop / 1e9:.1f} GFLOP/launch, {k_ms:.3f} ms avg "
                          f"over {k_n} launches in the timed region",
                # all executed matrix-core FLOPs of one step (every conv launch counts the multiplications
                # it really performs: Winograd-domain GEMMs their own, not the direct algorithm's)
                # / (step wall time x peak): the whole step's MFMA utilisation
                "step_mfma_frac": round(mfma_flop_step / (dt / a.steps) / 1e12 / peak, 4),
                "step_mfma_tflop": round(mfma_flop_step / 1e12, 3)}
        if wino and op_ms:
            # the same convolution as ONE op: input transform + GEMM + output transform
            roof["op_frac"] = round(k_flop / (op_ms * 1e-3) / 1e12 / peak, 4)
            roof["op_ms"] = round(op_ms, 3)
        base_frac = GFLOP_PER_IMAGE_STEP * (H * W / 65536.0) * imgs_per_s / world / 1e3 / peak
        line = {
            "metric": "256x256 hand images/sec (G+D step)", "value": round(imgs_per_s, 3),
            "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"RHD-shaped {H}x{W}, per-GPU batch {a.batch}, "
                       f"{'bf16 MFMA compute' if a.dtype == 'bf16' else 'fp32'}"
                       f"{f' (Winograd F({wtile}x{wtile},3x3) on the 3x3 stack)' if wino else ''}, G(9 PATBlocks,"
                       f" ngf64)+D_PB+D_PP+L1/perceptual/GAN+Adam, --norm {a.norm}, dropout on; VGG19[:4] "
                       f"weights: {vgg_source}",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}"},
            # BASELINE.md §4's formula (2444.4 GFLOP of DIRECT convolution per image and step x images/s
            # / peak).  With Winograd on the 3x3 stack it exceeds 1.0, because F(6x6,3x3) executes 5.06x
            # fewer multiplications than the direct algorithm that formula prices - it is a throughput
            # yardstick against BASELINE.md's "<= 64 img/s" bound, NOT a utilisation.  Utilisations:
            # roofline.frac (dominant kernel), roofline.op_frac (that conv incl. its transforms),
            # roofline.step_mfma_frac (whole step).
            "direct_equiv_tflops": round(GFLOP_PER_IMAGE_STEP * (H * W / 65536.0) * imgs_per_s / world / 1e3, 1),
            "direct_equiv_frac_of_peak": round(base_frac, 4),
            "peak_hbm_gib": peak_gib,
            "roofline": roof,
            "losses": {k: round(v, 5) for k, v in losses.items()},
            # ranks the collective backend really connects (an all-reduce of ones); 0 = single process, no
            # process group.  N > 1: gradients of G / D_PB / D_PP all-reduced over RCCL (mmhand_amd/dp.py)
            "rccl_ranks": n_rccl, "backend": (("nccl(RCCL)" if dist.get_backend() == "nccl" else dist.get_backend()) if dist.is_initialized() else None),
        }
        line.update(dp_info)
        line.update(side)
        if cpu_child is not None:
            line["cpu_baseline"] = cpu_baseline_collect(cpu_child)
        if world == 1 and not force_dp and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_with_sweep(H, W, a.norm)
            # SURVEY.md §8(d) asks for os.cpu_count() threads.  On the GPU box's 256 logical CPUs oneDNN is far slower
            # at 128+ threads than at 16 (tools/cpu_probe.py; one B=2 step did not finish in 240 s): opt-in
            # (--cpu-all-cores SECONDS; the round-4 measurement, profiles/r04_cpu_all_cores.json, timed out at 256 threads), not paid by every default run
            if a.cpu_all_cores and (os.cpu_count() or 0) > CPU_THREADS:
                allc = cpu_baseline(H, W, a.norm, budget_s=a.cpu_all_cores, hard_timeout_s=int(a.cpu_all_cores * 2 + 60),
                                    threads=os.cpu_count(), max_steps=2)
                line["cpu_baseline"]["all_cores"] = {k: allc.get(k) for k in ("value", "cores", "sample")}
        line["summary"] = line_summary(line)       # LAST key: what a truncated tail of this line must still show
        print(json.dumps(line), file=_OUT, flush=True)
    if dist.is_initialized():
        dist.barrier()          # the other ranks wait here while rank 0 collects the CPU baseline and prints
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
