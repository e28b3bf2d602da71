"""Host-side enqueue time of one optimize_parameters() against its GPU time: is a configuration launch-bound?

    python tools/host_overhead.py                 # the table of VERDICT r3 #5 (256x256 B=32 f32 / bf16, 512x512 B=4 bf16,
                                                  # single process and through the data-parallel path on RCCL, world 1)
    python tools/host_overhead.py --size 512 --batch 4 --dtype bf16 [--norm batch] [--dp]   # one row

enqueue ms : wall time of the optimize_parameters() CALL with the GPU idle when it starts (synchronised before) - what
             the host needs to issue one iteration (Python + autograd + ctypes + hipLaunchKernel);
step ms    : steady state, 10 calls back to back and one synchronise - what bench.py measures;
gpu ms     : HIP events around one iteration in steady state (the GPU's own time for it);
calls      : C-ABI calls per iteration.
enqueue / step >= 0.7: the host is the bottleneck or close to it (the GPU idles whenever the host hiccups).
Every row runs in a child process of its own (a data-parallel row needs its own process group)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(a):
    import torch
    import torch.distributed as dist
    if a.dp:
        os.environ.update(MMH_FORCE_DP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          NCCL_SOCKET_IFNAME="lo")
        os.environ.setdefault("MASTER_PORT", "29731")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
    from bench import synthetic_batch_gpu
    from mmhand_amd import lib as L
    from mmhand_amd.mmhand_model import MMHandModel
    from mmhand_amd.options import default_train_opt
    dev = torch.device("cuda:0")
    model = MMHandModel(default_train_opt(batchSize=a.batch, norm=a.norm, name="host", checkpoints_dir="/tmp/mmh_bench",
                                          distributed=bool(a.dp), opt_level="O1" if a.dtype == "bf16" else "O0"))
    model.set_input(synthetic_batch_gpu(a.batch, a.size, a.size, 49, dev))
    for _ in range(3):
        model.optimize_parameters()
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze()
    calls = [0]
    real = L.call

    def counting(name, *args):
        calls[0] += 1
        return real(name, *args)
    L.call = counting
    model.optimize_parameters()
    L.call = real
    torch.cuda.synchronize()
    enq = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.optimize_parameters()
        enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        model.optimize_parameters()
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        model.optimize_parameters()
    e0.record()
    for _ in range(4):
        model.optimize_parameters()
    e1.record()
    torch.cuda.synchronize()
    gpu = e0.elapsed_time(e1) / 4
    enq_ms = 1e3 * sorted(enq)[len(enq) // 2]
    print(json.dumps({"size": a.size, "batch": a.batch, "dtype": a.dtype, "norm": a.norm, "dp": bool(a.dp),
                      "enqueue_ms": round(enq_ms, 1), "step_ms": round(step * 1e3, 1), "gpu_ms": round(gpu, 1),
                      "calls": calls[0], "enqueue_over_step": round(enq_ms / (step * 1e3), 2)}), flush=True)
    if a.dp:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--norm", default="instance")
    ap.add_argument("--dp", action="store_true")
    a = ap.parse_args()
    if a.size:
        return one(a)
    rows = [(256, 32, "f32", "instance", False), (256, 32, "bf16", "instance", False), (256, 32, "bf16", "instance", True),
            (512, 4, "bf16", "instance", False), (512, 4, "bf16", "instance", True), (512, 4, "bf16", "batch", True)]
    print(f"{'config':44s} {'calls':>6s} {'enqueue ms':>11s} {'step ms':>9s} {'gpu ms':>8s} {'enqueue/step':>13s}")
    for i, (size, B, dt, norm, dp) in enumerate(rows):
        cmd = [sys.executable, os.path.abspath(__file__), "--size", str(size), "--batch", str(B), "--dtype", dt, "--norm", norm]
        if dp:
            cmd.append("--dp")
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_PORT=str(29731 + i)))
        js = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not js:
            print(f"{size}x{size} B={B} {dt} {norm} dp={dp}: FAILED rc {out.returncode}: {out.stderr[-300:]}")
            continue
        j = json.loads(js[-1])
        name = f"{size}x{size} B={B} {dt} --norm {norm}" + (" DP(RCCL world 1)" if dp else "")
        print(f"{name:44s} {j['calls']:6d} {j['enqueue_ms']:11.1f} {j['step_ms']:9.1f} {j['gpu_ms']:8.1f} {j['enqueue_over_step']:13.2f}")


if __name__ == "__main__":
    main()
