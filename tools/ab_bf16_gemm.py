"""A/B a knob on the bf16 Winograd NT GEMM alone (16 planes, B=32, 64x64 -> 32768 tiles)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import lib
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2:]]
L = lib.load(); dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for (K, N) in [(512, 512), (256, 256), (512, 256), (256, 512)]:
    tiles = 32768
    V = torch.randn(16, tiles, K, device=dev).bfloat16(); U = torch.randn(16, N, K, device=dev).bfloat16()
    M = torch.empty(16, tiles, N, dtype=torch.bfloat16, device=dev)
    ref = None
    res = {v: [] for v in vals}
    for v in vals:
        lib.check(L.mmh_set_option(key.encode(), v), "set")
        lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, K, N, 16, lib.BF16, st())
        torch.cuda.synchronize()
        if ref is None: ref = M.clone()
        else: assert torch.equal(ref, M), f"{key}={v} changes the result"
    for r in range(5):
        for v in vals:
            lib.check(L.mmh_set_option(key.encode(), v), "set")
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, K, N, 16, lib.BF16, st())
            e1.record(); torch.cuda.synchronize(); res[v].append(e0.elapsed_time(e1) / 5)
    gf = 16 * 2.0 * tiles * K * N / 1e9
    print(f"K={K} N={N}: " + " | ".join(f"{key}={v}: {statistics.median(res[v]):.3f} ms ({gf / statistics.median(res[v]):.0f} TF)" for v in vals), flush=True)
