"""bf16 Winograd F(2x2,3x3) vs the direct bf16 kernels on the K4 shapes (whole op, HIP events), and the
bf16 GEMM stages alone."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = 32
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (H, Cin, Cout) in [(64, 512, 512), (64, 256, 256), (64, 512, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dy = torch.randn(B, H, H, Cout, device=dev)
    gf = 2.0 * B * H * H * Cin * Cout * 9 / 1e9
    for name, f in (("fprop", lambda: ops.raw_conv_fprop(x, w, None, 1, 1, True, 0, bf16=True)),
                    ("dgrad", lambda: ops.raw_conv_dgrad(dy, w, x.shape, 1, 1, True, bf16=True)),
                    ("wgrad", lambda: ops.raw_conv_wgrad(x, dy, 3, 1, 1, True, bf16=True))):
        res = {}
        for wb in (False, True):
            ops.USE_WINOGRAD_BF16 = wb
            res[wb] = timeit(f)
        print(f"{Cin}->{Cout}@{H} {name}: direct bf16 {res[False]:.3f} ms ({gf / res[False]:.0f} TF direct-equiv) | "
              f"bf16 Winograd F(2,3) {res[True]:.3f} ms ({res[False] / res[True]:.2f}x)", flush=True)
    tiles = B * (H // 2) ** 2
    V = torch.randn(16, tiles, Cin, device=dev).bfloat16(); U = torch.randn(16, Cout, Cin, device=dev).bfloat16()
    M = torch.empty(16, tiles, Cout, dtype=torch.bfloat16, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, Cin, Cout, 16, lib.BF16, st()))
    print(f"   NT GEMM 16x[{tiles}x{Cin}].[{Cin}x{Cout}]: {t:.3f} ms = {16 * 2.0 * tiles * Cin * Cout / t / 1e9:.0f} TF")
    if Cin % 128 == 0 and Cout % 128 == 0:
        Y = torch.randn(16, tiles, Cout, device=dev).bfloat16()
        nws = lib.load().mmh_wino_wgrad_gemm_ws_bytes(tiles, Cin, Cout, 16)
        ws = torch.empty(nws // 4 + 4, device=dev); dU = torch.empty(16, Cin, Cout, device=dev)
        t = timeit(lambda: lib.call("mmh_wino_wgrad_gemm", V.data_ptr(), Y.data_ptr(), tiles, Cin, Cout, 16, lib.BF16,
                                    ws.data_ptr(), nws, dU.data_ptr(), st()))
        print(f"   TN GEMM (+slab reduce): {t:.3f} ms = {16 * 2.0 * tiles * Cin * Cout / t / 1e9:.0f} TF")
