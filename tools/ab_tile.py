"""F(4x4,3x3) vs F(6x6,3x3) on the K4 shapes: whole op per pass and per stage (HIP events)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmhand_amd import ops, lib
dev = torch.device("cuda:0"); B = int(os.environ.get("B", "32"))
def timeit(fn, iters=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
st = lambda: torch.cuda.current_stream().cuda_stream
for (H, Cin, Cout) in [(64, 512, 512), (64, 256, 256), (64, 512, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.05
    dy = torch.randn(B, H, H, Cout, device=dev)
    for name, f in (("fprop", lambda t: ops.raw_conv_fprop_wino(x, w, None, True, 0, t)),
                    ("dgrad", lambda t: ops.raw_conv_dgrad_wino(dy, w, x.shape, True, t)),
                    ("wgrad", lambda t: ops.raw_conv_wgrad_wino(x, dy, True, t))):
        r = {t: timeit(lambda: f(t)) for t in (4, 6)}
        print(f"{Cin}->{Cout}@{H} {name}: F(4,3) {r[4]:.3f} ms | F(6,3) {r[6]:.3f} ms ({r[4] / r[6]:.2f}x)", flush=True)
    for t in (4, 6):
        P = (t + 2) ** 2; tiles = B * (-(-H // t)) ** 2
        V = torch.empty(P, tiles, Cin, device=dev); M = torch.empty(P, tiles, Cout, device=dev)
        U = torch.randn(P, Cin, Cout, device=dev); y = torch.empty(B, H, H, Cout, device=dev)
        ti = timeit(lambda: lib.call("mmh_wino_input", x.data_ptr(), B, H, H, Cin, 1, t, lib.F32, V.data_ptr(), st()))
        tg = timeit(lambda: lib.call("mmh_wino_gemm", V.data_ptr(), U.data_ptr(), M.data_ptr(), tiles, Cin, Cout, P, lib.F32, st()))
        to = timeit(lambda: lib.call("mmh_wino_output", M.data_ptr(), y.data_ptr(), None, B, H, H, Cout, 0, t, lib.F32, None, 0, st()))
        gf = P * 2.0 * tiles * Cin * Cout / 1e9
        print(f"   tile {t}: input {ti*1e3:.0f} us ({(x.numel()+V.numel())*4/ti/1e9:.2f} TB/s) | gemm {tg*1e3:.0f} us ({gf/tg:.0f} TF) | "
              f"output {to*1e3:.0f} us ({(M.numel()+y.numel())*4/to/1e9:.2f} TB/s)", flush=True)
