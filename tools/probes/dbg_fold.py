"""Where does mode 2 (reflect fold inside the halo kernel) differ from mode 1 + border?  Per region of the image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmhand_amd import ops, lib
L = lib.load(); dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 1, 32, 32, 256, 64
torch.manual_seed(0)
dy = torch.randn(B, H, W, Cout, device=dev); w = torch.randn(3, 3, Cin, Cout, device=dev) * 0.1
dyb = ops.lp16_twin(dy, True)
def run(fold):
    ops.USE_LP16_FOLD = fold
    return ops.raw_conv_dgrad(None, w, (B, H, W, Cin), 1, 1, True, bf16=True, dy16=dyb, out16=False)
a, b = run(True), run(False)
d = (a - b).abs().amax(dim=(0, 3)).cpu()
sc = float(b.abs().max())
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
print("scale", sc)
print((d / sc * 100).round().int())
q = (d / sc * 100).round().int()
print("row 1   ", q[1].tolist()); print("row H-2 ", q[H - 2].tolist()); print("col 1   ", q[:, 1].tolist()); print("col W-2 ", q[:, W - 2].tolist())
for bits in (8, 16, 1024):
    lib.check(L.mmh_set_option(b"lp16_dbg", bits), "set"); a2 = run(True); lib.check(L.mmh_set_option(b"lp16_dbg", 0), "set")
    d2 = (a2 - a).abs().amax(dim=(0, 3)).cpu(); q2 = (d2 / sc * 100).round().int()
    print("dbg", bits, "changes: row 1", q2[1].tolist(), "row H-2", q2[H - 2].tolist(), "col 1", q2[:, 1].tolist())
def opt(bits):
    lib.check(L.mmh_set_option(b"lp16_dbg", bits), "set"); r = run(True); lib.check(L.mmh_set_option(b"lp16_dbg", 0), "set"); return r
a_nc, a_ncol, a_nrow, a_none, a_zc = opt(1024), opt(8), opt(16), opt(4), opt(2048)
ops.USE_LP16_FOLD = True
main = ops.raw_conv3x3_lp16(dyb, w, None, False, 0, True, 1, out16=False)
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for nm, t in (("border", b), ("fold", a), ("no corner", a_nc), ("no col", a_ncol), ("no row", a_nrow), ("no folds", a_none), ("zero corner", a_zc), ("main", main)):
    print(f"{nm:10s} row 1 ch 0 cols 0..7:", (t[0, 1, :8, 0] - main[0, 1, :8, 0]).cpu().numpy().round(3), " row 5 col 1:", float(t[0, 5, 1, 0] - main[0, 5, 1, 0]))
