"""Per-shape launch durations of one kernel from a rocprofv3 --kernel-trace CSV: shows that the
HIP-event average bench.py reports for its roofline kernel agrees with the profiler.
usage: python tools/roofline_from_trace.py <dir with *_kernel_trace.csv> [kernel substring]
The substring is matched with all blanks removed on both sides, so give the FULL template argument list of the kernel you
mean - "wino_gemm_kernel<128,2>" (the two-level forward GEMMs bench.py times) is not "wino_gemm_kernel<128,1>" (the
one-level dgrad twin, 5 % faster): a prefix that matches both mixes them."""
import collections, csv, glob, statistics, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm_kernel<128, 2, 2, false, false>"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat.replace(" ", "") in r["Kernel_Name"].replace(" ", ""):
        wg = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        acc[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"kernel: {pat}\ntrace:  {f.split('/')[-1]}")
print(f"{'workgroups (x,y,z)':>22s} {'launches':>9s} {'mean us':>10s} {'min us':>10s} {'max us':>10s} {'total ms':>10s}")
def clusters(v, gap=1.12):
    """Split sorted durations where two neighbours differ by more than `gap`x: one grid can carry
    several GEMM shapes (the persistent Winograd kernel always launches 8 x 96 workgroups)."""
    v = sorted(v)
    out = [[v[0]]]
    for x in v[1:]:
        if x > out[-1][-1] * gap:
            out.append([])
        out[-1].append(x)
    return out
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{str(k):>22s} {len(v):9d} {statistics.mean(v):10.1f} {min(v):10.1f} {max(v):10.1f} {sum(v)/1e3:10.1f}")
    if ("batched" in pat or "wino_gemm" in pat or "lp16h2" in pat) and max(v) > 1.5 * min(v):
        for c in clusters(v):
            if len(c) >= 5:
                print(f"{'  duration cluster':>22s} {len(c):9d} {statistics.mean(c):10.1f} {min(c):10.1f} {max(c):10.1f} {sum(c)/1e3:10.1f}")
if "wino_gemm" in pat:
    print("wino_gemm_kernel is persistent: every launch has 8 XCDs x 96 workgroups (3 per CU), whatever the GEMM shape, so\n"
          "shapes are told apart by duration.  The slowest cluster is the Winograd-domain GEMM set of a 3x3\n"
          "512->512 conv at 64x64, B=32 (fprop and dgrad launches are the same GEMM): F(6x6,3x3) = 64 x\n"
          "[3872x512].[512x512] = 129.9 GFLOP per launch (F(4x4,3x3), MMH_WINOGRAD_TILE=4: 36 x [8192x512].[512x512]\n"
          "= 154.6 GFLOP); bench.py's roofline.achieved = that FLOP count / its HIP-event mean over the fprop\n"
          "launches.  The other clusters: 512->256 / 256->512 and 256->256 convs (1/2 and 1/4 of the FLOPs).")
elif "lp16" in pat:
    print("conv_lp16h2_kernel<H16, SIGN, FOLD> is persistent since round 4: 8 XCDs x min(tiles / 8, CUs / 8) workgroups walk tile\n"
          "lists (tiles of 16x16 pixels x 256 channels), so a 256-workgroup grid carries every shape and shapes are told apart\n"
          "by duration.  Give the full template list: <false, 1, false> = bf16 fprop (bench.py --dtype bf16 / bf16_path times\n"
          "its 512->512 launches @64x64, B=32, with HIP events: 618.5 GFLOP per launch, 16 per step = the SLOWEST cluster of\n"
          "the B=32 launches; the two discriminators' merged real+fake passes run B=64 at 256 channels), <false, -1, true> = the\n"
          "complete reflect dgrad (ring folded), <false, -1, false> = zero-pad dgrad.  Clusters below it: 512->256 (half the\n"
          "FLOPs) and 256->256 (a quarter; B=64 launches of it take as long as 512->256).")
elif "batched" in pat:
    print("workgroups (4, 64, 36) = 4 column tiles x 64 row tiles x 36 Winograd planes = [8192x512].[512x512] per\n"
          "plane: the F(4x4,3x3) GEMMs of a 3x3 512->512 conv at 64x64, B=32 (fprop and dgrad launches both have this\n"
          "shape); 154.6 GFLOP per launch; bench.py's roofline.achieved = 154.6 GFLOP / its HIP-event mean over the\n"
          "fprop launches.  The same grid also carries the dgrad of the 512->256 convs ([8192x256].[256x512], half\n"
          "the contraction): per step 32 launches with K=512 (upper duration cluster) and 16 with K=256 (lower).")
else:
    print("workgroups (4, 1024, 1) = 4 column tiles x 1024 row tiles = the 3x3 512->512 fprop at 64x64, B=32:\n"
          "618.5 GFLOP per launch; bench.py's roofline.achieved = 618.5 GFLOP / its HIP-event mean.")
