"""Per-shape launch durations of one kernel from a rocprofv3 --kernel-trace CSV: shows that the
HIP-event average bench.py reports for its roofline kernel agrees with the profiler.
usage: python tools/roofline_from_trace.py <dir with *_kernel_trace.csv> [kernel substring]"""
import collections, csv, glob, statistics, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm_kernel<128, 2, 2, false, false>"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        wg = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        acc[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"kernel: {pat}\ntrace:  {f.split('/')[-1]}")
print(f"{'workgroups (x,y,z)':>22s} {'launches':>9s} {'mean us':>10s} {'min us':>10s} {'max us':>10s} {'total ms':>10s}")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{str(k):>22s} {len(v):9d} {statistics.mean(v):10.1f} {min(v):10.1f} {max(v):10.1f} {sum(v)/1e3:10.1f}")
print("workgroups (4, 1024, 1) = 4 column tiles x 1024 row tiles = the 3x3 512->512 fprop at 64x64, B=32:\n"
      "618.5 GFLOP per launch; bench.py's roofline.achieved = 618.5 GFLOP / its HIP-event mean.")
