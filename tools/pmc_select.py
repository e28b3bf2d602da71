"""pmc_select.py <rocprofv3 --pmc output dir> <kernel name pattern>...: the per-kernel counter means (tools/pmc_summary.py's
format) of the kernels whose name contains one of the patterns; EXIT CODE 1 when a pattern matches no kernel with counter rows
(a collector must fail loudly: round 5 committed an empty counter file)."""
import collections
import csv
import glob
import sys

d, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
missing = []
for p in pats:
    hit = [k for k in acc if p in k and acc[k]]
    if not hit:
        missing.append(p)
    for k in sorted(hit):
        print(k[:110])
        for c, v in sorted(acc[k].items()):
            print(f"   {c:32s} mean={sum(v) / len(v):16.1f} n={len(v)}")
if missing:
    print("pmc_select.py: no counter rows for: " + ", ".join(missing) + f" (kernels seen: {len(acc)})", file=sys.stderr)
    sys.exit(1)
