#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) per dispatch of the round-3 fp32 kernels; every dispatch listed (two shapes per kernel)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r03_f32.py >/dev/null 2>&1
  echo "== --pmc $c (KB per dispatch, in dispatch order per kernel)"
  python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for f in glob.glob("/tmp/fs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("dgrad_s2_kernel", "wgrad_s2_kernel", "conv_stem_f32_kernel", "wino_wgrad_dma_kernel", "slab_reduce")):
            acc.setdefault(k[:80], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, v in acc.items():
    v.sort()
    print(k); print("   " + " ".join(f"{x:.0f}" for _, x in v))
PY
done
