#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) for the three passes of the dominant conv, defaults.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_conv.py >/dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/fs | grep -A1 "conv_\|reflect_fold\|slab_red" | grep -v "^--"
done
