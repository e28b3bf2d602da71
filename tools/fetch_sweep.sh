#!/bin/bash
# FETCH_SIZE of the dominant conv passes under different K orders / tile maps (run on the GPU box)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cw in 1 2 4 8; do for xcd in 0 1; do
  export MMH_OPTS="conv_cw=$cw,conv_xcd=$xcd"
  rm -rf /tmp/fs; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_conv.py >/dev/null 2>&1
  echo "== cw=$cw xcd=$xcd"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A1 "conv_igemm" | grep -v "^--"
done; done
