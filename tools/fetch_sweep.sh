#!/bin/bash
# FETCH_SIZE of the dominant conv passes under different K orders (run on the GPU box)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cw in 1 2 4; do
  export MMH_OPTS="conv_cw=$cw,conv_xcd=1"
  rm -rf /tmp/fs; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_conv.py >/dev/null 2>&1
  echo "== cw=$cw"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A1 "conv_igemm\|conv_wgrad" | grep -v "^--"
done
