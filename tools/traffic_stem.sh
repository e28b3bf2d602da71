#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fp32 7x7 stem kernel (44->64), one counter per pass, without and with the
# XCD-banded row tiles (conv_xcd1; the default).  Algorithmic bytes: x 369.1 MB + y 536.9 MB = 906 MB.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for o in "conv_xcd1=0" "conv_xcd1=1"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fs; MMH_OPTS=$o rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_stem.py >/dev/null 2>&1
    echo "== MMH_OPTS=$o --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A2 "conv_igemm" | grep -v "^--"
  done
done
