#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) for the Winograd passes of the dominant conv, then the SQ
# MFMA-busy counters; every kernel of tools/pmc_wino.py is listed.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_wino.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A2 "conv_\|wino\|slab_red\|border" | grep -v "^--"
done
