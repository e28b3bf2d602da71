#!/bin/bash
# Round 2: FETCH_SIZE / WRITE_SIZE / MFMA-busy / instruction mix / LDS conflicts of the dominant kernels,
# every counter set in its own --pmc pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share one).
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r02.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A2 "lp16\|wino\|slab_red\|cvt" | grep -v "^--"
done
