#!/bin/bash
# Round 4: MFMA-busy, instruction-mix and LDS counters of the 16-bit 3x3 kernels as they stand at the end of the round
# (tools/pmc_r03.py launches the 512->512 @64x64 B=32 fprop / plain dgrad / reflect-fold dgrad / wgrad a few times each).
# One rocprofv3 --pmc pass per counter group, no trace domains.  usage (GPU box): bash tools/pmc_r04.sh > gpurun_out/r04_pmc_lp16.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r03.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A4 "lp16" | grep -v "^--"
done
