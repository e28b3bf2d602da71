#!/bin/bash
# Round 5: MFMA-busy and instruction mix of the halo kernel's two forms (VERDICT r4 #1 asked for MFMA-busy >= 75 % of the
# one-wave-per-SIMD form: it reaches less than the shipped form) and of the new stride-2 dgrad.  One --pmc pass per
# group, the program directly behind `--`; the A/B library through MMH_LIB_PATH.   bash tools/pmc_r05.sh > gpurun_out/r05_pmc_lp16.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
[ -f $R/mmhand_amd/libmmhand_hip_ab.so ] && export MMH_LIB_PATH=$R/mmhand_amd/libmmhand_hip_ab.so
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r05.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A5 "conv_lp16h2\|conv_lp16q\|conv_s2d" | grep -v "^--"
done
