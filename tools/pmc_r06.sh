#!/bin/bash
# Round 6: hardware counters of the 16-bit kernels and of the fp32 headline GEMM - one `--pmc` pass per counter
# group, the program DIRECTLY behind `--`, no tracing domain beside it - into gpurun_out/r6prof/pmc_lp16.txt.
# FAILS LOUDLY: a pass that exits non-zero, or a kernel of EXPECT without counter rows in any pass, ends the script with exit 1
# (round 5's collector piped "whatever it got" into profiles/ and committed four header lines).
#     bash tools/pmc_r06.sh          (on the GPU box, from the repo root)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r6prof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
EXPECT="conv_lp16h2_kernel conv_lp16h2_nbr_kernel wgrad_lp16t_kernel conv_lp16g conv_s2f_kernel wino_gemm_kernel"
: > $OUT/pmc_lp16.txt
fail=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/fs
  if ! rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r06.py > /tmp/pmc_pass.log 2>&1; then
    echo "pmc_r06.sh: the pass [$c] FAILED:" >&2; tail -20 /tmp/pmc_pass.log >&2; fail=1; continue
  fi
  echo "== --pmc $c" >> $OUT/pmc_lp16.txt
  if ! python3 $R/tools/pmc_select.py /tmp/fs $EXPECT >> $OUT/pmc_lp16.txt; then
    echo "pmc_r06.sh: the pass [$c] is incomplete" >&2; fail=1
  fi
done
lines=$(wc -l < $OUT/pmc_lp16.txt)
if [ "$lines" -lt 60 ]; then echo "pmc_r06.sh: only $lines lines of counters collected" >&2; fail=1; fi
[ $fail -eq 0 ] && echo "pmc_r06.sh: ok, $lines lines -> $OUT/pmc_lp16.txt" || { echo "pmc_r06.sh: INCOMPLETE" >&2; exit 1; }
