# where do the stride-2 16-bit dgrad's (conv_s2d_kernel) stores go?  Counters of the kernel beside a plain fill of
# the same output (torch fill_ : 7 TB/s of writes on these boxes), one --pmc pass per group (the program directly behind `--`).
#   bash tools/probes/pmc_s2d_r05.sh > gpurun_out/r05_pmc_s2d.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== counters offered (rocprofv3 -L, filtered)"
rocprofv3 -L 2>/dev/null | grep -o "\bTCC_EA0\?_WRREQ[A-Z0-9_]*\|\bTCP_TCC_[A-Z_]*REQ[A-Z_]*\|\bTCC_HIT[a-z_]*\|\bTCC_MISS[a-z_]*\|\bSQ_WAIT_INST_ANY\|\bSQ_INSTS_VMEM_WR\|\bSQ_INSTS_VMEM_RD\|\bSQ_WAVE_CYCLES\|\bSQ_BUSY_CYCLES\|\bTCC_EA0\?_WR_UNCACHED[A-Z0-9_]*\|\bTCC_WRITEBACK[a-z_]*\|\bTCC_EA0\?_RDREQ[A-Z0-9_]*\|\bTCP_PENDING_STALL_CYCLES[a-z_]*\|\bTCC_TAG_STALL[a-z_]*\|\bTCC_NORMAL_WRITEBACK[a-z_]*\|\bTCC_ALL_TC_OP_WB_WRITEBACK[a-z_]*\|\bTCC_REQ[a-z_]*\|\bTCC_WRITE[a-z_]*\|\bTCC_READ[a-z_]*\|\bTCP_TA_TCP_STATE_READ[a-z_]*\|\bTA_BUSY[a-z_]*\|\bTCP_TCC_NC_WRITE_REQ[a-z_]*" | sort -u | tr '\n' ' '
echo
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TCC_WRITE_sum TCC_READ_sum TCC_REQ_sum" "TCC_WRITEBACK_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/probes/pmc_s2d_r05.py >/tmp/fs.log 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A6 "s2d\|s2f\|FillFunctor\|fill\|vectorized_elementwise" | grep -v "^--"
done
