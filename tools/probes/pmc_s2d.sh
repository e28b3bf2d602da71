cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/probes/pmc_s2d.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A3 "s2d" | grep -v "^--"
done
