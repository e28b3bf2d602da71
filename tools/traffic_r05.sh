#!/bin/bash
# Round 5: HBM-side traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes; MI355X_MICROARCH.md) of the headline roofline
# kernel wino_gemm_kernel<128,2>, of the 16-bit kernels of the same conv, and - new - of the stride-2 16-bit dgrad
# conv_s2d_kernel (an `s2d` entry).  usage (GPU box): bash tools/traffic_r05.sh <out.json>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export PMC_WHICH=f32
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -- python3 $R/tools/pmc_r02.py >/dev/null 2>&1
done
python3 $R/tools/traffic_r04.py /tmp/tr_FETCH_SIZE /tmp/tr_WRITE_SIZE | sed 's/traffic_r04.sh/traffic_r05.sh/' > ${1:-$R/gpurun_out/r05_traffic.json}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/trb_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/trb_$c -- python3 $R/tools/pmc_r03.py >/dev/null 2>&1
  rm -rf /tmp/trs_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/trs_$c -- python3 $R/tools/probes/pmc_s2d_r05.py >/dev/null 2>&1
done
O1=${1:-$R/gpurun_out/r05_traffic.json}
python3 $R/tools/traffic_r05_bf16.py /tmp/trb_FETCH_SIZE /tmp/trb_WRITE_SIZE /tmp/trs_FETCH_SIZE /tmp/trs_WRITE_SIZE > ${O1%.json}_bf16.json
