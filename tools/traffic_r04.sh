#!/bin/bash
# Round 4: HBM-side traffic of the headline roofline kernel, wino_gemm_kernel<128,2> on the 512->512 @64x64 B=32 fprop:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md), summarised into profiles-format JSON by
# tools/traffic_r04.py.  usage (on the GPU box): bash tools/traffic_r04.sh <out.json>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export PMC_WHICH=f32
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -- python3 $R/tools/pmc_r02.py >/dev/null 2>&1
done
python3 $R/tools/traffic_r04.py /tmp/tr_FETCH_SIZE /tmp/tr_WRITE_SIZE > ${1:-$R/gpurun_out/r04_traffic.json}
# the 16-bit kernels of the same conv (bench.py --dtype bf16 / bf16_path): same two passes over tools/pmc_r03.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/trb_$c; rocprofv3 --pmc $c --output-format csv -d /tmp/trb_$c -- python3 $R/tools/pmc_r03.py >/dev/null 2>&1
done
O1=${1:-$R/gpurun_out/r04_traffic.json}
python3 $R/tools/traffic_r04_bf16.py /tmp/trb_FETCH_SIZE /tmp/trb_WRITE_SIZE > ${O1%.json}_bf16.json
