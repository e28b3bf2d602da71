#!/bin/bash
# copy the outputs of tools/profile_r06.sh (gpurun_out/r6prof/, merged back from the GPU box) into profiles/r06_*.
# Refuses to write an EMPTY evidence file: every source must exist and hold data, else exit 1 and nothing is overwritten.
set -u
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r6prof; P=$R/profiles
fail=0
put() {       # put <source> <destination> [grep pattern to keep | - for a filter of noise lines]
  local src=$1 dst=$2 mode=${3:-copy} tmp=$(mktemp)
  if [ ! -s "$src" ]; then echo "collect_r06.sh: missing or empty $src" >&2; fail=1; rm -f $tmp; return; fi
  case $mode in
    copy) cp "$src" $tmp ;;
    json) grep -h "^{" "$src" > $tmp ;;
    text) grep -v "amdgpu.ids\|^WARNING" "$src" > $tmp ;;
  esac
  if [ "$(wc -c < $tmp)" -lt 100 ]; then echo "collect_r06.sh: $src holds nothing to commit as $dst" >&2; fail=1; rm -f $tmp; return; fi
  mv $tmp "$dst"
}
put $O/prof_f32_line.log $P/r06_bench_f32_line.json json
put $O/prof_bf16_line.log $P/r06_bench_bf16_line.json json
put $O/infer_bf16_line.log $P/r06_infer_b64_bf16_line.json json
put $O/prof_f32.kernel_stats.csv $P/r06_bench_f32_kernel_stats.csv
put $O/prof_bf16.kernel_stats.csv $P/r06_bench_bf16_kernel_stats.csv
put $O/prof_infer.kernel_stats.csv $P/r06_infer_b64_bf16_kernel_stats.csv
put $O/prof_512.kernel_stats.csv $P/r06_size512_bf16_b4_kernel_stats.csv
put $O/roofline_f32_trace.txt $P/r06_roofline_winograd_kernel_trace.txt
put $O/roofline_bf16_trace.txt $P/r06_roofline_lp16_kernel_trace.txt
put $O/breakdown_f32.txt $P/r06_step_breakdown_f32.txt text
put $O/breakdown_bf16.txt $P/r06_step_breakdown_bf16.txt text
put $O/breakdown_bf16_512.txt $P/r06_step_breakdown_bf16_512.txt text
put $O/host_overhead.txt $P/r06_host_overhead.txt text
put $O/graph_step.txt $P/r06_graph_step_vs_eager.txt text
put $O/r06_traffic.json $P/r06_traffic.json
put $O/r06_traffic_bf16.json $P/r06_traffic_bf16.json
put $O/run_512.log $P/r06_size512_bf16_b4_run.txt text
put $O/full_f32_line.log $P/r06_bench_f32_full_line.json json
put $O/full_bf16_line.log $P/r06_bench_bf16_full_line.json json
put $O/pmc_lp16.txt $P/r06_pmc_lp16.txt text
[ $fail -eq 0 ] && echo "collect_r06.sh: ok" || { echo "collect_r06.sh: INCOMPLETE" >&2; exit 1; }
