#!/bin/bash
# Copy the judged summaries of tools/profile_r02.sh (gpurun_out/r02final/) into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r02final; P=profiles
line() { grep -h "^{" "$1" | tail -1; }
line $O/bench_f32_full.log > $P/r02_bench_f32_full_line.json
line $O/bench_bf16_full.log > $P/r02_bench_bf16_full_line.json
line $O/prof_f32_line.log > $P/r02_bench_f32_line.json
line $O/prof_bf16_line.log > $P/r02_bench_bf16_line.json
cp $O/prof_f32.kernel_stats.csv $P/r02_bench_f32_kernel_stats.csv
cp $O/prof_bf16.kernel_stats.csv $P/r02_bench_bf16_kernel_stats.csv
cp $O/roofline_f32_trace.txt $P/r02_roofline_winograd_kernel_trace.txt
cp $O/roofline_bf16_trace.txt $P/r02_roofline_lp16_kernel_trace.txt
cp $O/breakdown_f32.txt $P/r02_step_breakdown_f32.txt
cp $O/breakdown_bf16.txt $P/r02_step_breakdown_bf16.txt
cp $O/pointwise_bf16.txt $P/r02_pointwise_bf16.txt
cp $O/ab_lp16_kernels.txt $P/r02_ab_lp16_kernels.txt
cp $O/ab_lp16_wgrad.txt $P/r02_ab_lp16_wgrad.txt
cp $O/pmc_bf16.txt $P/r02_pmc_lp16h.txt
: > $P/r02_side_configs.jsonl
for n in bench_f32_bn bench_bf16_bn infer_f32 infer_bf16 bench_f32_512 bench_bf16_512; do
  echo "# $n" >> $P/r02_side_configs.jsonl; line $O/$n.log >> $P/r02_side_configs.jsonl
done
grep -h "^{" $P/r02_bench_f32_full_line.json $P/r02_bench_bf16_full_line.json | cut -c1-160
python - <<'PY'
import json
for n in ("bench_f32_bn", "bench_bf16_bn", "infer_f32", "infer_bf16", "bench_f32_512", "bench_bf16_512"):
    pass
rows = [l for l in open("profiles/r02_side_configs.jsonl") if l.startswith("{")]
for l in rows:
    d = json.loads(l); print(d["config"].get("workload", "")[:60], d["dtype"], d["value"], d["ms_per_step"])
PY
