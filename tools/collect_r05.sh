#!/bin/bash
# copy the outputs of tools/profile_r05.sh (gpurun_out/r05prof/, merged back from the GPU box) into profiles/r05_*
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r05prof; P=$R/profiles
grep -h "^{" $O/prof_f32_line.log > $P/r05_bench_f32_line.json
grep -h "^{" $O/prof_bf16_line.log > $P/r05_bench_bf16_line.json
grep -h "^{" $O/infer_bf16_line.log > $P/r05_infer_b64_bf16_line.json
cp $O/prof_f32.kernel_stats.csv $P/r05_bench_f32_kernel_stats.csv
cp $O/prof_bf16.kernel_stats.csv $P/r05_bench_bf16_kernel_stats.csv
cp $O/prof_infer.kernel_stats.csv $P/r05_infer_b64_bf16_kernel_stats.csv
cp $O/prof_512.kernel_stats.csv $P/r05_size512_bf16_b4_kernel_stats.csv
cp $O/roofline_f32_trace.txt $P/r05_roofline_winograd_kernel_trace.txt
cp $O/roofline_bf16_trace.txt $P/r05_roofline_lp16_kernel_trace.txt
for f in breakdown_f32 breakdown_bf16 host_overhead; do grep -v "amdgpu.ids\|^WARNING" $O/$f.txt > $P/r05_${f/breakdown/step_breakdown}.txt; done
grep -v "amdgpu.ids\|^WARNING" $O/breakdown_bf16_512.txt > $P/r05_step_breakdown_bf16_512.txt
cp $O/r05_traffic.json $P/r05_traffic.json
cp $O/r05_traffic_bf16.json $P/r05_traffic_bf16.json
grep -v "amdgpu.ids\|^WARNING" $O/run_512.log > $P/r05_size512_bf16_b4_run.txt
grep -h "^{" $O/full_f32_line.log > $P/r05_bench_f32_full_line.json
grep -h "^{" $O/full_bf16_line.log > $P/r05_bench_bf16_full_line.json
grep -v "amdgpu.ids\|^WARNING" $O/pmc_lp16.txt > $P/r05_pmc_lp16.txt
