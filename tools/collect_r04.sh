#!/bin/bash
# copy the outputs of tools/profile_r04.sh (gpurun_out/r04prof/, merged back from the GPU box) into profiles/r04_*
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r04prof; P=$R/profiles
grep -h "^{" $O/prof_f32_line.log > $P/r04_bench_f32_line.json
grep -h "^{" $O/prof_bf16_line.log > $P/r04_bench_bf16_line.json
grep -h "^{" $O/infer_bf16_line.log > $P/r04_infer_b64_bf16_line.json
cp $O/prof_f32.kernel_stats.csv $P/r04_bench_f32_kernel_stats.csv
cp $O/prof_bf16.kernel_stats.csv $P/r04_bench_bf16_kernel_stats.csv
cp $O/prof_infer.kernel_stats.csv $P/r04_infer_b64_bf16_kernel_stats.csv
cp $O/prof_512.kernel_stats.csv $P/r04_size512_bf16_b4_kernel_stats.csv
cp $O/roofline_f32_trace.txt $P/r04_roofline_winograd_kernel_trace.txt
cp $O/roofline_bf16_trace.txt $P/r04_roofline_lp16_kernel_trace.txt
for f in breakdown_f32 breakdown_bf16 host_overhead; do grep -v "amdgpu.ids\|^WARNING" $O/$f.txt > $P/r04_${f/breakdown/step_breakdown}.txt; done
grep -v "amdgpu.ids\|^WARNING" $O/breakdown_bf16_512.txt > $P/r04_step_breakdown_bf16_512.txt
cp $O/r04_traffic.json $P/r04_traffic.json
cp $O/r04_traffic_bf16.json $P/r04_traffic_bf16.json
grep -v "amdgpu.ids\|^WARNING" $O/run_512.log > $P/r04_size512_bf16_b4_run.txt
grep -h "^{" $O/full_f32_line.log > $P/r04_bench_f32_full_line.json
grep -h "^{" $O/full_bf16_line.log > $P/r04_bench_bf16_full_line.json
python3 - "$O/cpu_all_cores_line.log" > $P/r04_cpu_all_cores.json <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
j = json.loads(l)
print(json.dumps({"command": "python bench.py --steps 2 --warmup 1 --no-side-runs --cpu-all-cores 200", "cpu_baseline": j.get("cpu_baseline")}, indent=1))
PY
