#!/bin/bash
# Round-2 final evidence run (one MI355X): bench lines, rocprofv3 kernel stats + traces for both dtypes,
# side configurations, PMC passes for the 16-bit kernels.  Outputs under gpurun_out/r02final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02final; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_f32_full.log 2>&1
python bench.py --dtype bf16 --steps 20 --warmup 5 > $O/bench_bf16_full.log 2>&1
python bench.py --norm batch --steps 10 --warmup 3 --no-cpu-baseline --no-side-runs > $O/bench_f32_bn.log 2>&1
python bench.py --norm batch --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-side-runs > $O/bench_bf16_bn.log 2>&1
python bench.py --mode infer > $O/infer_f32.log 2>&1
python bench.py --mode infer --dtype bf16 > $O/infer_bf16.log 2>&1
python bench.py --size 512 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-side-runs > $O/bench_f32_512.log 2>&1
python bench.py --size 512 --batch 4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-side-runs > $O/bench_bf16_512.log 2>&1
python tools/step_breakdown.py --dtype bf16 > $O/breakdown_bf16.txt 2>&1
python tools/step_breakdown.py --dtype f32 > $O/breakdown_f32.txt 2>&1
python tools/bench_pointwise.py --lp 1 > $O/pointwise_bf16.txt 2>&1
python tools/ab_lp16_pipe.py > $O/ab_lp16_kernels.txt 2>&1
python tools/ab_lp16_wgrad.py > $O/ab_lp16_wgrad.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_f32_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_bf16_line.log 2>&1
cd $R
python tools/roofline_from_trace.py $O/prof_f32 "wino_gemm_kernel<128" > $O/roofline_f32_trace.txt 2>&1
python tools/roofline_from_trace.py $O/prof_bf16 conv_lp16h_kernel > $O/roofline_bf16_trace.txt 2>&1
for f in $O/prof_f32 $O/prof_bf16; do find $f -name "*kernel_stats.csv" -exec cp {} $f.kernel_stats.csv \; ; done
PMC_WHICH=bf16 bash tools/traffic_r02.sh > $O/pmc_bf16.txt 2>&1
# keep the merged output small: drop the raw traces
rm -rf $O/prof_f32 $O/prof_bf16
grep -h "^{" $O/bench_f32_full.log $O/bench_bf16_full.log | cut -c1-200
