#!/bin/bash
# Round-5 evidence run (one MI355X).  Outputs under gpurun_out/r05prof/ (copied into profiles/r05_* by tools/collect_r05.sh).
#   part A: configs[3] and configs[4] - kernel stats of the hipGraph inference run, per-entry-point breakdown
#                           of the 512x512 B=4 bf16 step, the host-enqueue table
#   part B               : the headline fp32 step and the bf16 step - kernel stats, traces, roofline from the trace, traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05prof; mkdir -p $O
cd $R
# counters first: the bench lines below read profiles/r05_traffic*.json for roofline.traffic
bash $R/tools/traffic_r05.sh $O/r05_traffic.json > $O/traffic.log 2>&1
cp $O/r05_traffic.json $O/r05_traffic_bf16.json $R/profiles/ 2>/dev/null
cd $R
python tools/host_overhead.py > $O/host_overhead.txt 2>&1
python tools/step_breakdown.py --dtype bf16 --size 512 --batch 4 --top 60 > $O/breakdown_bf16_512.txt 2>&1
python tools/step_breakdown.py --dtype bf16 > $O/breakdown_bf16.txt 2>&1
python tools/step_breakdown.py --dtype f32 > $O/breakdown_f32.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python3 $R/bench.py --mode infer --dtype bf16 --steps 10 --warmup 3 > $O/infer_bf16_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_512 -- python3 $R/tools/run_512.py > $O/run_512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_f32_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_bf16_line.log 2>&1
cd $R
python tools/roofline_from_trace.py $O/prof_f32 "wino_gemm_kernel<128, 2>" > $O/roofline_f32_trace.txt 2>&1
python tools/roofline_from_trace.py $O/prof_bf16 "conv_lp16h2_kernel<false, 1, false>" > $O/roofline_bf16_trace.txt 2>&1
for f in prof_infer prof_512 prof_f32 prof_bf16; do find $O/$f -name "*kernel_stats.csv" -exec cp {} $O/$f.kernel_stats.csv \; ; rm -rf $O/$f; done
# part C: the driver's own command (every side key on one line), and the same with the 16-bit step as the headline region
cd $R
python bench.py > $O/full_f32_line.log 2>&1
python bench.py --dtype bf16 --no-side-runs > $O/full_bf16_line.log 2>&1
# MFMA-busy of the two forms of the halo kernel (A/B build: two waves per SIMD, one wave per SIMD) and of the new stride-2 dgrad
bash $R/tools/pmc_r05.sh > $O/pmc_lp16.txt 2>&1
ls -la $O
