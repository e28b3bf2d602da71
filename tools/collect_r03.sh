#!/bin/bash
# copy the outputs of tools/profile_r03.sh (gpurun_out/r03final/, merged back from the GPU box) into profiles/r03_*
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r03final; P=$R/profiles
grep -h "^{" $O/bench_f32_full.log > $P/r03_bench_f32_full_line.json
grep -h "^{" $O/bench_bf16_full.log > $P/r03_bench_bf16_full_line.json
grep -h "^{" $O/prof_f32_line.log > $P/r03_bench_f32_line.json
grep -h "^{" $O/prof_bf16_line.log > $P/r03_bench_bf16_line.json
cp $O/prof_f32.kernel_stats.csv $P/r03_bench_f32_kernel_stats.csv
cp $O/prof_bf16.kernel_stats.csv $P/r03_bench_bf16_kernel_stats.csv
cp $O/roofline_f32_trace.txt $P/r03_roofline_winograd_kernel_trace.txt
cp $O/roofline_bf16_trace.txt $P/r03_roofline_lp16_kernel_trace.txt
cp $O/breakdown_f32.txt $P/r03_step_breakdown_f32.txt
cp $O/breakdown_bf16.txt $P/r03_step_breakdown_bf16.txt
for f in bench_lp16h bench_conv7_n4 bench_stem_wgrad bench_stem_fprop ab_lp16_wgrad mfma_peak pmc_lp16 bench_dgrad_s2 ablate_dgrad_s2 bench_wgrad_s2 bench_stem_f32 ablate_stem_f32 wino_grad_split ab_wino_wgrad_dma; do grep -v amdgpu.ids $O/$f.txt > $P/r03_$f.txt; done
grep -v amdgpu.ids $O/bench_lp16_fold.txt > $P/r03_lp16_fold.txt

# r03_traffic_f32.json is written by hand from this listing (FETCH_SIZE doubled, KB -> MB; the MFMA-busy pass is appended to the listing)
[ -f $O/traffic_f32.txt ] && cp $O/traffic_f32.txt $P/r03_traffic_f32_listing.txt || true
