#!/bin/bash
# Round-3 final evidence run (one MI355X): the default bench line (side keys included), rocprofv3 kernel stats + traces
# for both dtypes, step breakdowns, the A/B tools of the round-3 kernels, PMC passes, the bare-MFMA ceiling probe.
# Outputs under gpurun_out/r03final/ (copied into profiles/r03_* by hand: see profiles/README.md).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_f32_full.log 2>&1
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-side-runs > $O/bench_bf16_full.log 2>&1
python tools/step_breakdown.py --dtype bf16 > $O/breakdown_bf16.txt 2>&1
python tools/step_breakdown.py --dtype f32 > $O/breakdown_f32.txt 2>&1
python tools/bench_lp16h.py > $O/bench_lp16h.txt 2>&1
python tools/bench_lp16_fold.py > $O/bench_lp16_fold.txt 2>&1
python tools/ab_lp16_wgrad.py > $O/ab_lp16_wgrad.txt 2>&1
python tools/bench_conv7_n4.py > $O/bench_conv7_n4.txt 2>&1
python tools/bench_stem_wgrad.py > $O/bench_stem_wgrad.txt 2>&1
python tools/bench_stem_fprop.py > $O/bench_stem_fprop.txt 2>&1
python tools/bench_dgrad_s2.py > $O/bench_dgrad_s2.txt 2>&1
python tools/ablate_dgrad_s2.py > $O/ablate_dgrad_s2.txt 2>&1
python tools/bench_wgrad_s2.py > $O/bench_wgrad_s2.txt 2>&1
python tools/bench_stem_f32.py > $O/bench_stem_f32.txt 2>&1
CIN=44 python tools/ablate_stem_f32.py > $O/ablate_stem_f32.txt 2>&1
python tools/probes/wino_grad_split.py > $O/wino_grad_split.txt 2>&1
python tools/ab_wino_wgrad_dma.py > $O/ab_wino_wgrad_dma.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > $O/mfma_peak.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_f32_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_bf16_line.log 2>&1
cd $R
python tools/roofline_from_trace.py $O/prof_f32 "wino_gemm_kernel<128" > $O/roofline_f32_trace.txt 2>&1
python tools/roofline_from_trace.py $O/prof_bf16 conv_lp16h2_kernel > $O/roofline_bf16_trace.txt 2>&1
for f in $O/prof_f32 $O/prof_bf16; do find $f -name "*kernel_stats.csv" -exec cp {} $f.kernel_stats.csv \; ; done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  rm -rf /tmp/fs; rocprofv3 --pmc $c --output-format csv -d /tmp/fs -- python3 $R/tools/pmc_r03.py >/dev/null 2>&1
  echo "== --pmc $c"; python3 $R/tools/pmc_summary.py /tmp/fs | grep -A3 "lp16\|cvt" | grep -v "^--"
done > $O/pmc_lp16.txt 2>&1
bash $R/tools/traffic_r03_f32.sh > $O/traffic_f32.txt 2>&1
cd $R
rm -rf $O/prof_f32 $O/prof_bf16
grep -h "^{" $O/bench_f32_full.log $O/bench_bf16_full.log | cut -c1-200
