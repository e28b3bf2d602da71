cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_bf16_kernel_stats.csv \;
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bf16
