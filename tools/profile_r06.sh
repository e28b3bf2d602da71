#!/bin/bash
# Round-6 evidence run (one MI355X).  Outputs under gpurun_out/r6prof/ (copied into profiles/r06_* by tools/collect_r06.sh).
# Every stage is CHECKED: a stage that fails or leaves an empty result ends the script with a non-zero exit code and names the
# stage (round 5's collector committed an empty counter file).
#   A: hardware counters (MFMA-busy, instruction mix, waits) - tools/pmc_r06.sh; HBM traffic - the round-5 passes, checked
#   B: per-entry-point breakdowns (bf16 256^2, bf16 512^2 B=4, fp32), host enqueue table, graph step against eager
#   C: kernel stats + traces of the fp32 headline and the bf16 step, roofline from the trace
#   D: the driver's own command (every side key on one line) and the bf16 headline region
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r6prof; mkdir -p $O
fail=0
need() {      # need <file> <min bytes> <stage>
  if [ ! -s "$1" ] || [ "$(wc -c < "$1")" -lt "$2" ]; then echo "profile_r06.sh: stage [$3] left no usable $1" >&2; fail=1; fi
}
cd $R
# ---- A
bash $R/tools/pmc_r06.sh > $O/pmc.log 2>&1 || { echo "profile_r06.sh: stage [pmc] failed" >&2; cat $O/pmc.log >&2; fail=1; }
bash $R/tools/traffic_r05.sh $O/r06_traffic.json > $O/traffic.log 2>&1
python3 - "$O/r06_traffic.json" "$O/r06_traffic_bf16.json" <<'PY' || fail=1
import json, sys
ok = True
for f in sys.argv[1:]:
    try:
        j = json.load(open(f))
    except Exception as e:
        print(f"profile_r06.sh: stage [traffic]: {f}: {e}", file=sys.stderr); ok = False; continue
    vals = [v.get("hbm_bytes_per_launch") for v in j.values() if isinstance(v, dict) and "hbm_bytes_per_launch" in v]
    if not vals or any(not v for v in vals):
        print(f"profile_r06.sh: stage [traffic]: {f} holds no counter traffic: {vals}", file=sys.stderr); ok = False
sys.exit(0 if ok else 1)
PY
cp $O/r06_traffic.json $O/r06_traffic_bf16.json $R/profiles/ 2>/dev/null
# ---- B
python tools/host_overhead.py > $O/host_overhead.txt 2>&1 || fail=1
python tools/step_breakdown.py --dtype bf16 --size 512 --batch 4 --top 60 > $O/breakdown_bf16_512.txt 2>&1 || fail=1
python tools/step_breakdown.py --dtype bf16 > $O/breakdown_bf16.txt 2>&1 || fail=1
python tools/step_breakdown.py --dtype f32 > $O/breakdown_f32.txt 2>&1 || fail=1
python tools/probes/graph_step_bench.py bf16 512 f32 > $O/graph_step.txt 2>&1 || fail=1
for f in host_overhead breakdown_bf16_512 breakdown_bf16 breakdown_f32 graph_step; do need $O/$f.txt 400 $f; done
# ---- C
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python3 $R/bench.py --mode infer --dtype bf16 --steps 10 --warmup 3 > $O/infer_bf16_line.log 2>&1 || fail=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_512 -- python3 $R/tools/run_512.py > $O/run_512.log 2>&1 || fail=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_f32_line.log 2>&1 || fail=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-side-runs > $O/prof_bf16_line.log 2>&1 || fail=1
cd $R
python tools/roofline_from_trace.py $O/prof_f32 "wino_gemm_kernel<128, 2>" > $O/roofline_f32_trace.txt 2>&1 || fail=1
python tools/roofline_from_trace.py $O/prof_bf16 "conv_lp16h2_kernel<false, 1, false>" > $O/roofline_bf16_trace.txt 2>&1 || fail=1
for f in prof_infer prof_512 prof_f32 prof_bf16; do
  find $O/$f -name "*kernel_stats.csv" -exec cp {} $O/$f.kernel_stats.csv \; ; rm -rf $O/$f
  need $O/$f.kernel_stats.csv 1000 $f
done
need $O/roofline_f32_trace.txt 200 roofline_f32; need $O/roofline_bf16_trace.txt 200 roofline_bf16
# ---- D
python bench.py > $O/full_f32_line.log 2>&1 || fail=1
python bench.py --dtype bf16 --no-side-runs > $O/full_bf16_line.log 2>&1 || fail=1
grep -q '^{' $O/full_f32_line.log || { echo "profile_r06.sh: stage [full line] printed no JSON line" >&2; fail=1; }
ls -la $O
[ $fail -eq 0 ] && echo "profile_r06.sh: ok" || { echo "profile_r06.sh: INCOMPLETE (see above)" >&2; exit 1; }
