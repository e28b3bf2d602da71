O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
cut -c1-600 $O/bench.json
timeout 900 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; cat $O/host_overhead.txt
bash tools/traffic_r04.sh $GRAFT_REPO_ROOT/$O/r04_traffic.json; cat $O/r04_traffic.json | head -20
