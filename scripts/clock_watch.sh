# Sample GPU clock and power while the bench runs (diagnostic): writes gpurun_out/clock_watch.log
O=$GRAFT_REPO_ROOT/gpurun_out/clock_watch.log
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $O 2>&1 &
W=$!
timeout -k 10 200 python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-200
wait $W
