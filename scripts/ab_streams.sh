R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
for i in 1 2; do
for s in 1 2 4; do
timeout -k 10 200 python3 $R/bench_scs.py --games 1024 --streams $s > $O/scs_streams$s.log 2>&1 || exit 1; echo streams $s; tail -1 $O/scs_streams$s.log | cut -c300-420
done
done
