R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cfg5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/scripts/cfg5_profile.py > $O/plain.log 2>&1 || { tail $O/plain.log; exit 1; }
tail -1 $O/plain.log | cut -c1-300
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $R/scripts/cfg5_profile.py > $O/rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_c5/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
head -14 $O/kernel_stats.csv | cut -c1-220
