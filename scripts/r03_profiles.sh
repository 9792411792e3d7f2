# Round-3 measurements on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes for the
# headline kernel and for the persistent SCS kernel, the network micro-benchmark with its timing-only builds.
# Every GPU step runs under its own timeout; steps are joined with || exit.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
C=${NZ_COMMIT:-unknown}
export NZ_COMMIT=$C
timeout -k 10 700 python3 $R/bench.py > $O/bench_stdout.log 2>$O/bench_stderr.log || { tail -20 $O/bench_stderr.log; exit 1; }
tail -c 300 $O/bench_stdout.log
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --no-cpu-baseline --no-live-traffic > $O/bench_rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_w.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $(ls /tmp/p_f/*/*counter_collection.csv | head -1) $(ls /tmp/p_w/*/*counter_collection.csv | head -1) "selfplay_kernel<false>" $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline" $C 4096 65536 100 2 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_s1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_s1.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_s1/*/*counter_collection.csv | head -1) "selfplay_kernel<false>" > $O/pmc_selfplay_kernel_sq.txt
cat $O/pmc_selfplay_kernel_sq.txt
# SCS configs[3] on the persistent route: kernel stats of one round, PMC of the persistent kernel
timeout -k 10 300 python3 $R/bench_scs.py --games 1024 > $O/scs_1024.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_scs -- python3 $R/bench_scs.py --games 1024 > $O/scs_rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_scs/*/*kernel_stats.csv | head -1) $O/scs_1024_kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_s2 -- python3 $R/bench_scs.py --games 1024 > $O/pmc_s2.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_s2/*/*counter_collection.csv | head -1) "persist_kernel" > $O/pmc_persist_kernel_sq.txt
cat $O/pmc_persist_kernel_sq.txt
timeout -k 10 300 python3 $R/bench_scs.py --games 1024 --round-games 4096 > $O/scs_round4.log 2>&1 || exit 1
timeout -k 10 300 python3 $R/bench_scs.py --games 1024 --cache 1048576 > $O/scs_cache.log 2>&1 || exit 1
# the network of the persistent route alone: its phases (stamped build), without the heads' side-by-side chains, and its
# timing-only builds
( echo "== product build"; timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  echo "== product build, heads layer by layer (NZ_SCS_PERSIST_NO_SOLO=1)"; NZ_SCS_PERSIST_NO_SOLO=1 timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  echo "== phases of a pass (-DNZ_PERSIST_STAMPS -DNZ_PERSIST_HEADSTAMP; the stamps cost ~10 %)"; NZ_NETBENCH_PHASES=1 NZ_LIB_PATH=$R/scripts/ablate/lib_phead.so timeout -k 10 200 python3 $R/scripts/persist_netbench.py
  for v in pnoa pnob pnoab pnoepi pnone; do echo "== $v"; NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 120 python3 $R/scripts/persist_netbench.py; done ) 2>&1 | grep -v amdgpu.ids > $O/persist_netbench.txt
cat $O/persist_netbench.txt
NZ_LIB_PATH=$R/scripts/ablate/lib_pstamps.so timeout -k 10 300 python3 $R/bench_scs.py --games 1024 > $O/scs_stamps.log 2>&1
tail -1 $O/scs_1024.log | cut -c1-600
# configs[4] (10x10, RecurrentNet 256 x 2 x 16 iterations): kernel stats of one decision of 1024 games, conv_wide_kernel's
# K-step stamps, and the round-2 form of the kernel on the same box
timeout -k 10 300 python3 $R/scripts/cfg5_standalone.py > $O/cfg5_plain.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python3 $R/scripts/cfg5_standalone.py > $O/cfg5_rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_c5/*/*kernel_stats.csv | head -1) $O/cfg5_kernel_stats.csv
( echo "== product build (scripts/cfg5_standalone.py: 1024 games x their first decision x 400 simulations)"; tail -1 $O/cfg5_plain.log
  echo "== K-step stamps (-DNZ_WIDE_STAMPS; ticks per step of workgroup 0's wavefront 0: a corner cell)"; NZ_LIB_PATH=$R/scripts/ablate/lib_wstamps.so timeout -k 10 300 python3 $R/scripts/cfg5_standalone.py 2>&1 | tail -1
  echo "== round-2 form of conv_wide_kernel (-DNZ_WIDE_OVERLAP=0 -DNZ_WIDE_XCD=0 -DNZ_WIDE_KQ_OUTER=0)"; NZ_LIB_PATH=$R/scripts/ablate/lib_wnoover.so timeout -k 10 300 python3 $R/scripts/cfg5_standalone.py 2>&1 | tail -1 ) > $O/cfg5_conv_wide.txt
cat $O/cfg5_conv_wide.txt
