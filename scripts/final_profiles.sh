set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench_stdout.log 2>$O/bench_stderr.log || exit 1
tail -c 600 $O/bench_stdout.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- python3 $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_w.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $(ls /tmp/p_f/*/*counter_collection.csv | head -1) $(ls /tmp/p_w/*/*counter_collection.csv | head -1) "selfplay_kernel<false>" $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline" || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_n1 -- python3 $R/scripts/net_only.py > $O/pmc_n1.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_n1/*/*counter_collection.csv | head -1) "net_kernel" > $O/pmc_net_kernel_sq.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d /tmp/p_n2 -- python3 $R/scripts/net_only.py > $O/pmc_n2.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_n2/*/*counter_collection.csv | head -1) "net_kernel" > $O/pmc_net_kernel_lds.txt
cat $O/pmc_net_kernel_sq.txt $O/pmc_net_kernel_lds.txt
timeout -k 10 400 python3 $R/bench_scs.py --games 1024 > $O/scs_1024.log 2>&1 || exit 1
timeout -k 10 400 python3 $R/bench_scs.py --games 8192 --nodes-per-sim 1536 > $O/scs_8192.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_scs -- python3 $R/bench_scs.py --games 1024 > $O/scs_rocprof.log 2>&1 || exit 1
cp $(ls /tmp/p_scs/*/*kernel_stats.csv | head -1) $O/scs_1024_kernel_stats.csv
tail -1 $O/scs_1024.log; tail -1 $O/scs_8192.log
