R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_s2 -- python3 $R/bench_scs.py --games 1024 > $O/pmc_s2.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_s2/*/*counter_collection.csv | head -1) "persist_kernel" > $O/pmc_persist_kernel_sq.txt
cat $O/pmc_persist_kernel_sq.txt
