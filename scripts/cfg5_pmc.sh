R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cfg5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export NZ_CFG5_SIMS=40
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/p_c5p -- python3 $R/scripts/cfg5_standalone.py > $O/pmc.log 2>&1 || { grep -v "^    @" $O/pmc.log | tail -5; exit 1; }
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_c5p/*/*counter_collection.csv | head -1) "conv_wide_kernel" > $O/pmc_conv_wide.txt
cat $O/pmc_conv_wide.txt
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/p_c5q -- python3 $R/scripts/cfg5_standalone.py > $O/pmc2.log 2>&1 || { grep -v "^    @" $O/pmc2.log | tail -5; exit 1; }
python3 $R/scripts/pmc_summary.py $(ls /tmp/p_c5q/*/*counter_collection.csv | head -1) "conv_wide_kernel" > $O/pmc_conv_wide2.txt
cat $O/pmc_conv_wide2.txt
