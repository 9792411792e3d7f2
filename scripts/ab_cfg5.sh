# same-box A/B of library variants on bench.py's scs_config5 (conv_wide_kernel is 96 % of it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cfg5; mkdir -p $O
for v in $BENCHV; do NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 300 python3 $R/scripts/cfg5_profile.py > $O/$v.log 2>&1 || { tail $O/$v.log; exit 1; }; echo $v; tail -1 $O/$v.log | cut -c1-120; done
timeout -k 10 300 python3 $R/scripts/cfg5_profile.py > $O/prod.log 2>&1 || { tail $O/prod.log; exit 1; }; echo product; tail -1 $O/prod.log | cut -c1-120
