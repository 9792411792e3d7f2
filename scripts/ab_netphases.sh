# network pass of the persistent route: phases of the pass (stamped builds) and variants, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; mkdir -p $O
( echo "== product"; timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  for v in $NETV; do echo "== $v"; NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 120 python3 $R/scripts/persist_netbench.py || exit 1; done
  for v in $PHASEV; do echo "== $v"; NZ_NETBENCH_PHASES=1 NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 200 python3 $R/scripts/persist_netbench.py || exit 1; done
) 2>&1 | grep -v amdgpu.ids > $O/netphases.txt
cat $O/netphases.txt
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_scs_persist.py $R/tests/test_gpu_scs_configs.py -x -q > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
for v in $BENCHV; do NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_$v.log 2>&1 || exit 1; echo $v; tail -1 $O/scs_$v.log | cut -c300-420; done
timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_prod.log 2>&1 || exit 1; echo product; tail -1 $O/scs_prod.log | cut -c300-420
done
