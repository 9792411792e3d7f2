R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; mkdir -p $O
( echo "== product"; timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  echo "== no solo chains"; NZ_SCS_PERSIST_NO_SOLO=1 timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  echo "== phead"; NZ_NETBENCH_PHASES=1 NZ_LIB_PATH=$R/scripts/ablate/lib_phead.so timeout -k 10 200 python3 $R/scripts/persist_netbench.py ) 2>&1 | grep -v amdgpu.ids > $O/netphases.txt
cat $O/netphases.txt
timeout -k 10 900 python3 -m pytest $R/tests/test_gpu_scs_persist.py $R/tests/test_gpu_scs_configs.py $R/tests/test_gpu_scs_pergame.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do
NZ_SCS_PERSIST_NO_SOLO=1 timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_nosolo.log 2>&1 || exit 1; echo nosolo; tail -1 $O/scs_nosolo.log | cut -c300-420
timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_prod.log 2>&1 || exit 1; echo product; tail -1 $O/scs_prod.log | cut -c300-420
done
