set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; mkdir -p $O
( echo "== paired (product)"; timeout -k 10 120 python3 $R/scripts/persist_netbench.py
  echo "== unpaired"; NZ_LIB_PATH=$R/scripts/ablate/lib_unpaired.so timeout -k 10 120 python3 $R/scripts/persist_netbench.py ) 2>&1 | grep -v amdgpu.ids > $O/netbench_paired.txt
cat $O/netbench_paired.txt
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_scs_persist.py $R/tests/test_gpu_scs_configs.py -x -q > $O/tests_paired.log 2>&1 || { tail -20 $O/tests_paired.log; exit 1; }
tail -2 $O/tests_paired.log
for i in 1 2; do
NZ_LIB_PATH=$R/scripts/ablate/lib_unpaired.so timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_unpaired_$i.log 2>&1 || exit 1
timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_paired_$i.log 2>&1 || exit 1
done
for f in $O/scs_*.log; do echo $f; tail -1 $f | cut -c1-300; done
