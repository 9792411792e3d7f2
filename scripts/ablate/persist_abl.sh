mkdir -p gpurun_out/r03
NZ_SCS_MOVE_TIMES=1 python bench_scs.py > gpurun_out/r03/scs_moves.log 2>&1
grep "^move" gpurun_out/r03/scs_moves.log | awk 'NR%5==1'
for v in pnob pnoa pnoab a3; do
  NZ_LIB_PATH=scripts/ablate/lib_$v.so python bench_scs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['games_per_s'],1), d['seconds'])"
done
