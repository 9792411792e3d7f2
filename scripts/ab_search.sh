# same-box A/B of library variants on the SCS configs[3] search (bench_scs.py, 1024 games): $BENCHV variants against the product
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03ab; mkdir -p $O
for i in 1 2; do
for v in $BENCHV; do NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_$v.log 2>&1 || exit 1; echo $v; tail -1 $O/scs_$v.log | cut -c300-420; done
timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_prod.log 2>&1 || exit 1; echo product; tail -1 $O/scs_prod.log | cut -c300-420
done
for v in $STAMPV; do NZ_LIB_PATH=$R/scripts/ablate/lib_$v.so timeout -k 10 200 python3 $R/bench_scs.py --games 1024 > $O/scs_$v.log 2>&1 || exit 1; echo $v; tail -1 $O/scs_$v.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['persist_kernel_ticks']; n=d['simulations_per_s']*d['seconds']; e=d['expansions_per_s']*d['seconds']
print({k:(round(v/n) if k in ('clone','descent','backup') else round(v/e)) for k,v in t.items() if k not in ('moves_total','slowest_game_move')}, t['moves_total'], t['slowest_game_move'], d['seconds'])
"; done
