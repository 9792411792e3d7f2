# configs[4] (scripts/cfg5_standalone.py) against the number of concurrent games: conv_wide_kernel's tiles are
# (cell, 128 channels, 256 positions), 200 workgroups per 256 games on 256 CUs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
for g in 256 512 768 1024 1280; do
NZ_CFG5_GAMES=$g NZ_CFG5_SIMS=${SIMS:-100} timeout -k 10 200 python3 $R/scripts/cfg5_standalone.py > $O/cfg5_games$g.log 2>&1 || { tail -5 $O/cfg5_games$g.log; exit 1; }; echo games $g; tail -1 $O/cfg5_games$g.log
done
