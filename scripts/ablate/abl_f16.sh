# Timing-only builds of the one-launch BF16 network (outputs are wrong): where its time goes.
# Build first (in the container):
#   python scripts/build_ablations.py fstamps=-DNZ_FUSED_STAMPS nofetch=-DNZ_ABL_F16_NOFETCH nok=-DNZ_ABL_F16_NOK noact=-DNZ_ABL_F16_NOACT
# then run on the GPU box:  bash scripts/ablate/abl_f16.sh
NZ_LIB_PATH=scripts/ablate/lib_fstamps.so timeout -k 10 120 python scripts/fused_stamps.py 2>&1 | grep -A12 "second launch" || exit 1
echo "== base"; FUSED_ONLY=1 timeout -k 10 120 python scripts/fused_only.py 200 2>&1 | grep "fused \(64\|1024\)" || exit 1
for v in nofetch nok noact; do
  export NZ_LIB_PATH=scripts/ablate/lib_$v.so
  echo "== $v (no next-layer weight staging / no K loops / no activation functions)"
  FUSED_ONLY=1 timeout -k 10 120 python scripts/fused_only.py 200 2>&1 | grep "fused \(64\|1024\)" || exit 1
done
