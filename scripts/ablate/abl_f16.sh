NZ_LIB_PATH=scripts/ablate/lib_fstamps.so timeout -k 10 120 python scripts/fused_stamps.py 2>&1 | grep -A12 "second launch" || exit 1
for v in nofetch nok noact; do
  export NZ_LIB_PATH=scripts/ablate/lib_$v.so
  echo "== $v"; FUSED_ONLY=1 timeout -k 10 120 python scripts/fused_only.py 200 2>&1 | grep "fused \(64\|1024\)" || exit 1
done
