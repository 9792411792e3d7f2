for v in base nofetch nok noact nokact; do
  if [ $v = base ]; then unset NZ_LIB_PATH; else export NZ_LIB_PATH=scripts/ablate/lib_$v.so; fi
  echo "== $v"; FUSED_ONLY=1 timeout -k 10 120 python scripts/fused_only.py 200 2>&1 | grep "fused \(64\|256\|1024\)" || exit 1
done
