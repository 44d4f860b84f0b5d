for L in sdf_amd/libsdfhip*.so; do
  case $L in *stats*) continue;; esac
  echo "== $L"
  SDFHIP_LIB=$PWD/$L timeout -k 10 120 python tools/perf_probe.py --n 64000000 ${AB_ARGS:---only nn} 2>&1 | grep -E "^(nn|sdf|contains)"
done
