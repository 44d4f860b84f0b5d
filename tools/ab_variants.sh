for o in "grid_per_face=16" "grid_per_face=40"; do
  echo "== $o"
  SDFHIP_DEBUG=1 timeout -k 10 100 python tools/perf_probe.py --mesh ico7 --n 64000000 --only contains --ray-mode 1 --opt $o 2>&1 | grep -E "^contains|column grid" | tail -2
done
