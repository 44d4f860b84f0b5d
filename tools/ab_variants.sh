for hb in 500 1000 2000 4000 6000; do
  echo "== heavy budget $hb"
  timeout -k 10 100 python tools/perf_probe.py --n 4000000 --only nn --heavy-budget $hb --reps 5 2>&1 | tail -1
done
