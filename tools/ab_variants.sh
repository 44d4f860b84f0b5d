for cb in 0 750 1000 1500 2000; do
  echo "== coop budget $cb"
  timeout -k 10 100 python tools/perf_probe.py --n 4000000 --only nn --reps 5 --opt coop_budget=$cb 2>&1 | tail -1
  timeout -k 10 100 python tools/perf_probe.py --grid 256 --only sdf --reps 5 --opt coop_budget=$cb 2>&1 | tail -1
  timeout -k 10 100 python tools/perf_probe.py --n 64000000 --only sdf --opt coop_budget=$cb 2>&1 | tail -1
done
