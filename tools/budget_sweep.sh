# sweep of the two work budgets of the packet pass on the GPU box: bash tools/budget_sweep.sh
for hb in 2048 4096 5618 8192 16384; do
 for cb in 750 1500 3000; do
  echo "== heavy_budget $hb coop_budget $cb"
  timeout -k 10 100 python tools/perf_probe.py --n 4000000 --only nn --reps 5 --opt heavy_budget=$hb --opt coop_budget=$cb 2>&1 | tail -1 || exit 1
  timeout -k 10 100 python tools/perf_probe.py --n 125000000 --only sdf --opt heavy_budget=$hb --opt coop_budget=$cb 2>&1 | tail -1 || exit 1
 done
done
