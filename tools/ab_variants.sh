# A/B of builds on the GPU box: bash tools/ab_variants.sh <lib suffix> ...   (sdf_amd/libsdfhip_<suffix>.so)
for v in "$@"; do
  echo "== $v"
  export SDFHIP_LIB=$GRAFT_REPO_ROOT/sdf_amd/libsdfhip_$v.so
  timeout -k 10 100 python tools/perf_probe.py --n 4000000 --only nn --reps 5 2>&1 | tail -1 || exit 1
  timeout -k 10 100 python tools/perf_probe.py --grid 256 --only sdf --reps 5 2>&1 | tail -1 || exit 1
  timeout -k 10 100 python tools/perf_probe.py --n 125000000 --only sdf 2>&1 | tail -1 || exit 1
done
