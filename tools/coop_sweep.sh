#!/bin/bash
# ON THE GPU BOX: hand-over budget of the cooperative continuation on a res x res torus (tools/scale_probe.py)
RES=${1:-1000}
shift
for B in "$@"; do
  echo "== coop_budget $B"
  timeout -k 10 200 python3 tools/scale_probe.py $RES profile=1 coop_budget=$B 2>&1 | grep -E "^(nn|sdf) " || exit 1
done
