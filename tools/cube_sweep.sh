#!/bin/bash
# Run ON THE GPU BOX: cube-list level / batch-size sweep (tools/cube_probe.py).
for spec in "125000000 4" "125000000 5" "125000000 6" "64000000 4" "64000000 5" "32000000 4" "32000000 5" "16000000 3" "16000000 4" "16000000 5" "8000000 3" "8000000 4" "4000000 3" "4000000 4"; do
  set -- $spec
  echo "== n=$1 level=$2"
  timeout -k 10 200 python3 tools/cube_probe.py --n $1 --level $2 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "== grid 256 level 4/5"
timeout -k 10 200 python3 tools/cube_probe.py --grid 256 --level 4 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/cube_probe.py --grid 256 --level 5 2>&1 | grep -v amdgpu.ids
