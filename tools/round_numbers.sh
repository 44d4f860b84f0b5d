#!/bin/bash
# The numbers of DESIGN.md §8 in one GPU call (run on the box through gpurun); every step under its own limit.
P="timeout -k 10 150 python"
echo "== torus 424^2, 125M random"; $P tools/perf_probe.py --n 125000000 --reps 3 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== torus 424^2, 256^3 grid"; $P tools/perf_probe.py --grid 256 --reps 5 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== torus 424^2, 4M random"; $P tools/perf_probe.py --n 4000000 --reps 5 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== icosphere-7 smooth, 64M random"; $P tools/perf_probe.py --mesh ico7 --noise 0 --n 64000000 --reps 3 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== icosphere-7 smooth, 256^3 grid"; $P tools/perf_probe.py --mesh ico7 --noise 0 --grid 256 --reps 3 --only sdf 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== icosphere-7 5% noise, 64M random"; $P tools/perf_probe.py --mesh ico7 --noise 0.05 --n 64000000 --reps 3 2>&1 | grep -E "^(nn|contains|sdf)" || exit 1
echo "== SMPL-scale torus (83x83), 256^3 grid"; $P tools/perf_probe.py --res 83 --grid 256 --reps 5 2>&1 | grep -E "^(mesh|nn|contains|sdf)" || exit 1
echo "== C5 frame"; $P tools/c5_probe.py 2>&1 | tail -1; $P tools/c5_probe.py --pipeline 1 2>&1 | tail -1
echo "== host API 32M"; $P tools/host_api_probe.py 32000000 2>&1 | grep Mevals
echo "== renderer"; $P tools/renderer_probe.py 2>&1 | tail -8
echo "== scale"; $P tools/scale_probe.py 2>&1 | tail -6
