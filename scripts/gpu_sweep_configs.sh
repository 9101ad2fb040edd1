#!/bin/bash
# passes in flight / pass size on the other configs (two steps each): gpu_sweep_configs.sh
cd $GRAFT_REPO_ROOT
run() { label=$1; shift; timeout -k 10 200 python3 bench.py "$@" --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); print('%-40s %9.1f Mrays/s %8.2f ms' % ('$label', b['value'], b['ms_per_step']))"; }
for c in "soup 64" "spheres 64" "instanced 32"; do set -- $c
  run "$1 default" --config $1 --spp $2
  run "$1 inflight 3" --config $1 --spp $2 --inflight 3
  run "$1 spp-per-pass /2" --config $1 --spp $2 --spp-per-pass $(( $2 / 8 ))
  run "$1 spp-per-pass x1" --config $1 --spp $2 --spp-per-pass $(( $2 / 4 ))
  run "$1 spp-per-pass x2" --config $1 --spp $2 --spp-per-pass $(( $2 / 2 ))
done
