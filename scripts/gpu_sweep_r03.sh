#!/bin/bash
# Launch-shape sweep on the headline frame at 256 spp (same pass sizes as the 1024-spp frame): passes in flight, pass size, shade grid.
cd $GRAFT_REPO_ROOT
run() { # label, env..., -- bench args
  label=$1; shift
  env "$@" timeout -k 10 120 python3 bench.py --spp 256 --steps 3 --warmup 1 --cpu-spp 0 --no-alt --no-others $BARGS 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); print('%-44s %9.1f Mrays/s %8.2f ms' % ('$label', b['value'], b['ms_per_step']))"
}
BARGS=""; run "default (2 in flight, auto pass size)" A=1
for n in 1 3 4; do BARGS="--inflight $n"; run "inflight $n" A=1; done
for s in 8 16 32 64; do BARGS="--spp-per-pass $s"; run "spp per pass $s" A=1; done
BARGS=""
for g in 7 10 14 20 28; do run "shade blocks per CU $g" HPT_SHADE_BLOCKS_PER_CU=$g; done
for g in 0 1 2; do run "extend spare $g" HPT_EXTEND_SPARE=$g; done
run "default again" A=1
