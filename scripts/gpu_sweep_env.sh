#!/bin/bash
# bench.py over combinations of tuning environment variables: gpu_sweep_env.sh "VAR=a,b,c" "VAR2=x,y" ... (cartesian product)
cd "$GRAFT_REPO_ROOT"
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}; next=()
  IFS=',' read -ra arr <<< "$vals"
  for c in "${combos[@]}"; do for v in "${arr[@]}"; do next+=("$c $var=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  out=$(env $c timeout -k 10 120 python bench.py --steps 1 --warmup 1 --spp ${SWEEP_SPP:-256} --cpu-spp 0 --no-alt --no-others 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$c => $out"
done
