#!/bin/bash
# Node64 vs Node32 on the full-size configs (bench.py --config, one step): value and ms per step
cd "$GRAFT_REPO_ROOT"
for c in instanced spheres soup; do
  for f in 64 32; do
    out=$(HPT_NODE_FORMAT=$f timeout -k 10 280 python bench.py --config $c --steps 1 --warmup 1 --cpu-spp 0 --no-alt --no-others 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('node_bytes'))") || exit 1
    echo "$c HPT_NODE_FORMAT=$f => $out"
  done
done
