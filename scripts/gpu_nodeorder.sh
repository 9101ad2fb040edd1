#!/bin/bash
# breadth-first vs treelet numbering of the interior nodes (HPT_NODE_ORDER): teapot bench (512 spp) and the full-size configs
cd "$GRAFT_REPO_ROOT"
for o in bfs treelet bfs treelet; do
  out=$(HPT_NODE_ORDER=$o timeout -k 10 120 python bench.py --steps 1 --warmup 1 --spp 512 --cpu-spp 0 --no-alt --no-others 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])") || exit 1
  echo "teapot HPT_NODE_ORDER=$o => $out"
done
for c in soup instanced spheres; do
  for o in bfs treelet; do
    out=$(HPT_NODE_ORDER=$o timeout -k 10 280 python bench.py --config $c --steps 1 --warmup 1 --cpu-spp 0 --no-alt --no-others 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])") || exit 1
    echo "$c HPT_NODE_ORDER=$o => $out"
  done
done
