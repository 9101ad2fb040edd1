#!/bin/bash
# base library vs variant builds (path-tracer_amd/variants_<v>.so) on full-size configs: gpu_variant_configs.sh "<configs>" <variant...>
cd "$GRAFT_REPO_ROOT"
CFGS=$1; shift
for c in $CFGS; do
  for v in base "$@" base "$@"; do
    unset HPT_LIB_PATH; if [ "$v" != base ]; then export HPT_LIB_PATH=$PWD/path-tracer_amd/variants_$v.so; fi
    out=$(timeout -k 10 280 python bench.py --config $c --steps 1 --warmup 1 --spp ${VSPP:-32} --cpu-spp 0 --no-alt --no-others 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])") || exit 1
    echo "$c $v => $out"
  done
done
