#!/bin/bash
# Kernel-trace A/B of library builds: gpu_ab_trace.sh <tag> <variant...>   (variant "base" = libhpt.so, "base4" = libhpt.so
# at 4 k_extend workgroups per CU, else path-tracer_amd/variants_<variant>.so).  One-stream frame of 128 spp per variant;
# kernel stats land in gpurun_out/ab_<tag>/<variant>_kernel_stats.csv.
set -e -o pipefail
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
OUT=$REPO/gpurun_out/ab_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  unset HPT_LIB_PATH HPT_EXTEND_BLOCKS_PER_CU
  if [ "$v" = "base4" ]; then export HPT_EXTEND_BLOCKS_PER_CU=4; elif [ "$v" != "base" ]; then export HPT_LIB_PATH=$REPO/path-tracer_amd/variants_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python3 $REPO/bench.py --inflight 1 --spp 128 --steps 1 --warmup 1 --cpu-spp 0 --no-alt --no-others > $OUT/$v.json 2> $OUT/$v.err
  find $OUT/$v -name "*kernel_stats.csv" -exec cp {} $OUT/${v}_kernel_stats.csv \;
  rm -rf $OUT/$v
  echo "== $v"; head -3 $OUT/${v}_kernel_stats.csv
done
