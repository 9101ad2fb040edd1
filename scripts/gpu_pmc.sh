#!/bin/bash
# SQ counter pass over a short render (scripts/gpu_prof.py): usage [PROF_SPP=8 PROF_S=4 PROF_CONFIG=teapot] gpu_pmc.sh <tag> "<counters>"
set -e -o pipefail
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d $OUT/raw -o p -- python3 $REPO/scripts/gpu_prof.py ${PROF_SPP:-8} ${PROF_S:-4} ${PROF_CONFIG:-teapot} > $OUT/run.log 2> $OUT/run.err
python3 $REPO/scripts/pmc_summary.py $OUT/raw > $OUT/summary.txt
rm -rf $OUT/raw
cat $OUT/summary.txt
