#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command, then HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs) of a shortened one.  The traced and counted runs keep ONE pass in flight
# (--inflight 1): their per-kernel figures are those of launches that have the GPU to themselves, like roofline.avg_launch_ms.  Results land under gpurun_out/prof_<tag>/.
set -e -o pipefail
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ -z "$SKIP_BENCH" ]; then timeout -k 10 400 python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err; fi
cat $OUT/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py --inflight 1 --steps 1 --warmup 1 --cpu-spp 0 --no-alt --no-others > $OUT/bench_traced.json 2> $OUT/trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $REPO/bench.py --inflight 1 --spp 64 --steps 1 --warmup 0 --cpu-spp 0 --no-alt --no-others > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $REPO/bench.py --inflight 1 --spp 64 --steps 1 --warmup 0 --cpu-spp 0 --no-alt --no-others > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $REPO/scripts/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch_summary.txt
python3 $REPO/scripts/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write_summary.txt
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
cat $OUT/kernel_stats.csv | head -12; cat $OUT/pmc_fetch_summary.txt $OUT/pmc_write_summary.txt
