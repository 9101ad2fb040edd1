#!/bin/bash
# One of the synthetic BASELINE configs at full size: gpu_profile_config.sh <tag> <soup|instanced|spheres> <spp>
# bench line (two passes in flight), kernel trace of the one-stream run, HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in
# separate runs).  Results land under gpurun_out/prof_<tag>_<config>/.
set -e -o pipefail
TAG=$1; CFG=$2; SPP=$3
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
OUT=$REPO/gpurun_out/prof_${TAG}_$CFG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $REPO/bench.py --config $CFG --spp $SPP --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py --config $CFG --spp $SPP --inflight 1 --steps 1 --warmup 1 > $OUT/bench_traced.json 2> $OUT/trace.err
PSPP=$(( SPP < 32 ? SPP : 32 ))
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $REPO/bench.py --config $CFG --spp $PSPP --inflight 1 --steps 1 --warmup 0 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $REPO/bench.py --config $CFG --spp $PSPP --inflight 1 --steps 1 --warmup 0 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $REPO/scripts/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch_summary.txt
python3 $REPO/scripts/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write_summary.txt
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
cat $OUT/bench.json | cut -c1-400; head -5 $OUT/kernel_stats.csv; grep -A2 "k_extend" $OUT/pmc_fetch_summary.txt $OUT/pmc_write_summary.txt
