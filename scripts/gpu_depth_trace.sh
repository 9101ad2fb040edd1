#!/bin/bash
# per-launch k_extend / k_shade durations (one pass, one stream) of a config: gpu_depth_trace.sh <config> <spp>   (kernel trace; launches in stream order = depth order)
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
CFG=$1; SPP=$2
export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out/r03f/depthtrace_$CFG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/scripts/gpu_prof.py $SPP $SPP $CFG > $OUT/run.log 2> $OUT/run.err
python3 - $OUT $CFG <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name in ("void k_extend<", "void k_shade<"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith(name)]
    print("%s %s per depth (ms): %s  sum %.2f" % (sys.argv[2], name[5:-1], " ".join("%.2f" % x for x in d), sum(d)))
PY
cat $OUT/run.log | tail -1
find $OUT \( -name "*.db" -o -name "*.csv" \) -delete
