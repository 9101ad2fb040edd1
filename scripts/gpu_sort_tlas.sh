#!/bin/bash
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
CFG=$1
export TMPDIR=/tmp; cd /tmp
OUT=$REPO/gpurun_out/r03f/sorttlas_$CFG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/scripts/gpu_sort_tlas.py $CFG > $OUT/run.log 2> $OUT/run.err
python3 - $OUT $CFG <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith("k_scene_closest")]
names = [l.split("order: ")[1].strip() for l in open(sys.argv[1] + "/run.log") if l.startswith("order:")]
print(sys.argv[2], "k_scene_closest (ms): warm-up %.3f" % d[0])
for nme, t in zip(names, d[1:]): print("   %-24s %8.3f ms   %.2fx" % (nme, t, d[1] / t))
PY
tail -3 $OUT/run.err
find $OUT \( -name "*.db" -o -name "*.csv" \) -delete
