#!/bin/bash
# per-launch k_extend / k_mq_* durations of one 16-spp soup pass with and without the binned mesh queue (kernel trace, one stream)
cd "$GRAFT_REPO_ROOT"; REPO=$PWD
export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  OUT=$REPO/gpurun_out/r03f/bintrace_$v; rm -rf $OUT; mkdir -p $OUT
  HPT_MQ_BIN=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/scripts/gpu_prof.py 16 16 soup > $OUT/run.log 2> $OUT/run.err
  python3 - $OUT $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ext = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith("void k_extend<false")]
binn = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].startswith("k_mq_")]
print("HPT_MQ_BIN=%s k_extend per depth (ms): %s  sum %.2f" % (sys.argv[2], " ".join("%.2f" % x for x in ext), sum(ext)))
if binn: print("            k_mq_hist + k_mq_scatter per depth: %s  sum %.2f" % (" ".join("%.2f" % (binn[i] + binn[i + 1]) for i in range(0, len(binn) - 1, 2)), sum(binn)))
PY
  find $OUT \( -name "*.db" -o -name "*.csv" \) -delete
done
