# like ab_bench.sh for any config: ab_bench_cfg.sh "<bench.py args>" <variants...>   (prints Mrays/s, ms per step, k_extend launch time)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then unset HPT_LIB_PATH; else export HPT_LIB_PATH=$PWD/path-tracer_amd/variants_$v.so; fi
  timeout -k 10 200 python3 -u bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print('%-10s %9.1f Mrays/s %8.2f ms/step | k_extend %.4f ms' % ('$v', b['value'], b['ms_per_step'], r['avg_launch_ms']))" || exit 1
done
