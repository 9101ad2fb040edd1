# A/B of library builds on the headline frame inside one GPU call: ab_bench.sh "<bench.py args>" <variant names...> ("base" = the default build;
# variant v = path-tracer_amd/variants_v.so through HPT_LIB_PATH).  Prints Mrays/s, ms per step and the one-stream kernel averages per variant.
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then unset HPT_LIB_PATH; else export HPT_LIB_PATH=$PWD/path-tracer_amd/variants_$v.so; fi
  timeout -k 10 200 python3 -u bench.py --cpu-spp 0 --no-alt --no-others $ARGS 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']; s=r.get('shade_stage') or {}
print('%-10s %9.1f Mrays/s %8.2f ms/step | k_extend %.4f ms  k_shade %s ms (%s ps/ray)' % ('$v', b['value'], b['ms_per_step'], r['avg_launch_ms'], s.get('avg_launch_ms'), s.get('ps_per_ray')))" || exit 1
done
