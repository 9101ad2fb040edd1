# A/B of library builds inside one GPU call: ab_variants.sh "<gpu_time args>" <variant names...> ("base" = the default build)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then unset HPT_LIB_PATH; else export HPT_LIB_PATH=$PWD/path-tracer_amd/variants_$v.so; fi
  echo "== $v"
  timeout -k 10 100 python3 -u scripts/gpu_time.py $ARGS || exit 1
done
