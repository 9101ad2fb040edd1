#!/bin/bash
# gpurun_out/{prof_<tag>*, pmc_<tag>_*, <tag>/share128.json} (scripts/gpu_profile.sh <tag> + scripts/gpu_round_counters.sh <tag>) -> profiles/<tag>_*
set -e
T=$1; cd "$(dirname "$0")/.."; P=gpurun_out/prof_$T
cp $P/bench.json profiles/${T}_bench.json; cp $P/kernel_stats.csv profiles/${T}_bench_kernel_stats.csv; cp $P/bench_traced.json profiles/${T}_bench_under_rocprof.json
cp $P/pmc_fetch_summary.txt profiles/${T}_pmc_fetch_size.txt; cp $P/pmc_write_summary.txt profiles/${T}_pmc_write_size.txt
python3 scripts/make_traffic_json.py $P profiles/$T > /dev/null
cp gpurun_out/pmc_${T}_sq32/summary.txt profiles/${T}_pmc_sq_counters.txt; cp gpurun_out/pmc_${T}_ta32/summary.txt profiles/${T}_pmc_ta_tcp_counters.txt
python3 scripts/make_valu_json.py profiles/${T}_pmc_sq_counters.txt profiles/$T "32 32 (1080p teapot, one pass of 32 spp = 66 M paths: the bench's launch size)" > /dev/null
cp gpurun_out/$T/share128.json profiles/${T}_share_of_8_gpus_128spp.json
[ -d gpurun_out/prof_${T}_soup ] || exit 0
: > profiles/${T}_configs_3to5.jsonl
for c in soup instanced spheres; do
  Q=gpurun_out/prof_${T}_$c
  cp $Q/kernel_stats.csv profiles/${T}_${c}_bench_kernel_stats.csv
  cp $Q/pmc_fetch_summary.txt profiles/${T}_${c}_pmc_fetch_size.txt; cp $Q/pmc_write_summary.txt profiles/${T}_${c}_pmc_write_size.txt
  python3 scripts/make_traffic_json.py $Q profiles/${T}_$c $c > /dev/null
  python3 scripts/fill_physical.py $Q/bench.json profiles/${T}_${c}_pmc_traffic.json >> profiles/${T}_configs_3to5.jsonl
done
for c in instanced spheres; do cp gpurun_out/pmc_${T}_${c}_sq/summary.txt profiles/${T}_${c}_pmc_sq_counters.txt; done
python3 scripts/make_valu_json.py profiles/${T}_instanced_pmc_sq_counters.txt profiles/${T}_instanced "8 8 instanced (3 375 teapots at 3840x2160, one pass of 8 spp = 66 M paths)" instanced > /dev/null
python3 scripts/make_valu_json.py profiles/${T}_spheres_pmc_sq_counters.txt profiles/${T}_spheres "8 8 spheres (10 k spheres at 1920x1080, depth 16, one pass of 8 spp = 17 M paths)" spheres > /dev/null
python3 - $T <<'PY'
import json, sys
T = sys.argv[1]
out = []
for l in open('profiles/%s_configs_3to5.jsonl' % T):
    b = json.loads(l); r = b['roofline']; c = b['metric'].split(', ')[1]
    try:
        rec = json.load(open('profiles/%s_%s_pmc_valu.json' % (T, c))); k = rec['k_extend']
        r['valu'] = {"frac": k['valu_busy_frac'], "valu_wave_insts_per_launch": k['valu_wave_insts_per_launch'], "lane_util": k['valu_lane_util'],
                     "source": 'profiles/%s_%s_pmc_valu.json' % (T, c), "stale": rec['source_sha256'] != r.get('source_sha256'),
                     "filled_in": "offline from the SQ pass of the same gpurun call (a one-pass render of scripts/gpu_prof.py: launch sizes differ from the bench's)"}
        r['limiter'] = 'valu_issue' if k['valu_busy_frac'] >= 0.7 else r.get('limiter')
    except FileNotFoundError:
        pass
    out.append(json.dumps(b))
    print(c, b['value'], b['ms_per_step'], 'k_extend', r['avg_launch_ms'], 'physical', r['hbm_physical']['GB/s'], r['hbm_physical']['frac'], r.get('limiter'))
open('profiles/%s_configs_3to5.jsonl' % T, 'w').write("\n".join(out) + "\n")
b = json.load(open('profiles/%s_bench.json' % T)); r = b['roofline']
print('teapot', b['value'], b['ms_per_step'], 'frac', r['frac'], 'k_extend', r['avg_launch_ms'], 'k_shade', r['shade_stage']['avg_launch_ms'], 'cpu', b['cpu_baseline']['value'], {k: v['Mrays_s'] for k, v in b['other_configs'].items()})
t = json.load(open('profiles/%s_bench_under_rocprof.json' % T))['roofline']; print('traced run, HIP events: k_extend', t['avg_launch_ms'], 'k_shade', t['shade_stage']['avg_launch_ms'])
print('128-spp share', json.load(open('profiles/%s_share_of_8_gpus_128spp.json' % T))['ms_per_step'])
PY
grep "k_shade<true, true, false\|k_extend<false, false\|k_raygen<true, false" profiles/${T}_bench_kernel_stats.csv | awk -F'",' '{print substr($1,1,40), $2}' | cut -c1-110
