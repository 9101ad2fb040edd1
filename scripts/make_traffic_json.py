"""profiles/<tag>_pmc_traffic.json from the two PMC summaries scripts/gpu_profile.sh wrote (FETCH_SIZE, WRITE_SIZE):
   python3 scripts/make_traffic_json.py gpurun_out/prof_<x> profiles/<tag>"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import source_sha256, STAMPED
src, dst = sys.argv[1], sys.argv[2]


def grab(path, kernel, counter):
    txt = open(path).read().split("\n")
    for i, l in enumerate(txt):
        if l.startswith(kernel):
            m = re.search(counter + r"\s+total (\S+)\s+per-dispatch (\S+)\s+\(n=(\d+)\)", txt[i + 1])
            return float(m.group(2)), int(m.group(3))
    raise SystemExit("no %s in %s" % (kernel, path))


b = json.load(open(src + "/pmc_fetch.json"))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --inflight 1 --spp 64 --steps 1 --warmup 0 --cpu-spp 0 --no-alt",
       "workload": b["config"]["workload"],
       "unit": "KB per launch as rocprofv3 reports; bytes = KB*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B reads at 64 B)",
       "rays_per_launch": b["rays_per_step"] / b["roofline"]["launches"],
       "source_sha256": source_sha256(), "source_files": list(STAMPED)}
for key, name in (("k_extend", "void k_extend<false, false"), ("k_extend_a", "void k_extend_a<false>"), ("k_shade", "void k_shade<true>")):
    f, n = grab(src + "/pmc_fetch_summary.txt", name, "FETCH_SIZE")
    w, _ = grab(src + "/pmc_write_summary.txt", name, "WRITE_SIZE")
    out[key] = {"launches": n, "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
