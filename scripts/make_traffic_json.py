"""profiles/<tag>_pmc_traffic.json from the two PMC summaries scripts/gpu_profile.sh (or gpu_profile_config.sh) wrote
(FETCH_SIZE, WRITE_SIZE):   python3 scripts/make_traffic_json.py gpurun_out/prof_<x> profiles/<tag> [config]"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import source_sha256, STAMPED
src, dst = sys.argv[1], sys.argv[2]
config = sys.argv[3] if len(sys.argv) > 3 else "teapot"


def grab(path, kernel, counter, optional=False):
    txt = open(path).read().split("\n")
    for i, l in enumerate(txt):
        if l.startswith(kernel):
            for m_ in txt[i + 1:]:
                m = re.search(counter + r"\s+total (\S+)\s+per-dispatch (\S+)\s+\(n=(\d+)\)", m_)
                if m:
                    return float(m.group(2)), int(m.group(3))
                if not m_.startswith(" "):
                    break
    if optional:
        return None, 0
    raise SystemExit("no %s in %s" % (kernel, path))


b = json.load(open(src + "/pmc_fetch.json"))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --inflight 1 --steps 1 --warmup 0 --cpu-spp 0 --no-alt" +
                  (" --spp 64" if config == "teapot" else " --config %s --spp %s" % (config, b["config"]["workload"].split(" spp")[0].split()[-1])),
       "config": config, "workload": b["config"]["workload"],
       "unit": "KB per launch as rocprofv3 reports; bytes = KB*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B line requests at 64 B -- "
               "checked for random 32-B and 64-B gathers too: profiles/r03c_fetch_size_gather.txt, one 64-B tally per missed record, 128 B moved)",
       "rays_per_launch": b["rays_per_step"] / b["roofline"]["launches"],
       "source_sha256": source_sha256(), "source_files": list(STAMPED)}
# the un-instrumented k_extend variant of this scene (template arguments differ by scene: node format, TLAS, work counter)
for key, name, optional in (("k_extend", "void k_extend<false,", False), ("k_extend_a", "void k_extend_a<false>", True), ("k_raygen", "void k_raygen<", True), ("k_shade", "void k_shade<", False)):
    f, n = grab(src + "/pmc_fetch_summary.txt", name, "FETCH_SIZE", optional)
    w, _ = grab(src + "/pmc_write_summary.txt", name, "WRITE_SIZE", optional)
    if f is None:
        continue
    out[key] = {"launches": n, "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
