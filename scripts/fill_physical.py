"""bench line of a gpu_profile_config.sh call with roofline.hbm_physical taken from the FETCH_SIZE / WRITE_SIZE passes of the SAME
call (bench.py itself could only see the recordings committed before the call):
   python3 scripts/fill_physical.py <bench.json> <profiles/.._pmc_traffic.json>   -> one JSON line on stdout"""
import json, sys
b = json.load(open(sys.argv[1])); t = json.load(open(sys.argv[2]))
r = b["roofline"]
rays_per_launch = b["rays_per_step"] * b["steps"] / r["launches"] if "rays_per_step" in b else t["rays_per_launch"]
per_ray = t["k_extend"]["traffic_bytes_per_launch"] / t["rays_per_launch"]
phys = per_ray * rays_per_launch
avg_s = r["avg_launch_ms"] * 1e-3
r["traffic"] = int(phys); r["traffic_source"] = sys.argv[2]
r["hbm_physical"] = {"bytes_per_launch": int(phys), "GB/s": round(phys / avg_s / 1e9, 1), "frac": round(phys / avg_s / 8.0e12, 4),
                     "bytes_per_ray": round(per_ray, 1), "source": sys.argv[2], "stale": t["source_sha256"] != r.get("source_sha256"),
                     "filled_in": "offline by scripts/fill_physical.py from the PMC passes of the same gpurun call"}
if r.get("valu") is None:
    r["limiter"] = "hbm" if r["hbm_physical"]["frac"] >= 0.6 else None
print(json.dumps(b))
