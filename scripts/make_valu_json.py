"""profiles/<tag>_pmc_valu.json from the SQ counter summary scripts/gpu_pmc.sh wrote:
   python3 scripts/make_valu_json.py profiles/<tag>_pmc_sq_counters.txt profiles/<tag> ["<gpu_prof.py arguments and what they mean>"]
VALU busy fraction = SQ_ACTIVE_INST_VALU (quad-cycles the VALU pipes of all SIMDs spent issuing) * 4
                     / (1024 SIMDs * kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (summed over the 32 shader engines).
VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): the share of the 64 lanes the average VALU
                     instruction had active (1.000 for k_raygen; VERDICT r02 weak 2: k_extend 0.506 in r02g).
Fourth argument (optional): the config the pass was taken on ("teapot" default; "instanced", "spheres", "soup": the two-level / big-tree kernels)."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import source_sha256, STAMPED
src, dst = sys.argv[1], sys.argv[2]
txt = open(src).read().split("\n")
SIMDS, SES = 256 * 4, 32


def block(prefix, optional=False):
    out = {}
    for i, l in enumerate(txt):
        if l.startswith(prefix):
            for m in txt[i + 1:]:
                g = re.match(r"\s+(\S+)\s+total (\S+)\s+per-dispatch (\S+)\s+\(n=(\d+)\)", m)
                if not g:
                    break
                out[g.group(1)] = float(g.group(3)); out["launches"] = int(g.group(4))
            return out
    if optional:
        return None
    raise SystemExit("no %s in %s" % (prefix, src))


what = sys.argv[3] if len(sys.argv) > 3 else "8 (1080p teapot, passes of 4 spp = 8 M paths)"
config = sys.argv[4] if len(sys.argv) > 4 else "teapot"
res = {"command": "rocprofv3 --pmc <8 SQ counters> -- python3 scripts/gpu_prof.py " + what, "config": config,
       "source_sha256": source_sha256(), "source_files": list(STAMPED),
       "formula": "valu_busy_frac = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * SQ_BUSY_CYCLES / 32 shader engines); "
                  "valu_lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)"}
# (instance-list scenes since r03e: the analytic stage runs inside k_raygen (depth 0) and k_shade (the later depths); k_extend_a only with HPT_FUSE_ANALYTIC=0)
kernels = (("k_extend", "void k_extend<false, false", False), ("k_extend_a", "void k_extend_a<false>", True), ("k_raygen", "void k_raygen<", True), ("k_shade", "void k_shade<", False)) if config == "teapot" else \
          (("k_extend", "void k_extend<false, ", False), ("k_shade", "void k_shade<", False))
for key, name, optional in kernels:
    b = block(name, optional)
    if b is None:
        continue
    cycles = b["SQ_BUSY_CYCLES"] / SES
    res[key] = {"launches": b["launches"], "valu_wave_insts_per_launch": b["SQ_INSTS_VALU"], "salu_wave_insts_per_launch": b["SQ_INSTS_SALU"],
                "kernel_cycles": round(cycles), "valu_busy_frac": min(1.0, round(b["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cycles), 4)),
                "valu_busy_frac_raw": round(b["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * cycles), 4),   # (SQ_BUSY_CYCLES is sampled per shader engine: a kernel that keeps every pipe busy can read 1-2 % above 1)
                "valu_lane_util": round(b["SQ_THREAD_CYCLES_VALU"] / (64.0 * b["SQ_ACTIVE_INST_VALU"]), 4),
                "wave_residency_frac": round(b["SQ_WAVE_CYCLES"] * 4 / (b["SQ_WAVES"] * cycles), 4)}
json.dump(res, open(dst + "_pmc_valu.json", "w"), indent=1)
print(json.dumps(res, indent=1))
