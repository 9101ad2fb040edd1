#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on its own configuration.

metric   : Mrays/sec (primary + bounce segments traced per second), whole job over all ranks
workload : configs[1] "Teapot scene 1920x1080, 1024 spp, max depth 8" -- tests/golden/scene.xml with the camera
           resized to 1920x1080, fixed 1024 samples per pixel, seed 1, host-built reference BVH.
step     : one full frame (all 1024 spp of every pixel), two wavefront passes in flight on two HIP streams (library default).  With N ranks the samples of every pixel are split into N
           contiguous ranges (strong scaling: the frame is fixed), each rank adds its range to an accumulator on its GPU and the
           accumulators -- 64-bit fixed point (include/hpt.h hpt_accum), so the image does not depend on N in any bit -- are
           sum-reduced onto rank 0 with RCCL (torch.distributed backend "nccl", int64), inside the timed region.
Inputs are resident in HBM before the timed region (the scene is committed once); the output stays on the device.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

  python bench.py --config soup|instanced|spheres ...     the other BASELINE.json configs at full size as a line of their own

The default N=1 run also carries "other_configs": configs[2] (1 M-triangle soup, 256 spp) and configs[4] (10 k spheres, 256 spp,
depth 16) at their BASELINE sizes and configs[3] (3 375 instanced teapots at 4K) with 512 spp -- one GPU's share of the 4096 spp that
config splits over 8 GPUs --, one step each after the headline (--no-others skips them).

Extra objects on the JSON line: "roofline" (k_extend, the BVH traversal kernel: algorithmic bytes / HIP-event kernel time vs the
8 TB/s HBM peak.  The analytic stage of the extend step -- spheres, planes, mesh root boxes -- runs inside the kernel that has just made
the ray (k_raygen at depth 0, k_shade later): analytic_share_of_step is 0 unless HPT_FUSE_ANALYTIC=0, shade_share_of_step covers the stage;
the event-bracketed launches are the same K steps run once more on ONE stream, right after the timed region: with two
passes in flight the event interval of a launch would also cover the other pass's kernels sharing the GPU with it;
`hbm_physical` / `valu` = the committed rocprofv3 counter passes of the same frame, flagged `stale` when they were
taken with other kernel sources than the tree holds; `limiter` = which of the two is closer to its peak; `shade_stage` = the same figures
for k_shade, the largest kernel of the step since round 3: HIP-event launch time, algorithmic and physical bytes, VALU counters)
and "cpu_baseline" (the C restatement of the reference -- oracle/liboracle_fast.so -- on the host cores, rank 0, N=1).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# queue bytes k_extend (the BVH traversal kernel) moves per ray it walks: mesh-queue slot 48 B read (item 16 B + the ray in the
# space of its first mesh 32 B), hit + hit_inst 20 B written.  Instance-list scenes walk only the rays that entered a mesh box
# (stats.rays_mesh); two-level scenes walk every ray in the one kernel (ray 32 B read, hit record 20 B written).
MESH_QUEUE_BYTES = 68
TLAS_QUEUE_BYTES = 52


def tree_sha256():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from source_stamp import source_sha256
    return source_sha256(ROOT)


def _profile_order(path):
    """(round, letter, name) of profiles/rNNx_... or rNN_x_...: r10a sorts after r02g"""
    import re
    m = re.match(r"r(\d+)_?([a-z]?)", os.path.basename(path))
    return (int(m.group(1)), m.group(2), os.path.basename(path)) if m else (-1, "", os.path.basename(path))


def _newest_profile(pattern, config="teapot"):
    """newest (by round number and letter of the file name) committed counter recording of `config`; the teapot's carry no config tag"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=_profile_order, reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if rec.get("config", "teapot") == config:
            return rec, os.path.relpath(f, ROOT)
    return None, None


def recorded_traffic(kernel, config="teapot"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share
    a pass, and counters cannot be read from inside this process): the newest profiles/*_pmc_traffic.json written by
    scripts/gpu_profile.sh for this same workload (same pass size -> same launches).  `stale`: the recording was made
    with other kernel / launch / tree sources than the tree holds now (sha256 stamp).  None if absent."""
    rec, src = _newest_profile("*_pmc_traffic.json", config)
    if rec is None or kernel not in rec:
        return None
    return {"bytes_per_launch": int(rec[kernel]["traffic_bytes_per_launch"]), "source": src, "rays_per_launch": rec.get("rays_per_launch"),
            "stale": rec.get("source_sha256") != tree_sha256()}


def recorded_valu(kernel, config="teapot"):
    """VALU-busy fraction of `kernel` from the committed SQ counter pass (profiles/*_pmc_valu.json, written by
    scripts/make_valu_json.py from a `rocprofv3 --pmc` run of the same frame).  None if absent."""
    rec, src = _newest_profile("*_pmc_valu.json", config)
    if rec is None or kernel not in rec:
        return None
    return {"frac": float(rec[kernel]["valu_busy_frac"]), "valu_wave_insts_per_launch": float(rec[kernel]["valu_wave_insts_per_launch"]),
            # share of the 64 lanes the average VALU instruction of the kernel had active: SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
            "lane_util": rec[kernel].get("valu_lane_util"),
            "source": src, "stale": rec.get("source_sha256") != tree_sha256()}


def parity_ray_rates(scene, desc, n_gpu=10_000_000, n_cpu=1_000_000):
    """SURVEY 8d.6 beside the frame metric: closest-hit rate of k_trace_rays on the 10 M seeded parity rays (device
    buffers, HIP-event kernel time) and of the oracle's Mesh::intersect restatement on the first 1 M of them (host cores),
    with the ids of those 1 M compared on the way."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import rays as R
    lib = ol.load(fast=True)
    osc = ol.OracleScene(lib, desc)
    org, dirs = R.parity_rays(osc.meshes[0].box(), n_gpu)
    o4 = torch.zeros((n_gpu, 4), dtype=torch.float32); o4[:, :3] = torch.from_numpy(org)
    d4 = torch.zeros((n_gpu, 4), dtype=torch.float32); d4[:, :3] = torch.from_numpy(dirs)
    o4, d4 = o4.cuda(), d4.cuda()
    hit = torch.zeros((n_gpu, 4), dtype=torch.float32, device="cuda"); side = torch.zeros(n_gpu, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    scene.trace_rays_device(0, n_gpu, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), side.data_ptr(), want_stats=True)
    st = scene.trace_rays_device(0, n_gpu, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), side.data_ptr(), want_stats=True)
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "1024")))
    t0 = time.perf_counter()
    ref = osc.meshes[0].intersect(org[:n_cpu], dirs[:n_cpu], threads=threads)
    dt = time.perf_counter() - t0
    prim = hit[:n_cpu, 3].contiguous().view(torch.int32).cpu().numpy()
    return {"gpu_Mrays_s": round(n_gpu / st.seconds_kernel / 1e6, 1), "gpu_rays": n_gpu, "cpu_Mrays_s": round(n_cpu / dt / 1e6, 3),
            "cpu_rays": n_cpu, "cpu_cores": threads, "id_mismatches_in_cpu_sample": int((prim != ref["prim"]).sum())}


def cpu_baseline(desc, spp, depth, seed, frame_spp):
    """The oracle (kind "port": C restatement of the reference's algorithm, faithful mode -- reference tree with loose
    boxes, no pruning, per-call matrix inversion) timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    lib = ol.load(fast=True)
    sc = ol.OracleScene(lib, desc)
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "1024")))
    r = sc.render(spp=spp, max_depth=depth, seed=seed, threads=threads)
    st = r["stats"]
    return {"value": round(st.rays / st.seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(st.threads), "kind": "port",
            "sample": "%dx%d, samples 1..%d of %d spp, depth %d (%.1f M rays, %.1f s)" % (
                desc.camera.width, desc.camera.height, spp, frame_spp, depth, st.rays / 1e6, st.seconds),
            "samples_per_px_per_s": round(spp / st.seconds, 4)}, r["rgb"]


CONFIGS = {
    # name: (BASELINE.json config, default width, height, spp, depth)
    "teapot": ("configs[1] teapot scene 1920x1080, 1024 spp, max depth 8", 1920, 1080, 1024, 8),
    "soup": ("configs[2] synthetic 1M-triangle random soup, 1080p 256 spp", 1920, 1080, 256, 8),
    "instanced": ("configs[3] 3 375 instanced teapots (21.3 M triangles), 4K, 4096 spp split over the ranks", 3840, 2160, 4096, 8),
    "spheres": ("configs[4] 10 k spheres, mixed diffuse / specular, depth 16, 1080p 256 spp", 1920, 1080, 256, 16),
}


def build_workload(args):
    """(scene description, host tree builder or None, text for config.workload, text for config.bvh).  The teapot frame is
    the shipped scene.xml with the camera resized and the reference's own bvh.rs tree built on the host; the
    synthetic configs (generators in tests/scenes.py, fixed seeds) hand no tree over -- bvh.rs degenerates
    on unordered input -- so the library builds it."""
    from hpt import host_bvh, scene_io
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    teapot = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "scene.xml"))
    ref_tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
    if args.config == "teapot":
        teapot.camera.width, teapot.camera.height = args.width, args.height
        return teapot, (ref_tree if args.tree == "host" else None), "teapot scene.xml %dx%d, %d spp, max depth %d, 1 frame per step" % (
            args.width, args.height, args.spp, args.depth)
    import scenes
    if args.config == "soup":
        d = scenes.soup(1_000_000, w=args.width, h=args.height)
        what = "1 M-triangle random soup (seed 2)"
    elif args.config == "instanced":
        d = scenes.instanced(teapot.meshes[0], 15, w=args.width, h=args.height)
        what = "%d teapot instances on a 15^3 lattice + floor (%.1f M triangles, seed 3)" % (len(d.nodes) - 1, (len(d.nodes) - 1) * len(teapot.meshes[0].tris) / 1e6)
        return d, (ref_tree if args.tree == "host" else None), "%s %dx%d, %d spp, max depth %d, 1 frame per step" % (what, args.width, args.height, args.spp, args.depth)
    else:
        d = scenes.spheres(10_000, w=args.width, h=args.height)
        what = "10 k spheres + floor (seed 4)"
    return d, None, "%s %dx%d, %d spp, max depth %d, 1 frame per step" % (what, args.width, args.height, args.spp, args.depth)


def measure(args, scene, desc, tree, workload, base, world, rank, backend, stream, default_frame, ctx):
    """K timed steps of `scene` (+ the kernel-timing and counter runs beside them); returns the JSON line (rank 0) or None"""
    import hpt
    from hpt.dist import sample_range
    npix = args.width * args.height
    first, count = sample_range(rank, world, args.spp)
    with torch.cuda.stream(stream):
        accum = torch.zeros((npix, 4), device="cuda", dtype=torch.int64)      # hpt_accum records: 64-bit fixed-point sums
        image = torch.zeros(npix * 3, device="cuda", dtype=torch.float32)
        reduce_events = []   # the one exchange step, also shown on its own (it is inside the timed region)

        def step():
            accum.zero_()
            st = scene.render_accum_device(accum.data_ptr(), 0, 0, npix, want_stats=True, spp=count, first_sample=first, max_depth=args.depth,
                                           seed=1, spp_per_pass=args.spp_per_pass)
            if world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                if backend == "nccl":
                    dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)     # RCCL sum-reduce (int64) of the accumulators over xGMI
                else:                                                   # rehearsal only: reduce through host memory
                    stream.synchronize()
                    host = accum.cpu()
                    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                    accum.copy_(host)
                e1.record(stream)
                reduce_events.append((e0, e1))
            if rank == 0:
                scene.accum_resolve_device(accum.data_ptr(), image.data_ptr(), num_samples=args.spp)        # main.rs:110
            return st

        for _ in range(args.warmup):
            step()
        reduce_events.clear()
        stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = [step() for _ in range(args.steps)]
        stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0

        rays_local = sum(s.rays_primary + s.rays_bounce for s in stats)
        vec = torch.tensor([elapsed, float(rays_local)], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            tmax = vec[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            rsum = vec[1:].clone(); dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
            elapsed, rays_total = float(tmax.item()), float(rsum.item())
        else:
            rays_total = float(rays_local)
        image_mean = [round(float(v), 6) for v in image.view(-1, 3).mean(dim=0).cpu()] if rank == 0 else None

        # kernel timing (rank 0 only; no collective inside): the same K steps on one stream, every launch bracketed by HIP
        # events on the stream it is launched on (HPT_RENDER_TIME_KERNELS) -- the same launches, same rays, serialised
        ext_s_local = ana_s_local = shade_s_local = serial_elapsed = 0.0
        ext_launches = 0
        if rank == 0:
            stream.synchronize(); t1 = time.perf_counter()
            kst = [scene.render_accum_device(accum.data_ptr(), 0, 0, npix, want_stats=True, spp=count, first_sample=first, max_depth=args.depth, seed=1,
                                             spp_per_pass=args.spp_per_pass, flags=hpt.RENDER_TIME_KERNELS) for _ in range(args.steps)]
            stream.synchronize(); serial_elapsed = time.perf_counter() - t1
            assert sum(s.rays_primary + s.rays_bounce for s in kst) == rays_local
            ext_s_local = sum(s.seconds_extend for s in kst)
            ana_s_local = sum(s.seconds_analytic for s in kst)
            ext_launches = sum(s.launches_extend for s in kst)
            shade_s_local = sum(s.seconds_shade for s in kst)

        # per-ray algorithmic bytes of k_extend from an instrumented (counters on), untimed run of the same kernel on
        # a slice of this rank's samples: node bytes per interior node popped + 48 B per triangle tested + the queue bytes the
        # kernel itself moves (DESIGN.md "Roofline").  Instance records (128 B) count per ray only when rays reach them
        # one by one through the TLAS; the linear instance loop of small scenes reads each record once per WORKGROUP
        # (wave-uniform loads), which is < 0.1 % of the bytes and left out.
        cst = scene.render_accum_device(accum.data_ptr(), 0, 0, npix, want_stats=True, spp=min(count, 4), first_sample=first, max_depth=args.depth, seed=1,
                                        spp_per_pass=args.spp_per_pass, flags=hpt.RENDER_COUNTERS)
        crays = cst.rays_primary + cst.rays_bounce
        info = scene.info()
        inst_bytes = 128.0 * cst.inst_tests if info.uses_tlas else 0.0
        walked_frac = 1.0 if info.uses_tlas else cst.rays_mesh / crays      # share of the rays k_extend walks
        queue_bytes = TLAS_QUEUE_BYTES if info.uses_tlas else MESH_QUEUE_BYTES * walked_frac
        bytes_per_ray = (info.node_bytes * float(cst.nodes_visited) + 48.0 * cst.tris_tested + inst_bytes) / crays + queue_bytes
        bvh_bytes_per_ray = (info.node_bytes * float(cst.nodes_visited) + 48.0 * cst.tris_tested) / crays
    del accum, image

    if rank != 0:
        return None
    achieved = bytes_per_ray * rays_local / ext_s_local
    avg_launch_s = ext_s_local / max(ext_launches, 1)
    rays_per_launch = rays_local / max(ext_launches, 1)
    # physical side, from the committed counter passes of this same frame (stamped with the sources they were taken with)
    traffic = recorded_traffic("k_extend", args.config) if default_frame else None
    valu = recorded_valu("k_extend", args.config) if default_frame else None
    hbm_physical = None
    if traffic:
        # the recording's launches may differ in size from this run's (e.g. 64-spp frame: same 66 M-path passes): scale per ray
        per_ray = traffic["bytes_per_launch"] / traffic["rays_per_launch"] if traffic.get("rays_per_launch") else None
        phys_bytes = per_ray * rays_per_launch if per_ray else float(traffic["bytes_per_launch"])
        hbm_physical = {"bytes_per_launch": int(phys_bytes), "GB/s": round(phys_bytes / avg_launch_s / 1e9, 1),
                        "frac": round(phys_bytes / avg_launch_s / HBM_PEAK, 4), "bytes_per_ray": round(phys_bytes / rays_per_launch, 1),
                        "source": traffic["source"], "stale": traffic["stale"]}
    limiter = None
    if hbm_physical and not valu:
        limiter = "hbm" if hbm_physical["frac"] >= 0.6 else None
    if valu and hbm_physical:
        # what the counters say: a pipe that is busy most of the time limits the kernel; when neither the vector ALUs nor
        # the HBM pins are, the waves are waiting on dependent fetches (node -> child node through L2)
        limiter = "valu_issue" if valu["frac"] >= 0.7 and valu["frac"] >= hbm_physical["frac"] else ("hbm" if hbm_physical["frac"] >= 0.6 else "latency")
    if args.config == "teapot":
        bvh_text = ("reference bvh.rs tree built on host; library restructures it at flatten time (SAH re-partition + insertion rounds), tight refit"
                    if info.trees_optimized else "reference bvh.rs tree built on host, tight refit") if args.tree == "host" else "library SAH"
    else:
        bvh_text = ("TLAS over instances + " if info.uses_tlas else "") + ("reference bvh.rs BLAS built on host, restructured at flatten time" if tree else "library SAH")
    nominal = bool(info.device_bytes < 32 << 20)
    # the other big kernel of the step: k_shade, which on instance-list scenes also runs the analytic stage of the ray it makes.  Its queue
    # traffic per shaded ray: 68 B read (three 16-B path records, hit 16 B, hit_inst 4 B), per surviving path 48 B of path records + 20 B
    # of analytic hit record written, per path that enters a mesh box a 48-B mesh-queue slot (scripts/micro/stream_state.hip: a kernel
    # that only moves these bytes takes 23 ps per path)
    survive = (rays_local - sum(s.rays_primary for s in stats)) / rays_local
    shade_bytes_per_ray = 68.0 + survive * (48.0 + (20.0 + 48.0 * walked_frac if not info.uses_tlas else 0.0))
    shade_valu = recorded_valu("k_shade", args.config) if default_frame else None
    shade_traffic = recorded_traffic("k_shade", args.config) if default_frame else None
    shade_physical = None
    if shade_traffic and shade_traffic.get("rays_per_launch") and shade_s_local:
        sb = shade_traffic["bytes_per_launch"] / shade_traffic["rays_per_launch"] * rays_per_launch      # scaled per ray like k_extend's
        ss = shade_s_local / max(ext_launches, 1)
        shade_physical = {"bytes_per_launch": int(sb), "GB/s": round(sb / ss / 1e9, 1), "frac": round(sb / ss / HBM_PEAK, 4),
                          "bytes_per_ray": round(sb / rays_per_launch, 1), "source": shade_traffic["source"], "stale": shade_traffic["stale"]}
    shade_stage = {"kernel": "k_shade" + ("" if info.uses_tlas else " (+ analytic stage of the next depth)"), "bound": "hbm", "hbm_physical": shade_physical,
                   "avg_launch_ms": round(shade_s_local / max(ext_launches, 1) * 1e3, 4), "bytes_per_ray": round(shade_bytes_per_ray, 1),
                   "GB/s": round(shade_bytes_per_ray * rays_local / shade_s_local / 1e9, 1) if shade_s_local else None,
                   "frac": round(shade_bytes_per_ray * rays_local / shade_s_local / HBM_PEAK, 4) if shade_s_local else None,
                   "ps_per_ray": round(shade_s_local / rays_local * 1e12, 2) if shade_s_local else None, "valu": shade_valu,
                   "limiter": ("valu_issue" if shade_valu["frac"] >= 0.7 else "hbm") if shade_valu else None}
    line = {
        "metric": "Mrays/sec (primary+bounce), %s" % ("teapot 1080p" if args.config == "teapot" else args.config), "value": round(rays_total / elapsed / 1e6, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": base[0], "spp_per_rank": count, "bvh": bvh_text, "node_bytes": int(info.node_bytes),
                   "parallelism": "spp-split x%d + RCCL sum-reduce (int64 fixed-point accumulators)" % world, "passes_in_flight": args.inflight if args.inflight > 0 else 2},
        "samples_per_px_per_s": round(args.spp * args.steps / elapsed, 3),
        "image_mean": image_mean,
        "rays_per_step": rays_total / args.steps,
        "reduce_ms_per_step": round(sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1), 4) if reduce_events else 0.0,
        # `achieved` / `frac`: ALGORITHMIC bytes (every node and triangle a ray touches + its queue records) over the
        # kernel's HIP-event time, priced against the HBM peak -- the figure the north_star target is stated in.  It is
        # NOMINAL when the scene lives in the L2s (the teapot: 1.4 MB): what the HBM pins really carry is
        # `hbm_physical`, and what limits the kernel is `limiter` (VALU issue on the teapot, see `valu`) -- `bound` says which
        # peak the nominal fraction is priced against, `limiter` what the counters show.
        "roofline": {"bound": "hbm", "kernel": "k_extend", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK, 4), "nominal": nominal,
                     "traffic": hbm_physical["bytes_per_launch"] if hbm_physical else None,
                     "traffic_source": traffic["source"] if traffic else None,
                     "hbm_physical": hbm_physical, "valu": valu, "limiter": limiter,
                     "algorithmic_bytes_per_launch": int(bytes_per_ray * rays_per_launch),
                     "bytes_per_ray": round(bytes_per_ray, 1), "bvh_bytes_per_ray": round(bvh_bytes_per_ray, 1),
                     "nodes_per_ray": round(cst.nodes_visited / crays, 3), "tris_per_ray": round(cst.tris_tested / crays, 3),
                     "inst_per_ray": round(cst.inst_tests / crays, 3),
                     "scene_bytes_in_hbm": int(info.device_bytes),
                     "bvh_fetch_frac": round(bvh_bytes_per_ray * rays_local / ext_s_local / HBM_PEAK, 4),
                     "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(ext_launches),
                     "kernel_timing": "HIP events around every launch of the same %d step(s) re-run on one stream (%.1f ms per step); the timed region keeps %s passes in flight" % (
                         args.steps, serial_elapsed / args.steps * 1e3, args.inflight if args.inflight > 0 else 2),
                     "rays_walked_frac": round(walked_frac, 4),
                     "extend_share_of_step": round(ext_s_local / serial_elapsed, 3), "analytic_share_of_step": round(ana_s_local / serial_elapsed, 3),
                     "shade_share_of_step": round(shade_s_local / serial_elapsed, 3), "shade_stage": shade_stage,
                     "source_sha256": tree_sha256()},
    }
    return line


def other_config_line(name, ctx, stream, spp=0):
    """One step of another BASELINE.json config at its full size, for the driver's default line: throughput, the traversal
    kernel's launch time and visits per ray, the committed counter figures of that config.  spp: samples of the step when it is not the
    config's whole frame (instanced: 512 = one GPU's share of the 4096 spp the config splits over 8 GPUs)."""
    import copy
    a = argparse.Namespace(config=name, width=CONFIGS[name][1], height=CONFIGS[name][2], spp=spp or CONFIGS[name][3], depth=CONFIGS[name][4],
                           spp_per_pass=0, steps=1, warmup=0, tree="host", inflight=0)
    t0 = time.perf_counter()
    desc, tree, workload = build_workload(a)
    import hpt
    scene = hpt.Scene(ctx, desc, host_bvh=tree)
    build_s = time.perf_counter() - t0
    with torch.cuda.stream(stream):      # untimed warm-up: first launches of this scene's kernel variants, workspace allocation
        w = torch.zeros((a.width * a.height, 4), device="cuda", dtype=torch.int64)
        scene.render_accum_device(w.data_ptr(), 0, 0, a.width * a.height, want_stats=True, spp=16, max_depth=a.depth, seed=1)
        del w
    line = measure(a, scene, desc, tree, workload, CONFIGS[name], 1, 0, "nccl", stream, True, ctx)
    scene.close()
    r = line["roofline"]
    return {"workload": workload, "Mrays_s": line["value"], "ms_per_step": line["ms_per_step"], "samples_per_px_per_s": line["samples_per_px_per_s"],
            "k_extend_ms": r["avg_launch_ms"], "nodes_per_ray": r["nodes_per_ray"], "tris_per_ray": r["tris_per_ray"], "inst_per_ray": r["inst_per_ray"],
            "roofline_frac_algorithmic": r["frac"], "nominal": r["nominal"], "hbm_physical": r["hbm_physical"], "valu": r["valu"], "limiter": r["limiter"],
            "scene_bytes_in_hbm": r["scene_bytes_in_hbm"], "scene_build_and_commit_s": round(build_s, 2), "image_mean": line["image_mean"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="teapot", help="BASELINE.json workload; teapot = the headline (configs[1])")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--cpu-spp", type=int, default=-1, help="samples per pixel of the CPU-baseline sample (0 = skip; default 12 for the teapot frame, 0 else)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurement with the handed-over tree left as it came")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configs (soup, instanced teapots, spheres) the default N=1 teapot run appends as other_configs")
    ap.add_argument("--tree", choices=["host", "sah"], default="host", help="host: reference bvh.rs tree handed over (drop-in); sah: library-built")
    ap.add_argument("--inflight", type=int, default=0, help="wavefront passes in flight (0 = library default, 2; 1 = one stream: every launch has the GPU to itself, "
                    "the setting of the rocprofv3 kernel-trace runs)")
    args = ap.parse_args()
    base = CONFIGS[args.config]
    full_default = args.config == "teapot" and not (args.width or args.height or args.spp or args.depth or args.spp_per_pass) and args.tree == "host"
    args.width, args.height = args.width or base[1], args.height or base[2]
    args.spp, args.depth = args.spp or base[3], args.depth or base[4]
    if args.cpu_spp < 0:
        args.cpu_spp = 12 if args.config == "teapot" else 0
    default_frame = (args.width, args.height, args.depth) == base[1:3] + base[4:5] and args.tree == "host"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render core has no CPU path")
    # rehearsal hooks (never set by the driver): HPT_BENCH_BACKEND=gloo lets 2+ ranks share one GPU on a 1-GPU box
    backend = os.environ.get("HPT_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    import hpt

    desc, tree, workload = build_workload(args)
    ctx = hpt.Context(device_index)
    if args.inflight > 0:
        ctx.set_passes_in_flight(args.inflight)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    scene = hpt.Scene(ctx, desc, host_bvh=tree)          # scene + flattened BVH replicated on every rank's HBM

    npix = args.width * args.height
    from hpt.dist import sample_range
    first, count = sample_range(rank, world, args.spp)
    line = measure(args, scene, desc, tree, workload, base, world, rank, backend, stream, default_frame, ctx)

    if rank == 0:
        if world == 1 and tree is not None and not args.no_alt:
            # the same frame with the handed-over tree laid out as it came (tight refit only, HPT_TREE_OPT=0 -- the round-1
            # behaviour): identical hits by the deterministic tie rule, more box steps.  Reported beside `value`.
            hpt.set_tree_opt(0)
            try:
                alt = hpt.Scene(ctx, desc, host_bvh=tree)
            finally:
                hpt.set_tree_opt(-1)
            with torch.cuda.stream(stream):
                image = torch.zeros(npix * 3, device="cuda", dtype=torch.float32)
                alt.render_device(image.data_ptr(), want_stats=True, spp=min(count, 64), first_sample=first, max_depth=args.depth, seed=1)
                stream.synchronize(); t1 = time.perf_counter()
                ast = alt.render_device(image.data_ptr(), want_stats=True, spp=count, first_sample=first, max_depth=args.depth, seed=1,
                                        spp_per_pass=args.spp_per_pass)
                stream.synchronize(); dt = time.perf_counter() - t1
                del image
            line["handed_over_tree_as_is"] = {"value": round((ast.rays_primary + ast.rays_bounce) / dt / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt * 1e3, 3)}
            alt.close()
        if world == 1 and args.config == "teapot" and args.cpu_spp > 0:
            line["parity_rays"] = parity_ray_rates(scene, desc)
        if world == 1 and args.cpu_spp > 0:
            cb, cpu_img = cpu_baseline(desc, args.cpu_spp, args.depth, 1, args.spp)
            line["cpu_baseline"] = cb
            # sanity: the CPU sample is the first cpu_spp samples of the same frame -> must agree with the GPU's samples 1..cpu_spp
            chk = scene.render(spp=args.cpu_spp, max_depth=args.depth, seed=1)["rgb"]
            line["cpu_baseline"]["image_rel_rmse_vs_gpu"] = float(np.sqrt(((chk - cpu_img) ** 2).mean()) / cpu_img.mean())
    scene.close()
    if rank == 0:
        if world == 1 and full_default and not args.no_others:
            # the other single-GPU BASELINE configs at their full sizes, one step each: driver-visible
            line["other_configs"] = {name: other_config_line(name, ctx, stream, spp) for name, spp in (("soup", 0), ("instanced", 512), ("spheres", 0))}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
