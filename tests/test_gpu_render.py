"""GPU parity of the wavefront path tracer (raygen / extend / shade / resolve through the C ABI) against the oracle's
restatement of Scene::intersect, Scene::cast and the main.rs pixel loop, with the SAME counter-based RNG on both sides.

Tolerances (north_star check 2, "stated per-pixel RMSE"):
  * closest-hit records (node, prim, t, side, pos, normal): bit-exact;  sphere uv: 2 ulp (atan2f/asinf differ per libm)
  * per-sample radiance: >= 99.9 % of samples within 1e-3 relative (forward throughput product vs the recursion's nested
    products, device sinf/cosf/asinf); the rest are paths where a last-bit difference flips a discrete decision
  * image at matched spp, same RNG: RMSE <= 1e-3 x mean luminance
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLDEN, host_tree, resized

pytestmark = pytest.mark.gpu


def rel_rmse(a, b):
    return float(np.sqrt(((a - b) ** 2).mean()) / b.mean())


def test_scene_intersect_exact(gscene, oscene):
    from golden.make_golden import scene_rays
    o, d = scene_rays(200_000, seed=11)
    # add the primary rays of a coarse pixel grid (coherent, grazing the walls)
    pr = [oscene.primary_ray(x, y, 1, 1) for x in range(0, 800, 16) for y in range(0, 600, 16)]
    o = np.concatenate([o, np.array([p[0] for p in pr], np.float32)]); d = np.concatenate([d, np.array([p[1] for p in pr], np.float32)])
    g = gscene.intersect(o, d); r = oscene.intersect(o, d)
    for k in ("node", "prim", "t", "side", "pos", "normal"):
        np.testing.assert_array_equal(g[k], r[k])
    np.testing.assert_allclose(g["uvw"], r["uvw"], rtol=0, atol=3e-7)
    assert set(np.unique(r["node"])) >= {0, 2, 3, 4, 5, 6, 7, 8, 9}     # every object of scene.xml is hit by some ray


def test_golden_scene_hits(gscene):
    from golden.make_golden import scene_rays
    g0 = np.load(os.path.join(GOLDEN, "scene_hits_2048.npz"))
    o, d = scene_rays(2048)
    g = gscene.intersect(o, d)
    for k in ("node", "prim", "t", "side", "pos", "normal"):
        np.testing.assert_array_equal(g[k], g0[k])


def test_per_sample_radiance(ctx, olib, desc):
    import hpt
    d = resized(desc, 96, 72)
    osc = ol.OracleScene(olib, d); gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    spp = 8
    ref = osc.render(spp=spp, max_depth=8, seed=1, want_samples=True)
    # HPT_RENDER_EXACT_SHADE: the bounce in the reference's own f32 operations -> every path as long as the oracle's.
    # Default: 1-ulp forms of the same formulas (include/hpt.h) -> a path differs only where a roulette / lobe / hit decision
    # sits within a few ulp; tolerance: at most 1e-4 of the segments, and the same per-sample bar.
    for flags in (hpt.RENDER_EXACT_SHADE, 0):
        got = gsc.render_samples(spp=spp, max_depth=8, seed=1, spp_per_pass=3, flags=flags)      # ragged passes: 3 + 3 + 2
        rel = np.abs(got - ref["per_sample"]).max(axis=2) / (np.abs(ref["per_sample"]).max(axis=2) + 1e-3)
        assert (rel <= 1e-3).mean() >= 0.999
        st = gsc.render(spp=spp, max_depth=8, seed=1, flags=flags)["stats"]
        if flags: assert st.rays_primary + st.rays_bounce == ref["stats"].rays             # same number of segments traced
        else: assert abs(st.rays_primary + st.rays_bounce - ref["stats"].rays) <= 1e-4 * ref["stats"].rays
        assert st.rays_primary == 96 * 72 * spp
    gsc.close()


def test_image_parity_and_golden(ctx, olib, desc):
    import hpt
    d = resized(desc, 64, 48)
    gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    img = gsc.render(spp=16, max_depth=8, seed=1)["rgb"]
    gold = np.load(os.path.join(GOLDEN, "image_64x48_16spp.npy"))
    assert rel_rmse(img, gold) <= 1e-3
    # spp_per_pass must not change the result (per-pixel sums run in sample order across passes)
    img2 = gsc.render(spp=16, max_depth=8, seed=1, spp_per_pass=5)["rgb"]
    np.testing.assert_array_equal(img, img2)
    # sample ranges compose (what the multi-GPU split relies on)
    a = gsc.render(spp=8, max_depth=8, seed=1, first_sample=0)["rgb"]
    b = gsc.render(spp=8, max_depth=8, seed=1, first_sample=8)["rgb"]
    np.testing.assert_allclose((a + b) / 2, img, rtol=3e-6, atol=1e-6)
    # a different seed is a different (but statistically equal) image
    other = gsc.render(spp=16, max_depth=8, seed=2)["rgb"]
    assert rel_rmse(other, gold) > 1e-2 and abs(other.mean() / gold.mean() - 1) < 0.05
    gsc.close()


def test_depth_caps(ctx, olib, desc):
    import hpt
    d = resized(desc, 48, 36)
    osc = ol.OracleScene(olib, d); gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    for md in (0, 1, 16):
        r = osc.render(spp=4, max_depth=md, seed=5)
        g = gsc.render(spp=4, max_depth=md, seed=5, flags=hpt.RENDER_EXACT_SHADE)
        assert g["stats"].rays_primary + g["stats"].rays_bounce == r["stats"].rays
        assert rel_rmse(g["rgb"], r["rgb"]) <= 1e-3
        g = gsc.render(spp=4, max_depth=md, seed=5)      # default shade arithmetic: see test_per_sample_radiance
        assert abs(g["stats"].rays_primary + g["stats"].rays_bounce - r["stats"].rays) <= max(1e-4 * r["stats"].rays, 0 if md == 0 else 2)
        assert rel_rmse(g["rgb"], r["rgb"]) <= 2e-3
    gsc.close()


def test_adaptive_mode(ctx, olib, desc):
    """main.rs:67-107: 8 -> 16 -> 32 samples where the per-channel contrast test fires"""
    import hpt
    d = resized(desc, 32, 24)
    gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    g = gsc.render(adaptive=True, initial=8, max_samples=32, max_depth=8, seed=1)
    gold = np.load(os.path.join(GOLDEN, "adaptive_32x24.npz"))
    same = g["samples"] == gold["samples"]
    assert same.mean() >= 0.99 and set(np.unique(g["samples"])) <= {8, 16, 32}
    np.testing.assert_allclose(g["rgb"][same], gold["rgb"][same], rtol=2e-3, atol=1e-4)
    assert g["stats"].samples == int(g["samples"].sum())
    # reference constants (128..512) run too
    g2 = gsc.render(adaptive=True, max_depth=2, seed=1)
    assert set(np.unique(g2["samples"])) <= {128, 256, 512}
    gsc.close()


def test_textures_environment_dof(ctx, olib):
    """checkerboard + image textures, background / environment maps, lens sampling, glossy + refractive lobes"""
    import hpt
    from hpt import scene_io
    f32 = np.float32
    I = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], f32)
    rng = np.random.default_rng(3)
    d = scene_io.SceneDesc()
    d.images.append(rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8))
    chk = scene_io.Texture(kind=1, color=np.array([1, 1, 1], f32), color1=np.array([.9, .1, .1], f32), color2=np.array([.1, .1, .9], f32),
                           m=(I * f32(0.25)), t=np.array([.1, .2, 0], f32))
    img = scene_io.Texture(kind=2, image=0, color=np.array([1, .9, .8], f32), m=I.copy(), t=np.zeros(3, f32))
    d.materials.append(scene_io.Material(name="floor", diffuse=chk))
    d.materials.append(scene_io.Material(name="ball", diffuse=img, reflection=scene_io.tex_color((.3, .3, .3)), reflection_glossiness=0.2))
    d.materials.append(scene_io.Material(name="glass", diffuse=scene_io.tex_color((0, 0, 0)), refraction=scene_io.tex_color((.9, .9, .9)),
                                         reflection=scene_io.tex_color((.5, .5, .5)), refraction_index=1.5, refraction_glossiness=0.05))
    d.nodes.append(scene_io.Node(parent=-1, geom=2, mesh=-1, material=0, m=I * f32(20), t=np.zeros(3, f32)))
    d.nodes.append(scene_io.Node(parent=-1, geom=0, mesh=-1, material=-1, m=I.copy(), t=np.array([0, 0, 1], f32)))
    d.nodes.append(scene_io.Node(parent=1, geom=1, mesh=-1, material=1, m=I.copy(), t=np.array([-1.2, 0, 0], f32)))
    d.nodes.append(scene_io.Node(parent=1, geom=1, mesh=-1, material=2, m=np.array([.8, 0, 0, 0, .8, 0, 0, 0, 1.1], f32), t=np.array([1.2, 0, .1], f32)))
    d.background = scene_io.Texture(kind=2, image=0, color=np.array([1, 1, 1], f32), m=I.copy(), t=np.zeros(3, f32))
    d.environment = scene_io.Texture(kind=1, color=np.array([2, 2, 2], f32), color1=np.array([1, 1, .5], f32), color2=np.array([.2, .3, 1], f32), m=I.copy(), t=np.zeros(3, f32))
    pos = np.array([0, -6, 2.5], f32); dr = scene_io.normalize(np.array([0, 0, 1], f32) - pos)
    d.camera = scene_io.Camera(pos=pos, dir=dr, up=scene_io.cross(scene_io.cross(dr, np.array([0, 0, 1], f32)), dr), fov=40.0, width=80, height=60, focaldist=6.0, dof=0.05)
    osc = ol.OracleScene(olib, d); gsc = hpt.Scene(ctx, d)
    ref = osc.render(spp=8, max_depth=6, seed=2, want_samples=True)
    got = gsc.render_samples(spp=8, max_depth=6, seed=2)
    rel = np.abs(got - ref["per_sample"]).max(axis=2) / (np.abs(ref["per_sample"]).max(axis=2) + 1e-3)
    assert (rel <= 1e-3).mean() >= 0.995
    g = gsc.render(spp=8, max_depth=6, seed=2)
    assert g["stats"].rays_primary + g["stats"].rays_bounce == ref["stats"].rays or abs(g["stats"].rays_primary + g["stats"].rays_bounce - ref["stats"].rays) < 50
    assert rel_rmse(g["rgb"], ref["rgb"]) <= 5e-3
    assert ref["rgb"].min() >= -5 and (ref["rgb"] != 0).mean() > 0.9      # open scene: paths escape to the environment
    gsc.close()


def test_render_device_matches_host_entry(ctx, desc):
    import torch
    import hpt
    d = resized(desc, 64, 48)
    gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    out = torch.zeros(64 * 48 * 3, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()      # torch filled `out` on ITS stream; the library works on the context's own (nothing orders the two)
    st = gsc.render_device(out.data_ptr(), want_stats=True, spp=8, max_depth=8, seed=1, flags=hpt.RENDER_TIME_KERNELS | hpt.RENDER_COUNTERS)
    img = gsc.render(spp=8, max_depth=8, seed=1)["rgb"]
    np.testing.assert_array_equal((out.cpu().numpy().reshape(48, 64, 3) / np.float32(8)), img)
    assert st.seconds_extend > 0 and st.launches_extend == 9 and st.nodes_visited > 0 and 0 < st.inst_tests <= 9 * (st.rays_primary + st.rays_bounce)
    gsc.close()


def test_adaptive_rounds_api(ctx, olib, desc):
    """hpt_render_accum_device / hpt_adaptive_select_device (the per-rank pieces of a multi-GPU adaptive render):
    driven by hpt.dist.render_adaptive on one rank they reproduce hpt_render's adaptive mode exactly; a round split
    into two sample sub-ranges gives the same sums and min/max; scene info is reported."""
    import torch
    import hpt
    from hpt.dist import DeviceRounds, render_adaptive
    d = resized(desc, 64, 48)
    sc = hpt.Scene(ctx, d, host_bvh=host_tree)
    info = sc.info()
    assert info.n_instances == 9 and info.n_mesh_instances == 1 and info.n_triangles == 6320 and info.uses_tlas == 0
    assert info.n_nodes == 6319 and info.device_bytes > 6319 * 64 + 6320 * 48
    ref = sc.render(adaptive=True, initial=8, max_samples=32, max_depth=4, seed=3)
    with pytest.raises(RuntimeError):
        DeviceRounds(sc, max_depth=4, seed=3)          # default stream: no handle to share
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    try:
        _adaptive_rounds_body(ctx, olib, d, sc, ref, stream)
    finally:      # whatever happens, the session context goes back to its own stream
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
        ctx.set_stream(0)
        sc.close()


def _adaptive_rounds_body(ctx, olib, d, sc, ref, stream):
    import torch
    import hpt
    from hpt.dist import DeviceRounds, render_adaptive
    eng = DeviceRounds(sc, max_depth=4, seed=3)
    rgb, spp = render_adaptive(eng, 0, 1, initial_samples=8, max_samples=32)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(spp.cpu().numpy().reshape(48, 64), ref["samples"])
    assert len(np.unique(ref["samples"])) >= 2
    np.testing.assert_array_equal(rgb.cpu().numpy().reshape(48, 64, 3), ref["rgb"])
    assert eng.rays == ref["stats"].rays_primary + ref["stats"].rays_bounce
    # one round of 8 samples == samples 0..3 then 4..7 into the same buffers, on a pixel subset
    npix = 64 * 48
    plist = torch.arange(5, npix, 7, dtype=torch.int32, device="cuda")
    bufs = []
    for split in ((8,), (4, 4), (3, 5)):
        acc, mm, _ = eng.new_buffers(8)
        first = 0
        for cnt in split:
            sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), plist.data_ptr(), len(plist), spp=cnt, first_sample=first, max_depth=4, seed=3,
                                   flags=hpt.RENDER_EXACT_SHADE)      # (the reference's own bounce arithmetic for the oracle comparisons below)
            first += cnt
        torch.cuda.synchronize()
        bufs.append((acc.cpu().numpy(), mm.cpu().numpy()))
    for acc, mm in bufs[1:]:
        np.testing.assert_array_equal(acc, bufs[0][0]); np.testing.assert_array_equal(mm, bufs[0][1])
    touched = np.zeros(npix, bool); touched[plist.cpu().numpy()] = True
    assert (bufs[0][0][~touched] == 0).all() and np.isinf(bufs[0][1][~touched, 0]).all()
    # against the oracle's samples: per-pixel sums of samples 1..8 (fixed point, 2^-24 quantum; no NaN sample in this scene)
    per = ol.OracleScene(olib, d).render(spp=8, max_depth=4, seed=3, want_samples=True)["per_sample"]
    want = per[touched].astype(np.float64).sum(axis=1)
    assert (bufs[0][0][:, 3] == 0).all()
    np.testing.assert_allclose(bufs[0][0][touched, :3] / 2.0 ** 24, want, rtol=1e-5, atol=1e-6)
    # ... and of the GPU's OWN samples bit for bit (tests/accum_ref.py restates the accumulator), resolved like the library does on the host
    import accum_ref
    own = sc.render_samples(spp=8, max_depth=4, seed=3, flags=hpt.RENDER_EXACT_SHADE)
    mine = accum_ref.accumulate(np.zeros((npix, 4), dtype=np.int64), own[touched], rows=np.flatnonzero(touched))
    np.testing.assert_array_equal(bufs[0][0], mine)
    np.testing.assert_array_equal(hpt.accum_resolve(mine, num_samples=8), accum_ref.resolve(mine, num_samples=8))
    np.testing.assert_allclose(bufs[0][1][touched, :3], per[touched].min(axis=1), rtol=2e-5, atol=1e-6)      # (libm vs ocml sin / cos / asin: a few ulp per bounce)
    np.testing.assert_allclose(bufs[0][1][touched, 3:], per[touched].max(axis=1), rtol=2e-5, atol=1e-6)
    acc, mm, _ = eng.new_buffers(8)
    with pytest.raises(hpt.HptError):
        sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), 0, 0, spp=4, adaptive=True)


def test_progressive_checkpoint_resume(ctx, desc, tmp_path):
    """checkpoint = accumulator + sample cursor: 5 + (save, new object, load) + 7 + 4 samples == one render of 16"""
    import hpt
    from hpt.progressive import ProgressiveRender
    d = resized(desc, 48, 36)
    sc = hpt.Scene(ctx, d, host_bvh=host_tree)
    want = sc.render(spp=16, max_depth=5, seed=11)["rgb"]
    a = ProgressiveRender(sc, max_depth=5, seed=11)
    a.advance(5)
    ck = str(tmp_path / "ck.npz")
    a.save(ck)
    b = ProgressiveRender(sc, max_depth=5, seed=11)
    assert b.load(ck) == 5
    b.advance(7); b.advance(4)
    np.testing.assert_array_equal(b.image(), want)
    c = ProgressiveRender(sc, max_depth=5, seed=12)
    with pytest.raises(ValueError):
        c.load(ck)
    for r in (a, b, c):
        r.close()
    sc.close()


def test_full_size_properties(ctx, desc):
    """BASELINE config-2 size (1920x1080, depth 8), where the oracle would take minutes: size-independent properties.
    (1) linearity: doubling every emission doubles every sample exactly (x2 is exact in f32, every path is unchanged), so the
        fixed-point sums double up to the rounding of each sample to the 2^-24 quantum: |acc(2E) - 2 acc(E)| <= spp;
    (2) additivity over sample ranges: sums of samples [0,2) + [2,5) == sums of [0,5), bit for bit;
    (3) ray accounting: primary rays = W*H*spp, bounce rays <= primary * depth, and equal between the runs."""
    import copy
    import torch
    import hpt
    d = resized(desc, 1920, 1080)
    d2 = copy.deepcopy(d)
    for m in d2.materials:
        m.emission = (m.emission * np.float32(2.0)).astype(np.float32)
    sc, sc2 = hpt.Scene(ctx, d, host_bvh=host_tree), hpt.Scene(ctx, d2, host_bvh=host_tree)
    npix = 1920 * 1080
    a, b, c = (torch.zeros((npix, 4), dtype=torch.int64, device="cuda") for _ in range(3))      # hpt_accum records
    img, img_c = torch.zeros(npix * 3, device="cuda"), torch.zeros(npix * 3, device="cuda")
    torch.cuda.synchronize()      # (torch's fills run on torch's stream, the library on the context's)
    st_a = sc.render_accum_device(a.data_ptr(), 0, 0, npix, want_stats=True, spp=5, max_depth=8, seed=2)
    st_b = sc2.render_accum_device(b.data_ptr(), 0, 0, npix, want_stats=True, spp=5, max_depth=8, seed=2)
    assert st_a.rays_primary == npix * 5 and st_a.rays_bounce <= st_a.rays_primary * 8
    assert (st_a.rays_primary, st_a.rays_bounce) == (st_b.rays_primary, st_b.rays_bounce)
    assert int((b[:, :3] - 2 * a[:, :3]).abs().max()) <= 5 and int(a[:, :3].sum()) > 0 and int(a[:, 3].sum()) == 0
    sc.render_accum_device(c.data_ptr(), 0, 0, npix, want_stats=True, spp=2, first_sample=0, max_depth=8, seed=2)
    sc.render_accum_device(c.data_ptr(), 0, 0, npix, want_stats=True, spp=3, first_sample=2, max_depth=8, seed=2)
    assert torch.equal(a, c)
    # the float entry returns exactly these sums, rounded once
    sc.render_device(img.data_ptr(), want_stats=True, spp=5, max_depth=8, seed=2)
    sc.accum_resolve_device(c.data_ptr(), img_c.data_ptr()); ctx.synchronize()
    assert torch.equal(img, img_c)
    sc.close(); sc2.close()


def test_independent_rng_image_parity(ctx, olib, desc):
    """SURVEY 8d.7, independent-RNG mode -- the stand-in for the reference's OS-seeded rand: the GPU image with ITS OWN
    seed must be as close to a CPU render as two CPU renders with different seeds are to each other:
    RMSE(GPU, CPU_A) <= 1.25 x RMSE(CPU_A, CPU_B), and the per-channel mean signed error <= 0.5 % of the mean."""
    import hpt
    d = resized(desc, 128, 96)
    osc = ol.OracleScene(olib, d)
    cpu_a = osc.render(spp=128, max_depth=8, seed=101)["rgb"]
    cpu_b = osc.render(spp=128, max_depth=8, seed=202)["rgb"]
    sc = hpt.Scene(ctx, d, host_bvh=host_tree)
    gpu = sc.render(spp=128, max_depth=8, seed=303)["rgb"]
    sc.close()
    rmse = lambda x, y: float(np.sqrt(((x - y) ** 2).mean()))
    noise = rmse(cpu_a, cpu_b)
    assert noise > 0 and rmse(gpu, cpu_a) <= 1.25 * noise and rmse(gpu, cpu_b) <= 1.25 * noise
    ref_mean = 0.5 * (cpu_a + cpu_b).reshape(-1, 3).mean(axis=0)
    signed = gpu.reshape(-1, 3).mean(axis=0) - ref_mean
    assert (np.abs(signed) <= 0.005 * ref_mean).all(), (signed, ref_mean)


def test_passes_in_flight_do_not_change_results(ctx, desc):
    """render_range keeps two wavefront passes in flight on two streams (pass buffers of their own; the per-pixel sums are
    chained in sample order through events): images, per-pixel min/max (adaptive sample counts) and ray totals
    must be bit-identical to the one-stream run, for even and odd numbers of passes, up to 4 in flight."""
    import torch
    import hpt
    d = resized(desc, 96, 64)
    gsc = hpt.Scene(ctx, d, host_bvh=host_tree)
    out = [torch.zeros(96 * 64 * 3, device="cuda", dtype=torch.float32) for _ in range(2)]
    torch.cuda.synchronize()
    try:
        for spp, per_pass in ((8, 4), (12, 4), (13, 3), (5, 1), (3, 4)):
            ctx.set_passes_in_flight(1)
            a = gsc.render_device(out[0].data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=3, spp_per_pass=per_pass, flags=hpt.RENDER_COUNTERS)
            for n in (2, 3, 4):
                ctx.set_passes_in_flight(n)
                out[1].fill_(-1.0)
                torch.cuda.synchronize()      # torch's fill before the library's memset + resolves on another stream
                b = gsc.render_device(out[1].data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=3, spp_per_pass=per_pass, flags=hpt.RENDER_COUNTERS)
                torch.cuda.synchronize()
                assert torch.equal(out[0], out[1]), (spp, per_pass, n)
                assert (a.rays_primary, a.rays_bounce, a.passes) == (b.rays_primary, b.rays_bounce, b.passes)
                # work counters are not results: how many boxes a ray visits depends on when its wave refreshes the culling bound
                assert abs(a.nodes_visited - b.nodes_visited) <= 0.01 * a.nodes_visited and abs(a.tris_tested - b.tris_tested) <= 0.01 * a.tris_tested
        ctx.set_passes_in_flight(1)
        g1 = gsc.render(adaptive=True, initial=8, max_samples=32, max_depth=8, seed=1, spp_per_pass=2)
        ctx.set_passes_in_flight(2)
        g2 = gsc.render(adaptive=True, initial=8, max_samples=32, max_depth=8, seed=1, spp_per_pass=2)
        np.testing.assert_array_equal(g1["samples"], g2["samples"])
        np.testing.assert_array_equal(g1["rgb"], g2["rgb"])
    finally:
        ctx.set_passes_in_flight(2)
        gsc.close()


def test_fused_analytic_stage_equals_separate_kernels(ctx, desc, monkeypatch):
    """instance-list scenes: k_raygen / k_shade run the analytic stage of the ray they make (hit records per depth parity); the
    separate k_extend_a launches (HPT_FUSE_ANALYTIC=0 at commit) must leave the same bits -- accumulators, per-pixel min/max,
    ray and test counts -- in both shade modes, with ragged passes and a pixel list."""
    import torch
    import hpt
    d = resized(desc, 80, 60)
    npix = 80 * 60
    plist = torch.arange(3, npix, 5, dtype=torch.int32, device="cuda")
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("HPT_FUSE_ANALYTIC", fuse)
        sc = hpt.Scene(ctx, d, host_bvh=host_tree)
        got = []
        for flags in (0, hpt.RENDER_EXACT_SHADE):
            acc = torch.zeros((npix, 4), dtype=torch.int64, device="cuda")
            mm = torch.empty((npix, 6), dtype=torch.float32, device="cuda"); mm[:, :3] = float("inf"); mm[:, 3:] = 0.0
            torch.cuda.synchronize()
            st = sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), 0, npix, want_stats=True, spp=7, max_depth=8, seed=11, spp_per_pass=3,
                                        flags=flags | hpt.RENDER_COUNTERS)
            st2 = sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), plist.data_ptr(), len(plist), want_stats=True, spp=2, first_sample=7, max_depth=3, seed=11,
                                         flags=flags)
            torch.cuda.synchronize()
            got.append((acc.cpu().numpy(), mm.cpu().numpy(), (st.rays_primary, st.rays_bounce, st.inst_tests, st.rays_mesh), (st2.rays_primary, st2.rays_bounce, st2.rays_mesh)))
        res.append(got)
        sc.close()
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2] and a[3] == b[3]
    rays = res[0][0][2][0] + res[0][0][2][1]
    assert int(res[0][0][0][:, :3].sum()) > 0 and 8 * rays <= res[0][0][2][2] <= 9 * rays      # every ray met the eight analytic instances (+ the mesh box)


def test_binned_mesh_queue_changes_no_result(ctx, desc, monkeypatch):
    """big trees walk the mesh queue of depths >= 1 in the order of the cell the ray starts in (k_mq_hist + k_mq_scatter -> DPass.mq_bin, one
    linear queue): the same rays, so accumulators, per-pixel min/max and ray counts keep their bits -- forced on here for a small scene
    (HPT_MQ_BIN=2) against off (0), in both shade modes, with ragged passes, a pixel list and the separate analytic kernels too."""
    import torch
    import hpt
    d = resized(desc, 80, 60)
    npix = 80 * 60
    plist = torch.arange(1, npix, 3, dtype=torch.int32, device="cuda")
    res = []
    for fuse, binned in (("1", "0"), ("1", "2"), ("0", "2")):
        monkeypatch.setenv("HPT_FUSE_ANALYTIC", fuse); monkeypatch.setenv("HPT_MQ_BIN", binned)
        sc = hpt.Scene(ctx, d, host_bvh=host_tree)
        got = []
        for flags in (0, hpt.RENDER_EXACT_SHADE):
            acc = torch.zeros((npix, 4), dtype=torch.int64, device="cuda")
            mm = torch.empty((npix, 6), dtype=torch.float32, device="cuda"); mm[:, :3] = float("inf"); mm[:, 3:] = 0.0
            torch.cuda.synchronize()
            st = sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), 0, npix, want_stats=True, spp=7, max_depth=8, seed=5, spp_per_pass=3, flags=flags)
            st2 = sc.render_accum_device(acc.data_ptr(), mm.data_ptr(), plist.data_ptr(), len(plist), want_stats=True, spp=2, first_sample=7, max_depth=4, seed=5, flags=flags)
            torch.cuda.synchronize()
            got.append((acc.cpu().numpy(), mm.cpu().numpy(), (st.rays_primary, st.rays_bounce, st.rays_mesh), (st2.rays_primary, st2.rays_bounce, st2.rays_mesh)))
        res.append(got)
        sc.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
            assert a[2] == b[2] and a[3] == b[3]
    assert int(res[0][0][0][:, :3].sum()) > 0 and res[0][0][2][2] > 0
