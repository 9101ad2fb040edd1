"""The multi-GPU path of the PRODUCT on a GPU: sample-range split + sum-reduce of the accumulators.

(a) one process per GPU, as bench.py runs under torch.distributed.run: two fresh child processes share GPU 0 (gloo
    rendezvous -- RCCL cannot put two ranks on one device), each with its own hpt.Context, rendering its sample_range
    through hpt_render_accum_device, and the adaptive rounds through hpt.dist.DeviceRounds / render_adaptive.
(b) one process, several GPUs: hpt_group_create / hpt_group_render behind the C ABI -- with RCCL (communicator of one: the
    1-GPU box has one device) and with the peer-copy transport over two and three ranks on GPU 0.
Reference for both: the one-context render of all samples.  Same set of paths for any rank count (RNG keyed by pixel and
sample) and integer sums (include/hpt.h hpt_accum): images and adaptive sample counts are IDENTICAL, bit for bit."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, host_tree, resized

pytestmark = pytest.mark.gpu
W, H = 64, 48


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hpt
    from hpt import host_bvh, scene_io
    from hpt.dist import DeviceRounds, reduce_sum, render_adaptive, sample_range
    torch.cuda.set_device(0)
    desc = resized(scene_io.load_scene(os.path.join(GOLDEN, "scene.xml")), W, H)
    ctx = hpt.Context(0)
    sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx.set_stream(stream.cuda_stream)
        # fixed spp: this rank's contiguous share of 11 samples (uneven split) of every pixel
        lo, n = sample_range(rank, world, 11)
        accum = torch.zeros((W * H, 4), dtype=torch.int64, device="cuda")      # hpt_accum records
        fixed = torch.zeros(W * H * 3, device="cuda")
        st = sc.render_accum_device(accum.data_ptr(), 0, 0, W * H, want_stats=True, spp=n, first_sample=lo, max_depth=6, seed=9)
        rays = torch.tensor([float(st.rays_primary + st.rays_bounce)], dtype=torch.float64)
        dist.all_reduce(rays)
        reduce_sum(accum, dst=0)
        sc.accum_resolve_device(accum.data_ptr(), fixed.data_ptr(), num_samples=11)
        # adaptive: rounds in lock step, min/max planes all-reduced, identical active sets
        eng = DeviceRounds(sc, max_depth=6, seed=9)
        rgb, spp = render_adaptive(eng, rank, world, initial_samples=8, max_samples=32)
        stream.synchronize()
        if rank == 0:
            np.savez(out_path, fixed=fixed.cpu().numpy(), rays=rays.numpy(), rgb=rgb.cpu().numpy(), spp=spp.cpu().numpy())
    ctx.set_stream(0)
    sc.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_gpu(tmp_path, ctx, desc):
    import hpt
    import torch.multiprocessing as mp
    out = str(tmp_path / "two.npz")
    mp.spawn(_worker, args=(2, 33000 + os.getpid() % 2000, out), nprocs=2, join=True)
    two = np.load(out)
    sc = hpt.Scene(ctx, resized(desc, W, H), host_bvh=host_tree)
    one = sc.render(spp=11, max_depth=6, seed=9)
    np.testing.assert_array_equal(two["fixed"].reshape(H, W, 3), one["rgb"])
    assert int(two["rays"][0]) == one["stats"].rays_primary + one["stats"].rays_bounce
    ad = sc.render(adaptive=True, initial=8, max_samples=32, max_depth=6, seed=9)
    np.testing.assert_array_equal(two["spp"].reshape(H, W), ad["samples"])
    assert len(np.unique(ad["samples"])) >= 2
    np.testing.assert_array_equal(two["rgb"].reshape(H, W, 3), ad["rgb"])
    sc.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_group_render_behind_the_c_abi(ctx, desc, devices):
    """hpt_group_render == hpt_render: [0] runs the RCCL path (ncclCommInitAll over one GPU, ncclReduce), [0, 0] and
    [0, 0, 0] the peer-copy transport with two / three ranks, each rendering on a host thread of its own"""
    import hpt
    d = resized(desc, W, H)
    ref = hpt.Scene(ctx, d, host_bvh=host_tree)
    one = ref.render(spp=10, first_sample=3, max_depth=6, seed=5)
    ad = ref.render(adaptive=True, initial=8, max_samples=32, max_depth=6, seed=5)
    ref.close()
    g = hpt.Group(devices)
    assert g.size() == len(devices)
    sc = hpt.Scene(g.context(0), d, host_bvh=host_tree)
    try:
        r = g.render(sc, spp=10, first_sample=3, max_depth=6, seed=5)
        assert r["stats"].n_devices == len(devices) and r["stats"].transport == (0 if len(devices) == 1 else 1)
        assert r["stats"].rays_primary + r["stats"].rays_bounce == one["stats"].rays_primary + one["stats"].rays_bounce
        assert (r["samples"] == 10).all()
        np.testing.assert_array_equal(r["rgb"], one["rgb"])      # any rank count: the one-context image, bit for bit
        # fewer samples than ranks: the surplus ranks contribute zeros
        few = g.render(sc, spp=2, max_depth=4, seed=5)
        np.testing.assert_array_equal(few["rgb"], hpt.Scene(ctx, d, host_bvh=host_tree).render(spp=2, max_depth=4, seed=5)["rgb"])
        a = g.render(sc, adaptive=True, initial=8, max_samples=32, max_depth=6, seed=5)
        np.testing.assert_array_equal(a["samples"], ad["samples"])
        np.testing.assert_array_equal(a["rgb"], ad["rgb"])
        # the scene committed again (another camera): the replicas follow
        d2 = resized(desc, 40, 30)
        sc.set_camera(d2.camera); sc.commit()
        again = g.render(sc, spp=4, max_depth=4, seed=2)
        ref2 = hpt.Scene(ctx, d2, host_bvh=host_tree)
        np.testing.assert_array_equal(again["rgb"], ref2.render(spp=4, max_depth=4, seed=2)["rgb"])
        ref2.close()
        # the scene destroyed and ANOTHER one built on the group's context (the allocator may well hand out the same address,
        # and its first commit is "commit number one" again): the replicas must follow -- commit stamps are process-unique
        sc.close()
        d3 = resized(desc, W, H)
        d3.nodes = [n for n in d3.nodes if n.geom != 1]      # no sphere
        sc = hpt.Scene(g.context(0), d3, host_bvh=host_tree)
        other = g.render(sc, spp=4, max_depth=4, seed=2)
        ref3 = hpt.Scene(ctx, d3, host_bvh=host_tree)
        np.testing.assert_array_equal(other["rgb"], ref3.render(spp=4, max_depth=4, seed=2)["rgb"])
        ref3.close()
    finally:
        sc.close(); g.close()


def test_group_errors(ctx, desc):
    import hpt
    with pytest.raises(hpt.HptError, match="out of range"):
        hpt.Group([0, 99])
    g = hpt.Group([0])
    foreign = hpt.Scene(ctx, resized(desc, 32, 24), host_bvh=host_tree)      # not on the group's context
    with pytest.raises(hpt.HptError, match="hpt_group_context"):
        g.render(foreign, spp=1)
    foreign.close(); g.close()
