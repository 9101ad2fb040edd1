"""N > 1 path on CPU: world_size-2 gloo.  Each rank renders its sample range of every pixel (here with the CPU oracle
standing in for the device, as the checker), the accumulators are sum-reduced, and the result must equal the
single-process render of all samples: the set of paths does not depend on the number of ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, resized


def test_sample_range_partition():
    from hpt.dist import sample_range
    for spp in (1, 7, 16, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, n = sample_range(r, world, spp)
                seen += list(range(lo, lo + n))
            assert seen == list(range(spp))
            sizes = [sample_range(r, world, spp)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sample_range(2, 2, 8)


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hpt import scene_io
    from hpt.dist import sample_range, reduce_sum
    import oracle_lib as ol
    desc = resized(scene_io.load_scene(os.path.join(GOLDEN, "scene.xml")), 32, 24)
    sc = ol.OracleScene(ol.load(), desc)
    lo, n = sample_range(rank, world, spp)
    part = sc.render(spp=n, first_sample=lo, max_depth=4, seed=9, threads=2)["rgb"] * np.float32(n)   # per-rank SUM
    acc = torch.from_numpy(part.copy())
    reduce_sum(acc, dst=0)
    if rank == 0:
        np.save(out_path, acc.numpy() / np.float32(spp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_matches_single(tmp_path, olib, desc):
    import oracle_lib as ol
    spp, port = 8, 29000 + os.getpid() % 2000
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, spp, out), nprocs=2, join=True)
    two = np.load(out)
    one = ol.OracleScene(olib, resized(desc, 32, 24)).render(spp=spp, max_depth=4, seed=9)["rgb"]
    np.testing.assert_allclose(two, one, rtol=2e-6, atol=1e-6)   # same paths; only the f32 summation grouping differs
