"""hpt_render_device_noalloc (include/hpt.h): the variant of hpt_render_device whose pass buffers come from the library's
static device array instead of hipMalloc.  Through the C ABI, against hpt_render_device on the same inputs: bit-identical
sums (same kernels, same pass order), on two calls in a row, the second with a smaller input -- the array keeps the
first call's contents, so anything that must start at zero has to be cleared by the call itself.  Collected last
(test_zz_*): a failure here must not hide the other GPU tests behind `pytest -x`."""
import numpy as np
import pytest

from conftest import host_tree, resized

pytestmark = pytest.mark.gpu


def _both(sc, torch, **kw):
    n = sc.width * sc.height * 3
    a = torch.full((n,), -7.0, device="cuda"); b = torch.full((n,), 9.0, device="cuda")     # outputs start dirty: the calls clear them
    torch.cuda.synchronize()
    sa = sc.render_device(a.data_ptr(), want_stats=True, **kw)
    sb = sc.render_device_noalloc(b.data_ptr(), want_stats=True, **kw)
    sc.ctx.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), sa, sb


def test_noalloc_matches_render_device_twice(ctx, desc):
    import torch
    import hpt
    big = hpt.Scene(ctx, resized(desc, 800, 600), host_bvh=host_tree)      # the reference's own frame size: 480 000 of 524 288 pixels
    small = hpt.Scene(ctx, resized(desc, 96, 54), host_bvh=host_tree)
    try:
        # first call: the largest frame, three passes of one sample; counters on in the second round
        a, b, sa, sb = _both(big, torch, spp=3, max_depth=8, seed=11)
        np.testing.assert_array_equal(a, b)
        assert sb.passes == 3 and sa.rays_bounce == sb.rays_bounce and sa.rays_primary == sb.rays_primary == 800 * 600 * 3
        # second call, smaller input, same device array (stale queue contents, counters and radiance of the big frame)
        a, b, sa, sb = _both(small, torch, spp=70, first_sample=5, max_depth=6, seed=4, flags=hpt.RENDER_COUNTERS)
        np.testing.assert_array_equal(a, b)
        assert a.sum() > 0 and sb.passes == 2      # 64 + 6 samples of every pixel per pass
        assert sa.rays_bounce == sb.rays_bounce
        # (box steps depend on how the append atomics of k_shade order the queue -- culling bounds are refreshed per batch --
        #  so the counters agree only roughly between two runs; they must have counted from zero, though)
        assert 0 < sb.nodes_visited < 1.2 * sa.nodes_visited and 0 < sb.tris_tested < 1.2 * sa.tris_tested
        # and the big frame again: nothing of the small call leaks into it
        a2, b2, _, _ = _both(big, torch, spp=3, max_depth=8, seed=11)
        np.testing.assert_array_equal(a2, b2)
        # without statistics the call only enqueues on the context stream, like hpt_render_device
        out = torch.zeros(96 * 54 * 3, device="cuda")
        small.render_device_noalloc(out.data_ptr(), spp=70, first_sample=5, max_depth=6, seed=4)
        ctx.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), b)
    finally:
        big.close(); small.close()


def test_noalloc_limits(ctx, desc):
    import torch
    import hpt
    sc = hpt.Scene(ctx, resized(desc, 1024, 768), host_bvh=host_tree)      # 786 432 pixels > 2^19
    out = torch.zeros(1024 * 768 * 3, device="cuda")
    with pytest.raises(hpt.HptError, match="exceeds the static workspace"):
        sc.render_device_noalloc(out.data_ptr(), spp=1)
    sc.close()
    sc = hpt.Scene(ctx, resized(desc, 64, 48), host_bvh=host_tree)
    with pytest.raises(hpt.HptError, match="samples per call exceed"):
        sc.render_device_noalloc(out.data_ptr(), spp=4097)
    with pytest.raises(hpt.HptError, match="fixed-spp only"):
        sc.render_device_noalloc(out.data_ptr(), adaptive=True)
    with pytest.raises(hpt.HptError, match="HPT_RENDER_TIME_KERNELS"):
        sc.render_device_noalloc(out.data_ptr(), spp=1, flags=hpt.RENDER_TIME_KERNELS)
    sc.close()
