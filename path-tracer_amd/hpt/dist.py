"""Multi-GPU partition of the render path: every (pixel, sample) path is independent (main.rs:57,72 are pure
maps), so rank r of W renders the sample range [lo, hi) of EVERY pixel -- the counter-based RNG is keyed by
(pixel, sample index), hence the set of paths does not depend on W -- and the one exchange step is a sum-reduce of
the accumulators (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).  The accumulators are 64-bit fixed point
(include/hpt.h hpt_accum; here: int64 tensors [npix, 4]): the reduced image does not depend on W in any bit."""


def sample_range(rank, world, spp):
    """Contiguous, balanced split of sample indices 0..spp-1: returns (first_sample, count) for `rank`."""
    if world <= 0 or not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank/world/spp")
    base, rem = divmod(spp, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def reduce_sum(tensor, dst=0):
    """Sum the per-rank accumulators onto rank `dst` (in place). No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if tensor.is_cuda and dist.get_backend() == "gloo":      # rehearsal on one GPU: gloo reduces host tensors only
            host = tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            tensor.copy_(host)
        else:
            dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


def allreduce_minmax(minmax):
    """Per-channel running min/max planes (npix x 6: min rgb, max rgb) combined over ranks with ONE collective:
    max over [-min, max] (exact: negation is exact, +inf -> -inf)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return minmax
    v = minmax.view(-1, 6)
    v[:, :3].neg_()
    if minmax.is_cuda and dist.get_backend() == "gloo":          # rehearsal on one GPU (see reduce_sum)
        host = minmax.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX)
        minmax.copy_(host)
    else:
        dist.all_reduce(minmax, op=dist.ReduceOp.MAX)
    v[:, :3].neg_()
    return minmax


class DeviceRounds:
    """The adaptive rounds of one rank on its GPU (hpt_render_accum_device / hpt_adaptive_select_device) behind the small
    interface render_adaptive() drives: new_buffers, render_round, select.  Work is enqueued on torch's current stream
    (the context is pointed at it) so that the collectives between rounds are ordered after the kernels."""

    def __init__(self, scene, max_depth=8, seed=1, spp_per_pass=0):
        import torch
        self.torch = torch
        self.scene, self.max_depth, self.seed, self.spp_per_pass = scene, max_depth, seed, spp_per_pass
        self.npix = scene.width * scene.height
        # The library's kernels go on torch's CURRENT stream so that collectives between rounds are ordered after them.
        # On the legacy default stream the handle is NULL, which would mean "the context's own stream": use a real one.
        if torch.cuda.current_stream().cuda_stream == 0:
            raise RuntimeError("DeviceRounds: run under `with torch.cuda.stream(torch.cuda.Stream()):` (the default stream "
                               "has no handle to share with the library)")
        scene.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.lists = [torch.empty(self.npix, dtype=torch.int32, device="cuda") for _ in range(2)]
        self.cur = -1
        self.rays = 0

    def new_buffers(self, initial_samples):
        t = self.torch
        accum = t.zeros((self.npix, 4), dtype=t.int64, device="cuda")      # hpt_accum records
        minmax = t.empty((self.npix, 6), dtype=t.float32, device="cuda")
        minmax[:, :3] = float("inf"); minmax[:, 3:] = 0.0          # main.rs:90-92
        spp = t.full((self.npix,), int(initial_samples), dtype=t.int32, device="cuda")
        return accum, minmax, spp

    def render_round(self, first_sample, count, n_active, accum, minmax):
        plist = 0 if self.cur < 0 else self.lists[self.cur].data_ptr()
        st = self.scene.render_accum_device(accum.data_ptr(), minmax.data_ptr(), plist, n_active, want_stats=True, spp=count,
                                            first_sample=first_sample, max_depth=self.max_depth, seed=self.seed, spp_per_pass=self.spp_per_pass)
        self.rays += st.rays_primary + st.rays_bounce

    def resolve(self, accum, spp):
        """main.rs:110 with the per-pixel sample counts: float32 [npix, 3] on the device"""
        out = self.torch.empty((self.npix, 3), dtype=self.torch.float32, device="cuda")
        self.scene.accum_resolve_device(accum.data_ptr(), out.data_ptr(), spp_ptr=spp.data_ptr())
        return out

    def select(self, n_active, minmax, thresholds, spp, next_num_samples):
        nxt = 0 if self.cur < 0 else 1 - self.cur
        n = self.scene.adaptive_select_device(0 if self.cur < 0 else self.lists[self.cur].data_ptr(), n_active, minmax.data_ptr(), thresholds,
                                              self.lists[nxt].data_ptr(), spp.data_ptr(), next_num_samples)
        self.cur = nxt
        return n


def render_adaptive(engine, rank=0, world=1, initial_samples=128, max_samples=512, thresholds=(0.4, 0.3, 0.6)):
    """main.rs:67-110 over `world` ranks: in every round (samples have+1..num of the still-active pixels) rank
    r renders its contiguous share of the round's samples; the min/max planes are all-reduced so that every rank runs
    the contrast test on identical data and keeps an identical active set; the (integer) sums are reduced once at the end.
    Returns (rgb [npix,3] on rank 0 else None, samples_per_px)."""
    accum, minmax, spp = engine.new_buffers(initial_samples)
    have, num, n_active = 0, int(initial_samples), engine.npix
    while n_active > 0:
        lo, cnt = sample_range(rank, world, num - have)
        if cnt > 0:
            engine.render_round(have + lo, cnt, n_active, accum, minmax)
        allreduce_minmax(minmax)
        have = num
        if num >= max_samples:          # main.rs:101, first clause
            break
        n_active = engine.select(n_active, minmax, thresholds, spp, num * 2)
        num *= 2
    reduce_sum(accum, dst=0)
    if rank != 0:
        return None, spp
    return engine.resolve(accum, spp), spp
