"""Multi-GPU partition of the render path: every (pixel, sample) path is independent (main.rs:57,72 are pure
maps), so rank r of W renders the sample range [lo, hi) of EVERY pixel -- the counter-based RNG is keyed by
(pixel, sample index), hence the set of paths does not depend on W -- and the one exchange step is a sum-reduce of
the float accumulators (RCCL over xGMI with backend "nccl"; gloo in the CPU tests)."""


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
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor
