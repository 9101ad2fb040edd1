"""numpy restatement of the radiance accumulator (include/hpt.h hpt_accum; k_resolve / k_accum_resolve of the kernels): test
infrastructure -- the CPU stand-in of the device in the world-size-2 gloo tests, and the checker of the GPU's accumulators."""
import numpy as np

FRAC_BITS = 24
SAMPLE_MAX = np.float32(1048576.0)


def fix(x):
    """f32 samples -> fixed point: clamp to +-2^20, scale by 2^24 (exact), round to nearest even; NaN -> 0"""
    x = np.asarray(x, dtype=np.float32)
    nan = np.isnan(x)
    y = np.clip(np.where(nan, np.float32(0), x), -SAMPLE_MAX, SAMPLE_MAX) * np.float32(2.0 ** FRAC_BITS)
    return np.rint(y).astype(np.int64), nan


def accumulate(accum, per_sample, rows=None):
    """accum: int64 [npix, 4], per_sample: f32 [len(rows) | npix, n, 3] -- added in place"""
    q, nan = fix(per_sample)
    add = np.concatenate([q.sum(axis=1), nan.any(axis=2).sum(axis=1, keepdims=True).astype(np.int64)], axis=1)
    if rows is None:
        accum += add
    else:
        accum[rows] += add
    return accum


def resolve(accum, num_samples=0, samples_per_px=None):
    """hpt_accum_resolve: f32(sum) / n"""
    accum = np.asarray(accum, dtype=np.int64).reshape(-1, 4)
    v = (accum[:, :3].astype(np.float64) * (1.0 / 2.0 ** FRAC_BITS)).astype(np.float32)
    v[accum[:, 3] != 0] = np.nan
    if samples_per_px is not None:
        v = v / np.asarray(samples_per_px).reshape(-1, 1).astype(np.float32)
    elif num_samples:
        v = v / np.float32(num_samples)
    return v
