"""Seeded synthetic ray sets for the parity tests: rays in mesh-local space against one mesh."""
import numpy as np


def parity_rays(box, n, seed=5):
    """80 % from a sphere of radius 2.5 R around the box centre aimed into the 0.8-scaled box, 15 % origins inside the
    box with uniform directions, 5 % axis-aligned directions with exact zeros (bvh.rs:228,244,260 branches)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(box[:3], np.float64), np.asarray(box[3:], np.float64)
    c, ext = (lo + hi) / 2, (hi - lo)
    R = np.linalg.norm(ext) / 2
    n_a = int(n * 0.80); n_b = int(n * 0.15); n_c = n - n_a - n_b
    o = rng.normal(size=(n_a, 3)); o = c + 2.5 * R * o / np.linalg.norm(o, axis=1, keepdims=True)
    tg = c + 0.8 * ext * (rng.random((n_a, 3)) - 0.5)
    org = [o]; dirs = [tg - o]
    ob = lo + ext * rng.random((n_b, 3))
    db = rng.normal(size=(n_b, 3)); db /= np.linalg.norm(db, axis=1, keepdims=True)
    org.append(ob); dirs.append(db)
    # axis aligned: origin on a face plane of the 1.5x box, pointing inward along one axis, other components exactly 0
    ax = rng.integers(0, 3, size=n_c); sign = rng.integers(0, 2, size=n_c) * 2 - 1
    oc = lo + ext * rng.random((n_c, 3))
    dc = np.zeros((n_c, 3))
    oc[np.arange(n_c), ax] = c[ax] - sign * 1.5 * ext[ax]
    dc[np.arange(n_c), ax] = sign
    org.append(oc); dirs.append(dc)
    return np.concatenate(org).astype(np.float32), np.concatenate(dirs).astype(np.float32)


def edge_distance(u, v):
    """min barycentric coordinate of a hit: rays within 1e-6 of an edge are excluded from the id comparison"""
    return np.minimum(np.minimum(u, v), 1.0 - u - v)
