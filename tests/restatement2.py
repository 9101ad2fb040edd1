"""A SECOND, independent restatement of the reference's intersect / shade semantics -- test infrastructure, like oracle/.

Written from the text of /root/reference/src/{geometry,scene,bvh,main}.rs (cited per function), NOT from oracle/oracle.c, in
numpy float32, vectorised over rays, with deliberately different structure:
  * Mesh::intersect (geometry.rs:132-158) by BRUTE FORCE over all triangles (the commented-out variant at geometry.rs:160-185)
    with the reference's replacement rule (`nearest.z < t -> continue`, so on equal t the later-tested triangle wins);
  * Node::intersect (scene.rs:242-275) as a recursion over the node tree with Matrix3::invert evaluated PER CALL
    (scene.rs:279,287), Geometry::intersect incl. its bounding-box pre-test (geometry.rs:30-97, bvh.rs:225-281);
  * Scene::cast (scene.rs:130-188) as a loop over bounces that keeps the recursion's nested evaluation order
    (colour = emission + n.dot(new_dir) * albedo (*) L_next / p), with the two shared deviations of DESIGN.md (depth cap,
    counter-based RNG of include/hpt.h);
  * the primary rays of main.rs:43-85.
It cannot pin parity to the Rust binary (nothing reference-held exists: SURVEY 4, 8c) -- it halves the single-reader risk:
tests/test_restatement2.py holds the C oracle to it.  cgmath arithmetic is restated from the crate's published
definitions (dot = x + y + z left to right, normalize = v * (1 / |v|), Matrix3::invert by cofactor columns / det).
"""
import numpy as np

F = np.float32
INF = F(np.inf)
BIAS, EPSILON = F(0.01), F(1.0e-8)     # scene.rs:122-123
PI = F(np.pi)                           # std::f32::consts::PI


def lm(fn, *x):
    """libm's f32 functions (Rust's f32::sin / cos / tan / asin / atan2 call them): evaluated in f64 and rounded -- numpy's own
    float32 loops are SIMD approximations that differ from libm in the last bit (tan(15 deg) already does)"""
    return fn(*[np.asarray(v, np.float64) for v in x]).astype(F)


# ------------------------------------------------------------------ cgmath pieces
def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def normalize(v):
    with np.errstate(all="ignore"):
        return v * (F(1.0) / np.sqrt(dot(v, v)))[..., None]


def invert_rows(m9):
    """Matrix3::invert of a column-major m9 -> the ROWS of the inverse (= cross products of the columns / det)."""
    c0, c1, c2 = m9[0:3].astype(F), m9[3:6].astype(F), m9[6:9].astype(F)
    det = (c0[0] * (c1[1] * c2[2] - c2[1] * c1[2]) - c1[0] * (c0[1] * c2[2] - c2[1] * c0[2])) + c2[0] * (c0[1] * c1[2] - c1[1] * c0[2])
    det = F(det)
    if det == 0:
        raise ValueError("singular transform (scene.rs:279 unwrap)")
    return [(cross(c1, c2) / det).astype(F), (cross(c2, c0) / det).astype(F), (cross(c0, c1) / det).astype(F)]


def to_local_space(m9, t3, v):
    """scene.rs:278-280: invert() * (vec - translate), inverse recomputed at every call"""
    r = invert_rows(m9)
    w = (v - t3.astype(F)).astype(F)
    return np.stack([(r[i][0] * w[..., 0] + r[i][1] * w[..., 1]) + r[i][2] * w[..., 2] for i in range(3)], axis=-1).astype(F)


def from_local_space(m9, t3, v):
    """scene.rs:282-284: transform * vec + translate (Matrix3 * Vector3 = col0 * x + col1 * y + col2 * z)"""
    m = m9.astype(F)
    out = np.stack([(m[0 + i] * v[..., 0] + m[3 + i] * v[..., 1]) + m[6 + i] * v[..., 2] for i in range(3)], axis=-1)
    return (out + t3.astype(F)).astype(F)


def normal_from_local_space(m9, n):
    """scene.rs:286-288: (invert().transpose() * normal).normalize(); columns of the transpose = rows of the inverse"""
    r = invert_rows(m9)
    out = np.stack([(r[0][i] * n[..., 0] + r[1][i] * n[..., 1]) + r[2][i] * n[..., 2] for i in range(3)], axis=-1).astype(F)
    return normalize(out)


# ------------------------------------------------------------------ bvh.rs:225-281
def box_intersect(p1, p2, pos, dir):
    """BoundingBox::intersect -> boolean "Some(_)" per ray (no t_out >= 0 test: boxes behind the origin still hit)"""
    ok = np.ones(len(pos), bool)
    t_in = np.full(len(pos), -INF, F); t_out = np.full(len(pos), INF, F)
    with np.errstate(all="ignore"):
        for k in range(3):
            zero = dir[:, k] == 0
            ok &= ~(zero & ((pos[:, k] < p1[k]) | (pos[:, k] > p2[k])))
            a = (p1[k] - pos[:, k]) / dir[:, k]
            b = (p2[k] - pos[:, k]) / dir[:, k]
            lo = np.where(b < a, b, a); hi = np.where(b < a, a, b)
            lo = np.where(zero, -INF, lo); hi = np.where(zero, INF, hi)
            t_in = np.fmax(t_in, lo); t_out = np.fmin(t_out, hi)       # f32::max / min ignore NaN
    return ok & (t_in <= t_out)


# ------------------------------------------------------------------ geometry.rs
def intersect_triangles(a, b, c, pos, dir):
    """geometry.rs:188-221 for rays [N,3] x triangles [T,3] -> t,u,v [N,T] (t = inf where None), side"""
    ab = (b - a)[None]; ac = (c - a)[None]
    d = dir[:, None, :]; p = pos[:, None, :]
    with np.errstate(all="ignore"):
        pvec = cross(d, ac)
        det = dot(ab, pvec)
        tvec = p - a[None]
        u = dot(tvec, pvec) / det
        qvec = cross(tvec, ab)
        v = dot(d, qvec) / det
        t = dot(ac, qvec) / det
        ok = ~(np.abs(det) < EPSILON)
        ok &= ~((u < 0) | (u > 1))
        ok &= ~((v < 0) | (u + v > 1))
        ok &= ~(t < 0)
    return np.where(ok, t, INF).astype(F), u.astype(F), v.astype(F), det > 0


def mesh_intersect(mesh, pos, dir, chunk=192):
    """Mesh::intersect by brute force (geometry.rs:160-185 is that loop): returns prim (-1 none), t, u, v, side.
    On exactly equal t the LATER index wins (`nearest.z < t -> continue`); the BVH version visits in another order, so
    tie winners may differ on shared edges (north_star excludes those rays)."""
    V = mesh.verts.astype(F); T = mesh.tris.astype(np.int64)
    a, b, c = V[T[:, 0]], V[T[:, 1]], V[T[:, 2]]
    n = len(pos)
    prim = np.full(n, -1, np.int64); tt = np.zeros(n, F); uu = np.zeros(n, F); vv = np.zeros(n, F); ss = np.zeros(n, bool)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        t, u, v, side = intersect_triangles(a, b, c, pos[s:e], dir[s:e])
        tmin = t.min(axis=1)
        hit = np.isfinite(tmin)
        last = t.shape[1] - 1 - np.argmax((t == tmin[:, None])[:, ::-1], axis=1)      # the last index that attains the minimum
        r = np.arange(e - s)
        prim[s:e] = np.where(hit, last, -1)
        tt[s:e] = np.where(hit, tmin, 0); uu[s:e] = np.where(hit, u[r, last], 0); vv[s:e] = np.where(hit, v[r, last], 0)
        ss[s:e] = hit & side[r, last]
    return prim, tt, uu, vv, ss


def bary(arr, idx, face, u, v):
    """geometry.rs:117-130: (1 - u - v) * A + u * B + v * C"""
    A, B, C = arr[idx[face, 0]], arr[idx[face, 1]], arr[idx[face, 2]]
    w = (F(1.0) - u - v)
    return ((w[:, None] * A + u[:, None] * B) + v[:, None] * C).astype(F)


class Hits:
    def __init__(self, n):
        self.t = np.full(n, INF, F); self.pos = np.zeros((n, 3), F); self.uv = np.zeros((n, 3), F)
        self.normal = np.zeros((n, 3), F); self.front = np.zeros(n, bool); self.node = np.full(n, -1, np.int64); self.prim = np.full(n, -1, np.int64)

    def take(self, other, mask):
        for k in ("t", "pos", "uv", "normal", "front", "node", "prim"):
            getattr(self, k)[mask] = getattr(other, k)[mask]


def geometry_intersect(desc, node_id, pos, dir):
    """Geometry::intersect (geometry.rs:30-97) of the object of one node, rays in the node's local space"""
    nd = desc.nodes[node_id]
    h = Hits(len(pos))
    with np.errstate(all="ignore"):
        if nd.geom == 1:
            pre = box_intersect(np.array([-1, -1, -1], F), np.array([1, 1, 1], F), pos, dir)
            pdd = dot(pos, dir); dl = dot(dir, dir)
            disc = F(4.0) * (pdd * pdd) - F(4.0) * dl * (dot(pos, pos) - F(1.0))
            ds = np.sqrt(disc)
            t1 = (F(-2.0) * pdd + ds) / (F(2.0) * dl); t2 = (F(-2.0) * pdd - ds) / (F(2.0) * dl)
            ok = pre & (disc > 0) & ((t1 > 0) | (t2 > 0))
            front = (t1 > 0) & (t2 > 0)
            t = np.where(front, np.fmin(t1, t2), np.fmax(t1, t2)).astype(F)
            hp = (pos + t[:, None] * dir).astype(F)
            nrm = normalize(hp)
            uv = np.stack([F(0.5) + lm(np.arctan2, nrm[:, 1], nrm[:, 0]) / (F(2.0) * PI), F(0.5) - lm(np.arcsin, -nrm[:, 2]) / PI, np.zeros(len(pos), F)], axis=-1).astype(F)
            h.t[ok] = t[ok]; h.pos[ok] = hp[ok]; h.normal[ok] = nrm[ok]; h.uv[ok] = uv[ok]; h.front[ok] = front[ok]
        elif nd.geom == 2:
            pre = box_intersect(np.array([-1, -1, 0], F), np.array([1, 1, 0], F), pos, dir)
            t = (-(pos[:, 2] / dir[:, 2])).astype(F)
            p = (pos + t[:, None] * dir).astype(F)
            ok = pre & (t > 0) & (F(-1.0) < p[:, 0]) & (p[:, 0] < F(1.0)) & (F(-1.0) < p[:, 1]) & (p[:, 1] < F(1.0))
            h.t[ok] = t[ok]; h.pos[ok] = p[ok]
            h.uv[ok] = np.stack([F(0.5) + F(0.5) * p[:, 0], F(0.5) + F(0.5) * p[:, 1], np.zeros(len(pos), F)], axis=-1).astype(F)[ok]
            h.normal[ok] = np.array([0, 0, 1], F); h.front[ok] = (pos[:, 2] > 0)[ok]
        elif nd.geom == 3:
            m = desc.meshes[nd.mesh]
            V = m.verts.astype(F)
            pre = box_intersect(V.min(axis=0), V.max(axis=0), pos, dir)       # load.rs:284-293 mesh box
            idx = np.nonzero(pre)[0]
            if len(idx):
                prim, t, u, v, front = mesh_intersect(m, pos[idx], dir[idx])
                ok = prim >= 0
                j = idx[ok]; f = prim[ok]
                h.t[j] = t[ok]; h.front[j] = front[ok]; h.prim[j] = f
                h.pos[j] = bary(V, m.tris.astype(np.int64), f, u[ok], v[ok])
                h.uv[j] = bary(m.uvw.astype(F), m.ttris.astype(np.int64), f, u[ok], v[ok])
                h.normal[j] = bary(m.normals.astype(F), m.ntris.astype(np.int64), f, u[ok], v[ok])
    h.node[np.isfinite(h.t)] = node_id
    return h


def node_intersect(desc, children, node_id, pos, dir):
    """Node::intersect (scene.rs:242-275)"""
    nd = desc.nodes[node_id]
    lp = to_local_space(nd.m, nd.t, pos)                              # scene.rs:228-232
    ld = (to_local_space(nd.m, nd.t, (pos + dir).astype(F)) - lp).astype(F)
    best = geometry_intersect(desc, node_id, lp, ld) if nd.geom != 0 else Hits(len(pos))
    for c in children[node_id]:
        h = node_intersect(desc, children, c, lp, ld)
        best.take(h, h.t < best.t)                                    # strictly nearer replaces: the earlier node keeps ties
    hit = np.isfinite(best.t)
    if hit.any():
        best.pos[hit] = from_local_space(nd.m, nd.t, best.pos[hit])
        best.normal[hit] = normal_from_local_space(nd.m, best.normal[hit])
    return best


def scene_children(desc):
    children = [[] for _ in desc.nodes]; top = []
    for i, nd in enumerate(desc.nodes):
        (top if nd.parent < 0 else children[nd.parent]).append(i)
    return top, children


def scene_intersect(desc, pos, dir):
    """Scene::intersect (scene.rs:190-208)"""
    top, children = scene_children(desc)
    pos = np.ascontiguousarray(pos, F); dir = np.ascontiguousarray(dir, F)
    best = Hits(len(pos))
    for nid in top:
        h = node_intersect(desc, children, nid, pos, dir)
        best.take(h, h.t < best.t)
    return best


# ------------------------------------------------------------------ include/hpt.h RNG, scene.rs:431 halton
def _mix(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def rng(seed, pixel, sample, dim):
    with np.errstate(over="ignore"):
        seed = np.uint32(seed); pixel = np.asarray(pixel, np.uint32); sample = np.asarray(sample, np.uint32)
        k1 = _mix(pixel + _mix(np.array([seed], np.uint32))[0])
        k2 = _mix(sample ^ np.uint32(0x5bd1e995) ^ np.uint32((int(seed) * 0x9E3779B9) & 0xffffffff))
        h = _mix(k1 ^ (k2 + np.uint32((dim * 0x9E3779B9) & 0xffffffff)))
    return ((h >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)).astype(F)


def halton(index, base):
    r = F(0.0); f = F(1.0); i = int(index)
    while i > 0:
        f = F(f / F(base)); r = F(r + F(f * F(i % base))); i //= base
    return r


# ------------------------------------------------------------------ scene.rs shading pieces
def texture_sample(tex, point):
    """scene.rs:298-358 for Blank and Checkerboard (the kinds the shipped scene and this cross-check use)"""
    n = len(point)
    if tex.kind == 0:
        data = np.ones((n, 3), F)
    elif tex.kind == 1:
        lp = to_local_space(tex.m, tex.t, point)
        with np.errstate(all="ignore"):
            c = lp - np.trunc(lp).astype(F); c = np.where(c < 0, c + F(1.0), c)        # unit_clamp scene.rs:378-392
        first = ((c[:, 0] < 0.5) & (c[:, 1] < 0.5)) | ((c[:, 0] > 0.5) & (c[:, 1] > 0.5))
        data = np.where(first[:, None], tex.color1.astype(F)[None], tex.color2.astype(F)[None])
    else:
        raise NotImplementedError("image textures are not part of this cross-check")
    return (tex.color.astype(F)[None] * data).astype(F)


def sample_environment(tex, d):
    if tex.kind == 0:
        return np.tile(tex.color.astype(F), (len(d), 1))
    p = np.stack([F(0.5) + lm(np.arctan2, d[:, 1], d[:, 0]) / (F(2.0) * PI), F(0.5) - lm(np.arcsin, -d[:, 2]) / PI, np.zeros(len(d), F)], axis=-1).astype(F)
    return texture_sample(tex, p)


def random_about(vec, z, theta):
    """the common tail of random_rotation / random_hemisphere_sample (scene.rs:394-429)"""
    ax, ay, az = np.abs(vec[:, 0]), np.abs(vec[:, 1]), np.abs(vec[:, 2])
    axis = np.where(((ax < ay) & (ax < az))[:, None], np.array([1, 0, 0], F), np.where((ay < az)[:, None], np.array([0, 1, 0], F), np.array([0, 0, 1], F))).astype(F)
    u = normalize(cross(vec, axis)); v = normalize(cross(vec, u))
    with np.errstate(all="ignore"):
        rad = lm(np.cos, lm(np.arcsin, z))
        out = vec * z[:, None] + rad[:, None] * (lm(np.cos, theta)[:, None] * u + lm(np.sin, theta)[:, None] * v)
    return normalize(out.astype(F))


def powi(x, n):
    """f32::powi as compiler-rt's __powisf2 evaluates it (square-and-multiply from the low bit)"""
    r = np.ones_like(x); a = x.copy(); b = n
    while True:
        if b & 1:
            r = (r * a).astype(F)
        b //= 2
        if b == 0:
            return r
        a = (a * a).astype(F)


def primary_rays(cam, pixels, sample, seed):
    """main.rs:43-85 for pixel indices `pixels` and ONE sample index (1-based)"""
    width_px, height_px = F(cam.width), F(cam.height)
    height = F(F(lm(np.tan, F(F(F(cam.fov) / F(2.0)) * F(F(2.0) * PI / F(360.0)))) * F(2.0)) * F(cam.focaldist))
    width = F(F(height * width_px) / height_px)
    ph, pw = F(height / height_px), F(width / width_px)
    d, up, pos = cam.dir.astype(F), cam.up.astype(F), cam.pos.astype(F)
    right = cross(d, up).astype(F)
    b = (pos + F(cam.focaldist) * d + F(height / F(2.0)) * up).astype(F)
    a = (b - F(width / F(2.0)) * right).astype(F)
    x = (pixels % cam.width).astype(F); y = (pixels // cam.width).astype(F)
    top_left = (a[None] + (x * pw)[:, None] * right[None] - (y * ph)[:, None] * up[None]).astype(F)
    xo, yo = halton(sample, 2), halton(sample, 3)
    r1 = rng(seed, pixels, sample, 0); r2 = rng(seed, pixels, sample, 1)
    ex = (F(2.0) * r1 - F(1.0)).astype(F); ey = (F(2.0) * r2 - F(1.0)).astype(F)
    p = (top_left + F(xo * pw) * right[None] - F(yo * ph) * up[None]).astype(F)
    eye = (pos[None] + (ex * F(cam.dof))[:, None] * right[None] + (ey * F(cam.dof))[:, None] * up[None]).astype(F)
    dirs = normalize((p - eye).astype(F))
    img = np.stack([(x + xo) / width_px, (y + yo) / height_px, np.zeros(len(x), F)], axis=-1).astype(F)
    return eye, dirs, img


def radiance_1spp(desc, sample, seed, max_depth):
    """Scene::sample (scene.rs:126) for sample index `sample` of every pixel -> [npix, 3].
    The recursion of Scene::cast is unrolled into a forward sweep that records each bounce's (emission, n.dot(new_dir),
    albedo, p) and a backward fold colour_k = emission_k + n.dot * (albedo (*) colour_{k+1}) / p -- the reference's order."""
    cam = desc.camera
    npix = cam.width * cam.height
    pixels = np.arange(npix, dtype=np.uint32)
    pos, d, img_xy = primary_rays(cam, pixels, sample, seed)
    weight = np.ones(npix, F)
    live = np.arange(npix)
    rec = []          # per depth: (indices, emission, ndot, albedo, p, continues)
    tail = {}         # depth -> (indices, colour) of paths that ended by a miss at that depth
    for depth in range(max_depth + 1):
        if len(live) == 0:
            break
        h = scene_intersect(desc, pos[live], d[live])
        hit = np.isfinite(h.t)
        miss = live[~hit]
        if len(miss):
            tail[depth] = (miss, texture_sample(desc.background, img_xy[miss]) if depth == 0 else sample_environment(desc.environment, d[miss]))
        idx = live[hit]
        n = len(idx)
        if n == 0:
            live = idx
            continue
        mats = np.array([desc.nodes[k].material for k in h.node[hit]])
        uv = h.uv[hit]; dd = d[idx]
        diffuse = np.zeros((n, 3), F); reflection = np.zeros((n, 3), F); refraction = np.zeros((n, 3), F)
        emission = np.zeros((n, 3), F); ior = np.ones(n, F); g_refl = np.zeros(n, F); g_refr = np.zeros(n, F)
        for mi in np.unique(mats):
            m = desc.materials[mi]; sel = mats == mi
            diffuse[sel] = texture_sample(m.diffuse, uv[sel]); reflection[sel] = texture_sample(m.reflection, uv[sel])
            refraction[sel] = texture_sample(m.refraction, uv[sel]); emission[sel] = m.emission.astype(F)
            ior[sel] = F(m.refraction_index); g_refl[sel] = F(m.reflection_glossiness); g_refr[sel] = F(m.refraction_glossiness)
        front = h.front[hit]
        normal = np.where(front[:, None], h.normal[hit], -h.normal[hit]).astype(F)
        n1 = np.where(front, F(1.0), ior).astype(F); n2 = np.where(front, ior, F(1.0)).astype(F)
        with np.errstate(all="ignore"):
            r0 = powi(((n1 - n2) / (n1 + n2)).astype(F), 2)
            ar = (r0 + (F(1.0) - r0) * powi((F(1.0) - dot(normal, -dd)).astype(F), 5)).astype(F)
            p_d = ((diffuse[:, 0] + diffuse[:, 1] + diffuse[:, 2]) / F(3.0)).astype(F)
            p_t = ((F(1.0) - ar) * (refraction[:, 0] + refraction[:, 1] + refraction[:, 2]) / F(3.0)).astype(F)
            p_r = ((F(1.0) + ar) * (reflection[:, 0] + reflection[:, 1] + reflection[:, 2]) / F(3.0)).astype(F)      # sic, scene.rs:154
            p_range = ((p_d + p_r) + p_t).astype(F)
            pix = pixels[idx]; d0 = 2 + 4 * depth
            u0 = rng(seed, pix, sample, d0); u1 = rng(seed, pix, sample, d0 + 1); u2 = rng(seed, pix, sample, d0 + 2); u3 = rng(seed, pix, sample, d0 + 3)
            go = (depth < max_depth) & ~((p_range == 0) | (u0 > weight[idx]))      # depth cap: shared deviation
            pd, pr, pt = (p_d / p_range).astype(F), (p_r / p_range).astype(F), (p_t / p_range).astype(F)
            lobe = np.where(u1 < pd, 0, np.where(u1 < pd + pr, 1, np.where(u1 < (pd + pr) + pt, 2, 3)))
            go &= lobe < 3
            theta = (u3 * F(2.0) * PI).astype(F)
            refl = normalize((-(-dd) + (F(2.0) * dot(normal, -dd))[:, None] * normal).astype(F))      # reflect_ray(-dir, normal)
            nn = (n1 / n2).astype(F); ndv = dot(normal, -dd)
            s = (nn[:, None] * (ndv[:, None] * normal - (-dd))).astype(F)
            cos_sqr = (F(1.0) - powi(nn, 2) * (F(1.0) - powi(ndv, 2))).astype(F)
            refr = normalize((s - np.sqrt(cos_sqr)[:, None] * normal).astype(F))
            refr = np.where((cos_sqr >= 0)[:, None], refr, refl)
            cr, ct = lm(np.cos, g_refl), lm(np.cos, g_refr)
            zr = (cr + u2 * (F(1.0) - cr)).astype(F); zt = (ct + u2 * (F(1.0) - ct)).astype(F)
            axis = np.where((lobe == 0)[:, None], normal, np.where((lobe == 1)[:, None], refl, refr)).astype(F)
            z = np.where(lobe == 0, u2, np.where(lobe == 1, zr, zt)).astype(F)
            new_dir = random_about(axis, z, theta)
            albedo = np.where((lobe == 0)[:, None], diffuse, np.where((lobe == 1)[:, None], reflection, refraction)).astype(F)
            p = np.where(lobe == 0, pd, np.where(lobe == 1, pr, pt)).astype(F)
            ndot = dot(normal, new_dir).astype(F)
        rec.append((idx, emission, ndot, albedo, p, go))
        nxt = idx[go]
        pos[nxt] = (h.pos[hit][go] + BIAS * new_dir[go]).astype(F)
        d[nxt] = new_dir[go]
        weight[nxt] = (weight[nxt] * p[go]).astype(F)
        live = nxt
    # backward fold, deepest bounce first
    colour = np.zeros((npix, 3), F); have = np.zeros(npix, bool)
    for depth in range(max_depth + 1, -1, -1):
        if depth in tail:
            idx, c = tail[depth]; colour[idx] = c; have[idx] = True
        if depth - 1 >= 0 and depth - 1 < len(rec):
            idx, emission, ndot, albedo, p, go = rec[depth - 1]
            with np.errstate(all="ignore"):
                nxt = (ndot[:, None] * (albedo * colour[idx]) / p[:, None]).astype(F)
            colour[idx] = np.where(go[:, None], (emission + nxt).astype(F), emission)
    return colour
