"""Seeded synthetic scenes of BASELINE.json configs 3-5, scaled by a size parameter so that the oracle
finishes in seconds in tests; the full sizes are used by scripts/gpu_configs.py on the GPU."""
import numpy as np

from hpt import scene_io

F = np.float32
I3 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], F)


def look_at_camera(pos, target, fov, w, h, up=(0, 0, 1)):
    pos = np.array(pos, F)
    d = scene_io.normalize((np.array(target, F) - pos).astype(F))
    u = scene_io.cross(scene_io.cross(d, np.array(up, F)), d)
    return scene_io.Camera(pos=pos, dir=d, up=u, fov=float(fov), width=w, height=h)


def soup(n_tris, seed=2, w=96, h=54, extent=None):
    """config 3: triangle soup, centres uniform in [0,E)^3 (E = 100 at 1M triangles, scaled to keep the density),
    vertices = centre + uniform(-0.5,0.5)^3, one diffuse 0.7 material, white environment and background,
    camera (E/2,-1.5E,E/2) -> cube centre, fov 40."""
    rng = np.random.default_rng(seed)
    E = float(extent) if extent else 100.0 * (n_tris / 1.0e6) ** (1.0 / 3.0)
    c = rng.random((n_tris, 1, 3)) * E
    v = (c + rng.random((n_tris, 3, 3)) - 0.5).reshape(-1, 3).astype(F)
    tris = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    z = np.zeros((n_tris, 3), np.uint32)
    mesh = scene_io.Mesh(verts=v, tris=tris, normals=np.array([[0, 0, 1]], F), ntris=z, uvw=np.array([[0, 0, 0]], F), ttris=z.copy())
    d = scene_io.SceneDesc()
    d.meshes.append(mesh)
    d.materials.append(scene_io.Material(name="m", diffuse=scene_io.tex_color((0.7, 0.7, 0.7))))
    d.nodes.append(scene_io.Node(parent=-1, geom=3, mesh=0, material=0, m=I3.copy(), t=np.zeros(3, F)))
    d.environment = scene_io.tex_color((1, 1, 1))
    d.background = scene_io.tex_color((1, 1, 1))
    d.camera = look_at_camera((E / 2, -1.5 * E, E / 2), (E / 2, E / 2, E / 2), 40, w, h)
    return d


def rot_z(deg, scale=1.0):
    c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    return (np.array([c, s, 0, -s, c, 0, 0, 0, 1], np.float64) * scale).astype(F)


def instanced(mesh, n_side, seed=3, spacing=40.0, w=96, h=54):
    """config 4: one mesh instanced on an n_side^3 lattice with random z-rotations (BLAS shared, TLAS over instances),
    a big diffuse floor, white environment."""
    rng = np.random.default_rng(seed)
    d = scene_io.SceneDesc()
    d.meshes.append(mesh)
    d.materials.append(scene_io.Material(name="a", diffuse=scene_io.tex_color((0.8, 0.3, 0.2))))
    d.materials.append(scene_io.Material(name="b", diffuse=scene_io.tex_color((0.3, 0.5, 0.8)), reflection=scene_io.tex_color((0.3, 0.3, 0.3)), reflection_glossiness=0.1))
    d.materials.append(scene_io.Material(name="floor", diffuse=scene_io.tex_color((0.6, 0.6, 0.6))))
    ext = n_side * spacing
    d.nodes.append(scene_io.Node(parent=-1, geom=2, mesh=-1, material=2, m=(I3 * F(ext * 2)), t=np.array([ext / 2, ext / 2, -5], F)))
    for ix in range(n_side):
        for iy in range(n_side):
            for iz in range(n_side):
                d.nodes.append(scene_io.Node(parent=-1, geom=3, mesh=0, material=int(rng.integers(0, 2)), m=rot_z(rng.uniform(0, 360)),
                                             t=np.array([ix * spacing, iy * spacing, iz * spacing], F)))
    d.environment = scene_io.tex_color((1, 1, 1))
    d.background = scene_io.tex_color((0.6, 0.7, 0.9))
    d.camera = look_at_camera((ext / 2, -1.2 * ext - 60, ext * 0.9), (ext / 2, ext / 2, ext / 3), 40, w, h)
    return d


def spheres(n, seed=4, w=96, h=54):
    """config 5: n spheres, centres uniform in [-50,50]^2 x [0,30], radius U[0.2,0.8]; 50 % diffuse random albedo, 25 %
    reflection 1 with gloss U[0,0.2], 25 % refraction ior 1.5; floor plane; white environment."""
    rng = np.random.default_rng(seed)
    d = scene_io.SceneDesc()
    d.materials.append(scene_io.Material(name="floor", diffuse=scene_io.tex_color((0.5, 0.5, 0.5))))
    d.nodes.append(scene_io.Node(parent=-1, geom=2, mesh=-1, material=0, m=(I3 * F(200)), t=np.zeros(3, F)))
    grp = len(d.nodes)
    d.nodes.append(scene_io.Node(parent=-1, geom=0, mesh=-1, material=-1, m=I3.copy(), t=np.array([0, 0, 0.9], F)))   # a group level
    for k in range(n):
        kind = rng.random()
        if kind < 0.5:
            m = scene_io.Material(name="d%d" % k, diffuse=scene_io.tex_color(rng.uniform(0.1, 0.9, 3)))
        elif kind < 0.75:
            m = scene_io.Material(name="r%d" % k, diffuse=scene_io.tex_color((0, 0, 0)), reflection=scene_io.tex_color((1, 1, 1)), reflection_glossiness=float(rng.uniform(0, 0.2)))
        else:
            m = scene_io.Material(name="t%d" % k, diffuse=scene_io.tex_color((0, 0, 0)), refraction=scene_io.tex_color((0.9, 0.9, 0.9)), refraction_index=1.5)
        d.materials.append(m)
        r = F(rng.uniform(0.2, 0.8))
        d.nodes.append(scene_io.Node(parent=grp, geom=1, mesh=-1, material=len(d.materials) - 1, m=(I3 * r),
                                     t=np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 30)], F)))
    d.environment = scene_io.tex_color((1, 1, 1))
    d.background = scene_io.tex_color((0.6, 0.7, 0.9))
    d.camera = look_at_camera((0, -120, 40), (0, 0, 10), 40, w, h)
    return d
