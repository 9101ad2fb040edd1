"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32P = C.POINTER(C.c_float)
U32P = C.POINTER(C.c_uint32)
I32P = C.POINTER(C.c_int32)
U8P = C.POINTER(C.c_uint8)


class BvhNode(C.Structure):
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("prim", C.c_int32), ("p1", C.c_float * 3), ("p2", C.c_float * 3)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image", C.c_int32), ("color", C.c_float * 3), ("color1", C.c_float * 3),
                ("color2", C.c_float * 3), ("m", C.c_float * 9), ("t", C.c_float * 3)]


class Material(C.Structure):
    _fields_ = [("diffuse", Texture), ("reflection", Texture), ("refraction", Texture), ("emission", C.c_float * 3),
                ("reflection_glossiness", C.c_float), ("refraction_glossiness", C.c_float), ("refraction_index", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("img_width", C.c_uint32), ("img_height", C.c_uint32), ("focaldist", C.c_float), ("dof", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("adaptive", C.c_int32), ("spp", C.c_int32), ("first_sample", C.c_int32), ("initial_samples", C.c_int32),
                ("max_samples", C.c_int32), ("thresholds", C.c_float * 3), ("max_depth", C.c_int32), ("seed", C.c_uint32),
                ("threads", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_pops", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("seconds", C.c_double), ("threads", C.c_int32)]


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def fill_texture(dst, tx):
    dst.kind, dst.image = int(tx.kind), int(tx.image)
    dst.color[:] = [float(x) for x in tx.color]
    dst.color1[:] = [float(x) for x in tx.color1]
    dst.color2[:] = [float(x) for x in tx.color2]
    dst.m[:] = [float(x) for x in tx.m]
    dst.t[:] = [float(x) for x in tx.t]


def fill_material(dst, m):
    fill_texture(dst.diffuse, m.diffuse)
    fill_texture(dst.reflection, m.reflection)
    fill_texture(dst.refraction, m.refraction)
    dst.emission[:] = [float(x) for x in m.emission]
    dst.reflection_glossiness = m.reflection_glossiness
    dst.refraction_glossiness = m.refraction_glossiness
    dst.refraction_index = m.refraction_index


def fill_camera(dst, c):
    dst.pos[:] = [float(x) for x in c.pos]
    dst.dir[:] = [float(x) for x in c.dir]
    dst.up[:] = [float(x) for x in c.up]
    dst.fov, dst.img_width, dst.img_height = c.fov, c.width, c.height
    dst.focaldist, dst.dof = c.focaldist, c.dof


def load(fast=False):
    path = os.path.join(ROOT, "oracle", "liboracle_fast.so" if fast else "liboracle.so")
    if not os.path.exists(path):
        raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    lib.orc_halton.restype = C.c_float
    lib.orc_halton.argtypes = [C.c_int32, C.c_int32]
    lib.orc_rng_f32.restype = C.c_float
    lib.orc_rng_f32.argtypes = [C.c_uint32] * 4
    lib.orc_tonemap_rgba8.argtypes = [F32P, C.c_uint32, U8P]
    lib.orc_mat3_invert.argtypes = [F32P, F32P]
    lib.orc_bvh_build.restype = C.c_int32
    lib.orc_bvh_build.argtypes = [F32P, F32P, C.c_int32, C.POINTER(BvhNode)]
    lib.orc_box_intersect.argtypes = [F32P, F32P, F32P, F32P]
    lib.orc_mesh_create.restype = C.c_void_p
    lib.orc_mesh_create.argtypes = [F32P, C.c_int32, U32P, F32P, C.c_int32, U32P, F32P, C.c_int32, U32P, C.c_int32]
    lib.orc_mesh_destroy.argtypes = [C.c_void_p]
    lib.orc_mesh_num_nodes.restype = C.c_int32
    lib.orc_mesh_num_nodes.argtypes = [C.c_void_p]
    lib.orc_mesh_get_nodes.argtypes = [C.c_void_p, C.POINTER(BvhNode)]
    lib.orc_mesh_get_box.argtypes = [C.c_void_p, F32P]
    lib.orc_mesh_intersect.argtypes = [C.c_void_p, C.c_int64, F32P, F32P, I32P, F32P, F32P, F32P, U8P, C.c_int32, C.POINTER(Stats)]
    lib.orc_intersect_triangle.argtypes = [F32P] * 5 + [F32P, F32P, F32P, C.POINTER(C.c_int)]
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_scene_add_image.argtypes = [C.c_void_p, U8P, C.c_int32, C.c_int32]
    lib.orc_scene_add_material.argtypes = [C.c_void_p, C.POINTER(Material)]
    lib.orc_scene_add_node.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, F32P, F32P]
    lib.orc_scene_set_background.argtypes = [C.c_void_p, C.POINTER(Texture)]
    lib.orc_scene_set_environment.argtypes = [C.c_void_p, C.POINTER(Texture)]
    lib.orc_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
    lib.orc_scene_intersect.argtypes = [C.c_void_p, C.c_int64, F32P, F32P, I32P, I32P, F32P, U8P, F32P, F32P, F32P, C.c_int32, C.POINTER(Stats)]
    lib.orc_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), F32P, U32P, F32P, C.POINTER(Stats)]
    lib.orc_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, F32P, F32P]
    return lib


class OracleMesh:
    def __init__(self, lib, mesh):
        self.lib, self.mesh = lib, mesh
        self._keep = [np.ascontiguousarray(a) for a in (mesh.verts, mesh.tris, mesh.normals, mesh.ntris, mesh.uvw, mesh.ttris)]
        v, t, n, nt, uv, tt = self._keep
        self.h = lib.orc_mesh_create(_p(v, F32P), len(v), _p(t, U32P), _p(n, F32P), len(n), _p(nt, U32P), _p(uv, F32P), len(uv),
                                     _p(tt, U32P), len(t))
        if not self.h:
            raise RuntimeError("Cannot construct empty BVH")

    def nodes(self):
        n = self.lib.orc_mesh_num_nodes(self.h)
        arr = (BvhNode * n)()
        self.lib.orc_mesh_get_nodes(self.h, arr)
        return np.frombuffer(arr, dtype=np.dtype([("left", "i4"), ("right", "i4"), ("prim", "i4"), ("p1", "f4", 3), ("p2", "f4", 3)])).copy()

    def box(self):
        b = np.zeros(6, dtype=np.float32)
        self.lib.orc_mesh_get_box(self.h, _p(b, F32P))
        return b

    def intersect(self, org, dirs, threads=0):
        n = len(org)
        org = np.ascontiguousarray(org, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32); u = np.empty(n, dtype=np.float32); v = np.empty(n, dtype=np.float32)
        side = np.empty(n, dtype=np.uint8)
        st = Stats()
        self.lib.orc_mesh_intersect(self.h, n, _p(org, F32P), _p(dirs, F32P), _p(prim, I32P), _p(t, F32P), _p(u, F32P), _p(v, F32P),
                                    _p(side, U8P), threads, C.byref(st))
        return dict(prim=prim, t=t, u=u, v=v, side=side, stats=st)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_mesh_destroy(self.h)
            self.h = None


class OracleScene:
    """Feeds a scene_io.SceneDesc to the oracle."""

    def __init__(self, lib, desc):
        self.lib, self.desc = lib, desc
        self.meshes = [OracleMesh(lib, m) for m in desc.meshes]
        self.h = lib.orc_scene_create()
        for m in self.meshes:
            lib.orc_scene_add_mesh(self.h, m.h)
        for img in desc.images:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            lib.orc_scene_add_image(self.h, _p(img, U8P), img.shape[1], img.shape[0])
        for m in desc.materials:
            cm = Material()
            fill_material(cm, m)
            lib.orc_scene_add_material(self.h, C.byref(cm))
        for nd in desc.nodes:
            m = np.ascontiguousarray(nd.m, dtype=np.float32); t = np.ascontiguousarray(nd.t, dtype=np.float32)
            lib.orc_scene_add_node(self.h, nd.parent, nd.geom, nd.mesh, nd.material, _p(m, F32P), _p(t, F32P))
        tx = Texture()
        fill_texture(tx, desc.background)
        lib.orc_scene_set_background(self.h, C.byref(tx))
        fill_texture(tx, desc.environment)
        lib.orc_scene_set_environment(self.h, C.byref(tx))
        cam = Camera()
        fill_camera(cam, desc.camera)
        lib.orc_scene_set_camera(self.h, C.byref(cam))

    def intersect(self, org, dirs, threads=0):
        n = len(org)
        org = np.ascontiguousarray(org, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        node = np.empty(n, dtype=np.int32); prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32); side = np.empty(n, dtype=np.uint8)
        pos = np.empty((n, 3), dtype=np.float32); nrm = np.empty((n, 3), dtype=np.float32); uvw = np.empty((n, 3), dtype=np.float32)
        st = Stats()
        self.lib.orc_scene_intersect(self.h, n, _p(org, F32P), _p(dirs, F32P), _p(node, I32P), _p(prim, I32P), _p(t, F32P),
                                     _p(side, U8P), _p(pos, F32P), _p(nrm, F32P), _p(uvw, F32P), threads, C.byref(st))
        return dict(node=node, prim=prim, t=t, side=side, pos=pos, normal=nrm, uvw=uvw, stats=st)

    def render(self, spp=0, max_depth=8, seed=1, first_sample=0, adaptive=False, initial=128, max_samples=512,
               thresholds=(0.4, 0.3, 0.6), threads=0, want_samples=False):
        cam = self.desc.camera
        npix = cam.width * cam.height
        p = RenderParams(adaptive=int(adaptive), spp=spp, first_sample=first_sample, initial_samples=initial, max_samples=max_samples,
                         max_depth=max_depth, seed=seed, threads=threads)
        p.thresholds[:] = thresholds
        rgb = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        cnt = np.zeros((cam.height, cam.width), dtype=np.uint32)
        per = np.zeros((npix, spp, 3), dtype=np.float32) if (want_samples and not adaptive) else None
        st = Stats()
        rc = self.lib.orc_render(self.h, C.byref(p), _p(rgb, F32P), _p(cnt, U32P), _p(per, F32P), C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return dict(rgb=rgb, samples=cnt, per_sample=per, stats=st)

    def primary_ray(self, x, y, i, seed):
        o = np.zeros(3, dtype=np.float32); d = np.zeros(3, dtype=np.float32)
        self.lib.orc_primary_ray(self.h, x, y, i, seed, _p(o, F32P), _p(d, F32P))
        return o, d

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_scene_destroy(self.h)
            self.h = None
