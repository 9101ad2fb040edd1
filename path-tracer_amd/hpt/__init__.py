"""ctypes binding of libhpt.so -- the C ABI declared in include/hpt.h.

This is plumbing above the boundary (tests, bench.py, the Python CLI); every compute call goes through the C ABI
into the HIP kernels.  There is no CPU fallback: importing works without a GPU (symbols only), but creating a
context raises when no MI355X is visible, and a missing library raises at import.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPT_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libhpt.so")   # HPT_LIB_PATH: A/B builds in tuning runs

F32P = C.POINTER(C.c_float)
U32P = C.POINTER(C.c_uint32)
I32P = C.POINTER(C.c_int32)
U8P = C.POINTER(C.c_uint8)

GEOM_NONE, GEOM_SPHERE, GEOM_PLANE, GEOM_MESH = 0, 1, 2, 3
RENDER_COUNTERS, RENDER_TIME_KERNELS, RENDER_EXACT_SHADE = 1, 2, 4


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
                ("spp_per_pass", C.c_int32), ("flags", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("inst_tests", C.c_uint64), ("seconds_total", C.c_double),
                ("seconds_raygen", C.c_double), ("seconds_extend", C.c_double), ("seconds_shade", C.c_double),
                ("seconds_resolve", C.c_double), ("launches_extend", C.c_uint32), ("passes", C.c_uint32), ("samples", C.c_uint64),
                ("seconds_analytic", C.c_double), ("rays_mesh", C.c_uint64)]


class Accum(C.Structure):
    """hpt_accum: the radiance accumulator of one pixel (64-bit fixed point, 2^-24 quantum) -- as numpy / torch: int64 [npix, 4]"""
    _fields_ = [("r", C.c_int64), ("g", C.c_int64), ("b", C.c_int64), ("nan", C.c_int64)]


ACCUM_FRAC_BITS = 24
ACCUM_SAMPLE_MAX = 1048576.0


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("seconds_kernel", C.c_double), ("seconds_total", C.c_double)]


class SceneInfo(C.Structure):
    _fields_ = [("n_instances", C.c_int32), ("n_mesh_instances", C.c_int32), ("n_meshes", C.c_int32), ("n_materials", C.c_int32),
                ("n_nodes", C.c_int32), ("tlas_nodes", C.c_int32), ("uses_tlas", C.c_int32), ("stack_entries", C.c_int32),
                ("n_triangles", C.c_int64), ("device_bytes", C.c_uint64), ("node_bytes", C.c_int32), ("trees_optimized", C.c_int32)]


class GroupStats(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("transport", C.c_int32), ("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("samples", C.c_uint64), ("seconds_render_max", C.c_double), ("seconds_reduce", C.c_double), ("seconds_total", C.c_double)]


# every symbol include/hpt.h declares (tests check that the library exports all of them)
EXPORTS = [
    "hpt_version", "hpt_last_error", "hpt_device_count", "hpt_rng_f32", "hpt_halton", "hpt_tonemap_rgba8",
    "hpt_context_create", "hpt_context_destroy", "hpt_context_set_stream", "hpt_context_synchronize",
    "hpt_context_set_passes_in_flight",
    "hpt_scene_create", "hpt_scene_destroy", "hpt_scene_add_mesh", "hpt_scene_add_image", "hpt_scene_add_material",
    "hpt_scene_add_node", "hpt_scene_set_background", "hpt_scene_set_environment", "hpt_scene_set_camera", "hpt_scene_commit",
    "hpt_trace_rays", "hpt_trace_rays_device", "hpt_scene_intersect", "hpt_render", "hpt_render_device", "hpt_render_samples",
    "hpt_scene_info", "hpt_render_accum_device", "hpt_render_accum", "hpt_accum_resolve_device", "hpt_accum_resolve",
    "hpt_adaptive_select_device", "hpt_render_device_noalloc",
    "hpt_group_create", "hpt_group_destroy", "hpt_group_size", "hpt_group_context", "hpt_group_render",
]


class HptError(RuntimeError):
    pass


def _preload_torch_hip():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7,
    requested by torch as plain "libamdhip64.so"), so loading libhpt.so first would pull /opt/rocm's copy in and a later
    `import torch` would bring a second runtime that finds no GPU.  Loading torch's copy first makes libhpt.so's
    NEEDED libamdhip64.so.7 resolve to it (SONAME match) and torch's later request dedupe to the same file.
    Without torch installed (a plain C / Rust host) the system runtime is used.  HPT_SYSTEM_HIP=1 skips this."""
    if os.environ.get("HPT_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def _load():
    _preload_torch_hip()
    if not os.path.exists(LIB_PATH):
        raise ImportError("libhpt.so is not built (%s): run `make -C path-tracer_amd` or __graft_entry__.build(); "
                          "there is no fallback path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.hpt_version.restype = C.c_char_p
    lib.hpt_last_error.restype = C.c_char_p
    lib.hpt_rng_f32.restype = C.c_float
    lib.hpt_rng_f32.argtypes = [C.c_uint32] * 4
    lib.hpt_halton.restype = C.c_float
    lib.hpt_halton.argtypes = [C.c_int32, C.c_int32]
    lib.hpt_tonemap_rgba8.argtypes = [F32P, C.c_uint32, U8P]
    lib.hpt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.hpt_context_destroy.argtypes = [C.c_void_p]
    lib.hpt_context_destroy.restype = None
    lib.hpt_context_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.hpt_context_set_passes_in_flight.argtypes = [C.c_void_p, C.c_int]
    lib.hpt_context_synchronize.argtypes = [C.c_void_p]
    lib.hpt_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hpt_scene_destroy.argtypes = [C.c_void_p]
    lib.hpt_scene_destroy.restype = None
    lib.hpt_scene_add_mesh.argtypes = [C.c_void_p, F32P, C.c_int32, U32P, F32P, C.c_int32, U32P, F32P, C.c_int32, U32P, C.c_int32,
                                       C.POINTER(BvhNode), C.c_int32, I32P]
    lib.hpt_scene_add_image.argtypes = [C.c_void_p, U8P, C.c_int32, C.c_int32, I32P]
    lib.hpt_scene_add_material.argtypes = [C.c_void_p, C.POINTER(Material), I32P]
    lib.hpt_scene_add_node.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, F32P, F32P, I32P]
    lib.hpt_scene_set_background.argtypes = [C.c_void_p, C.POINTER(Texture)]
    lib.hpt_scene_set_environment.argtypes = [C.c_void_p, C.POINTER(Texture)]
    lib.hpt_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
    lib.hpt_scene_commit.argtypes = [C.c_void_p]
    lib.hpt_trace_rays.argtypes = [C.c_void_p, C.c_int32, C.c_int64, F32P, F32P, I32P, F32P, F32P, F32P, U8P, C.POINTER(TraceStats)]
    lib.hpt_trace_rays_device.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.POINTER(TraceStats)]
    lib.hpt_scene_intersect.argtypes = [C.c_void_p, C.c_int64, F32P, F32P, I32P, I32P, F32P, U8P, F32P, F32P, F32P]
    lib.hpt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), F32P, U32P, C.POINTER(RenderStats)]
    lib.hpt_render_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
    lib.hpt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
    lib.hpt_group_create.argtypes = [I32P, C.c_int, C.POINTER(C.c_void_p)]
    lib.hpt_group_destroy.argtypes = [C.c_void_p]
    lib.hpt_group_destroy.restype = None
    lib.hpt_group_size.argtypes = [C.c_void_p]
    lib.hpt_group_context.argtypes = [C.c_void_p, C.c_int]
    lib.hpt_group_context.restype = C.c_void_p
    lib.hpt_group_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RenderParams), F32P, U32P, C.POINTER(GroupStats)]
    lib.hpt_render_samples.argtypes = [C.c_void_p, C.POINTER(RenderParams), F32P]
    lib.hpt_scene_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
    lib.hpt_render_accum_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(RenderStats)]
    lib.hpt_render_accum.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(RenderStats)]
    lib.hpt_accum_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hpt_accum_resolve.argtypes = [C.c_void_p, C.c_uint32, U32P, C.c_uint32, F32P]
    lib.hpt_adaptive_select_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, F32P, C.c_void_p, C.c_void_p, C.c_uint32, U32P]
    return lib


lib = _load()


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def check(rc):
    if rc != 0:
        raise HptError("hpt error %d: %s" % (rc, lib.hpt_last_error().decode()))


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


def device_count():
    return lib.hpt_device_count()


def set_tree_opt(mode):
    """tuning / test hook (not in include/hpt.h): 0 keeps the handed-over tree topology at later commits, 1 restructures
    it (default), -1 back to the environment (HPT_TREE_OPT)"""
    lib.hpt_debug_set_tree_opt(int(mode))


class Context:
    def __init__(self, device=0, _borrowed=None):
        if _borrowed is not None:      # a context owned by a Group (hpt_group_context): never destroyed from here
            self.h, self.borrowed = C.c_void_p(_borrowed), True
            return
        h = C.c_void_p()
        check(lib.hpt_context_create(device, C.byref(h)))
        self.h, self.borrowed = h, False

    def set_stream(self, stream_ptr):
        check(lib.hpt_context_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_passes_in_flight(self, n):
        check(lib.hpt_context_set_passes_in_flight(self.h, int(n)))

    def synchronize(self):
        check(lib.hpt_context_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None) and lib is not None and not self.borrowed:
            lib.hpt_context_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()


class Group:
    """hpt_group_*: one process, several GPUs.  Build ONE scene on `group.context(0)`; `render` replicates it, splits the
    samples of every pixel over the GPUs and sum-reduces the accumulators onto GPU 0 (RCCL, or peer copies + ordered adds
    when a device is listed twice / HPT_GROUP_TRANSPORT=p2p)."""

    def __init__(self, devices=None):
        devs = np.ascontiguousarray(devices if devices is not None else [], dtype=np.int32)
        h = C.c_void_p()
        check(lib.hpt_group_create(_p(devs, I32P) if len(devs) else None, len(devs), C.byref(h)))
        self.h = h

    def size(self):
        return lib.hpt_group_size(self.h)

    def context(self, rank=0):
        p = lib.hpt_group_context(self.h, rank)
        if not p:
            raise HptError(lib.hpt_last_error().decode())
        return Context(_borrowed=p)

    def render(self, scene, **kw):
        p = Scene.params(**kw)
        rgb = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
        cnt = np.zeros((scene.height, scene.width), dtype=np.uint32)
        st = GroupStats()
        check(lib.hpt_group_render(self.h, scene.h, C.byref(p), _p(rgb, F32P), _p(cnt, U32P), C.byref(st)))
        return dict(rgb=rgb, samples=cnt, stats=st)

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.hpt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Scene:
    """Hands a scene_io.SceneDesc over the C ABI: hpt_scene_add_* then hpt_scene_commit.

    host_bvh: callable(mesh) -> numpy structured array of hpt_bvh_node (the tree the host built, bvh.rs:35), or None
    to let the library build its own tree.
    """

    def __init__(self, ctx, desc=None, host_bvh=None):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib.hpt_scene_create(ctx.h, C.byref(h)))
        self.h = h
        self.desc = desc
        if desc is not None:
            self.build(desc, host_bvh)

    def build(self, desc, host_bvh=None):
        for m in desc.meshes:
            self.add_mesh(m, host_bvh(m) if host_bvh else None)
        for img in desc.images:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            check(lib.hpt_scene_add_image(self.h, _p(img, U8P), img.shape[1], img.shape[0], None))
        for m in desc.materials:
            cm = Material()
            fill_material(cm, m)
            check(lib.hpt_scene_add_material(self.h, C.byref(cm), None))
        for nd in desc.nodes:
            self.add_node(nd.parent, nd.geom, nd.mesh, nd.material, nd.m, nd.t)
        tx = Texture()
        fill_texture(tx, desc.background)
        check(lib.hpt_scene_set_background(self.h, C.byref(tx)))
        fill_texture(tx, desc.environment)
        check(lib.hpt_scene_set_environment(self.h, C.byref(tx)))
        self.set_camera(desc.camera)
        self.commit()

    def add_mesh(self, m, nodes=None):
        v = np.ascontiguousarray(m.verts, dtype=np.float32); t = np.ascontiguousarray(m.tris, dtype=np.uint32)
        n = np.ascontiguousarray(m.normals, dtype=np.float32); nt = np.ascontiguousarray(m.ntris, dtype=np.uint32)
        uv = np.ascontiguousarray(m.uvw, dtype=np.float32); tt = np.ascontiguousarray(m.ttris, dtype=np.uint32)
        mid = C.c_int32(-1)
        if nodes is not None:
            nodes = np.ascontiguousarray(nodes)
            assert nodes.dtype.itemsize == C.sizeof(BvhNode)
            np_ptr, nn = nodes.ctypes.data_as(C.POINTER(BvhNode)), len(nodes)
        else:
            np_ptr, nn = None, 0
        check(lib.hpt_scene_add_mesh(self.h, _p(v, F32P), len(v), _p(t, U32P), _p(n, F32P), len(n), _p(nt, U32P), _p(uv, F32P), len(uv),
                                     _p(tt, U32P), len(t), np_ptr, nn, C.byref(mid)))
        return mid.value

    def add_node(self, parent, geom, mesh, material, m, t):
        m = np.ascontiguousarray(m, dtype=np.float32); t = np.ascontiguousarray(t, dtype=np.float32)
        nid = C.c_int32(-1)
        check(lib.hpt_scene_add_node(self.h, parent, geom, mesh, material, _p(m, F32P), _p(t, F32P), C.byref(nid)))
        return nid.value

    def set_camera(self, cam):
        c = Camera()
        fill_camera(c, cam)
        check(lib.hpt_scene_set_camera(self.h, C.byref(c)))
        self.width, self.height = cam.width, cam.height

    def commit(self):
        check(lib.hpt_scene_commit(self.h))

    def trace_rays(self, mesh_id, org, dirs, want_stats=True):
        org = np.ascontiguousarray(org, dtype=np.float32); dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = len(org)
        prim = np.empty(n, dtype=np.int32); t = np.empty(n, dtype=np.float32); u = np.empty(n, dtype=np.float32)
        v = np.empty(n, dtype=np.float32); side = np.empty(n, dtype=np.uint8)
        st = TraceStats()
        check(lib.hpt_trace_rays(self.h, mesh_id, n, _p(org, F32P), _p(dirs, F32P), _p(prim, I32P), _p(t, F32P), _p(u, F32P), _p(v, F32P),
                                 _p(side, U8P), C.byref(st) if want_stats else None))
        return dict(prim=prim, t=t, u=u, v=v, side=side, stats=st)

    def trace_rays_device(self, mesh_id, n, org4_ptr, dir4_ptr, hit4_ptr, side_ptr, count=False, want_stats=False):
        st = TraceStats()
        check(lib.hpt_trace_rays_device(self.h, mesh_id, n, C.c_void_p(org4_ptr), C.c_void_p(dir4_ptr), C.c_void_p(hit4_ptr),
                                        C.c_void_p(side_ptr), int(count), C.byref(st) if want_stats else None))
        return st

    def intersect(self, org, dirs):
        org = np.ascontiguousarray(org, dtype=np.float32); dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = len(org)
        node = np.empty(n, dtype=np.int32); prim = np.empty(n, dtype=np.int32); t = np.empty(n, dtype=np.float32)
        side = np.empty(n, dtype=np.uint8)
        pos = np.empty((n, 3), dtype=np.float32); nrm = np.empty((n, 3), dtype=np.float32); uvw = np.empty((n, 3), dtype=np.float32)
        check(lib.hpt_scene_intersect(self.h, n, _p(org, F32P), _p(dirs, F32P), _p(node, I32P), _p(prim, I32P), _p(t, F32P), _p(side, U8P),
                                      _p(pos, F32P), _p(nrm, F32P), _p(uvw, F32P)))
        return dict(node=node, prim=prim, t=t, side=side, pos=pos, normal=nrm, uvw=uvw)

    @staticmethod
    def params(spp=0, max_depth=8, seed=1, first_sample=0, adaptive=False, initial=128, max_samples=512,
               thresholds=(0.4, 0.3, 0.6), spp_per_pass=0, flags=0):
        p = RenderParams(adaptive=int(adaptive), spp=spp, first_sample=first_sample, initial_samples=initial, max_samples=max_samples,
                         max_depth=max_depth, seed=seed, spp_per_pass=spp_per_pass, flags=flags)
        p.thresholds[:] = thresholds
        return p

    def render(self, **kw):
        p = self.params(**kw)
        rgb = np.zeros((self.height, self.width, 3), dtype=np.float32)
        cnt = np.zeros((self.height, self.width), dtype=np.uint32)
        st = RenderStats()
        check(lib.hpt_render(self.h, C.byref(p), _p(rgb, F32P), _p(cnt, U32P), C.byref(st)))
        return dict(rgb=rgb, samples=cnt, stats=st)

    def render_device(self, out_ptr, want_stats=False, **kw):
        p = self.params(**kw)
        st = RenderStats()
        check(lib.hpt_render_device(self.h, C.byref(p), C.c_void_p(out_ptr), C.byref(st) if want_stats else None))
        return st

    def render_device_noalloc(self, out_ptr, want_stats=False, **kw):
        """hpt_render_device_noalloc: pass buffers from the library's static device array, nothing allocated"""
        p = self.params(**kw)
        st = RenderStats()
        check(lib.hpt_render_device_noalloc(self.h, C.byref(p), C.c_void_p(out_ptr), C.byref(st) if want_stats else None))
        return st

    def info(self):
        out = SceneInfo()
        check(lib.hpt_scene_info(self.h, C.byref(out)))
        return out

    def render_accum_device(self, accum_ptr, minmax_ptr, pixel_list_ptr, n_pixels, want_stats=False, **kw):
        """a sample range added to a device accumulator (hpt_render_accum_device: int64 [npix, 4]); pixel_list_ptr = 0 for every pixel"""
        p = self.params(**kw)
        st = RenderStats()
        check(lib.hpt_render_accum_device(self.h, C.byref(p), C.c_void_p(pixel_list_ptr or None), n_pixels, C.c_void_p(accum_ptr),
                                          C.c_void_p(minmax_ptr or None), C.byref(st) if want_stats else None))
        return st

    def render_accum(self, accum, want_stats=False, **kw):
        """hpt_render_accum: a sample range added to a HOST accumulator (numpy int64 [npix, 4], modified in place)"""
        assert accum.dtype == np.int64 and accum.flags["C_CONTIGUOUS"] and accum.size == self.width * self.height * 4
        p = self.params(**kw)
        st = RenderStats()
        check(lib.hpt_render_accum(self.h, C.byref(p), accum.ctypes.data_as(C.c_void_p), C.byref(st) if want_stats else None))
        return st

    def accum_resolve_device(self, accum_ptr, out_ptr, num_samples=0, spp_ptr=0):
        """hpt_accum_resolve_device: out (float32 [npix, 3], device) = f32(sum) / n; n = spp[pixel] | num_samples | no division"""
        check(lib.hpt_accum_resolve_device(self.h, C.c_void_p(accum_ptr), C.c_void_p(spp_ptr or None), int(num_samples), C.c_void_p(out_ptr)))

    def adaptive_select_device(self, list_in_ptr, n_in, minmax_ptr, thresholds, list_out_ptr, spp_ptr, next_num_samples):
        thr = (C.c_float * 3)(*[float(t) for t in thresholds])
        n_out = C.c_uint32(0)
        check(lib.hpt_adaptive_select_device(self.h, C.c_void_p(list_in_ptr or None), n_in, C.c_void_p(minmax_ptr), thr, C.c_void_p(list_out_ptr),
                                             C.c_void_p(spp_ptr), next_num_samples, C.byref(n_out)))
        return n_out.value

    def render_samples(self, **kw):
        p = self.params(**kw)
        out = np.zeros((self.height * self.width, p.spp, 3), dtype=np.float32)
        check(lib.hpt_render_samples(self.h, C.byref(p), _p(out, F32P)))
        return out

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.hpt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def accum_resolve(accum, num_samples=0, samples_per_px=None):
    """hpt_accum_resolve on the host: int64 [npix, 4] -> float32 [npix, 3]"""
    accum = np.ascontiguousarray(accum, dtype=np.int64).reshape(-1, 4)
    out = np.empty((len(accum), 3), dtype=np.float32)
    spp = np.ascontiguousarray(samples_per_px, dtype=np.uint32).reshape(-1) if samples_per_px is not None else None
    check(lib.hpt_accum_resolve(accum.ctypes.data_as(C.c_void_p), len(accum), _p(spp, U32P), int(num_samples), _p(out, F32P)))
    return out


def tonemap_rgba8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    out = np.zeros((h, w, 4), dtype=np.uint8)
    check(lib.hpt_tonemap_rgba8(_p(rgb, F32P), h * w, _p(out, U8P)))
    return out
