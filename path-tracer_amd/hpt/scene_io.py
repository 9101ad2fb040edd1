"""Host-side scene loading: a Python mirror of the reference loader (load.rs), kept OFF the hot path.

Produces a plain description (numpy arrays + dicts) that is handed, unchanged, to the C-ABI of
libhpt.so (the product) and -- in tests only -- to the CPU oracle.  Arithmetic that shapes the data
crossing the boundary is done in f32 in the reference's operation order:

* load_transform   load.rs:367-411   (scale/translate/rotate accumulation, document order)
* load_camera      load.rs:241-268   (up = (dir x up) x dir, not renormalised)
* load_material    load.rs:118-202   (only the fields the integrator reads are kept)
* load_texture     load.rs:330-365
* load_obj         load.rs:270-328   (objects[0] only; triangles only; vt and vn mandatory)
* polygon fan      wavefront_obj 5.1 (NOT in /root/reference; restated from the crate's published
                   behaviour: a polygon is fanned around its LAST vertex -- parity unpinned)

Matrices are 3x3 COLUMN-major float32[9] (m[c*3+r]) like cgmath::Matrix3.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
import zlib
from dataclasses import dataclass, field

import numpy as np

F = np.float32
GEOM_NONE, GEOM_SPHERE, GEOM_PLANE, GEOM_MESH = 0, 1, 2, 3
TEX_BLANK, TEX_CHECKER, TEX_IMAGE = 0, 1, 2


def _ident():
    return np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=F)


def _matmul(a, b):
    """cgmath Matrix3 * Matrix3: out[c][r] = row_r(a) . col_c(b), summed as (x + y) + z in f32."""
    out = np.zeros(9, dtype=F)
    for c in range(3):
        for r in range(3):
            out[c * 3 + r] = F(F(a[0 * 3 + r] * b[c * 3 + 0]) + F(a[1 * 3 + r] * b[c * 3 + 1])) + F(a[2 * 3 + r] * b[c * 3 + 2])
    return out


def _matvec(a, v):
    out = np.zeros(3, dtype=F)
    for r in range(3):
        out[r] = F(F(a[r] * v[0]) + F(a[3 + r] * v[1])) + F(a[6 + r] * v[2])
    return out


def _pf(s):
    """Rust's str::parse::<f32>().ok()"""
    try:
        return F(float(s.strip())) if s is not None else None
    except ValueError:
        return None


def _vec3(attrs, default):
    out = np.array(default, dtype=F)
    for k, name in enumerate("xyz"):
        v = _pf(attrs.get(name))
        if v is not None:
            out[k] = v
    return out


def _color(attrs):
    r, g, b = _pf(attrs.get("r")), _pf(attrs.get("g")), _pf(attrs.get("b"))
    if r is None or g is None or b is None:
        return None
    return np.array([r, g, b], dtype=F)


def load_transform(el):
    """load.rs:367-411"""
    m, t = _ident(), np.zeros(3, dtype=F)
    for ch in el:
        if ch.tag == "scale":
            if "value" in ch.attrib:
                s = F(float(ch.attrib["value"]))
                mat = (s * _ident()).astype(F)
            else:
                d = _vec3(ch.attrib, (1, 1, 1))
                mat = np.array([d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], dtype=F)
            m = _matmul(mat, m)
            t = _matvec(mat, t)
        elif ch.tag == "translate":
            t = (t + _vec3(ch.attrib, (0, 0, 0))).astype(F)
        elif ch.tag == "rotate":
            if "angle" not in ch.attrib:
                raise ValueError("no angle given for rotate")
            # cgmath Deg -> Rad: deg * (PI/180 as f32); sin_cos in f32
            rad = F(F(float(ch.attrib["angle"])) * F(math.pi / 180.0))
            s, c = F(np.sin(rad)), F(np.cos(rad))
            if "x" in ch.attrib:
                rot = np.array([1, 0, 0, 0, c, s, 0, -s, c], dtype=F)
            elif "y" in ch.attrib:
                rot = np.array([c, 0, -s, 0, 1, 0, s, 0, c], dtype=F)
            else:
                rot = np.array([c, s, 0, -s, c, 0, 0, 0, 1], dtype=F)
            m = _matmul(rot, m)
            t = _matvec(rot, t)
    return m, t


@dataclass
class Texture:
    kind: int = TEX_BLANK
    image: int = -1
    color: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F))
    color1: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F))
    color2: np.ndarray = field(default_factory=lambda: np.ones(3, dtype=F))
    m: np.ndarray = field(default_factory=_ident)
    t: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F))


@dataclass
class Material:
    name: str = ""
    diffuse: Texture = field(default_factory=Texture)
    reflection: Texture = field(default_factory=Texture)
    refraction: Texture = field(default_factory=Texture)
    emission: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F))
    reflection_glossiness: float = 0.0
    refraction_glossiness: float = 0.0
    refraction_index: float = 1.0


@dataclass
class Mesh:
    verts: np.ndarray
    tris: np.ndarray
    normals: np.ndarray
    ntris: np.ndarray
    uvw: np.ndarray
    ttris: np.ndarray


@dataclass
class Node:
    parent: int
    geom: int
    mesh: int
    material: int
    m: np.ndarray
    t: np.ndarray
    name: str = ""


@dataclass
class Camera:
    pos: np.ndarray
    dir: np.ndarray
    up: np.ndarray
    fov: float
    width: int
    height: int
    focaldist: float = 1.0
    dof: float = 0.0


@dataclass
class SceneDesc:
    nodes: list = field(default_factory=list)
    materials: list = field(default_factory=list)
    meshes: list = field(default_factory=list)
    images: list = field(default_factory=list)  # (rgb8 ndarray[h,w,3])
    background: Texture = field(default_factory=Texture)
    environment: Texture = field(default_factory=Texture)
    camera: Camera = None


def tex_color(color):
    return Texture(color=np.array(color, dtype=F))


def read_png_rgb8(path):
    """Minimal PNG reader (8-bit RGB/RGBA/grey, non-interlaced) over zlib; load.rs:437-443 assumes RGB8."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a png: %s" % path)
    pos, idat, w = 8, b"", None
    while pos < len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
                raise ValueError("unsupported png format")
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = zlib.decompress(idat)
    stride = w * ch
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for y in range(h):
        ft = raw[p]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=p + 1).astype(np.int32)
        p += 1 + stride
        cur = np.zeros(stride, dtype=np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 1:
                    pr = a
                elif ft == 3:
                    pr = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if ch == 1:
        img = np.repeat(img, 3, axis=2)
    elif ch == 2:
        img = np.repeat(img[:, :, :1], 3, axis=2)
    elif ch == 4:
        img = img[:, :, :3]
    return np.ascontiguousarray(img)


def write_png_rgba8(path, rgba):
    """main.rs:120-130 save_img: RGBA, 8 bit."""
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(typ, body):
        c = struct.pack(">I", len(body)) + typ + body
        return c + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def load_texture(el, default_color, scene, base_dir):
    """load.rs:330-365"""
    tx = Texture()
    val, col = el.attrib.get("value"), _color(el.attrib)
    if val is not None or col is not None:
        v = _pf(val)
        v = F(1.0) if v is None else v
        tx.color = (v * (col if col is not None else np.ones(3, dtype=F))).astype(F)
    else:
        tx.color = np.array(default_color, dtype=F)
    name = el.attrib.get("texture")
    if name is not None:
        if name == "checkerboard":
            tx.kind = TEX_CHECKER
            c1, c2 = el.find("color1"), el.find("color2")
            if c1 is not None and _color(c1.attrib) is not None:
                tx.color1 = _color(c1.attrib)
            if c2 is not None and _color(c2.attrib) is not None:
                tx.color2 = _color(c2.attrib)
        else:
            tx.kind = TEX_IMAGE
            scene.images.append(read_png_rgb8(os.path.join(base_dir, name)))
            tx.image = len(scene.images) - 1
    tx.m, tx.t = load_transform(el)
    return tx


def load_material(el, scene, base_dir):
    """load.rs:118-202 (type must be blinn; <specular> mandatory although unused)"""
    if "type" not in el.attrib:
        raise ValueError("no type for material")
    if "name" not in el.attrib:
        raise ValueError("no name for material")
    if el.attrib["type"] != "blinn":
        raise ValueError("unknown material type")
    m = Material(name=el.attrib["name"])
    m.diffuse = tex_color((0.5, 0.5, 0.5))
    d = el.find("diffuse")
    if d is not None:
        m.diffuse = load_texture(d, (1, 1, 1), scene, base_dir)
    if el.find("specular") is None:
        raise ValueError("no specular found for <material>")
    e = el.find("emission")
    if e is not None:
        val, col = e.attrib.get("value"), _color(e.attrib)
        if val is not None or col is not None:
            v = _pf(val)
            v = F(1.0) if v is None else v
            m.emission = (v * (col if col is not None else np.ones(3, dtype=F))).astype(F)
    r = el.find("reflection")
    if r is not None:
        m.reflection = load_texture(r, (1, 1, 1), scene, base_dir)
        g = _pf(r.attrib.get("glossiness"))
        m.reflection_glossiness = float(g) if g is not None else 0.0
    r = el.find("refraction")
    if r is not None:
        m.refraction = load_texture(r, (1, 1, 1), scene, base_dir)
        i = _pf(r.attrib.get("index"))
        m.refraction_index = float(i) if i is not None else 1.0
        g = _pf(r.attrib.get("glossiness"))
        m.refraction_glossiness = float(g) if g is not None else 0.0
    return m


def parse_obj(path):
    """load.rs:270-328 over a wavefront_obj-5.1-like parse: first object only, polygons fanned around
    their LAST vertex ((a,b,c,d) -> (d,a,b),(d,b,c); (a,b,c) -> (c,a,b)); vt/vn indices mandatory."""
    verts, normals, uvw = [], [], []
    tris, ntris, ttris = [], [], []
    n_objects = 0
    with open(path, "r") as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "o":
                n_objects += 1
                if n_objects > 1:
                    break
            elif p[0] == "v":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "vn":
                normals.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "vt":
                uvw.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0, float(p[3]) if len(p) > 3 else 0.0])
            elif p[0] == "f":
                idx = []
                for tok in p[1:]:
                    q = tok.split("/")
                    if len(q) < 3 or q[1] == "" or q[2] == "":
                        raise ValueError("face without vt/vn index (load.rs:303-304 unwrap)")
                    vi, ti, ni = int(q[0]), int(q[1]), int(q[2])
                    vi = vi - 1 if vi > 0 else len(verts) + vi
                    ti = ti - 1 if ti > 0 else len(uvw) + ti
                    ni = ni - 1 if ni > 0 else len(normals) + ni
                    idx.append((vi, ti, ni))
                if len(idx) < 3:
                    continue  # points/lines are not Primitive::Triangle (load.rs:301)
                last = idx[-1]
                for k in range(len(idx) - 2):
                    a, b = idx[k], idx[k + 1]
                    tris.append((last[0], a[0], b[0]))
                    ttris.append((last[1], a[1], b[1]))
                    ntris.append((last[2], a[2], b[2]))
    if not verts:
        raise ValueError("no objects found in file")
    return Mesh(verts=np.array(verts, dtype=np.float64).astype(F), tris=np.array(tris, dtype=np.uint32).reshape(-1, 3),
                normals=np.array(normals, dtype=np.float64).astype(F).reshape(-1, 3), ntris=np.array(ntris, dtype=np.uint32).reshape(-1, 3),
                uvw=np.array(uvw, dtype=np.float64).astype(F).reshape(-1, 3), ttris=np.array(ttris, dtype=np.uint32).reshape(-1, 3))


def load_camera(el):
    """load.rs:241-268"""
    def need(tag):
        c = el.find(tag)
        if c is None:
            raise ValueError("no <%s> tag found in <camera>" % tag)
        return c
    pos = _vec3(need("position").attrib, (0, 0, 0))
    default_target = (pos + np.array([0, 0, -1], dtype=F)).astype(F)
    d = (_vec3(need("target").attrib, default_target) - pos).astype(F)
    d = normalize(d)
    up = _vec3(need("up").attrib, (0, 1, 0))
    fov = float(F(float(need("fov").attrib["value"])))
    w = int(need("width").attrib["value"])
    h = int(need("height").attrib["value"])
    fd = el.find("focaldist")
    focaldist = float(_pf(fd.attrib.get("value")) or 1.0) if fd is not None and _pf(fd.attrib.get("value")) is not None else 1.0
    df = el.find("dof")
    dof = float(_pf(df.attrib.get("value"))) if df is not None and _pf(df.attrib.get("value")) is not None else 0.0
    up = cross(cross(d, up), d)
    return Camera(pos=pos, dir=d, up=up, fov=fov, width=w, height=h, focaldist=focaldist, dof=dof)


def cross(a, b):
    return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], dtype=F)


def normalize(a):
    d = F(F(a[0] * a[0]) + F(a[1] * a[1])) + F(a[2] * a[2])
    return (a * F(F(1.0) / np.sqrt(d))).astype(F)


def load_scene(filename):
    """load.rs:17-76.  OBJ / PNG paths are resolved relative to the scene file's directory (the reference
    resolves them against the CWD, load.rs:89,271 -- run from that directory for identical behaviour)."""
    base_dir = os.path.dirname(os.path.abspath(filename))
    root = ET.parse(filename).getroot()
    scene_xml = root.find("scene")
    if scene_xml is None:
        raise ValueError("no <scene> tag found")
    sc = SceneDesc()
    bg = scene_xml.find("background")
    if bg is not None:
        sc.background = load_texture(bg, (1, 1, 1), sc, base_dir)
    env = scene_xml.find("environment")
    if env is not None:
        sc.environment = load_texture(env, (1, 1, 1), sc, base_dir)

    pending = []  # (node index, material name)

    def load_node(el, parent):
        geom, mesh = GEOM_NONE, -1
        typ = el.attrib.get("type")
        if typ is not None:
            if typ == "sphere":
                geom = GEOM_SPHERE
            elif typ == "plane":
                geom = GEOM_PLANE
            elif typ == "obj":
                if "name" not in el.attrib:
                    raise ValueError("no filename given for obj")
                sc.meshes.append(parse_obj(os.path.join(base_dir, el.attrib["name"])))
                geom, mesh = GEOM_MESH, len(sc.meshes) - 1
            else:
                raise ValueError("unknown object type")
            if "material" not in el.attrib:
                raise ValueError("no material given for object")
        m, t = load_transform(el)
        sc.nodes.append(Node(parent=parent, geom=geom, mesh=mesh, material=-1, m=m, t=t, name=el.attrib.get("name", "")))
        me = len(sc.nodes) - 1
        if typ is not None:
            pending.append((me, el.attrib["material"]))
        for ch in el:
            if ch.tag == "object":
                load_node(ch, me)

    names = {}
    for ch in scene_xml:
        if ch.tag == "object":
            load_node(ch, -1)
        elif ch.tag == "material":
            mat = load_material(ch, sc, base_dir)
            if mat.name in names:          # HashMap::insert replaces
                sc.materials[names[mat.name]] = mat
            else:
                names[mat.name] = len(sc.materials)
                sc.materials.append(mat)
    for idx, name in pending:
        if name not in names:
            raise ValueError("material does not exist for object")   # scene.rs:133
        sc.nodes[idx].material = names[name]
    cam = root.find("camera")
    if cam is None:
        raise ValueError("no <camera> tag found")
    sc.camera = load_camera(cam)
    return sc
