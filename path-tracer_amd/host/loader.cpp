// loader.cpp -- load.rs in C++ (scene XML, OBJ, textures, camera), the host-side BVH build of bvh.rs, and the
// hand-over through the C ABI.  Compiled with -ffp-contract=off: the f32 arithmetic that shapes the data crossing the
// boundary keeps the reference's operation order (cgmath sums (x + y) + z).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include "hpt_host.h"

namespace hpt_host {
namespace {

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("file not found: " + path);   // load.rs:18,271
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}
std::string dir_of(const std::string& path) {
    const size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
// Rust's str::parse::<f32>().ok()
bool parse_f32(const std::string* s, float& out) {
    if (!s) return false;
    const std::string t = trim(*s);
    if (t.empty()) return false;
    char* end = nullptr;
    const float v = std::strtof(t.c_str(), &end);
    if (end != t.c_str() + t.size()) return false;
    out = v;
    return true;
}
float need_f32(const std::string* s, const char* what) {
    float v;
    if (!parse_f32(s, v)) throw Error(what);
    return v;
}
void read_vec3(const Element& e, const float def[3], float out[3]) {   // load.rs read_vector3_default
    static const char* names[3] = {"x", "y", "z"};
    for (int k = 0; k < 3; ++k) { out[k] = def[k]; float v; if (parse_f32(e.attr(names[k]), v)) out[k] = v; }
}
bool read_color(const Element& e, float out[3]) {                       // load.rs read_color
    float r, g, b;
    if (!parse_f32(e.attr("r"), r) || !parse_f32(e.attr("g"), g) || !parse_f32(e.attr("b"), b)) return false;
    out[0] = r; out[1] = g; out[2] = b;
    return true;
}
void ident(float m[9]) { for (int i = 0; i < 9; ++i) m[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
// cgmath Matrix3 * Matrix3, column-major: out[c][r] = row_r(a) . col_c(b)
void matmul(const float a[9], const float b[9], float out[9]) {
    float o[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) o[c * 3 + r] = (a[0 * 3 + r] * b[c * 3 + 0] + a[1 * 3 + r] * b[c * 3 + 1]) + a[2 * 3 + r] * b[c * 3 + 2];
    std::memcpy(out, o, sizeof o);
}
void matvec(const float a[9], const float v[3], float out[3]) {
    float o[3];
    for (int r = 0; r < 3; ++r) o[r] = (a[r] * v[0] + a[3 + r] * v[1]) + a[6 + r] * v[2];
    std::memcpy(out, o, sizeof o);
}
void cross(const float a[3], const float b[3], float out[3]) {
    const float o[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    std::memcpy(out, o, sizeof o);
}

// load.rs:367-411
void load_transform(const Element& e, float m[9], float t[3]) {
    ident(m); t[0] = t[1] = t[2] = 0.0f;
    for (const Element& ch : e.children) {
        if (ch.tag == "scale") {
            float mat[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (ch.attr("value")) {
                const float s = need_f32(ch.attr("value"), "could not parse scale value");
                mat[0] = mat[4] = mat[8] = s;
            } else {
                const float one[3] = {1, 1, 1}; float d[3];
                read_vec3(ch, one, d);
                mat[0] = d[0]; mat[4] = d[1]; mat[8] = d[2];
            }
            matmul(mat, m, m); matvec(mat, t, t);
        } else if (ch.tag == "translate") {
            const float zero[3] = {0, 0, 0}; float d[3];
            read_vec3(ch, zero, d);
            for (int k = 0; k < 3; ++k) t[k] = t[k] + d[k];
        } else if (ch.tag == "rotate") {
            if (!ch.attr("angle")) throw Error("no angle given for rotate");
            const float rad = need_f32(ch.attr("angle"), "could not parse rotate angle") * (float)(M_PI / 180.0);   // cgmath Deg -> Rad
            const float s = sinf(rad), c = cosf(rad);
            float rot[9];
            if (ch.attr("x")) { const float r[9] = {1, 0, 0, 0, c, s, 0, -s, c}; std::memcpy(rot, r, sizeof r); }
            else if (ch.attr("y")) { const float r[9] = {c, 0, -s, 0, 1, 0, s, 0, c}; std::memcpy(rot, r, sizeof r); }
            else { const float r[9] = {c, s, 0, -s, c, 0, 0, 0, 1}; std::memcpy(rot, r, sizeof r); }
            matmul(rot, m, m); matvec(rot, t, t);
        }
    }
}

hpt_texture blank_texture(float r, float g, float b) {   // load.rs:30-34
    hpt_texture t; std::memset(&t, 0, sizeof t);
    t.kind = HPT_TEX_BLANK; t.image = -1;
    t.color[0] = r; t.color[1] = g; t.color[2] = b;
    t.color2[0] = t.color2[1] = t.color2[2] = 1.0f;
    ident(t.m);
    return t;
}

// load.rs:330-365
hpt_texture load_texture(const Element& e, const float def[3], SceneDesc& sc, const std::string& base_dir) {
    hpt_texture tx = blank_texture(def[0], def[1], def[2]);
    float col[3]; const bool have_col = read_color(e, col);
    if (e.attr("value") || have_col) {
        float v = 1.0f; parse_f32(e.attr("value"), v);
        for (int k = 0; k < 3; ++k) tx.color[k] = v * (have_col ? col[k] : 1.0f);
    }
    if (const std::string* name = e.attr("texture")) {
        if (*name == "checkerboard") {
            tx.kind = HPT_TEX_CHECKER;
            if (const Element* c1 = e.child("color1")) read_color(*c1, tx.color1);
            if (const Element* c2 = e.child("color2")) read_color(*c2, tx.color2);
        } else {
            tx.kind = HPT_TEX_IMAGE;
            sc.images.push_back(read_png_rgb8(base_dir + "/" + *name));   // load.rs:437-443 load_img
            tx.image = (int32_t)sc.images.size() - 1;
        }
    }
    load_transform(e, tx.m, tx.t);
    return tx;
}

// load.rs:118-202 (type must be blinn; <specular> is mandatory although the integrator never reads it)
Material load_material(const Element& e, SceneDesc& sc, const std::string& base_dir) {
    if (!e.attr("type")) throw Error("no type for material");
    if (!e.attr("name")) throw Error("no name for material");
    if (*e.attr("type") != "blinn") throw Error("unknown material type");
    Material out; out.name = *e.attr("name");
    hpt_material& m = out.m; std::memset(&m, 0, sizeof m);
    const float one[3] = {1, 1, 1};
    m.diffuse = blank_texture(0.5f, 0.5f, 0.5f);
    m.reflection = blank_texture(0, 0, 0); m.refraction = blank_texture(0, 0, 0);
    m.refraction_index = 1.0f;
    if (const Element* d = e.child("diffuse")) m.diffuse = load_texture(*d, one, sc, base_dir);
    if (!e.child("specular")) throw Error("no specular found for <material>");
    if (const Element* em = e.child("emission")) {
        float col[3]; const bool have_col = read_color(*em, col);
        if (em->attr("value") || have_col) {
            float v = 1.0f; parse_f32(em->attr("value"), v);
            for (int k = 0; k < 3; ++k) m.emission[k] = v * (have_col ? col[k] : 1.0f);
        }
    }
    if (const Element* r = e.child("reflection")) {
        m.reflection = load_texture(*r, one, sc, base_dir);
        float g; m.reflection_glossiness = parse_f32(r->attr("glossiness"), g) ? g : 0.0f;
    }
    if (const Element* r = e.child("refraction")) {
        m.refraction = load_texture(*r, one, sc, base_dir);
        float v; m.refraction_index = parse_f32(r->attr("index"), v) ? v : 1.0f;
        m.refraction_glossiness = parse_f32(r->attr("glossiness"), v) ? v : 0.0f;
    }
    return out;
}

// load.rs:241-268
hpt_camera load_camera(const Element& e) {
    auto need = [&](const char* tag) -> const Element& {
        const Element* c = e.child(tag);
        if (!c) throw Error(std::string("no <") + tag + "> tag found in <camera>");
        return *c;
    };
    auto need_value = [&](const char* tag) -> const std::string& {
        const std::string* v = need(tag).attr("value");
        if (!v) throw Error(std::string("no value attribute found on <") + tag + "> tag");
        return *v;
    };
    hpt_camera c; std::memset(&c, 0, sizeof c);
    const float zero[3] = {0, 0, 0};
    read_vec3(need("position"), zero, c.pos);
    const float def_target[3] = {c.pos[0] + 0.0f, c.pos[1] + 0.0f, c.pos[2] + -1.0f};   // scene.rs:107-120 default dir
    float target[3];
    read_vec3(need("target"), def_target, target);
    float d[3] = {target[0] - c.pos[0], target[1] - c.pos[1], target[2] - c.pos[2]};
    const float inv = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);   // cgmath normalize: v * (1 / magnitude)
    for (int k = 0; k < 3; ++k) c.dir[k] = d[k] * inv;
    const float def_up[3] = {0, 1, 0}; float up[3];
    read_vec3(need("up"), def_up, up);
    c.fov = need_f32(&need_value("fov"), "could not parse camera fov");
    char* end = nullptr;
    const std::string ws = trim(need_value("width")), hs = trim(need_value("height"));
    c.img_width = (uint32_t)std::strtoul(ws.c_str(), &end, 10);
    if (ws.empty() || *end) throw Error("could not parse camera width");
    c.img_height = (uint32_t)std::strtoul(hs.c_str(), &end, 10);
    if (hs.empty() || *end) throw Error("could not parse camera height");
    c.focaldist = 1.0f; c.dof = 0.0f;
    float v;
    if (const Element* f = e.child("focaldist")) if (parse_f32(f->attr("value"), v)) c.focaldist = v;
    if (const Element* f = e.child("dof")) if (parse_f32(f->attr("value"), v)) c.dof = v;
    float tmp[3];
    cross(c.dir, up, tmp); cross(tmp, c.dir, c.up);   // orthogonal to dir, NOT renormalised (load.rs:265)
    return c;
}

}   // namespace

// load.rs:270-328 over a wavefront_obj-5.1-like parse (the crate is not under /root/reference; restated from its
// published behaviour, parity unpinned): first object only, f64 coordinates cast to f32, polygons fanned around
// their LAST vertex ((a,b,c,d) -> (d,a,b),(d,b,c); (a,b,c) -> (c,a,b)), vt/vn indices mandatory (load.rs:303-304).
Mesh load_obj(const std::string& path) {
    std::istringstream in(read_file(path));
    Mesh m;
    std::string line;
    int n_objects = 0;
    struct Idx { long v, t, n; };
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string kind;
        if (!(ls >> kind)) continue;
        if (kind == "o") { if (++n_objects > 1) break; }
        else if (kind == "v" || kind == "vn") {
            double x = 0, y = 0, z = 0; ls >> x >> y >> z;
            std::vector<float>& dst = kind == "v" ? m.verts : m.normals;
            dst.push_back((float)x); dst.push_back((float)y); dst.push_back((float)z);
        } else if (kind == "vt") {
            double u = 0, v = 0, w = 0; ls >> u; if (!(ls >> v)) v = 0; else if (!(ls >> w)) w = 0;
            m.uvw.push_back((float)u); m.uvw.push_back((float)v); m.uvw.push_back((float)w);
        } else if (kind == "f") {
            std::vector<Idx> idx;
            std::string tok;
            while (ls >> tok) {
                const size_t s1 = tok.find('/'), s2 = s1 == std::string::npos ? s1 : tok.find('/', s1 + 1);
                if (s1 == std::string::npos || s2 == std::string::npos || s2 == s1 + 1 || s2 + 1 >= tok.size())
                    throw Error("face without vt/vn index (load.rs:303-304 unwrap)");
                long vi = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10), ti = std::strtol(tok.substr(s1 + 1, s2 - s1 - 1).c_str(), nullptr, 10),
                     ni = std::strtol(tok.substr(s2 + 1).c_str(), nullptr, 10);
                vi = vi > 0 ? vi - 1 : (long)(m.verts.size() / 3) + vi;
                ti = ti > 0 ? ti - 1 : (long)(m.uvw.size() / 3) + ti;
                ni = ni > 0 ? ni - 1 : (long)(m.normals.size() / 3) + ni;
                if (vi < 0 || ti < 0 || ni < 0 || vi > 0x7fffffffL || ti > 0x7fffffffL || ni > 0x7fffffffL) throw Error("obj: face index out of range");
                idx.push_back({vi, ti, ni});
            }
            if (idx.size() < 3) continue;   // points / lines are not Primitive::Triangle (load.rs:301)
            const Idx last = idx.back();
            for (size_t k = 0; k + 2 < idx.size(); ++k) {
                const Idx a = idx[k], b = idx[k + 1];
                m.tris.push_back((uint32_t)last.v); m.tris.push_back((uint32_t)a.v); m.tris.push_back((uint32_t)b.v);
                m.ttris.push_back((uint32_t)last.t); m.ttris.push_back((uint32_t)a.t); m.ttris.push_back((uint32_t)b.t);
                m.ntris.push_back((uint32_t)last.n); m.ntris.push_back((uint32_t)a.n); m.ntris.push_back((uint32_t)b.n);
            }
        }
    }
    if (m.verts.empty()) throw Error("no objects found in file");   // load.rs:278
    // the reference indexes its Vecs with these (geometry.rs:117-130, :226): out of range is a panic there
    auto check = [](const std::vector<uint32_t>& idx, size_t n, const char* what) {
        for (uint32_t i : idx) if ((size_t)i >= n) throw Error(std::string("obj: ") + what + " index out of range");
    };
    check(m.tris, m.verts.size() / 3, "vertex"); check(m.ntris, m.normals.size() / 3, "normal"); check(m.ttris, m.uvw.size() / 3, "texture vertex");
    return m;
}

// Mesh::build_bvh (geometry.rs:223-239) + BVH::build / split (bvh.rs:35-146), both quirks kept: the partition key is
// the HALF EXTENT (p2 - p1) / 2 (bvh.rs:68,82,96 call it "center"), and the right child inherits the parent's box
// (bvh.rs:144) while the left child gets the tight union of its elements (bvh.rs:119-125).  Pre-order node array.
void build_reference_bvh(Mesh& mesh) {
    const size_t n = mesh.tris.size() / 3;
    if (n == 0) throw Error("Cannot construct empty BVH");   // bvh.rs:37
    std::vector<std::array<float, 3>> p1(n), p2(n), half(n);
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            const float a = mesh.verts[mesh.tris[3 * i] * 3 + k], b = mesh.verts[mesh.tris[3 * i + 1] * 3 + k], c = mesh.verts[mesh.tris[3 * i + 2] * 3 + k];
            p1[i][k] = std::fmin(std::fmin(a, b), c); p2[i][k] = std::fmax(std::fmax(a, b), c);
            half[i][k] = (p2[i][k] - p1[i][k]) / 2.0f;
        }
    }
    std::array<float, 3> r1, r2;   // load.rs:284-293
    for (int k = 0; k < 3; ++k) { r1[k] = r2[k] = mesh.verts[k]; }
    for (size_t v = 0; v < mesh.verts.size() / 3; ++v)
        for (int k = 0; k < 3; ++k) { r1[k] = std::fmin(r1[k], mesh.verts[3 * v + k]); r2[k] = std::fmax(r2[k], mesh.verts[3 * v + k]); }
    mesh.bvh.assign(2 * n - 1, hpt_bvh_node{});
    std::vector<int64_t> elems(n);
    for (size_t i = 0; i < n; ++i) elems[i] = (int64_t)i;
    struct Job { size_t lo, hi; std::array<float, 3> b1, b2; int64_t parent; bool is_right; };
    std::vector<Job> stack{{0, n, r1, r2, -1, false}};
    size_t count = 0;
    while (!stack.empty()) {
        const Job j = stack.back(); stack.pop_back();
        const size_t idx = count++;
        if (j.parent >= 0 && j.is_right) mesh.bvh[j.parent].right = (int32_t)idx;
        hpt_bvh_node& nd = mesh.bvh[idx];
        if (j.hi - j.lo == 1) {
            const int64_t e = elems[j.lo];
            nd.left = nd.right = -1; nd.prim = (int32_t)e;
            for (int k = 0; k < 3; ++k) { nd.p1[k] = p1[e][k]; nd.p2[k] = p2[e][k]; }
            continue;
        }
        const float sx = j.b2[0] - j.b1[0], sy = j.b2[1] - j.b1[1], sz = j.b2[2] - j.b1[2];
        const int ax = (sx > sy && sx > sz) ? 0 : (sy > sz ? 1 : 2);
        const float size = ax == 0 ? sx : ax == 1 ? sy : sz;
        const float pivot = j.b1[ax] + size / 2.0f;
        size_t split = 0;   // the swap loop of bvh.rs:66-76
        for (size_t i = j.lo; i < j.hi; ++i)
            if (half[elems[i]][ax] < pivot) { std::swap(elems[i], elems[j.lo + split]); ++split; }
        if (split == 0 || split == j.hi - j.lo) split = (j.hi - j.lo) / 2;
        nd.left = (int32_t)(idx + 1); nd.right = 0; nd.prim = -1;
        for (int k = 0; k < 3; ++k) { nd.p1[k] = j.b1[k]; nd.p2[k] = j.b2[k]; }
        std::array<float, 3> l1 = p1[elems[j.lo]], l2 = p2[elems[j.lo]];
        for (size_t i = j.lo; i < j.lo + split; ++i)
            for (int k = 0; k < 3; ++k) { l1[k] = std::fmin(l1[k], p1[elems[i]][k]); l2[k] = std::fmax(l2[k], p2[elems[i]][k]); }
        stack.push_back({j.lo + split, j.hi, j.b1, j.b2, (int64_t)idx, true});
        stack.push_back({j.lo, j.lo + split, l1, l2, (int64_t)idx, false});
    }
}

// load.rs:17-76.  OBJ / PNG paths resolve against the scene file's directory (the reference resolves them against
// the CWD, load.rs:89,271 -- run from that directory for identical behaviour).
SceneDesc load_scene(const std::string& filename, bool build_bvh) {
    const std::string base_dir = dir_of(filename);
    const Element root = parse_xml(read_file(filename));
    const Element* scene_xml = root.child("scene");
    if (!scene_xml) throw Error("no <scene> tag found");
    SceneDesc sc;
    const float one[3] = {1, 1, 1};
    sc.background = blank_texture(0, 0, 0); sc.environment = blank_texture(0, 0, 0);
    if (const Element* b = scene_xml->child("background")) sc.background = load_texture(*b, one, sc, base_dir);
    if (const Element* b = scene_xml->child("environment")) sc.environment = load_texture(*b, one, sc, base_dir);
    std::vector<std::pair<int32_t, std::string>> pending;
    // load.rs:78-116; nodes are numbered parents first, children in Vec order (the order hpt_scene_add_node wants)
    struct Rec { static void node(const Element& e, int32_t parent, SceneDesc& sc, std::vector<std::pair<int32_t, std::string>>& pending,
                                  const std::string& base_dir, bool build_bvh) {
        Node nd; nd.parent = parent; nd.geom = HPT_GEOM_NONE; nd.mesh = -1; nd.material = -1;
        const std::string* typ = e.attr("type");
        if (typ) {
            if (*typ == "sphere") nd.geom = HPT_GEOM_SPHERE;
            else if (*typ == "plane") nd.geom = HPT_GEOM_PLANE;
            else if (*typ == "obj") {
                if (!e.attr("name")) throw Error("no filename given for obj");
                sc.meshes.push_back(load_obj(base_dir + "/" + *e.attr("name")));
                if (build_bvh) build_reference_bvh(sc.meshes.back());   // load.rs:310
                nd.geom = HPT_GEOM_MESH; nd.mesh = (int32_t)sc.meshes.size() - 1;
            } else throw Error("unknown object type");
            if (!e.attr("material")) throw Error("no material given for object");
        }
        if (const std::string* nm = e.attr("name")) nd.name = *nm;
        load_transform(e, nd.m, nd.t);
        sc.nodes.push_back(nd);
        const int32_t me = (int32_t)sc.nodes.size() - 1;
        if (typ) pending.emplace_back(me, *e.attr("material"));
        for (const Element& ch : e.children) if (ch.tag == "object") node(ch, me, sc, pending, base_dir, build_bvh);
    } };
    std::map<std::string, int32_t> names;
    for (const Element& ch : scene_xml->children) {
        if (ch.tag == "object") Rec::node(ch, -1, sc, pending, base_dir, build_bvh);
        else if (ch.tag == "material") {
            Material m = load_material(ch, sc, base_dir);
            auto it = names.find(m.name);
            if (it != names.end()) sc.materials[it->second] = m;   // HashMap::insert replaces
            else { names[m.name] = (int32_t)sc.materials.size(); sc.materials.push_back(m); }
        }   // <light>: parsed by the reference (load.rs:204) but never read by its integrator
    }
    for (const auto& pr : pending) {
        auto it = names.find(pr.second);
        if (it == names.end()) throw Error("material does not exist for object");   // scene.rs:133
        sc.nodes[pr.first].material = it->second;
    }
    const Element* cam = root.child("camera");
    if (!cam) throw Error("no <camera> tag found");
    sc.camera = load_camera(*cam);
    return sc;
}

static void check(int rc, const char* what) {
    if (rc != HPT_OK) throw Error(std::string(what) + ": " + hpt_last_error());
}

void hand_over(const SceneDesc& d, hpt_scene* s, bool library_bvh) {
    for (const Mesh& m : d.meshes) {
        const bool give = !library_bvh && !m.bvh.empty();
        check(hpt_scene_add_mesh(s, m.verts.data(), (int32_t)(m.verts.size() / 3), m.tris.data(), m.normals.data(), (int32_t)(m.normals.size() / 3),
                                 m.ntris.data(), m.uvw.data(), (int32_t)(m.uvw.size() / 3), m.ttris.data(), (int32_t)(m.tris.size() / 3),
                                 give ? m.bvh.data() : nullptr, give ? (int32_t)m.bvh.size() : 0, nullptr), "hpt_scene_add_mesh");
    }
    for (const ImageRGB8& im : d.images) check(hpt_scene_add_image(s, im.px.data(), im.w, im.h, nullptr), "hpt_scene_add_image");
    for (const Material& m : d.materials) check(hpt_scene_add_material(s, &m.m, nullptr), "hpt_scene_add_material");
    for (const Node& n : d.nodes) check(hpt_scene_add_node(s, n.parent, n.geom, n.mesh, n.material, n.m, n.t, nullptr), "hpt_scene_add_node");
    check(hpt_scene_set_background(s, &d.background), "hpt_scene_set_background");
    check(hpt_scene_set_environment(s, &d.environment), "hpt_scene_set_environment");
    check(hpt_scene_set_camera(s, &d.camera), "hpt_scene_set_camera");
    check(hpt_scene_commit(s), "hpt_scene_commit");
}

uint64_t fnv1a(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

std::string dump_scene(const SceneDesc& d) {
    std::ostringstream o;
    auto bits = [&](const float* v, int n) { for (int i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, &v[i], 4); char b[12]; std::snprintf(b, sizeof b, " %08x", u); o << b; } };
    auto tex = [&](const char* name, const hpt_texture& t) {
        o << name << " " << t.kind << " " << t.image; bits(t.color, 3); bits(t.color1, 3); bits(t.color2, 3); bits(t.m, 9); bits(t.t, 3); o << "\n";
    };
    const hpt_camera& c = d.camera;
    o << "camera " << c.img_width << " " << c.img_height; bits(c.pos, 3); bits(c.dir, 3); bits(c.up, 3); bits(&c.fov, 1); bits(&c.focaldist, 1); bits(&c.dof, 1); o << "\n";
    tex("background", d.background); tex("environment", d.environment);
    for (size_t i = 0; i < d.materials.size(); ++i) {
        const hpt_material& m = d.materials[i].m;
        o << "material " << i << " " << d.materials[i].name; bits(m.emission, 3); bits(&m.reflection_glossiness, 1); bits(&m.refraction_glossiness, 1); bits(&m.refraction_index, 1); o << "\n";
        tex(" diffuse", m.diffuse); tex(" reflection", m.reflection); tex(" refraction", m.refraction);
    }
    for (size_t i = 0; i < d.nodes.size(); ++i) {
        const Node& n = d.nodes[i];
        o << "node " << i << " " << n.parent << " " << n.geom << " " << n.mesh << " " << n.material; bits(n.m, 9); bits(n.t, 3); o << "\n";
    }
    for (size_t i = 0; i < d.meshes.size(); ++i) {
        const Mesh& m = d.meshes[i];
        char b[512];
        std::snprintf(b, sizeof b, "mesh %zu nv %zu nn %zu nt %zu ntri %zu verts %016llx normals %016llx uvw %016llx tris %016llx ntris %016llx ttris %016llx bvh %zu %016llx\n",
                      i, m.verts.size() / 3, m.normals.size() / 3, m.uvw.size() / 3, m.tris.size() / 3,
                      (unsigned long long)fnv1a(m.verts.data(), m.verts.size() * 4), (unsigned long long)fnv1a(m.normals.data(), m.normals.size() * 4),
                      (unsigned long long)fnv1a(m.uvw.data(), m.uvw.size() * 4), (unsigned long long)fnv1a(m.tris.data(), m.tris.size() * 4),
                      (unsigned long long)fnv1a(m.ntris.data(), m.ntris.size() * 4), (unsigned long long)fnv1a(m.ttris.data(), m.ttris.size() * 4),
                      m.bvh.size(), (unsigned long long)fnv1a(m.bvh.data(), m.bvh.size() * sizeof(hpt_bvh_node)));
        o << b;
    }
    for (size_t i = 0; i < d.images.size(); ++i) {
        char b[128];
        std::snprintf(b, sizeof b, "image %zu %d %d %016llx\n", i, d.images[i].w, d.images[i].h, (unsigned long long)fnv1a(d.images[i].px.data(), d.images[i].px.size()));
        o << b;
    }
    return o.str();
}

}   // namespace hpt_host
