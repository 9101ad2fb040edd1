// hpt_host.h -- C++ host side above the C ABI (include/hpt.h): what stays on the host in the reference program.
//
// The reference is a compiled (Rust) program and no Rust toolchain exists in this image, so the part of it that the
// north_star keeps on the host is mirrored here in C++ (hpt/scene_io.py + hpt/host_bvh.py are the Python twins the
// tests and bench.py use; tests/test_host_cli.py holds the two to the same bits):
//   load_scene / load_node / load_material / load_texture / load_transform / load_camera / load_obj   load.rs:17-411
//   load_img (PNG -> RGB8)                                                                             load.rs:437-443
//   Mesh::build_bvh + BVH::build / split (the tree is still built on the host)      geometry.rs:223-239, bvh.rs:35-146
//   main (argument handling, constants, save_img)                                              main.rs:27-41,117-130
// and the hand-over of the loaded scene through hpt_scene_add_* / hpt_scene_commit.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/hpt.h"

namespace hpt_host {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };   // the reference panics (exit 101)

// ---- minimal XML (xmltree 0.4 as load.rs uses it: element tree + attributes; text and comments are ignored)
struct Element {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<Element> children;
    const std::string* attr(const char* name) const;
    const Element* child(const char* tag) const;   // first child with that tag (xmltree get_child)
};
Element parse_xml(const std::string& text);

// ---- PNG (png 0.9 in the reference): 8-bit grey / grey+alpha / RGB / RGBA, non-interlaced -> RGB8; RGBA8 writer
struct ImageRGB8 { int32_t w = 0, h = 0; std::vector<uint8_t> px; };
ImageRGB8 read_png_rgb8(const std::string& path);
void write_png_rgba8(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgba);   // main.rs:120-130 save_img

// ---- scene description = what load_scene returns (scene.rs:10-85), in the shapes the C ABI takes
struct Mesh {
    std::vector<float> verts, normals, uvw;          // n x 3
    std::vector<uint32_t> tris, ntris, ttris;        // ntri x 3
    std::vector<hpt_bvh_node> bvh;                   // BVH.nodes, pre-order (bvh.rs:35)
};
struct Node { int32_t parent, geom, mesh, material; float m[9], t[3]; std::string name; };
struct Material { std::string name; hpt_material m; };
struct SceneDesc {
    std::vector<Node> nodes;
    std::vector<Material> materials;
    std::vector<Mesh> meshes;
    std::vector<ImageRGB8> images;
    hpt_texture background, environment;
    hpt_camera camera;
};

SceneDesc load_scene(const std::string& filename, bool build_bvh = true);     // load.rs:17-76
Mesh load_obj(const std::string& path);                                       // load.rs:270-328
void build_reference_bvh(Mesh& mesh);                                         // geometry.rs:223-239 + bvh.rs:35-146
// hpt_scene_add_* in the reference's own order, then hpt_scene_commit; library_bvh = hand no tree (nodes = NULL)
void hand_over(const SceneDesc& desc, hpt_scene* scene, bool library_bvh = false);
// text dump with exact float bit patterns + FNV-1a digests of the big arrays (tests compare it with the Python loader)
std::string dump_scene(const SceneDesc& desc);
uint64_t fnv1a(const void* data, size_t n);

}   // namespace hpt_host
