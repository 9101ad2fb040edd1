// Drop-in replacement for main.rs:55-115 of glowcoil/path-tracer (the image buffer + rayon pixel loop around
// scene.sample): hand the loaded Scene / Camera to libhpt.so and make ONE hpt_group_render call -- every GPU of the node
// (HPT_GPUS=n limits it to the first n), like the rayon loops used every core.  SOURCE ONLY -- no Rust
// toolchain in the build image; path-tracer_amd/host/main.cpp is the same sequence in C++, compiled and tested.
//
// Needs in the reference: `extern crate hpt_sys;` + `hpt-sys = { path = "../rust/hpt-sys" }` in Cargo.toml, and `pub`
// on the fields this reads (BVH.nodes, Mesh.*, Node.*, Texture.*, Material.*) -- they are private to their modules today.
use hpt_sys::*;
use std::collections::HashMap;
use std::ffi::CStr;
use std::ptr;

fn chk(rc: i32) {
    if rc != 0 {
        panic!("{}", unsafe { CStr::from_ptr(hpt_last_error()) }.to_string_lossy());   // keeps the reference's exit code 101
    }
}

fn cols(m: &Matrix3<f32>) -> [f32; 9] { [m.x.x, m.x.y, m.x.z, m.y.x, m.y.y, m.y.z, m.z.x, m.z.y, m.z.z] }   // cgmath is column-major
fn v3(v: &Vector3<f32>) -> [f32; 3] { [v.x, v.y, v.z] }

unsafe fn to_texture(hs: *mut hpt_scene, t: &Texture) -> hpt_texture {
    let mut out = hpt_texture { kind: HPT_TEX_BLANK, image: -1, color: v3(&t.color), color1: [0.0; 3], color2: [1.0; 3],
                                m: cols(&t.transform.transform), t: v3(&t.transform.translate) };
    match t.data {
        TextureData::Blank => {}
        TextureData::Checkerboard { color1, color2 } => { out.kind = HPT_TEX_CHECKER; out.color1 = v3(&color1); out.color2 = v3(&color2); }
        TextureData::Image { ref pixels, width, height } => {   // RGB8 (scene.rs:56, load.rs:437-443)
            out.kind = HPT_TEX_IMAGE;
            chk(hpt_scene_add_image(hs, pixels.as_ptr(), width as i32, height as i32, &mut out.image));
        }
    }
    out
}

unsafe fn add_mesh(hs: *mut hpt_scene, m: &Mesh) -> i32 {
    let flat3 = |v: &Vec<Vector3<f32>>| -> Vec<f32> { v.iter().flat_map(|p| vec![p.x, p.y, p.z]).collect() };
    let idx3 = |v: &Vec<(usize, usize, usize)>| -> Vec<u32> { v.iter().flat_map(|t| vec![t.0 as u32, t.1 as u32, t.2 as u32]).collect() };
    // bvh.rs:13-23: Node{left_child,right_child,bounding_box} / Leaf{index,bounding_box}, pre-order Vec
    let nodes: Vec<hpt_bvh_node> = m.bvh.nodes.iter().map(|n| match *n {
        BVHNode::Node { left_child, right_child, bounding_box } =>
            hpt_bvh_node { left: left_child as i32, right: right_child as i32, prim: -1, p1: v3(&bounding_box.p1), p2: v3(&bounding_box.p2) },
        BVHNode::Leaf { index, bounding_box } =>
            hpt_bvh_node { left: -1, right: -1, prim: index as i32, p1: v3(&bounding_box.p1), p2: v3(&bounding_box.p2) },
    }).collect();
    let (v, n, t) = (flat3(&m.vertices), flat3(&m.normals), flat3(&m.texture_vertices));
    let (ti, ni, tti) = (idx3(&m.triangles), idx3(&m.normal_triangles), idx3(&m.texture_triangles));
    let mut id = -1;
    chk(hpt_scene_add_mesh(hs, v.as_ptr(), m.vertices.len() as i32, ti.as_ptr(), n.as_ptr(), m.normals.len() as i32, ni.as_ptr(),
                           t.as_ptr(), m.texture_vertices.len() as i32, tti.as_ptr(), m.triangles.len() as i32,
                           nodes.as_ptr(), nodes.len() as i32, &mut id));
    id
}

// scene.rs:193,245-248: a node's own object first, then its children in Vec order; parents before children
unsafe fn add_node(hs: *mut hpt_scene, node: &Node, parent: i32, ids: &HashMap<String, i32>) {
    let (geom, mesh, material) = match node.object {
        None => (HPT_GEOM_NONE, -1, -1),
        Some(ref o) => {
            let mat = *ids.get(&o.material).expect("material does not exist for object");   // scene.rs:133
            match o.geometry {
                Geometry::Sphere => (HPT_GEOM_SPHERE, -1, mat),
                Geometry::Plane => (HPT_GEOM_PLANE, -1, mat),
                Geometry::Mesh(ref m) => (HPT_GEOM_MESH, add_mesh(hs, m), mat),
            }
        }
    };
    let (m, t) = (cols(&node.transform.transform), v3(&node.transform.translate));
    let mut id = -1;
    chk(hpt_scene_add_node(hs, parent, geom, mesh, material, m.as_ptr(), t.as_ptr(), &mut id));
    for c in &node.children { add_node(hs, c, id, ids); }
}

/// main.rs:55-115 -> img: Vec<u8> (RGBA8), ready for save_img (main.rs:117)
pub fn render_on_gpu(scene: &Scene, camera: &Camera) -> Vec<u8> {
    unsafe {
        // one context per GPU + the RCCL communicator; the scene is built once, on GPU 0's context, and replicated by the library
        let gpus: Vec<i32> = match std::env::var("HPT_GPUS").ok().and_then(|v| v.parse::<i32>().ok()) {
            Some(n) if n > 0 => (0..n).collect(),
            _ => Vec::new(),                                    // empty list = every GPU the process sees
        };
        let (mut group, mut hs) = (ptr::null_mut(), ptr::null_mut());
        chk(hpt_group_create(if gpus.is_empty() { ptr::null() } else { gpus.as_ptr() }, gpus.len() as i32, &mut group));
        let ctx = hpt_group_context(group, 0);
        chk(hpt_scene_create(ctx, &mut hs));
        let mut ids = HashMap::new();
        for (name, m) in &scene.materials {   // HashMap<String, Material> -> integer ids (scene.rs:132 lookup per hit disappears)
            let hm = hpt_material { diffuse: to_texture(hs, &m.diffuse), reflection: to_texture(hs, &m.reflection), refraction: to_texture(hs, &m.refraction),
                                    emission: v3(&m.emission), reflection_glossiness: m.reflection_glossiness,
                                    refraction_glossiness: m.refraction_glossiness, refraction_index: m.refraction_index };
            let mut id = -1;
            chk(hpt_scene_add_material(hs, &hm, &mut id));
            ids.insert(name.clone(), id);
        }
        for n in &scene.nodes { add_node(hs, n, -1, &ids); }
        chk(hpt_scene_set_background(hs, &to_texture(hs, &scene.background)));
        chk(hpt_scene_set_environment(hs, &to_texture(hs, &scene.environment)));
        chk(hpt_scene_set_camera(hs, &hpt_camera { pos: v3(&camera.pos), dir: v3(&camera.dir), up: v3(&camera.up), fov: camera.fov,
                                                   img_width: camera.img_width, img_height: camera.img_height,
                                                   focaldist: camera.focaldist, dof: camera.dof }));
        chk(hpt_scene_commit(hs));
        let params = hpt_render_params { adaptive: 1, spp: 0, first_sample: 0, initial_samples: INITIAL_SAMPLES, max_samples: MAX_SAMPLES,
                                         thresholds: [R_THRESHOLD, G_THRESHOLD, B_THRESHOLD],
                                         max_depth: 8, seed: 1, spp_per_pass: 0, flags: 0 };
        let npix = (camera.img_width * camera.img_height) as usize;
        let mut rgb = vec![0f32; npix * 3];
        chk(hpt_group_render(group, hs, &params, rgb.as_mut_ptr(), ptr::null_mut(), ptr::null_mut()));
        let mut img = vec![0u8; npix * 4];
        chk(hpt_tonemap_rgba8(rgb.as_ptr(), npix as u32, img.as_mut_ptr()));   // = color_as_u8_array (scene.rs:362)
        hpt_scene_destroy(hs);
        hpt_group_destroy(group);
        img
    }
}
// main(): `let img = render_on_gpu(&scene, &camera); save_img(&args[2], camera.img_width, camera.img_height, &img);`
