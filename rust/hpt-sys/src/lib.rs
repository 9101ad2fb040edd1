//! Raw bindings to libhpt.so: one item per declaration of include/hpt.h, same order, `#[repr(C)]` structs with the
//! header's field order.  Written for the 2017-era toolchain the reference targets (no edition, `extern crate`).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub enum hpt_context {}
pub enum hpt_scene {}
pub enum hpt_group {}

pub const HPT_OK: c_int = 0;
pub const HPT_ERR_INVALID: c_int = -1;
pub const HPT_ERR_DEVICE: c_int = -2;
pub const HPT_ERR_NOMEM: c_int = -3;
pub const HPT_ERR_STATE: c_int = -4;
pub const HPT_ERR_LIMIT: c_int = -5;

pub const HPT_GEOM_NONE: i32 = 0;
pub const HPT_GEOM_SPHERE: i32 = 1;
pub const HPT_GEOM_PLANE: i32 = 2;
pub const HPT_GEOM_MESH: i32 = 3;
pub const HPT_TEX_BLANK: i32 = 0;
pub const HPT_TEX_CHECKER: i32 = 1;
pub const HPT_TEX_IMAGE: i32 = 2;
pub const HPT_RENDER_COUNTERS: u32 = 1;
pub const HPT_RENDER_TIME_KERNELS: u32 = 2;
pub const HPT_RENDER_EXACT_SHADE: u32 = 4;

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct hpt_bvh_node { pub left: i32, pub right: i32, pub prim: i32, pub p1: [f32; 3], pub p2: [f32; 3] }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct hpt_texture {
    pub kind: i32, pub image: i32,
    pub color: [f32; 3], pub color1: [f32; 3], pub color2: [f32; 3],
    pub m: [f32; 9], pub t: [f32; 3],
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct hpt_material {
    pub diffuse: hpt_texture, pub reflection: hpt_texture, pub refraction: hpt_texture,
    pub emission: [f32; 3],
    pub reflection_glossiness: f32, pub refraction_glossiness: f32, pub refraction_index: f32,
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct hpt_camera {
    pub pos: [f32; 3], pub dir: [f32; 3], pub up: [f32; 3],
    pub fov: f32, pub img_width: u32, pub img_height: u32, pub focaldist: f32, pub dof: f32,
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct hpt_render_params {
    pub adaptive: i32, pub spp: i32, pub first_sample: i32, pub initial_samples: i32, pub max_samples: i32,
    pub thresholds: [f32; 3], pub max_depth: i32, pub seed: u32, pub spp_per_pass: i32, pub flags: u32,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct hpt_render_stats {
    pub rays_primary: u64, pub rays_bounce: u64,
    pub nodes_visited: u64, pub tris_tested: u64, pub inst_tests: u64,
    pub seconds_total: f64,
    pub seconds_raygen: f64, pub seconds_extend: f64, pub seconds_shade: f64, pub seconds_resolve: f64,
    pub launches_extend: u32, pub passes: u32,
    pub samples: u64,
    pub seconds_analytic: f64, pub rays_mesh: u64,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct hpt_group_stats {
    pub n_devices: i32, pub transport: i32,
    pub rays_primary: u64, pub rays_bounce: u64, pub samples: u64,
    pub seconds_render_max: f64, pub seconds_reduce: f64, pub seconds_total: f64,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct hpt_accum { pub r: i64, pub g: i64, pub b: i64, pub nan: i64 }   // 64-bit fixed point, 2^-24 quantum: integer sums do not depend on any split

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct hpt_trace_stats { pub rays: u64, pub nodes_visited: u64, pub tris_tested: u64, pub seconds_kernel: f64, pub seconds_total: f64 }

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct hpt_scene_info_t {
    pub n_instances: i32, pub n_mesh_instances: i32, pub n_meshes: i32, pub n_materials: i32,
    pub n_nodes: i32, pub tlas_nodes: i32, pub uses_tlas: i32, pub stack_entries: i32,
    pub n_triangles: i64, pub device_bytes: u64,
    pub node_bytes: i32, pub trees_optimized: i32,
}

extern "C" {
    pub fn hpt_version() -> *const c_char;
    pub fn hpt_last_error() -> *const c_char;
    pub fn hpt_device_count() -> c_int;
    pub fn hpt_rng_f32(seed: u32, pixel: u32, sample: u32, dim: u32) -> f32;
    pub fn hpt_halton(index: i32, base: i32) -> f32;
    pub fn hpt_tonemap_rgba8(rgb: *const f32, n_pixels: u32, rgba: *mut u8) -> c_int;

    pub fn hpt_context_create(device: c_int, out: *mut *mut hpt_context) -> c_int;
    pub fn hpt_context_destroy(ctx: *mut hpt_context);
    pub fn hpt_context_set_stream(ctx: *mut hpt_context, hip_stream: *mut c_void) -> c_int;
    pub fn hpt_context_set_passes_in_flight(ctx: *mut hpt_context, n: c_int) -> c_int;
    pub fn hpt_context_synchronize(ctx: *mut hpt_context) -> c_int;

    pub fn hpt_scene_create(ctx: *mut hpt_context, out: *mut *mut hpt_scene) -> c_int;
    pub fn hpt_scene_destroy(scene: *mut hpt_scene);
    pub fn hpt_scene_add_mesh(scene: *mut hpt_scene, verts: *const f32, nv: i32, tri_idx: *const u32,
                              normals: *const f32, nn: i32, nrm_idx: *const u32,
                              uvw: *const f32, nt: i32, uv_idx: *const u32, ntri: i32,
                              nodes: *const hpt_bvh_node, nnodes: i32, mesh_id: *mut i32) -> c_int;
    pub fn hpt_scene_add_image(scene: *mut hpt_scene, rgb8: *const u8, width: i32, height: i32, image_id: *mut i32) -> c_int;
    pub fn hpt_scene_add_material(scene: *mut hpt_scene, material: *const hpt_material, material_id: *mut i32) -> c_int;
    pub fn hpt_scene_add_node(scene: *mut hpt_scene, parent: i32, geom: i32, mesh_id: i32, material_id: i32,
                              m: *const f32, t: *const f32, node_id: *mut i32) -> c_int;
    pub fn hpt_scene_set_background(scene: *mut hpt_scene, tex: *const hpt_texture) -> c_int;
    pub fn hpt_scene_set_environment(scene: *mut hpt_scene, tex: *const hpt_texture) -> c_int;
    pub fn hpt_scene_set_camera(scene: *mut hpt_scene, cam: *const hpt_camera) -> c_int;
    pub fn hpt_scene_commit(scene: *mut hpt_scene) -> c_int;
    pub fn hpt_scene_info(scene: *mut hpt_scene, out: *mut hpt_scene_info_t) -> c_int;

    pub fn hpt_trace_rays(scene: *mut hpt_scene, mesh_id: i32, n: i64, org3: *const f32, dir3: *const f32,
                          prim: *mut i32, t: *mut f32, u: *mut f32, v: *mut f32, side: *mut u8, stats: *mut hpt_trace_stats) -> c_int;
    pub fn hpt_trace_rays_device(scene: *mut hpt_scene, mesh_id: i32, n: i64, org4: *const c_void, dir4: *const c_void,
                                 hit4: *mut c_void, side_out: *mut c_void, count_visits: c_int, stats: *mut hpt_trace_stats) -> c_int;
    pub fn hpt_scene_intersect(scene: *mut hpt_scene, n: i64, org3: *const f32, dir3: *const f32,
                               node: *mut i32, prim: *mut i32, t: *mut f32, side: *mut u8,
                               pos3: *mut f32, normal3: *mut f32, uvw3: *mut f32) -> c_int;

    pub fn hpt_render(scene: *mut hpt_scene, params: *const hpt_render_params, rgb: *mut f32, samples_per_px: *mut u32,
                      stats: *mut hpt_render_stats) -> c_int;
    pub fn hpt_render_device(scene: *mut hpt_scene, params: *const hpt_render_params, rgb_sum_device: *mut f32,
                             stats: *mut hpt_render_stats) -> c_int;
    /// `hpt_render_device` for callers that cannot allocate: pass buffers from a static 98 MB device array, at most 2^19
    /// pixels and 4096 samples per call; calls that overlap on one GPU race.
    pub fn hpt_render_device_noalloc(scene: *mut hpt_scene, params: *const hpt_render_params, rgb_sum_device: *mut f32,
                                     stats: *mut hpt_render_stats) -> c_int;
    /// A sample range added to a caller-held accumulator on the device (ranks, adaptive rounds, resumed renders).
    pub fn hpt_render_accum_device(scene: *mut hpt_scene, params: *const hpt_render_params, pixel_list_device: *const u32,
                                   n_pixels: u32, accum_device: *mut hpt_accum, minmax_device: *mut f32,
                                   stats: *mut hpt_render_stats) -> c_int;
    /// The same for every pixel with the accumulator in host memory: a checkpoint is this array + the sample cursor.
    pub fn hpt_render_accum(scene: *mut hpt_scene, params: *const hpt_render_params, accum: *mut hpt_accum,
                            stats: *mut hpt_render_stats) -> c_int;
    pub fn hpt_accum_resolve_device(scene: *mut hpt_scene, accum_device: *const hpt_accum, samples_per_px_device: *const u32,
                                    num_samples: u32, rgb_device: *mut f32) -> c_int;
    pub fn hpt_accum_resolve(accum: *const hpt_accum, n_pixels: u32, samples_per_px: *const u32, num_samples: u32,
                             rgb: *mut f32) -> c_int;
    pub fn hpt_adaptive_select_device(scene: *mut hpt_scene, list_in_device: *const u32, n_in: u32, minmax_device: *const f32,
                                      thresholds: *const f32, list_out_device: *mut u32, samples_per_px_device: *mut u32,
                                      next_num_samples: u32, n_out: *mut u32) -> c_int;
    pub fn hpt_render_samples(scene: *mut hpt_scene, params: *const hpt_render_params, per_sample: *mut f32) -> c_int;

    /// One process, several GPUs: contexts + RCCL communicator (`n_devices == 0`: every GPU the process sees).
    pub fn hpt_group_create(devices: *const c_int, n_devices: c_int, out: *mut *mut hpt_group) -> c_int;
    pub fn hpt_group_destroy(group: *mut hpt_group);
    pub fn hpt_group_size(group: *const hpt_group) -> c_int;
    pub fn hpt_group_context(group: *mut hpt_group, rank: c_int) -> *mut hpt_context;
    /// `hpt_render` over the group: samples split over the GPUs, accumulators sum-reduced onto GPU 0 (RCCL).
    pub fn hpt_group_render(group: *mut hpt_group, scene: *mut hpt_scene, params: *const hpt_render_params, rgb: *mut f32,
                            samples_per_px: *mut u32, stats: *mut hpt_group_stats) -> c_int;
}
