//! Raw FFI surface of `include/rtmi.h`, one item per C declaration, same order.
//! Written without a Rust toolchain at hand: it has never been compiled. The same prototypes are
//! exercised from C++ (`raytracing_amd/host/`) and from Python ctypes (`raytracing_amd/_lib.py`).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RTMI_NBINS: usize = 60;
pub const RTMI_MISS: u32 = 0xFFFF_FFFF;
pub const RTMI_MAX_DEPTH: u32 = 32;

pub const RTMI_OK: c_int = 0;
pub const RTMI_BAD_ARG: c_int = -1;
pub const RTMI_NO_DEVICE: c_int = -2;
pub const RTMI_OOM: c_int = -3;
pub const RTMI_HIP_ERROR: c_int = -4;
pub const RTMI_UNSUPPORTED: c_int = -5;
pub const RTMI_RCCL_ERROR: c_int = -6;
pub const RTMI_INTERNAL: c_int = -7;

pub const RTMI_OBJ_SPHERE: u32 = 0;
pub const RTMI_OBJ_TRIANGLE: u32 = 1;
pub const RTMI_OBJ_MODEL: u32 = 2;
pub const RTMI_MAT_EMITTER: u32 = 0;
pub const RTMI_MAT_LAMBERTIAN: u32 = 1;
pub const RTMI_MAT_SPECULAR: u32 = 2;
pub const RTMI_MAT_GLOSSY: u32 = 3; // extension
pub const RTMI_MAT_DIELECTRIC: u32 = 4; // extension
pub const RTMI_HAS_AMBIENT: u32 = 1;
pub const RTMI_HAS_EMITTED: u32 = 2;
pub const RTMI_HAS_DIFFUSED: u32 = 4;
pub const RTMI_HAS_SPECULAR: u32 = 8;
pub const RTMI_HIT_FIRST_INDEX: u32 = 0;
pub const RTMI_HIT_NEAREST: u32 = 1;
pub const RTMI_INT_RANDOM_WALK: u32 = 0;
pub const RTMI_INT_ALBEDO: u32 = 1;
pub const RTMI_INT_NORMAL: u32 = 2;
pub const RTMI_SPEC_REFLECTANCE: c_int = 0;
pub const RTMI_SPEC_ILLUMINANT: c_int = 1;
pub const RTMI_FLAG_TIME_KERNELS: u32 = 1;
pub const RTMI_FLAG_COUNT_TRAVERSAL: u32 = 2;
/// This call's share of the render is empty (a rank beyond the samples of a sample partition): nothing is rendered.
pub const RTMI_FLAG_EMPTY_SHARE: u32 = 4;
pub const RTMI_SCENE_GPU_BVH: u32 = 1;
pub const RTMI_PARTITION_SAMPLES: u32 = 0;
pub const RTMI_PARTITION_TILES: u32 = 1;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_ray { pub o: [f64; 3], pub d: [f64; 3] }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_hit { pub obj: u32, pub prim: u32, pub t: f64, pub p: [f64; 3], pub n: [f64; 3] }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_object { pub kind: u32, pub first: u32, pub count: u32, pub material: u32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rtmi_sphere { pub l2w: [f64; 12], pub radius: f64 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rtmi_triangle { pub v: [f64; 9] }

#[repr(C)] #[derive(Clone, Copy)]
pub struct rtmi_material {
    pub type_: u32,
    pub present: u32,
    pub ambient: [f32; RTMI_NBINS],
    pub emitted: [f32; RTMI_NBINS],
    pub diffused: [f32; RTMI_NBINS],
    pub specular: [f32; RTMI_NBINS],
    pub specular_coefficient: f32,
    pub optical_density: f32,
}

#[repr(C)]
pub struct rtmi_scene_desc {
    pub objects: *const rtmi_object, pub n_objects: u32,
    pub spheres: *const rtmi_sphere, pub n_spheres: u32,
    pub triangles: *const rtmi_triangle, pub n_triangles: u32,
    pub materials: *const rtmi_material, pub n_materials: u32,
    pub hit_rule: u32,
    pub flags: u32,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_camera {
    pub focus: [f64; 3], pub p1: [f64; 3], pub p2: [f64; 3], pub p3: [f64; 3], pub p4: [f64; 3],
    pub width: u32, pub height: u32,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_render_params {
    pub integrator: u32, pub max_depth: u32, pub spp: u32, pub flags: u32,
    pub seed: u64,
    pub sample_begin: u32, pub sample_end: u32,
    pub tile_rank: u32, pub tile_world: u32,
    pub tile_size: u32, pub reserved: u32,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_stats {
    pub segments: u64, pub paths: u64, pub nodes_visited: u64, pub prims_tested: u64,
    pub ms_total: f64, pub ms_kernels: f64,
    pub launches: u32, pub batches: u32,
    pub mesh_rays: u64,
    pub ms_traverse: f64,
    pub traverse_launches: u32, pub reserved: u32,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rtmi_scene_info {
    pub n_nodes: u32, pub max_depth: u32, pub n_mesh_triangles: u32, pub n_spectra: u32,
    pub node_bytes: u64, pub triangle_bytes: u64,
    pub bvh_build_ms: f64, pub bvh_device_ms: f64, pub upload_ms: f64,
}

#[repr(C)] #[derive(Clone, Copy)]
pub struct rtmi_multi_info_t { pub n_ranks: u32, pub uses_rccl: u32, pub rccl_library: [c_char; 240] }

pub enum rtmi_ctx {}
pub enum rtmi_scene {}
pub enum rtmi_multi {}

extern "C" {
    pub fn rtmi_ctx_create(device_id: c_int, max_batch_paths: u64, out: *mut *mut rtmi_ctx) -> c_int;
    pub fn rtmi_ctx_destroy(ctx: *mut rtmi_ctx);
    pub fn rtmi_last_error() -> *const c_char;
    pub fn rtmi_version() -> *const c_char;

    pub fn rtmi_spectrum_from_rgb(r: f32, g: f32, b: f32, type_: c_int, out: *mut f32) -> c_int;
    pub fn rtmi_spectrum_to_xyz(spectrum: *const f32, xyz: *mut f32) -> c_int;
    pub fn rtmi_xyz_to_rgb(xyz: *const f32, rgb: *mut f32) -> c_int;
    pub fn rtmi_spectrum_cie_tables(x: *mut f32, y: *mut f32, z: *mut f32) -> c_int;

    pub fn rtmi_camera_new(focus: *const f64, focal_distance: f64, field_of_view_deg: f64, rotation_rpy: *const f64,
                           width: u32, height: u32, out: *mut rtmi_camera) -> c_int;
    pub fn rtmi_camera_get_ray(cam: *const rtmi_camera, i: f64, j: f64, out: *mut rtmi_ray) -> c_int;

    pub fn rtmi_scene_create(ctx: *mut rtmi_ctx, desc: *const rtmi_scene_desc, out: *mut *mut rtmi_scene) -> c_int;
    pub fn rtmi_scene_destroy(scene: *mut rtmi_scene);
    pub fn rtmi_scene_get_info(scene: *const rtmi_scene, out: *mut rtmi_scene_info) -> c_int;

    pub fn rtmi_intersect_batch(scene: *mut rtmi_scene, host_rays: *const rtmi_ray, n: usize, host_hits: *mut rtmi_hit,
                                stats: *mut rtmi_stats) -> c_int;
    pub fn rtmi_intersect_batch_device(scene: *mut rtmi_scene, dev_rays: *const rtmi_ray, n: usize, dev_hits: *mut rtmi_hit,
                                       stream: *mut c_void) -> c_int;
    pub fn rtmi_trace_batch(scene: *mut rtmi_scene, host_rays: *const rtmi_ray, n: usize, integrator: u32, max_depth: u32,
                            seed: u64, pixel_base: u32, sample_index: u32, host_spectra: *mut f32,
                            stats: *mut rtmi_stats) -> c_int;

    pub fn rtmi_render(scene: *mut rtmi_scene, cam: *const rtmi_camera, params: *const rtmi_render_params,
                       host_rgb8: *mut u8, host_xyz: *mut f32, stats: *mut rtmi_stats) -> c_int;
    pub fn rtmi_render_accumulate(scene: *mut rtmi_scene, cam: *const rtmi_camera, params: *const rtmi_render_params,
                                  dev_xyz_sum: *mut f32, stream: *mut c_void, stats: *mut rtmi_stats) -> c_int;
    pub fn rtmi_render_collect_stats(scene: *mut rtmi_scene, stream: *mut c_void, stats: *mut rtmi_stats) -> c_int;
    pub fn rtmi_tonemap(ctx: *mut rtmi_ctx, dev_xyz_sum: *const f32, spp: u32, width: u32, height: u32,
                        dev_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rtmi_render_accumulate_noalloc(scene: *mut rtmi_scene, cam: *const rtmi_camera, params: *const rtmi_render_params,
                                          dev_xyz_sum: *mut f32, stream: *mut c_void) -> c_int;

    pub fn rtmi_multi_create(device_ids: *const c_int, n: c_int, max_batch_paths: u64, desc: *const rtmi_scene_desc,
                             out: *mut *mut rtmi_multi) -> c_int;
    pub fn rtmi_multi_destroy(multi: *mut rtmi_multi);
    pub fn rtmi_multi_info(multi: *const rtmi_multi, out: *mut rtmi_multi_info_t) -> c_int;
    pub fn rtmi_multi_render(multi: *mut rtmi_multi, cam: *const rtmi_camera, params: *const rtmi_render_params, partition: u32,
                             host_rgb8: *mut u8, host_xyz: *mut f32, stats: *mut rtmi_stats, per_rank_stats: *mut rtmi_stats) -> c_int;
    pub fn rtmi_multi_render_pass(multi: *mut rtmi_multi, cam: *const rtmi_camera, params: *const rtmi_render_params, partition: u32,
                                  pass: u32, n_passes: u32, host_rgb8: *mut u8, host_xyz: *mut f32, stats: *mut rtmi_stats) -> c_int;
}
