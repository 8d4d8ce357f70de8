//! A fourth `Integrator` variant for the reference's crate: `src/integrators/gpu.rs`.
//! Written without a Rust toolchain at hand (never compiled); it shows where each field of the
//! reference's types goes. Register it in `enum Integrator` (src/integrators/integrator.rs:145-151)
//! and build it in `main` where the three CPU integrators are built (src/main.rs:74-95).
//!
//! What it reads from the reference's types, and where that comes from:
//!   exists today          Triangle::vértice(i) (src/shapes/triangle.rs:59), ShapeOperations::material()
//!                         (src/shapes/shape.rs:13), Camera::width()/height() (src/camera.rs:63-69), the public fields
//!                         of Material (src/material.rs:12-23), Rectangle's public tuple fields (src/geometry/mod.rs:12)
//!   added by the patch    Scene::shapes(), Camera::focus()/screen(), Sphere::local_to_world()/radius(),
//!   reference_accessors.patch   ModelObj::triangles(), CoefficientSpectrum::coefficients()
//! The render goes through rtmi_multi_*: every GPU of the node (RTMI_DEVICES=0,1,.. or all visible ones), the scene
//! replicated, tiles shared out as src/integrators/integrator.rs:39-50 numbers them, one RCCL reduce at the end.
use super::integrator::SamplerIntegrator;
use crate::{camera::Camera, geometry::Ray, material::{Material, Type}, scene::Scene, shapes::shape::Shape,
            spectrum::SampledSpectrum};
use image::{ImageBuffer, Rgb};
use rtmi_sys::*;
use std::{ffi::CStr, ptr};

pub struct GpuIntegrator { camera: Camera, scene: Scene, kind: u32, depth: usize, iterations: usize }

impl GpuIntegrator {
    pub fn random_walk(camera: &Camera, scene: &Scene, depth: usize, iterations: usize) -> Self {
        Self { camera: camera.clone(), scene: scene.clone(), kind: RTMI_INT_RANDOM_WALK, depth, iterations }
    }
    pub fn albedo(camera: &Camera, scene: &Scene, iterations: usize) -> Self {
        Self { camera: camera.clone(), scene: scene.clone(), kind: RTMI_INT_ALBEDO, depth: 0, iterations }
    }
    pub fn normal(camera: &Camera, scene: &Scene, iterations: usize) -> Self {
        Self { camera: camera.clone(), scene: scene.clone(), kind: RTMI_INT_NORMAL, depth: 0, iterations }
    }
}

fn check(rc: i32) -> Result<(), anyhow::Error> {
    if rc == RTMI_OK { return Ok(()); }
    let msg = unsafe { CStr::from_ptr(rtmi_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow::anyhow!("rtmi: {}", msg))
}

fn flatten_material(m: &Material) -> rtmi_material {
    let mut out = rtmi_material {
        type_: match m.tipo { Type::Emitter => RTMI_MAT_EMITTER, Type::Lambertian => RTMI_MAT_LAMBERTIAN, Type::Specular => RTMI_MAT_SPECULAR },
        present: 0,
        ambient: [0.0; RTMI_NBINS], emitted: [0.0; RTMI_NBINS], diffused: [0.0; RTMI_NBINS], specular: [0.0; RTMI_NBINS],
        specular_coefficient: m.specular_coefficient.unwrap_or(0.0) as f32,
        optical_density: m.optical_density.unwrap_or(1.0) as f32,
    };
    let mut put = |s: &Option<SampledSpectrum>, bit: u32, dst: &mut [f32; RTMI_NBINS]| {
        if let Some(s) = s { out.present |= bit; dst.copy_from_slice(s.coefficients()); }
    };
    put(&m.ambient_color, RTMI_HAS_AMBIENT, &mut out.ambient);
    put(&m.emitted_color, RTMI_HAS_EMITTED, &mut out.emitted);
    put(&m.diffused_color, RTMI_HAS_DIFFUSED, &mut out.diffused);
    put(&m.specular_color, RTMI_HAS_SPECULAR, &mut out.specular);
    out
}

impl SamplerIntegrator for GpuIntegrator {
    fn camera(&self) -> &Camera { &self.camera }
    fn scene(&self) -> &Scene { &self.scene }
    fn max_depth(&self) -> usize { self.depth }
    fn total_samples(&self) -> usize { self.iterations }
    fn incident_light(&self, _: &Ray, _: usize) -> SampledSpectrum { unreachable!("rendered on the GPU") }

    fn render(&self) -> Result<image::RgbImage, anyhow::Error> {
        // one rtmi_object per Shape, in scene order (Scene::objetos, src/scene.rs:7-9)
        let (mut objects, mut spheres, mut triangles, mut materials) = (vec![], vec![], vec![], vec![]);
        for shape in self.scene.shapes() {
            let material = materials.len() as u32;
            materials.push(flatten_material(shape.material()));
            match shape {
                Shape::Sphere(s) => {
                    // rows 0..3 of the 4x4 homogeneous local->world matrix, row-major
                    let m = s.local_to_world().matrix();
                    let mut l2w = [0.0; 12];
                    for r in 0..3 { for c in 0..4 { l2w[r * 4 + c] = m[(r, c)]; } }
                    objects.push(rtmi_object { kind: RTMI_OBJ_SPHERE, first: spheres.len() as u32, count: 1, material });
                    spheres.push(rtmi_sphere { l2w, radius: s.radius() });
                }
                Shape::Triangle(t) => {
                    objects.push(rtmi_object { kind: RTMI_OBJ_TRIANGLE, first: triangles.len() as u32, count: 1, material });
                    triangles.push(tri(t));
                }
                Shape::ModelObj(m) => {
                    // file order matters: the reference returns the first triangle in this order that the ray hits
                    objects.push(rtmi_object { kind: RTMI_OBJ_MODEL, first: triangles.len() as u32, count: m.triangles().len() as u32, material });
                    triangles.extend(m.triangles().iter().map(tri));
                }
                _ => anyhow::bail!("shape kind not carried by rtmi"),
            }
        }
        let desc = rtmi_scene_desc {
            objects: objects.as_ptr(), n_objects: objects.len() as u32,
            spheres: spheres.as_ptr(), n_spheres: spheres.len() as u32,
            triangles: triangles.as_ptr(), n_triangles: triangles.len() as u32,
            materials: materials.as_ptr(), n_materials: materials.len() as u32,
            hit_rule: RTMI_HIT_FIRST_INDEX, flags: 0,
        };
        let (w, h) = (self.camera.width() as usize, self.camera.height() as usize);
        let screen = self.camera.screen();   // (p1 top-left, p2 top-right, p3 bottom-left, p4 bottom-right), src/camera.rs:4-9
        let cam = rtmi_camera { focus: xyz(&self.camera.focus()), p1: xyz(&screen.0), p2: xyz(&screen.1), p3: xyz(&screen.2),
                                p4: xyz(&screen.3), width: w as u32, height: h as u32 };
        let params = rtmi_render_params { integrator: self.kind, max_depth: self.depth as u32, spp: self.iterations as u32,
                                          seed: 0, ..Default::default() };
        let mut rgb8 = vec![0u8; w * h * 3];
        // RTMI_DEVICES=0,1,2,3 picks the GPUs; default: device 0 only
        let devices: Vec<i32> = std::env::var("RTMI_DEVICES").ok()
            .map(|v| v.split(',').filter_map(|t| t.trim().parse().ok()).collect()).filter(|v: &Vec<i32>| !v.is_empty())
            .unwrap_or_else(|| vec![0]);
        unsafe {
            let mut multi = ptr::null_mut();
            check(rtmi_multi_create(devices.as_ptr(), devices.len() as i32, 0, &desc, &mut multi))?;
            // tiles: bit-identical to one GPU; RTMI_PARTITION_SAMPLES balances better when spp >= number of GPUs
            let result = check(rtmi_multi_render(multi, &cam, &params, RTMI_PARTITION_TILES, rgb8.as_mut_ptr(), ptr::null_mut(),
                                                 ptr::null_mut(), ptr::null_mut()));
            rtmi_multi_destroy(multi);
            result?;
        }
        // row-major, y = 0 at the top: the layout SamplerIntegrator::render fills (src/integrators/integrator.rs:110-112)
        ImageBuffer::<Rgb<u8>, _>::from_raw(w as u32, h as u32, rgb8).ok_or_else(|| anyhow::anyhow!("image size"))
    }
}

fn xyz(p: &crate::geometry::Point) -> [f64; 3] { [p.x, p.y, p.z] }
fn tri(t: &crate::shapes::Triangle) -> rtmi_triangle {
    // the loaders give every triangle the identity local->world map (src/scene_config.rs:189, src/shapes/model_obj.rs:54),
    // so the stored vertices are world-space
    let (a, b, c) = (t.vértice(0), t.vértice(1), t.vértice(2));
    rtmi_triangle { v: [a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z] }
}
