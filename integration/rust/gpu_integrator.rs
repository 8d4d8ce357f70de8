//! A fourth `Integrator` variant for the reference's crate: `src/integrators/gpu.rs`.
//! Written without a Rust toolchain at hand (never compiled); it shows where each field of the
//! reference's types goes. Register it in `enum Integrator` (src/integrators/integrator.rs:145-151)
//! and build it in `main` where the three CPU integrators are built (src/main.rs:74-95).
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
                    triangles.push(tri(t.vertices()));
                }
                Shape::ModelObj(m) => {
                    // file order matters: the reference returns the first triangle in this order that the ray hits
                    objects.push(rtmi_object { kind: RTMI_OBJ_MODEL, first: triangles.len() as u32, count: m.triangles().len() as u32, material });
                    triangles.extend(m.triangles().iter().map(|t| tri(t.vertices())));
                }
                _ => anyhow::bail!("shape kind not carried by rtmi"),
            }
        }
        let desc = rtmi_scene_desc {
            objects: objects.as_ptr(), n_objects: objects.len() as u32,
            spheres: spheres.as_ptr(), n_spheres: spheres.len() as u32,
            triangles: triangles.as_ptr(), n_triangles: triangles.len() as u32,
            materials: materials.as_ptr(), n_materials: materials.len() as u32,
            hit_rule: RTMI_HIT_FIRST_INDEX, reserved: 0,
        };
        let (w, h) = self.camera.resolution();
        let screen = self.camera.screen();   // (p1 top-left, p2 top-right, p3 bottom-left, p4 bottom-right), src/camera.rs:4-9
        let cam = rtmi_camera { focus: xyz(self.camera.focus()), p1: xyz(&screen.0), p2: xyz(&screen.1), p3: xyz(&screen.2),
                                p4: xyz(&screen.3), width: w as u32, height: h as u32 };
        let params = rtmi_render_params { integrator: self.kind, max_depth: self.depth as u32, spp: self.iterations as u32,
                                          seed: 0, ..Default::default() };
        let mut rgb8 = vec![0u8; w * h * 3];
        unsafe {
            let (mut ctx, mut scene) = (ptr::null_mut(), ptr::null_mut());
            check(rtmi_ctx_create(-1, 0, &mut ctx))?;
            let result = check(rtmi_scene_create(ctx, &desc, &mut scene))
                .and_then(|_| check(rtmi_render(scene, &cam, &params, rgb8.as_mut_ptr(), ptr::null_mut(), ptr::null_mut())));
            rtmi_scene_destroy(scene);
            rtmi_ctx_destroy(ctx);
            result?;
        }
        // row-major, y = 0 at the top: the layout SamplerIntegrator::render fills (src/integrators/integrator.rs:110-112)
        ImageBuffer::<Rgb<u8>, _>::from_raw(w as u32, h as u32, rgb8).ok_or_else(|| anyhow::anyhow!("image size"))
    }
}

fn xyz(p: &crate::geometry::Point) -> [f64; 3] { [p.x, p.y, p.z] }
fn tri(v: &[crate::geometry::Point; 3]) -> rtmi_triangle {
    rtmi_triangle { v: [v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z] }
}
