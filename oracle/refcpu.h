/* refcpu.h -- CPU oracle for the path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C restatement of the reference
 * renderer's algorithm (lukeovalle/raytracing, Rust, f64 geometry + 60-bin f32
 * spectra, linear scans, no acceleration structure). It exists to check the
 * HIP library and to provide the timed CPU baseline. The product library never
 * links, loads or calls anything in this directory.
 *
 * Parity pinning: every known-answer test the reference holds for this path
 * (camera, ray, transforms, AABB, Moller-Trumbore, triangle, sphere) is
 * replayed against this file by tests/test_oracle_kat.py. The parts of the
 * path the reference has no tests for (scene ordering, first-hit rule, shading,
 * spectra, tonemap, integrators) are restated from the reference's source text;
 * for those, parity with the compiled Rust binary is unpinned (no Rust
 * toolchain exists in this pipeline and the reference RNG is unseedable).
 */
#ifndef REFCPU_H
#define REFCPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REF_NBINS 60
#define REF_MISS 0xFFFFFFFFu

enum { REF_OBJ_SPHERE = 0, REF_OBJ_TRIANGLE = 1, REF_OBJ_MODEL = 2 };
/* GLOSSY and DIELECTRIC are extensions: the reference parses Ns and Ni (src/material.rs:20-22,49-50) but no
 * material type reads them. Their behaviour is defined here, by this restatement, and nothing in the reference
 * pins it ("parity unpinned"). */
enum { REF_MAT_EMITTER = 0, REF_MAT_LAMBERTIAN = 1, REF_MAT_SPECULAR = 2, REF_MAT_GLOSSY = 3, REF_MAT_DIELECTRIC = 4 };
enum { REF_HAS_AMBIENT = 1, REF_HAS_EMITTED = 2, REF_HAS_DIFFUSED = 4, REF_HAS_SPECULAR = 8 };
enum { REF_HIT_FIRST_INDEX = 0, REF_HIT_NEAREST = 1 };
enum { REF_INT_RANDOM_WALK = 0, REF_INT_ALBEDO = 1, REF_INT_NORMAL = 2 };
enum { REF_SPEC_REFLECTANCE = 0, REF_SPEC_ILLUMINANT = 1 };

/* Row-major 3x4 affine map: x' = m[0..2].x + m[3], ... */
typedef struct { double m[12]; } ref_affine;

typedef struct { double o[3]; double d[3]; } ref_ray; /* d normalised; max_t = +inf */

typedef struct {
  uint32_t obj;   /* index in the scene's object list */
  uint32_t prim;  /* triangle index inside a model (file order); 0 otherwise; REF_MISS = no hit */
  double t;       /* as the reference reports it (sphere: along the re-normalised local ray) */
  double p[3];
  double n[3];    /* geometric outward normal, not yet turned towards the ray */
} ref_hit;

typedef struct { uint32_t kind, first, count, material; } ref_object;
typedef struct { double l2w[12]; double radius; } ref_sphere;
typedef struct { double v[9]; } ref_triangle;
typedef struct {
  uint32_t type;
  uint32_t present; /* REF_HAS_* bits */
  float ambient[REF_NBINS], emitted[REF_NBINS], diffused[REF_NBINS], specular[REF_NBINS];
  float specular_coefficient; /* Ns: exponent of the glossy lobe */
  float optical_density;      /* Ni: refractive index of a dielectric */
} ref_material;

typedef struct {
  const ref_object* objects; uint32_t n_objects;
  const ref_sphere* spheres; uint32_t n_spheres;
  const ref_triangle* triangles; uint32_t n_triangles;
  const ref_material* materials; uint32_t n_materials;
  uint32_t hit_rule;
  uint32_t reserved;
} ref_scene_desc;

typedef struct { double focus[3], p1[3], p2[3], p3[3], p4[3]; uint32_t width, height; } ref_camera;

typedef struct {
  uint32_t integrator, max_depth, spp, tile_size;
  uint64_t seed;
  uint32_t sample_begin, sample_end; /* samples [begin,end) of every pixel; end==0 -> [0,spp) */
  uint32_t x0, y0, x1, y1;           /* pixel window; x1==0 -> whole image */
  uint32_t tile_rank, tile_world;    /* tile k (x-major) is rendered iff k % world == rank; world==0 -> all */
} ref_render_params;

typedef struct { uint64_t segments; uint64_t paths; double seconds; uint32_t threads; uint32_t pad; } ref_stats;

typedef struct ref_scene ref_scene;

/* extension BSDFs (no counterpart in the reference)
 * Phong lobe about the unit vector `axis`: cos(theta) = r1^(1/(ns+1)), phi = 2 pi r2, in the same basis
 * construction as the cosine sampler. */
void ref_phong_lobe_direction(const double axis[3], double ns, double r1, double r2, double out[3]);
/* Smooth dielectric: d = unit incident direction, nf = unit normal facing against d, entering = the ray comes from
 * outside, u in [0,1) picks reflection (u < Schlick reflectance, or total internal reflection) or refraction.
 * Returns 1 when the ray is reflected; out = the (unnormalised) new direction. */
int ref_dielectric_direction(const double d[3], const double nf[3], int entering, double ni, double u, double out[3]);

/* spectrum (src/spectrum/mod.rs) */
void ref_spectrum_init(void);
void ref_spectrum_tables(float cie_x[REF_NBINS], float cie_y[REF_NBINS], float cie_z[REF_NBINS]);
void ref_smits_table(int type, int which /*0..6 white,cyan,magenta,yellow,red,green,blue*/, float out[REF_NBINS]);
void ref_from_rgb(float r, float g, float b, int type, float out[REF_NBINS]);
void ref_to_xyz(const float spec[REF_NBINS], float xyz[3]);
void ref_xyz_to_rgb(const float xyz[3], float rgb[3]);
void ref_tonemap(const float rgb[3], uint8_t out[3]);
float ref_average_spectrum_sample(const float* lambdas, const float* samples, int n, float l0, float l1);

/* geometry (src/geometry/) */
void ref_affine_identity(ref_affine* out);
void ref_affine_translation(const double v[3], ref_affine* out);
void ref_affine_scaling(const double v[3], ref_affine* out);
void ref_affine_rotation(const double axis[3], double angle, ref_affine* out);
void ref_affine_mul(const ref_affine* a, const ref_affine* b, ref_affine* out);
int  ref_affine_inverse(const ref_affine* a, ref_affine* out);
void ref_affine_point(const ref_affine* a, const double p[3], double out[3]);
void ref_affine_vector(const ref_affine* a, const double v[3], double out[3]);
void ref_ray_new(const double o[3], const double d[3], ref_ray* out);
int  ref_ray_at(const ref_ray* r, double t, double out[3]);
void ref_affine_ray(const ref_affine* a, const ref_ray* r, ref_ray* out);
int  ref_intersect_ray_and_triangle(const double v[9], const ref_ray* r, double tuv[3]);
int  ref_aabb_intersect_ray(const double bmin[3], const double bmax[3], const ref_ray* r, double* t);
double ref_smaller_of_three(double a, double b, double c);   /* src/auxiliar.rs:2-4 */
double ref_bigger_of_three(double a, double b, double c);    /* src/auxiliar.rs:7-9 */
void ref_aabb_new(const double p1[3], const double p2[3], double bmin[3], double bmax[3]);                 /* aabb.rs:26-39 */
void ref_aabb_union(const double amin[3], const double amax[3], const double bmin[3], const double bmax[3],
                    double omin[3], double omax[3]);                                                      /* aabb.rs:49-61, 77-79 */
void ref_random_versor_cos_density(const double n[3], double r1, double r2, double out[3]);

/* shapes (src/shapes/) */
int ref_sphere_intersect(const ref_sphere* s, const ref_ray* r, ref_hit* out);
int ref_triangle_intersect(const ref_triangle* tri, const ref_affine* l2w, const ref_ray* r, ref_hit* out);
void ref_sphere_bounds(const ref_sphere* s, double bmin[3], double bmax[3]);

/* camera (src/camera.rs) */
void ref_camera_new(const double focus[3], double focal_distance, double fov_deg,
                    const double rotation[3], uint32_t width, uint32_t height, ref_camera* out);
void ref_camera_get_ray(const ref_camera* c, double i, double j, ref_ray* out);

/* rng: Philox4x32-10, counter (pixel, sample, bounce, stream), key = seed */
void ref_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void ref_uniform_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream, double out[2]);

/* scene + integrators (src/scene.rs, src/integrators/) */
ref_scene* ref_scene_create(const ref_scene_desc* desc);
void ref_scene_destroy(ref_scene* s);
/* NOT the reference's algorithm: switches the models' triangle scan to a median-split BVH that returns the same hits,
 * so that an accelerated CPU baseline can be timed beside the faithful linear scan. Parity tests never enable it
 * (except the one that checks it against the linear scan). */
void ref_scene_enable_bvh(ref_scene* s, int on);
void ref_scene_model_bounds(const ref_scene* s, uint32_t obj, double bmin[3], double bmax[3]);
int  ref_scene_intersect(const ref_scene* s, const ref_ray* r, ref_hit* out);
void ref_intersect_batch(const ref_scene* s, const ref_ray* rays, size_t n, ref_hit* out, int threads);
/* radiance along given rays; RNG counter pixel = pixel_base + i, sample = sample_index */
void ref_trace_batch(const ref_scene* s, const ref_ray* rays, size_t n, uint32_t integrator,
                     uint32_t max_depth, uint64_t seed, uint32_t pixel_base, uint32_t sample_index,
                     float* out_spectra /* n*60 */, uint64_t* segments, int threads);
/* one (pixel, sample): the reference's evaluate_pixel_sample with Philox jitter */
void ref_evaluate_pixel_sample(const ref_scene* s, const ref_camera* c, const ref_render_params* p,
                               uint32_t x, uint32_t y, uint32_t sample, float out[REF_NBINS], uint64_t* segments);
/* Full render. rgb8: W*H*3 row-major, y = 0 top. xyz_ref: mean XYZ computed as the
 * reference does (sum 60-bin spectra, divide by the number of samples rendered, then to_XYZ).
 * xyz_lin: mean XYZ computed by converting each sample to XYZ first and summing those in sample
 * order; xyz_sum: that sum before the division, ADDED to the buffer's contents (for partitioned
 * renders). Any output may be NULL; pixels of tiles that are not rendered are left untouched. */
void ref_render(const ref_scene* s, const ref_camera* c, const ref_render_params* p,
                uint8_t* rgb8, float* xyz_ref, float* xyz_lin, float* xyz_sum, ref_stats* stats, int threads);

#ifdef __cplusplus
}
#endif
#endif
