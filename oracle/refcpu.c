/* refcpu.c -- CPU oracle: plain-C restatement of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see refcpu.h). Every function cites the reference
 * file:line it follows (paths relative to the reference repository root).
 * Build with -ffp-contract=off: the reference (Rust) never fuses a*b+c, except
 * where it calls mul_add explicitly.
 *
 * Two deliberate departures from the reference, neither of which changes results:
 *  - the random numbers come from Philox4x32-10 keyed by (pixel, sample, bounce)
 *    instead of rand::random (thread-local, OS-seeded, unseedable);
 *  - inverses of the local->world maps are computed once per scene instead of once
 *    per intersection test, and hit records carry ids instead of shape copies.
 */
#include "refcpu.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ref_spectral_tables.inc"

/* ------------------------------------------------------------------ vectors */

typedef struct { double x, y, z; } v3;

static inline v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const double* p) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_to(v3 a, double* p) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, double s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, double s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline double v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
  return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double v3_norm(v3 a) { return sqrt(v3_dot(a, a)); }
/* nalgebra normalize(): self / self.norm() */
static inline v3 v3_normalize(v3 a) { return v3_div(a, v3_norm(a)); }

/* ------------------------------------------------- affine maps (geometry/common.rs:8-24) */

void ref_affine_identity(ref_affine* out) {
  memset(out, 0, sizeof *out);
  out->m[0] = out->m[5] = out->m[10] = 1.0;
}

/* src/geometry/common.rs:14 */
void ref_affine_translation(const double v[3], ref_affine* out) {
  ref_affine_identity(out);
  out->m[3] = v[0]; out->m[7] = v[1]; out->m[11] = v[2];
}

/* src/geometry/common.rs:22 */
void ref_affine_scaling(const double v[3], ref_affine* out) {
  ref_affine_identity(out);
  out->m[0] = v[0]; out->m[5] = v[1]; out->m[10] = v[2];
}

/* src/geometry/common.rs:18 -- Rotation3::from_axis_angle (Rodrigues) */
void ref_affine_rotation(const double axis[3], double angle, ref_affine* out) {
  v3 u = v3_normalize(v3_from(axis));
  double s = sin(angle), c = cos(angle), k = 1.0 - c;
  ref_affine_identity(out);
  out->m[0] = u.x * u.x * k + c;       out->m[1] = u.x * u.y * k - u.z * s; out->m[2] = u.x * u.z * k + u.y * s;
  out->m[4] = u.x * u.y * k + u.z * s; out->m[5] = u.y * u.y * k + c;       out->m[6] = u.y * u.z * k - u.x * s;
  out->m[8] = u.x * u.z * k - u.y * s; out->m[9] = u.y * u.z * k + u.x * s; out->m[10] = u.z * u.z * k + c;
}

/* Composition a*b (apply b first), as Affine3 * Affine3. */
void ref_affine_mul(const ref_affine* a, const ref_affine* b, ref_affine* out) {
  ref_affine r;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double s = (a->m[i * 4 + 0] * b->m[0 * 4 + j] + a->m[i * 4 + 1] * b->m[1 * 4 + j]) + a->m[i * 4 + 2] * b->m[2 * 4 + j];
      if (j == 3) s += a->m[i * 4 + 3];
      r.m[i * 4 + j] = s;
    }
  }
  *out = r;
}

/* Affine3::inverse() (used at src/shapes/sphere.rs:39,52 and src/shapes/triangle.rs:74):
 * adjugate of the linear part, then -A^-1 t. Returns 0 when singular. */
int ref_affine_inverse(const ref_affine* a, ref_affine* out) {
  const double* m = a->m;
  double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
  double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  double det = (a00 * c00 + a01 * c01) + a02 * c02;
  if (det == 0.0) return 0;
  double id = 1.0 / det;
  ref_affine r;
  r.m[0] = c00 * id; r.m[1] = (a02 * a21 - a01 * a22) * id; r.m[2] = (a01 * a12 - a02 * a11) * id;
  r.m[4] = c01 * id; r.m[5] = (a00 * a22 - a02 * a20) * id; r.m[6] = (a02 * a10 - a00 * a12) * id;
  r.m[8] = c02 * id; r.m[9] = (a01 * a20 - a00 * a21) * id; r.m[10] = (a00 * a11 - a01 * a10) * id;
  for (int i = 0; i < 3; ++i)
    r.m[i * 4 + 3] = -((r.m[i * 4 + 0] * m[3] + r.m[i * 4 + 1] * m[7]) + r.m[i * 4 + 2] * m[11]);
  *out = r;
  return 1;
}

static inline v3 affine_vec(const ref_affine* a, v3 v) {
  const double* m = a->m;
  return v3_make((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
                 (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
static inline v3 affine_pt(const ref_affine* a, v3 p) {
  v3 r = affine_vec(a, p);
  return v3_make(r.x + a->m[3], r.y + a->m[7], r.z + a->m[11]);
}
/* (A^-1)^T applied to a vector, given A^-1: src/shapes/sphere.rs:38-42 */
static inline v3 affine_vec_transposed(const ref_affine* a, v3 v) {
  const double* m = a->m;
  return v3_make((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
                 (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}

void ref_affine_point(const ref_affine* a, const double p[3], double out[3]) { v3_to(affine_pt(a, v3_from(p)), out); }
void ref_affine_vector(const ref_affine* a, const double v[3], double out[3]) { v3_to(affine_vec(a, v3_from(v)), out); }

/* ------------------------------------------------------------- rays (geometry/ray.rs) */

typedef struct { v3 o, d; } rayi;

static inline rayi rayi_from(const ref_ray* r) { rayi q = {v3_from(r->o), v3_from(r->d)}; return q; }

/* src/geometry/ray.rs:13-19 */
static inline rayi rayi_new(v3 o, v3 d) { rayi q = {o, v3_normalize(d)}; return q; }

void ref_ray_new(const double o[3], const double d[3], ref_ray* out) {
  rayi q = rayi_new(v3_from(o), v3_from(d));
  v3_to(q.o, out->o); v3_to(q.d, out->d);
}

/* src/geometry/ray.rs:30-36 (max_t is always +inf on this path) */
int ref_ray_at(const ref_ray* r, double t, double out[3]) {
  if (t < 0.0 || t > INFINITY) return 0;
  v3_to(v3_add(v3_from(r->o), v3_scale(v3_from(r->d), t)), out);
  return 1;
}

/* src/geometry/ray.rs:39-49: origin as a point, direction as a vector, re-normalised */
static inline rayi affine_ray(const ref_affine* a, rayi r) {
  rayi q = {affine_pt(a, r.o), v3_normalize(affine_vec(a, r.d))};
  return q;
}

void ref_affine_ray(const ref_affine* a, const ref_ray* r, ref_ray* out) {
  rayi q = affine_ray(a, rayi_from(r));
  v3_to(q.o, out->o); v3_to(q.d, out->d);
}

/* --------------------------------------- Moller-Trumbore (geometry/mod.rs:16-80) */

static inline int moller_trumbore(const double v[9], rayi r, double* t_out, double* u_out, double* v_out) {
  v3 v0 = v3_from(v), v1 = v3_from(v + 3), v2 = v3_from(v + 6);
  v3 e1 = v3_sub(v1, v0), e2 = v3_sub(v2, v0), tv = v3_sub(r.o, v0), d = r.d;
  v3 p = v3_cross(d, e2), q = v3_cross(tv, e1);
  double det = v3_dot(p, e1);
  if (fabs(det) < 1e-10) return 0;                 /* :58 */
  double inv = 1.0 / det;                          /* :61 det.recip() */
  double t = inv * v3_dot(q, e2);
  if (t < 0.0) return 0;                           /* :64 */
  double u = inv * v3_dot(p, tv);
  if (!(u >= 0.0 && u <= 1.0)) return 0;           /* :69 !(0.0..=1.0).contains(&u) */
  double w = inv * v3_dot(q, d);
  if (w < 0.0 || u + w > 1.0) return 0;            /* :75 */
  *t_out = t; *u_out = u; *v_out = w;
  return 1;
}

int ref_intersect_ray_and_triangle(const double v[9], const ref_ray* r, double tuv[3]) {
  return moller_trumbore(v, rayi_from(r), &tuv[0], &tuv[1], &tuv[2]);
}

/* ---------------------------------------------------- AABB slabs (geometry/aabb.rs:136-247) */

static inline int aabb_hit(v3 bmin, v3 bmax, rayi r, double* t_out) {
  double lo = 0.0, hi = INFINITY;
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  const double mn[3] = {bmin.x, bmin.y, bmin.z}, mx[3] = {bmax.x, bmax.y, bmax.z};
  for (int k = 0; k < 3; ++k) {
    if (d[k] == 0.0) {
      if (o[k] < mn[k] || o[k] > mx[k]) return 0;
    } else {
      double t0 = (mn[k] - o[k]) / d[k], t1 = (mx[k] - o[k]) / d[k];
      double tmin = d[k] > 0.0 ? t0 : t1, tmax = d[k] > 0.0 ? t1 : t0;
      if (tmin > lo) lo = tmin;
      if (tmax < hi) hi = tmax;
      if (lo > hi) return 0;
    }
  }
  *t_out = lo;
  return 1;
}

int ref_aabb_intersect_ray(const double bmin[3], const double bmax[3], const ref_ray* r, double* t) {
  return aabb_hit(v3_from(bmin), v3_from(bmax), rayi_from(r), t);
}

/* src/auxiliar.rs:1-9 (used by Triangle::get_box, shapes/triangle.rs:44-54) */
double ref_smaller_of_three(double a, double b, double c) { return fmin(a, fmin(b, c)); }
double ref_bigger_of_three(double a, double b, double c) { return fmax(a, fmax(b, c)); }

/* AABB::new / union (geometry/aabb.rs:26-61) */
static inline void aabb_new(v3 a, v3 b, v3* mn, v3* mx) {
  *mn = v3_make(fmin(a.x, b.x), fmin(a.y, b.y), fmin(a.z, b.z));
  *mx = v3_make(fmax(a.x, b.x), fmax(a.y, b.y), fmax(a.z, b.z));
}
static inline void aabb_union(v3* mn, v3* mx, v3 omn, v3 omx) {
  *mn = v3_make(fmin(mn->x, omn.x), fmin(mn->y, omn.y), fmin(mn->z, omn.z));
  *mx = v3_make(fmax(mx->x, omx.x), fmax(mx->y, omx.y), fmax(mx->z, omx.z));
}

/* AABB::new(p_1, p_2) followed by AABB::union(&other) -- the pair ModelObj::new folds over its triangles' boxes
 * (geometry/aabb.rs:26-61, shapes/model_obj.rs:62-66); resize_box (aabb.rs:77-79) is the same union in place. */
void ref_aabb_new(const double p1[3], const double p2[3], double bmin[3], double bmax[3]) {
  v3 mn, mx;
  aabb_new(v3_from(p1), v3_from(p2), &mn, &mx);
  v3_to(mn, bmin); v3_to(mx, bmax);
}
void ref_aabb_union(const double amin[3], const double amax[3], const double bmin[3], const double bmax[3], double omin[3], double omax[3]) {
  v3 mn = v3_from(amin), mx = v3_from(amax);
  aabb_union(&mn, &mx, v3_from(bmin), v3_from(bmax));
  v3_to(mn, omin); v3_to(mx, omax);
}

/* ------------------------------------- cosine hemisphere sampling (geometry/mod.rs:85-116) */

void ref_random_versor_cos_density(const double n[3], double r1, double r2, double out[3]) {
  v3 normal = v3_from(n);
  double sin_t = sqrt(r1);                         /* :87 */
  double cos_t = sqrt(1.0 - sin_t * sin_t);        /* :89 */
  double phi = 2.0 * M_PI * r2;                    /* :91 */
  v3 v = v3_make(cos(phi) * sin_t, sin(phi) * sin_t, cos_t);
  /* create_base_using_normal :102-116 */
  v3 b1 = fabs(normal.x) > 0.9 ? v3_make(0, 1, 0) : v3_make(1, 0, 0);
  b1 = v3_sub(b1, v3_scale(normal, v3_dot(b1, normal)));
  b1 = v3_scale(b1, 1.0 / v3_norm(b1));
  v3 b2 = v3_cross(normal, b1);
  /* Matrix3::from_columns([b1, b2, n]) * v, accumulated column by column */
  v3 r = v3_make((b1.x * v.x + b2.x * v.y) + normal.x * v.z, (b1.y * v.x + b2.y * v.y) + normal.y * v.z,
                 (b1.z * v.x + b2.z * v.y) + normal.z * v.z);
  v3_to(r, out);
}

/* Extension (no counterpart in the reference): Phong lobe about `axis`, same basis construction as above. */
void ref_phong_lobe_direction(const double axis[3], double ns, double r1, double r2, double out[3]) {
  v3 a = v3_from(axis);
  double cos_t = pow(r1, 1.0 / (ns + 1.0));
  double sin_t = sqrt(1.0 - cos_t * cos_t);
  double phi = 2.0 * M_PI * r2;
  v3 v = v3_make(cos(phi) * sin_t, sin(phi) * sin_t, cos_t);
  v3 b1 = fabs(a.x) > 0.9 ? v3_make(0, 1, 0) : v3_make(1, 0, 0);
  b1 = v3_sub(b1, v3_scale(a, v3_dot(b1, a)));
  b1 = v3_scale(b1, 1.0 / v3_norm(b1));
  v3 b2 = v3_cross(a, b1);
  v3 r = v3_make((b1.x * v.x + b2.x * v.y) + a.x * v.z, (b1.y * v.x + b2.y * v.y) + a.y * v.z,
                 (b1.z * v.x + b2.z * v.y) + a.z * v.z);
  v3_to(r, out);
}

/* Extension (no counterpart in the reference; the Schlick form is the one sketched at src/scene.rs:103-107). */
int ref_dielectric_direction(const double d[3], const double nf_[3], int entering, double ni, double u, double out[3]) {
  v3 i = v3_from(d), nf = v3_from(nf_);
  if (!(ni > 0.0)) ni = 1.0;
  double cosi = -v3_dot(i, nf);
  double eta = entering ? 1.0 / ni : ni;
  double sin2t = (eta * eta) * (1.0 - cosi * cosi);
  int reflect;
  double cost = 0.0;
  if (sin2t > 1.0) {
    reflect = 1;                                   /* total internal reflection */
  } else {
    cost = sqrt(1.0 - sin2t);
    double r0 = (1.0 - ni) / (1.0 + ni);
    r0 = r0 * r0;
    double c = entering ? cosi : cost;             /* the angle on the vacuum side */
    double x = 1.0 - c, x2 = x * x;
    double refl = r0 + (1.0 - r0) * ((x2 * x2) * x);
    reflect = u < refl;
  }
  v3 r = reflect ? v3_add(i, v3_scale(nf, 2.0 * cosi))
                 : v3_add(v3_scale(i, eta), v3_scale(nf, eta * cosi - cost));
  v3_to(r, out);
  return reflect;
}

/* ---------------------------------------------------------------- shapes */

typedef struct { uint32_t obj, prim; double t; v3 p, n; } hiti;

static inline void hiti_to(const hiti* h, ref_hit* out) {
  out->obj = h->obj; out->prim = h->prim; out->t = h->t;
  v3_to(h->p, out->p); v3_to(h->n, out->n);
}

/* src/shapes/sphere.rs:50-103 and :37-43, given the precomputed inverse map */
static inline int sphere_hit(const ref_affine* l2w, const ref_affine* w2l, double radius, rayi ray, hiti* out) {
  rayi lr = affine_ray(w2l, ray);                  /* :52 */
  v3 dir = lr.d, orig = lr.o;
  double h = v3_dot(dir, orig);                    /* :63 */
  double c = v3_dot(orig, orig) - radius * radius; /* :64 */
  double disc = h * h - c;                         /* :66 */
  if (disc < 0.0) return 0;
  double t1 = -h - sqrt(disc), t2 = -h + sqrt(disc);
  if (t2 < 0.0) return 0;                          /* :78 */
  double t = t1 < 0.0 ? t2 : t1;                   /* :82-88 */
  if (t < 0.0 || t > INFINITY) return 0;           /* local_ray.at(t) :90-93 */
  v3 pl = v3_add(lr.o, v3_scale(lr.d, t));
  out->p = affine_pt(l2w, pl);                     /* :98 */
  out->n = v3_normalize(affine_vec_transposed(w2l, pl)); /* :100, :38-42 */
  out->t = t;                                      /* :101 local-space t */
  out->prim = 0;
  return 1;
}

int ref_sphere_intersect(const ref_sphere* s, const ref_ray* r, ref_hit* out) {
  ref_affine l2w, w2l;
  memcpy(l2w.m, s->l2w, sizeof l2w.m);
  if (!ref_affine_inverse(&l2w, &w2l)) return 0;
  hiti h; h.obj = 0;
  if (!sphere_hit(&l2w, &w2l, s->radius, rayi_from(r), &h)) return 0;
  hiti_to(&h, out);
  return 1;
}

/* Sphere::new bounding box (src/shapes/sphere.rs:22-33) */
void ref_sphere_bounds(const ref_sphere* s, double bmin[3], double bmax[3]) {
  ref_affine l2w; memcpy(l2w.m, s->l2w, sizeof l2w.m);
  v3 mx = v3_make(s->radius, s->radius, s->radius), mn = v3_neg(mx), a, b;
  aabb_new(affine_pt(&l2w, mn), affine_pt(&l2w, mx), &a, &b);
  v3_to(a, bmin); v3_to(b, bmax);
}

/* src/shapes/triangle.rs:73-97 with the flat normal of :30; l2w and its inverse given */
static inline int triangle_hit(const double v[9], v3 flat_normal, const ref_affine* l2w, const ref_affine* w2l,
                               rayi ray, hiti* out) {
  rayi lr = affine_ray(w2l, ray);                  /* :74 */
  double t, u, w;
  if (!moller_trumbore(v, lr, &t, &u, &w)) return 0;
  if (t < 0.0 || t > INFINITY) return 0;           /* local_ray.at(t) :78-81 */
  out->p = affine_pt(l2w, v3_add(lr.o, v3_scale(lr.d, t)));
  v3 normal = affine_vec(l2w, flat_normal);        /* :83 */
  out->n = v3_normalize(normal);                   /* :91 */
  out->t = t / v3_norm(normal);                    /* :92 */
  out->prim = 0;
  return 1;
}

static inline v3 triangle_flat_normal(const double v[9]) { /* src/shapes/triangle.rs:30 */
  v3 p1 = v3_from(v), p2 = v3_from(v + 3), p3 = v3_from(v + 6);
  return v3_normalize(v3_cross(v3_sub(p2, p1), v3_sub(p3, p1)));
}

int ref_triangle_intersect(const ref_triangle* tri, const ref_affine* l2w, const ref_ray* r, ref_hit* out) {
  ref_affine w2l;
  if (!ref_affine_inverse(l2w, &w2l)) return 0;
  hiti h; h.obj = 0;
  if (!triangle_hit(tri->v, triangle_flat_normal(tri->v), l2w, &w2l, rayi_from(r), &h)) return 0;
  hiti_to(&h, out);
  return 1;
}

/* ---------------------------------------------------------------- camera (camera.rs) */

/* src/camera.rs:19-51. Rotation3::from_euler_angles(roll, pitch, yaw) = Rz(yaw) Ry(pitch) Rx(roll). */
void ref_camera_new(const double focus[3], double focal_distance, double fov_deg, const double rotation[3],
                    uint32_t width, uint32_t height, ref_camera* out) {
  double delta_y = focal_distance * tan(fov_deg / 2.0 * M_PI / 180.0);
  double delta_z = delta_y * (double)height / (double)width;
  double sr = sin(rotation[0]), cr = cos(rotation[0]);
  double sp = sin(rotation[1]), cp = cos(rotation[1]);
  double sy = sin(rotation[2]), cy = cos(rotation[2]);
  ref_affine m;
  ref_affine_identity(&m);
  m.m[0] = cy * cp; m.m[1] = cy * sp * sr - sy * cr; m.m[2] = cy * sp * cr + sy * sr;
  m.m[4] = sy * cp; m.m[5] = sy * sp * sr + cy * cr; m.m[6] = sy * sp * cr - cy * sr;
  m.m[8] = -sp;     m.m[9] = cp * sr;                m.m[10] = cp * cr;
  m.m[3] = focus[0]; m.m[7] = focus[1]; m.m[11] = focus[2];
  v3_to(affine_pt(&m, v3_make(focal_distance, -delta_y, delta_z)), out->p1);
  v3_to(affine_pt(&m, v3_make(focal_distance, delta_y, delta_z)), out->p2);
  v3_to(affine_pt(&m, v3_make(focal_distance, -delta_y, -delta_z)), out->p3);
  v3_to(affine_pt(&m, v3_make(focal_distance, delta_y, -delta_z)), out->p4);
  memcpy(out->focus, focus, 3 * sizeof(double));
  out->width = width; out->height = height;
}

/* src/camera.rs:71-85 */
static inline rayi camera_ray(const ref_camera* c, double i, double j) {
  v3 p1 = v3_from(c->p1), p2 = v3_from(c->p2), p3 = v3_from(c->p3), f = v3_from(c->focus);
  v3 right = v3_div(v3_scale(v3_sub(p2, p1), i), (double)c->width);
  v3 down = v3_div(v3_scale(v3_sub(p3, p1), j), (double)c->height);
  v3 point = v3_add(v3_add(p1, right), down);
  return rayi_new(f, v3_sub(point, f));
}

void ref_camera_get_ray(const ref_camera* c, double i, double j, ref_ray* out) {
  rayi q = camera_ray(c, i, j);
  v3_to(q.o, out->o); v3_to(q.d, out->d);
}

/* ---------------------------------------------------------------- spectrum (spectrum/mod.rs) */

static float g_cie[3][REF_NBINS];
static float g_smits[2][7][REF_NBINS];
static int g_spectrum_ready = 0;

/* src/auxiliar.rs:13 */
static inline float lerpf(float a, float b, float t) { return (1.0f - t) * a + t * b; }

/* src/spectrum/mod.rs:320-383 */
float ref_average_spectrum_sample(const float* lambdas, const float* samples, int n, float l0, float l1) {
  if (n == 0) return 0.0f;
  if (n == 1) return samples[0];
  float lf = lambdas[0], ll = lambdas[n - 1], sf = samples[0], sl = samples[n - 1];
  if (l1 <= lf) return sf;
  if (l0 >= ll) return sl;
  float sum = 0.0f;
  if (l0 < lf) sum += sf * (lf - l0);
  if (l1 > ll) sum += sl * (l1 - ll);
  int i = 0;
  while (l0 > lambdas[i + 1]) i += 1;
  while ((i + 1) < n && l1 >= lambdas[i]) {
    float a = fmaxf(lambdas[i], l0), b = fminf(lambdas[i + 1], l1);
    float span = lambdas[i + 1] - lambdas[i];
    float ia = lerpf(samples[i], samples[i + 1], (a - lambdas[i]) / span);
    float ib = lerpf(samples[i], samples[i + 1], (b - lambdas[i]) / span);
    sum += 0.5f * (ia + ib) * (b - a);
    i += 1;
  }
  return sum / (l1 - l0);
}

/* src/spectrum/mod.rs:386-522 */
void ref_spectrum_init(void) {
  if (g_spectrum_ready) return;
  const float* refl[7] = {kSmitsReflWhite, kSmitsReflCyan, kSmitsReflMagenta, kSmitsReflYellow,
                          kSmitsReflRed, kSmitsReflGreen, kSmitsReflBlue};
  const float* illum[7] = {kSmitsIllumWhite, kSmitsIllumCyan, kSmitsIllumMagenta, kSmitsIllumYellow,
                           kSmitsIllumRed, kSmitsIllumGreen, kSmitsIllumBlue};
  for (int i = 0; i < REF_NBINS; ++i) {
    float l0 = lerpf(400.0f, 700.0f, (float)i / (float)REF_NBINS);
    float l1 = lerpf(400.0f, 700.0f, (float)(i + 1) / (float)REF_NBINS);
    g_cie[0][i] = ref_average_spectrum_sample(kCieLambda, kCieX, RT_N_CIE_SAMPLES, l0, l1);
    g_cie[1][i] = ref_average_spectrum_sample(kCieLambda, kCieY, RT_N_CIE_SAMPLES, l0, l1);
    g_cie[2][i] = ref_average_spectrum_sample(kCieLambda, kCieZ, RT_N_CIE_SAMPLES, l0, l1);
    for (int k = 0; k < 7; ++k) {
      g_smits[0][k][i] = ref_average_spectrum_sample(kSmitsLambda, refl[k], RT_N_SMITS_SAMPLES, l0, l1);
      g_smits[1][k][i] = ref_average_spectrum_sample(kSmitsLambda, illum[k], RT_N_SMITS_SAMPLES, l0, l1);
    }
  }
  g_spectrum_ready = 1;
}

void ref_spectrum_tables(float cie_x[REF_NBINS], float cie_y[REF_NBINS], float cie_z[REF_NBINS]) {
  ref_spectrum_init();
  memcpy(cie_x, g_cie[0], sizeof g_cie[0]);
  memcpy(cie_y, g_cie[1], sizeof g_cie[1]);
  memcpy(cie_z, g_cie[2], sizeof g_cie[2]);
}

void ref_smits_table(int type, int which, float out[REF_NBINS]) {
  ref_spectrum_init();
  memcpy(out, g_smits[type ? 1 : 0][which], sizeof(float) * REF_NBINS);
}

static inline void spec_add_scaled(float* acc, const float* basis, float k) {
  for (int i = 0; i < REF_NBINS; ++i) acc[i] += basis[i] * k; /* result += basis * k */
}

/* src/spectrum/mod.rs:525-581 */
void ref_from_rgb(float r, float g, float b, int type, float out[REF_NBINS]) {
  ref_spectrum_init();
  float(*s)[REF_NBINS] = g_smits[type == REF_SPEC_ILLUMINANT ? 1 : 0];
  const float *white = s[0], *cyan = s[1], *magenta = s[2], *yellow = s[3], *red = s[4], *green = s[5], *blue = s[6];
  for (int i = 0; i < REF_NBINS; ++i) out[i] = 0.0f;
  if (r < g && r < b) {
    spec_add_scaled(out, white, r);
    if (g < b) { spec_add_scaled(out, cyan, g - r); spec_add_scaled(out, blue, b - g); }
    else       { spec_add_scaled(out, cyan, b - r); spec_add_scaled(out, green, g - b); }
  } else if (g < r && g < b) {
    spec_add_scaled(out, white, g);
    if (r < b) { spec_add_scaled(out, magenta, r - g); spec_add_scaled(out, blue, b - r); }
    else       { spec_add_scaled(out, magenta, b - g); spec_add_scaled(out, red, r - b); }
  } else {
    spec_add_scaled(out, white, b);
    if (r < g) { spec_add_scaled(out, yellow, r - b); spec_add_scaled(out, green, g - r); }
    else       { spec_add_scaled(out, yellow, g - b); spec_add_scaled(out, red, r - g); }
  }
}

/* src/spectrum/mod.rs:590-609 */
void ref_to_xyz(const float spec[REF_NBINS], float xyz[3]) {
  ref_spectrum_init();
  float x = 0.0f, y = 0.0f, z = 0.0f;
  for (int i = 0; i < REF_NBINS; ++i) {
    x += g_cie[0][i] * spec[i];
    y += g_cie[1][i] * spec[i];
    z += g_cie[2][i] * spec[i];
  }
  float scale = (700.0f - 400.0f) / (kCieYIntegral * (float)REF_NBINS);
  xyz[0] = x * scale; xyz[1] = y * scale; xyz[2] = z * scale;
}

/* src/spectrum/mod.rs:630-636 */
void ref_xyz_to_rgb(const float xyz[3], float rgb[3]) {
  float x = xyz[0], y = xyz[1], z = xyz[2];
  rgb[0] = 3.2404790f * x - 1.537150f * y - 0.498535f * z;
  rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
  rgb[2] = 0.0556480f * x - 0.204043f * y + 1.057311f * z;
}

/* Rust `as u8` from f32: saturating, NaN -> 0 */
static inline uint8_t f32_as_u8(float v) {
  if (!(v == v)) return 0;
  if (v <= 0.0f) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)v;
}

/* src/integrators/integrator.rs:80-90 */
void ref_tonemap(const float rgb[3], uint8_t out[3]) {
  for (int k = 0; k < 3; ++k) {
    float c = rgb[k];
    if (c < 0.0f) c = 0.0f;       /* f32::clamp keeps NaN */
    if (c > 1.0f) c = 1.0f;
    out[k] = f32_as_u8(256.0f * powf(c, 1.0f / 2.2f));
  }
}

/* ---------------------------------------------------------------- Philox4x32-10 */

static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
  uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void ref_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k[2] = {key[0], key[1]};
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  memcpy(out, c, sizeof c);
}

/* Two uniforms in [0,1) with 53 random bits each, as rand's f64 sampling does ((x >> 11) * 2^-53). */
void ref_uniform_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream, double out[2]) {
  uint32_t ctr[4] = {pixel, sample, bounce, stream}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
  ref_philox4x32_10(ctr, key, r);
  uint64_t a = ((uint64_t)r[1] << 32) | r[0], b = ((uint64_t)r[3] << 32) | r[2];
  out[0] = (double)(a >> 11) * 0x1.0p-53;
  out[1] = (double)(b >> 11) * 0x1.0p-53;
}

/* ---------------------------------------------------------------- scene (scene.rs, model_obj.rs) */

/* Optional median-split BVH over a model's triangles. It is NOT part of the restatement of the reference (which scans
 * every triangle, model_obj.rs:80-84) and is never used for parity: it exists so that a CPU baseline with the same
 * kind of acceleration the GPU path uses can be timed beside the faithful linear scan (ref_scene_enable_bvh). It
 * returns the linear scan's results: boxes are padded, distance never prunes under the first-index rule. */
typedef struct { double lo[3], hi[3]; uint32_t left, right, first, count, min_index; } cpu_bvh_node;
typedef struct { cpu_bvh_node* nodes; uint32_t n_nodes; uint32_t* order; } cpu_bvh;

typedef struct {
  ref_affine l2w, w2l;  /* spheres and top-level triangles */
  v3 bmin, bmax;        /* models: box grown from the origin point (model_obj.rs:62-66) */
  cpu_bvh bvh;          /* models, only after ref_scene_enable_bvh */
} obj_aux;

struct ref_scene {
  ref_object* objects; uint32_t n_objects;
  ref_sphere* spheres; uint32_t n_spheres;
  ref_triangle* triangles; uint32_t n_triangles;
  v3* tri_normals;
  ref_material* materials; uint32_t n_materials;
  obj_aux* aux;
  uint32_t hit_rule;
  ref_affine identity;
  int use_bvh;
};

static void* dup_mem(const void* p, size_t n) {
  void* q = malloc(n ? n : 1);
  if (n) memcpy(q, p, n);
  return q;
}

ref_scene* ref_scene_create(const ref_scene_desc* d) {
  ref_spectrum_init();
  ref_scene* s = (ref_scene*)calloc(1, sizeof *s);
  s->n_objects = d->n_objects; s->n_spheres = d->n_spheres; s->n_triangles = d->n_triangles; s->n_materials = d->n_materials;
  s->objects = (ref_object*)dup_mem(d->objects, sizeof(ref_object) * d->n_objects);
  s->spheres = (ref_sphere*)dup_mem(d->spheres, sizeof(ref_sphere) * d->n_spheres);
  s->triangles = (ref_triangle*)dup_mem(d->triangles, sizeof(ref_triangle) * d->n_triangles);
  s->materials = (ref_material*)dup_mem(d->materials, sizeof(ref_material) * d->n_materials);
  s->tri_normals = (v3*)malloc(sizeof(v3) * (d->n_triangles ? d->n_triangles : 1));
  s->aux = (obj_aux*)calloc(d->n_objects ? d->n_objects : 1, sizeof(obj_aux));
  s->hit_rule = d->hit_rule;
  ref_affine_identity(&s->identity);
  for (uint32_t i = 0; i < s->n_triangles; ++i) s->tri_normals[i] = triangle_flat_normal(s->triangles[i].v);
  for (uint32_t i = 0; i < s->n_objects; ++i) {
    const ref_object* o = &s->objects[i];
    obj_aux* a = &s->aux[i];
    ref_affine_identity(&a->l2w); ref_affine_identity(&a->w2l);
    if (o->kind == REF_OBJ_SPHERE) {
      memcpy(a->l2w.m, s->spheres[o->first].l2w, sizeof a->l2w.m);
      ref_affine_inverse(&a->l2w, &a->w2l);
    } else if (o->kind == REF_OBJ_TRIANGLE) {
      ref_affine_inverse(&a->l2w, &a->w2l); /* identity: scene_config.rs:189 */
    } else {
      /* AABB::empty() is the origin point: geometry/aabb.rs:15-20, model_obj.rs:62-66 */
      a->bmin = a->bmax = v3_make(0, 0, 0);
      for (uint32_t k = 0; k < o->count; ++k) {
        const double* v = s->triangles[o->first + k].v;
        v3 p1 = v3_from(v), p2 = v3_from(v + 3), p3 = v3_from(v + 6), mn, mx;
        /* Triangle::get_box triangle.rs:35-57 (identity map) */
        v3 lo = v3_make(ref_smaller_of_three(p1.x, p2.x, p3.x), ref_smaller_of_three(p1.y, p2.y, p3.y), ref_smaller_of_three(p1.z, p2.z, p3.z));
        v3 hi = v3_make(ref_bigger_of_three(p1.x, p2.x, p3.x), ref_bigger_of_three(p1.y, p2.y, p3.y), ref_bigger_of_three(p1.z, p2.z, p3.z));
        aabb_new(lo, hi, &mn, &mx);
        aabb_union(&a->bmin, &a->bmax, mn, mx);
      }
    }
  }
  return s;
}

void ref_scene_destroy(ref_scene* s) {
  if (!s) return;
  for (uint32_t i = 0; i < s->n_objects; ++i) { free(s->aux[i].bvh.nodes); free(s->aux[i].bvh.order); }
  free(s->objects); free(s->spheres); free(s->triangles); free(s->materials); free(s->tri_normals); free(s->aux);
  free(s);
}

void ref_scene_model_bounds(const ref_scene* s, uint32_t obj, double bmin[3], double bmax[3]) {
  v3_to(s->aux[obj].bmin, bmin); v3_to(s->aux[obj].bmax, bmax);
}

/* src/shapes/model_obj.rs:77-87: box prefilter, then the FIRST triangle in file order that
 * reports a hit (not the nearest). REF_HIT_NEAREST is the physically correct alternative:
 * smallest t, ties to the smaller index. */
/* ---- the optional CPU BVH (not the reference's algorithm; see cpu_bvh above) ---- */
typedef struct { const ref_triangle* tris; double (*cent)[3]; uint32_t* order; cpu_bvh_node* nodes; uint32_t n_nodes; double pad; } bvh_builder;

static int cmp_axis;
static double (*cmp_cent)[3];
static int cmp_order(const void* a, const void* b) {
  double x = cmp_cent[*(const uint32_t*)a][cmp_axis], y = cmp_cent[*(const uint32_t*)b][cmp_axis];
  return x < y ? -1 : (x > y ? 1 : (*(const uint32_t*)a < *(const uint32_t*)b ? -1 : 1));
}

static uint32_t bvh_build_range(bvh_builder* b, uint32_t first, uint32_t count) {
  uint32_t id = b->n_nodes++;
  cpu_bvh_node* n = &b->nodes[id];
  for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
  n->min_index = 0xFFFFFFFFu;
  double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = first; i < first + count; ++i) {
    uint32_t t = b->order[i];
    if (t < n->min_index) n->min_index = t;
    for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) {
      double x = b->tris[t].v[v * 3 + k];
      if (x - b->pad < n->lo[k]) n->lo[k] = x - b->pad;
      if (x + b->pad > n->hi[k]) n->hi[k] = x + b->pad;
    }
    for (int k = 0; k < 3; ++k) { if (b->cent[t][k] < clo[k]) clo[k] = b->cent[t][k]; if (b->cent[t][k] > chi[k]) chi[k] = b->cent[t][k]; }
  }
  n->first = first; n->count = count; n->left = n->right = 0;
  if (count <= 4) return id;
  int axis = 0;
  for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
  cmp_axis = axis; cmp_cent = b->cent;
  qsort(b->order + first, count, sizeof(uint32_t), cmp_order);
  uint32_t half = count / 2;
  uint32_t l = bvh_build_range(b, first, half);
  uint32_t r = bvh_build_range(b, first + half, count - half);
  n = &b->nodes[id];
  n->left = l; n->right = r; n->count = 0;
  return id;
}

void ref_scene_enable_bvh(ref_scene* s, int on) {
  s->use_bvh = on;
  if (!on) return;
  for (uint32_t i = 0; i < s->n_objects; ++i) {
    const ref_object* o = &s->objects[i];
    obj_aux* a = &s->aux[i];
    if (o->kind != REF_OBJ_MODEL || a->bvh.nodes || o->count == 0) continue;
    bvh_builder b;
    b.tris = s->triangles + o->first;
    b.cent = (double(*)[3])malloc(sizeof(double[3]) * o->count);
    b.order = (uint32_t*)malloc(sizeof(uint32_t) * o->count);
    b.nodes = (cpu_bvh_node*)malloc(sizeof(cpu_bvh_node) * 2 * (size_t)o->count);
    b.n_nodes = 0;
    double ext = 0.0;
    for (uint32_t t = 0; t < o->count; ++t) {
      b.order[t] = t;
      for (int k = 0; k < 3; ++k) {
        b.cent[t][k] = (b.tris[t].v[k] + b.tris[t].v[3 + k] + b.tris[t].v[6 + k]) / 3.0;
        for (int v = 0; v < 3; ++v) if (fabs(b.tris[t].v[v * 3 + k]) > ext) ext = fabs(b.tris[t].v[v * 3 + k]);
      }
    }
    b.pad = ext * 1e-9 + 1e-30;   /* Moller-Trumbore accepts points a few ulps outside the exact triangle */
    bvh_build_range(&b, 0, o->count);
    free(b.cent);
    a->bvh.nodes = b.nodes; a->bvh.n_nodes = b.n_nodes; a->bvh.order = b.order;
  }
}

static inline int bvh_box_hit(const cpu_bvh_node* n, const double o[3], const double inv[3], double tmax) {
  double t0 = 0.0, t1 = tmax;
  for (int k = 0; k < 3; ++k) {
    double a = (n->lo[k] - o[k]) * inv[k], b = (n->hi[k] - o[k]) * inv[k];
    double lo = a < b ? a : b, hi = a < b ? b : a;
    if (lo != lo || hi != hi) continue;           /* 0 * inf: the ray runs inside this slab's plane; do not reject */
    if (lo > t0) t0 = lo;
    if (hi < t1) t1 = hi;
  }
  return t0 <= t1 * (1.0 + 1e-12) + 1e-300;
}

static int model_hit_bvh(const ref_scene* s, const ref_object* o, const obj_aux* a, rayi ray, hiti* out) {
  const cpu_bvh* bv = &a->bvh;
  const double org[3] = {ray.o.x, ray.o.y, ray.o.z};
  const double inv[3] = {1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z};
  uint32_t stack[128];
  int sp = 0, found = 0;
  uint32_t best_index = 0xFFFFFFFFu;
  stack[sp++] = 0;
  while (sp) {
    const cpu_bvh_node* n = &bv->nodes[stack[--sp]];
    if (s->hit_rule == REF_HIT_FIRST_INDEX && n->min_index >= best_index) continue;
    /* the box test is a necessary condition only; slack on tmax because triangle t and box t round differently */
    double tmax = (s->hit_rule == REF_HIT_NEAREST && found) ? out->t * (1.0 + 1e-9) + 1e-9 : INFINITY;
    if (!bvh_box_hit(n, org, inv, tmax)) continue;
    if (n->count) {
      for (uint32_t i = n->first; i < n->first + n->count; ++i) {
        uint32_t k = bv->order[i];
        if (s->hit_rule == REF_HIT_FIRST_INDEX && k >= best_index) continue;
        hiti h;
        if (triangle_hit(s->triangles[o->first + k].v, s->tri_normals[o->first + k], &s->identity, &s->identity, ray, &h)) {
          h.prim = k;
          if (s->hit_rule == REF_HIT_FIRST_INDEX) { *out = h; best_index = k; found = 1; }
          else if (!found || h.t < out->t || (h.t == out->t && k < out->prim)) { *out = h; found = 1; }
        }
      }
    } else if (sp + 2 <= 128) {
      stack[sp++] = n->right; stack[sp++] = n->left;
    }
  }
  return found;
}

static inline int model_hit(const ref_scene* s, const ref_object* o, const obj_aux* a, rayi ray, hiti* out) {
  double tbox;
  if (!aabb_hit(a->bmin, a->bmax, ray, &tbox)) return 0;
  if (s->use_bvh && a->bvh.nodes) return model_hit_bvh(s, o, a, ray, out);
  int found = 0;
  for (uint32_t k = 0; k < o->count; ++k) {
    hiti h;
    if (triangle_hit(s->triangles[o->first + k].v, s->tri_normals[o->first + k], &s->identity, &s->identity, ray, &h)) {
      h.prim = k;
      if (s->hit_rule == REF_HIT_FIRST_INDEX) { *out = h; return 1; }
      if (!found || h.t < out->t) { *out = h; found = 1; }
    }
  }
  return found;
}

/* src/scene.rs:135-150: closest over the object list, strict '<' keeps the earlier object */
static inline int scene_hit(const ref_scene* s, rayi ray, hiti* out) {
  int found = 0;
  for (uint32_t i = 0; i < s->n_objects; ++i) {
    const ref_object* o = &s->objects[i];
    const obj_aux* a = &s->aux[i];
    hiti h;
    int ok;
    if (o->kind == REF_OBJ_SPHERE) ok = sphere_hit(&a->l2w, &a->w2l, s->spheres[o->first].radius, ray, &h);
    else if (o->kind == REF_OBJ_TRIANGLE) ok = triangle_hit(s->triangles[o->first].v, s->tri_normals[o->first], &a->l2w, &a->w2l, ray, &h);
    else ok = model_hit(s, o, a, ray, &h);
    if (!ok) continue;
    h.obj = i;
    if (!found) { *out = h; found = 1; }
    else if (h.t < out->t) *out = h;
  }
  return found;
}

int ref_scene_intersect(const ref_scene* s, const ref_ray* r, ref_hit* out) {
  hiti h;
  if (!scene_hit(s, rayi_from(r), &h)) { memset(out, 0, sizeof *out); out->obj = REF_MISS; out->prim = REF_MISS; return 0; }
  hiti_to(&h, out);
  return 1;
}

/* ---------------------------------------------------------------- integrators */

typedef struct { uint64_t seed; uint32_t pixel, sample; uint64_t segments; } path_ctx;

static inline void spec_set(float* s, float v) { for (int i = 0; i < REF_NBINS; ++i) s[i] = v; }
static inline void spec_mul(float* s, const float* c) { for (int i = 0; i < REF_NBINS; ++i) s[i] *= c[i]; }

static void trace_ray(const ref_scene* s, rayi ray, uint32_t iter, uint32_t max_depth, path_ctx* ctx, float* out);

/* src/scene.rs:45-131. `iter` is the reference's `iteraciones`; the hit being shaded is the
 * (max_depth - iter)-th of the path, which indexes the Philox counter. */
static void shade_point(const ref_scene* s, const hiti* hit, rayi incident, int entering, uint32_t iter, uint32_t max_depth,
                        path_ctx* ctx, float* out) {
  const ref_material* m = &s->materials[s->objects[hit->obj].material];
  switch (m->type) {
    case REF_MAT_EMITTER:                                             /* :56-62 */
      if (m->present & REF_HAS_EMITTED) memcpy(out, m->emitted, sizeof(float) * REF_NBINS);
      else spec_set(out, 0.0f);
      return;
    case REF_MAT_LAMBERTIAN: {                                        /* :63-79 */
      double u[2], dir[3], n[3];
      ref_uniform_pair(ctx->seed, ctx->pixel, ctx->sample, max_depth - iter, 1, u);
      v3_to(hit->n, n);
      ref_random_versor_cos_density(n, u[0], u[1], dir);
      rayi next = rayi_new(v3_add(hit->p, v3_scale(hit->n, 1e-10)), v3_from(dir));
      if (m->present & REF_HAS_AMBIENT) {
        trace_ray(s, next, iter - 1, max_depth, ctx, out);
        spec_mul(out, m->ambient);
      } else {
        spec_set(out, 0.0f);
      }
      return;
    }
    case REF_MAT_GLOSSY: {                                            /* extension: Phong lobe about the mirror direction */
      v3 i = incident.d, n = hit->n;
      rayi mirror = rayi_new(hit->p, v3_sub(i, v3_scale(n, 2.0 * v3_dot(i, n))));
      double u[2], axis[3], dir[3];
      ref_uniform_pair(ctx->seed, ctx->pixel, ctx->sample, max_depth - iter, 1, u);
      v3_to(mirror.d, axis);
      ref_phong_lobe_direction(axis, (double)m->specular_coefficient, u[0], u[1], dir);
      v3 dv = v3_from(dir);
      if (!(v3_dot(dv, n) > 0.0)) { spec_set(out, 0.0f); return; }   /* the lobe dips under the surface: absorbed */
      rayi next = rayi_new(v3_add(hit->p, v3_scale(n, 1e-10)), dv);
      trace_ray(s, next, iter - 1, max_depth, ctx, out);
      if (m->present & REF_HAS_SPECULAR) spec_mul(out, m->specular);
      return;
    }
    case REF_MAT_DIELECTRIC: {                                        /* extension: smooth dielectric, Schlick reflectance */
      v3 i = incident.d;
      v3 nf = v3_dot(hit->n, i) > 0.0 ? v3_neg(hit->n) : hit->n;
      double u[2], d[3], n[3], dir[3];
      ref_uniform_pair(ctx->seed, ctx->pixel, ctx->sample, max_depth - iter, 1, u);
      v3_to(i, d); v3_to(nf, n);
      int reflected = ref_dielectric_direction(d, n, entering, (double)m->optical_density, u[0], dir);
      v3 origin = reflected ? v3_add(hit->p, v3_scale(nf, 1e-10)) : v3_sub(hit->p, v3_scale(nf, 1e-10));
      rayi next = rayi_new(origin, v3_from(dir));
      trace_ray(s, next, iter - 1, max_depth, ctx, out);
      if (m->present & REF_HAS_SPECULAR) spec_mul(out, m->specular);
      return;
    }
    default: {                                                        /* Specular :80-112 */
      v3 i = incident.d, n = hit->n;
      v3 dir = v3_sub(i, v3_scale(n, 2.0 * v3_dot(i, n)));
      rayi next = rayi_new(v3_add(hit->p, v3_scale(n, 1e-10)), dir);
      trace_ray(s, next, iter - 1, max_depth, ctx, out);
      if (m->present & REF_HAS_SPECULAR) spec_mul(out, m->specular);
      else { float one[REF_NBINS]; spec_set(one, 1.0f); spec_mul(out, one); }
      return;
    }
  }
}

/* src/scene.rs:23-43 */
static void trace_ray(const ref_scene* s, rayi ray, uint32_t iter, uint32_t max_depth, path_ctx* ctx, float* out) {
  if (iter == 0) { spec_set(out, 0.0f); return; }
  hiti hit;
  ctx->segments += 1;
  if (!scene_hit(s, ray, &hit)) { spec_set(out, 0.0f); return; }
  const int entering = v3_dot(hit.n, ray.d) < 0.0;                  /* against the outward normal (extension materials) */
  if (v3_dot(hit.n, ray.d) > 0.0) hit.n = v3_neg(hit.n);            /* :30-32 */
  shade_point(s, &hit, ray, entering, iter, max_depth, ctx, out);
}

/* incident_light of the three integrators */
static void incident_light(const ref_scene* s, uint32_t integrator, rayi ray, uint32_t depth, path_ctx* ctx, float* out) {
  hiti hit;
  ctx->segments += 1;
  int ok = scene_hit(s, ray, &hit);
  if (integrator == REF_INT_RANDOM_WALK) {                            /* random_walk.rs:49-75 */
    if (!ok || depth == 0) { spec_set(out, 0.0f); return; }
    shade_point(s, &hit, ray, v3_dot(hit.n, ray.d) < 0.0, depth, depth, ctx, out);   /* normal NOT turned: :74 */
  } else if (integrator == REF_INT_ALBEDO) {                          /* albedo.rs:46-71 */
    if (!ok) { spec_set(out, 0.0f); return; }
    const ref_material* m = &s->materials[s->objects[hit.obj].material];
    if (m->present & REF_HAS_AMBIENT) memcpy(out, m->ambient, sizeof(float) * REF_NBINS);
    else if (m->present & REF_HAS_EMITTED) memcpy(out, m->emitted, sizeof(float) * REF_NBINS);
    else if (m->present & REF_HAS_DIFFUSED) memcpy(out, m->diffused, sizeof(float) * REF_NBINS);
    else if (m->present & REF_HAS_SPECULAR) memcpy(out, m->specular, sizeof(float) * REF_NBINS);
    else spec_set(out, 0.0f);
  } else {                                                            /* normal.rs:45-63 */
    if (!ok) { spec_set(out, 0.0f); return; }
    ref_from_rgb((float)fma(hit.n.x, 0.5, 0.5), (float)fma(hit.n.y, 0.5, 0.5), (float)fma(hit.n.z, 0.5, 0.5),
                 REF_SPEC_REFLECTANCE, out);
  }
}

/* src/integrators/integrator.rs:121-139; the two jitter uniforms come from counter
 * (pixel, sample, 0, stream 0), x first. AOV integrators report max_depth 0. */
void ref_evaluate_pixel_sample(const ref_scene* s, const ref_camera* c, const ref_render_params* p, uint32_t x,
                               uint32_t y, uint32_t sample, float out[REF_NBINS], uint64_t* segments) {
  path_ctx ctx = {p->seed, y * c->width + x, sample, 0};
  double u[2];
  ref_uniform_pair(p->seed, ctx.pixel, sample, 0, 0, u);
  double v1 = (double)x + u[0] - 0.5, v2 = (double)y + u[1] - 0.5;
  rayi ray = camera_ray(c, v1, v2);
  uint32_t depth = p->integrator == REF_INT_RANDOM_WALK ? p->max_depth : 0;
  incident_light(s, p->integrator, ray, depth, &ctx, out);
  if (segments) *segments += ctx.segments;
}

/* ------------------------------------------------ thread pool (parallel.rs) + batches */

typedef struct {
  void (*fn)(void* arg, size_t item, int worker);
  void* arg;
  size_t n_items;
  size_t next;
  pthread_mutex_t lock;
} pool_job;

typedef struct { pool_job* job; int worker; } pool_worker;

/* src/parallel.rs:44-57: workers pop tasks from one FIFO until it is empty */
static void* pool_main(void* p) {
  pool_worker* w = (pool_worker*)p;
  for (;;) {
    pthread_mutex_lock(&w->job->lock);
    size_t item = w->job->next++;
    pthread_mutex_unlock(&w->job->lock);
    if (item >= w->job->n_items) break;
    w->job->fn(w->job->arg, item, w->worker);
  }
  return NULL;
}

static void pool_run(void (*fn)(void*, size_t, int), void* arg, size_t n_items, int threads) {
  if (threads < 1) threads = 1;
  pool_job job = {fn, arg, n_items, 0, PTHREAD_MUTEX_INITIALIZER};
  if (threads == 1) { pool_worker w = {&job, 0}; pool_main(&w); return; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  pool_worker* ws = (pool_worker*)malloc(sizeof(pool_worker) * threads);
  for (int i = 0; i < threads; ++i) { ws[i].job = &job; ws[i].worker = i; pthread_create(&th[i], NULL, pool_main, &ws[i]); }
  for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
  free(th); free(ws);
}

#define BATCH_CHUNK 256

typedef struct { const ref_scene* s; const ref_ray* rays; size_t n; ref_hit* out; } isect_job;

static void isect_chunk(void* a, size_t item, int worker) {
  (void)worker;
  isect_job* j = (isect_job*)a;
  size_t lo = item * BATCH_CHUNK, hi = lo + BATCH_CHUNK < j->n ? lo + BATCH_CHUNK : j->n;
  for (size_t i = lo; i < hi; ++i) ref_scene_intersect(j->s, &j->rays[i], &j->out[i]);
}

void ref_intersect_batch(const ref_scene* s, const ref_ray* rays, size_t n, ref_hit* out, int threads) {
  isect_job j = {s, rays, n, out};
  pool_run(isect_chunk, &j, (n + BATCH_CHUNK - 1) / BATCH_CHUNK, threads);
}

typedef struct {
  const ref_scene* s; const ref_ray* rays; size_t n; uint32_t integrator, max_depth; uint64_t seed;
  uint32_t pixel_base, sample; float* out; uint64_t* seg_per_worker;
} trace_job;

static void trace_chunk(void* a, size_t item, int worker) {
  trace_job* j = (trace_job*)a;
  size_t lo = item * BATCH_CHUNK, hi = lo + BATCH_CHUNK < j->n ? lo + BATCH_CHUNK : j->n;
  for (size_t i = lo; i < hi; ++i) {
    path_ctx ctx = {j->seed, j->pixel_base + (uint32_t)i, j->sample, 0};
    uint32_t depth = j->integrator == REF_INT_RANDOM_WALK ? j->max_depth : 0;
    incident_light(j->s, j->integrator, rayi_from(&j->rays[i]), depth, &ctx, j->out + i * REF_NBINS);
    j->seg_per_worker[worker] += ctx.segments;
  }
}

void ref_trace_batch(const ref_scene* s, const ref_ray* rays, size_t n, uint32_t integrator, uint32_t max_depth,
                     uint64_t seed, uint32_t pixel_base, uint32_t sample_index, float* out_spectra,
                     uint64_t* segments, int threads) {
  if (threads < 1) threads = 1;
  uint64_t* seg = (uint64_t*)calloc((size_t)threads, sizeof(uint64_t));
  trace_job j = {s, rays, n, integrator, max_depth, seed, pixel_base, sample_index, out_spectra, seg};
  pool_run(trace_chunk, &j, (n + BATCH_CHUNK - 1) / BATCH_CHUNK, threads);
  if (segments) { *segments = 0; for (int i = 0; i < threads; ++i) *segments += seg[i]; }
  free(seg);
}

/* ---------------------------------------------------------------- render (integrator.rs:30-118) */

typedef struct {
  const ref_scene* s; const ref_camera* c; const ref_render_params* p;
  uint32_t x0, y0, x1, y1, tile, tiles_y, s0, s1;
  uint8_t* rgb8; float* xyz_ref; float* xyz_lin; float* xyz_sum;
  uint64_t* seg_per_worker; uint64_t* pix_per_worker;
} render_job;

static void render_tile(void* a, size_t item, int worker) {
  render_job* j = (render_job*)a;
  /* tiles enumerated x-major: cartesian_product(0..n_x, 0..n_y) :50 */
  uint32_t tx = (uint32_t)(item / j->tiles_y), ty = (uint32_t)(item % j->tiles_y);
  if (j->p->tile_world && item % j->p->tile_world != j->p->tile_rank) return;
  uint32_t xa = j->x0 + tx * j->tile, ya = j->y0 + ty * j->tile;
  uint32_t xb = xa + j->tile < j->x1 ? xa + j->tile : j->x1, yb = ya + j->tile < j->y1 ? ya + j->tile : j->y1;
  uint32_t W = j->c->width;
  uint32_t n_samples = j->s1 - j->s0;
  uint64_t segs = 0;
  for (uint32_t x = xa; x < xb; ++x) {
    for (uint32_t y = ya; y < yb; ++y) {                               /* :67 x outer, y inner */
      float sum[REF_NBINS], lin[3] = {0.0f, 0.0f, 0.0f}, run[3] = {0.0f, 0.0f, 0.0f};
      size_t pix = (size_t)y * W + x;
      spec_set(sum, 0.0f);
      /* xyz_sum continues the per-pixel running sum in sample order, as one long sequential sum would */
      if (j->xyz_sum) { run[0] = j->xyz_sum[pix * 3]; run[1] = j->xyz_sum[pix * 3 + 1]; run[2] = j->xyz_sum[pix * 3 + 2]; }
      for (uint32_t k = j->s0; k < j->s1; ++k) {                       /* :69-78 */
        float smp[REF_NBINS], xyz[3];
        ref_evaluate_pixel_sample(j->s, j->c, j->p, x, y, k, smp, &segs);
        for (int i = 0; i < REF_NBINS; ++i) sum[i] += smp[i];
        if (j->xyz_lin || j->xyz_sum) {
          ref_to_xyz(smp, xyz);
          lin[0] += xyz[0]; lin[1] += xyz[1]; lin[2] += xyz[2];
          run[0] += xyz[0]; run[1] += xyz[1]; run[2] += xyz[2];
        }
      }
      float nf = (float)n_samples;
      for (int i = 0; i < REF_NBINS; ++i) sum[i] /= nf;                /* :79 */
      float xyz[3], rgb[3];
      ref_to_xyz(sum, xyz);
      ref_xyz_to_rgb(xyz, rgb);
      if (j->rgb8) ref_tonemap(rgb, j->rgb8 + pix * 3);
      if (j->xyz_ref) { j->xyz_ref[pix * 3] = xyz[0]; j->xyz_ref[pix * 3 + 1] = xyz[1]; j->xyz_ref[pix * 3 + 2] = xyz[2]; }
      if (j->xyz_lin) { j->xyz_lin[pix * 3] = lin[0] / nf; j->xyz_lin[pix * 3 + 1] = lin[1] / nf; j->xyz_lin[pix * 3 + 2] = lin[2] / nf; }
      if (j->xyz_sum) { j->xyz_sum[pix * 3] = run[0]; j->xyz_sum[pix * 3 + 1] = run[1]; j->xyz_sum[pix * 3 + 2] = run[2]; }
      j->pix_per_worker[worker] += 1;
    }
  }
  j->seg_per_worker[worker] += segs;
}

void ref_render(const ref_scene* s, const ref_camera* c, const ref_render_params* p, uint8_t* rgb8, float* xyz_ref,
                float* xyz_lin, float* xyz_sum, ref_stats* stats, int threads) {
  if (threads < 1) threads = 1;
  render_job j;
  memset(&j, 0, sizeof j);
  j.s = s; j.c = c; j.p = p;
  j.x0 = p->x0; j.y0 = p->y0;
  j.x1 = p->x1 ? p->x1 : c->width; j.y1 = p->x1 ? p->y1 : c->height;
  j.tile = p->tile_size ? p->tile_size : 16;                          /* :39 */
  j.s0 = p->sample_end ? p->sample_begin : 0; j.s1 = p->sample_end ? p->sample_end : p->spp;
  uint32_t tiles_x = (j.x1 - j.x0 + j.tile - 1) / j.tile;
  j.tiles_y = (j.y1 - j.y0 + j.tile - 1) / j.tile;
  j.rgb8 = rgb8; j.xyz_ref = xyz_ref; j.xyz_lin = xyz_lin; j.xyz_sum = xyz_sum;
  j.seg_per_worker = (uint64_t*)calloc((size_t)threads, sizeof(uint64_t));
  j.pix_per_worker = (uint64_t*)calloc((size_t)threads, sizeof(uint64_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pool_run(render_tile, &j, (size_t)tiles_x * j.tiles_y, threads);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (stats) {
    stats->segments = 0;
    for (int i = 0; i < threads; ++i) stats->segments += j.seg_per_worker[i];
    stats->paths = 0;
    for (int i = 0; i < threads; ++i) stats->paths += j.pix_per_worker[i] * (uint64_t)(j.s1 - j.s0);
    stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    stats->threads = (uint32_t)threads; stats->pad = 0;
  }
  free(j.seg_per_worker); free(j.pix_per_worker);
}
