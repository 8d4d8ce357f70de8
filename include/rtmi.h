/* rtmi.h -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * Drop-in boundary for the per-pixel hot path of the lukeovalle/raytracing renderer:
 * camera-ray generation, ray/scene intersection and the Monte-Carlo bounce loop with 60-bin
 * spectral weights. A host program (the reference's Rust host through an FFI shim, or the C++
 * host in this repository) flattens its Camera and Scene into the plain structs below and
 * calls these entry points. No C++ or torch types cross this boundary. Citations are
 * file:line in the reference repository.
 *
 * Conventions
 *  - every function returns 0 (RTMI_OK) or a negative rtmi_status; rtmi_last_error() gives the
 *    message of the calling thread's last failure; nothing aborts or throws across the ABI;
 *  - the caller owns every input array (copied during the call) and every output buffer;
 *  - a context is bound to one GPU; one call at a time per context; distinct contexts may be
 *    used from different threads or processes (one process per GPU, or rtmi_multi_* inside one process);
 *  - "host" pointers are ordinary memory, "device" pointers are HIP device memory of the
 *    context's GPU; `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *  - images are row-major with y = 0 at the top (src/camera.rs:26-29).
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_NBINS 60          /* SampledSpectrum = 60 bins over 400..700 nm (src/spectrum/mod.rs:262-265) */
#define RTMI_MISS 0xFFFFFFFFu
#define RTMI_MAX_DEPTH 32      /* longest path (ray casts per path) the kernels carry state for */

typedef enum {
  RTMI_OK = 0, RTMI_BAD_ARG = -1, RTMI_NO_DEVICE = -2, RTMI_OOM = -3, RTMI_HIP_ERROR = -4, RTMI_UNSUPPORTED = -5,
  RTMI_RCCL_ERROR = -6,  /* the collective library could not be loaded, or one of its calls failed */
  RTMI_INTERNAL = -7     /* a consistency check inside the library failed; the output of the call is not valid */
} rtmi_status;

/* enum Shape (src/shapes/shape.rs:34-39); BoxAABB is never built by any loader and is not carried */
enum { RTMI_OBJ_SPHERE = 0, RTMI_OBJ_TRIANGLE = 1, RTMI_OBJ_MODEL = 2 };
/* enum Type (src/material.rs:5-9). GLOSSY and DIELECTRIC are extensions with no counterpart in the
 * reference, which parses Ns and Ni (src/material.rs:20-22,49-50) but never reads them:
 *   GLOSSY      Phong lobe of exponent specular_coefficient about the mirror direction; a sample under the
 *               surface ends the path; weight = specular colour (1 if absent), like Specular
 *   DIELECTRIC  smooth interface of index optical_density: reflection with Schlick's probability (the form
 *               sketched at src/scene.rs:103-107) or on total internal reflection, refraction otherwise;
 *               weight = specular colour (1 if absent) */
enum { RTMI_MAT_EMITTER = 0, RTMI_MAT_LAMBERTIAN = 1, RTMI_MAT_SPECULAR = 2, RTMI_MAT_GLOSSY = 3, RTMI_MAT_DIELECTRIC = 4 };
/* which Option<SampledSpectrum> fields of Material are Some (src/material.rs:15-19) */
enum { RTMI_HAS_AMBIENT = 1, RTMI_HAS_EMITTED = 2, RTMI_HAS_DIFFUSED = 4, RTMI_HAS_SPECULAR = 8 };
/* Hit rule inside a mesh. FIRST_INDEX is the reference's behaviour (first triangle in file order
 * that the ray hits, src/shapes/model_obj.rs:80-84); NEAREST is smallest t, ties to the smaller index. */
enum { RTMI_HIT_FIRST_INDEX = 0, RTMI_HIT_NEAREST = 1 };
/* enum Integrator (src/integrators/integrator.rs:145-151) */
enum { RTMI_INT_RANDOM_WALK = 0, RTMI_INT_ALBEDO = 1, RTMI_INT_NORMAL = 2 };
/* enum SpectrumType (src/spectrum/mod.rs:8-11) */
enum { RTMI_SPEC_REFLECTANCE = 0, RTMI_SPEC_ILLUMINANT = 1 };
/* rtmi_scene_desc.flags. The reference has no acceleration structure at all (src/shapes/model_obj.rs:77-87 scans every
 * triangle); both builders only prune work and give the same hits. */
enum {
  RTMI_SCENE_GPU_BVH = 1   /* build the meshes' BVHs on the GPU (Morton order + radix tree: built in a tenth of the time of
                              the default host SAH build, slower to traverse), for scenes that change between renders */
};
/* rtmi_render_params.flags */
enum {
  RTMI_FLAG_TIME_KERNELS = 1,    /* bracket every closest-hit kernel with HIP events (adds host syncs) */
  RTMI_FLAG_COUNT_TRAVERSAL = 2, /* count BVH nodes visited and triangles tested (slower kernel variant) */
  RTMI_FLAG_EMPTY_SHARE = 4      /* this call's share of the render is empty (a rank beyond the samples of a sample partition):
                                    the arguments are checked and nothing is rendered */
};

/* Ray (src/geometry/ray.rs:5-10): direction must be normalised; max_t is always +inf on this path */
typedef struct { double o[3]; double d[3]; } rtmi_ray;

/* Intersection (src/shapes/common.rs:9-17) without the by-value shape copy */
typedef struct {
  uint32_t obj;   /* index into rtmi_scene_desc.objects, RTMI_MISS when nothing was hit */
  uint32_t prim;  /* triangle index inside a model, in file order; 0 for spheres and lone triangles */
  double t;       /* as the reference reports it; spheres: along the re-normalised local ray (src/shapes/sphere.rs:101) */
  double p[3];    /* world-space hit point */
  double n[3];    /* geometric outward normal, not turned towards the ray */
} rtmi_hit;

/* one entry of Scene::objetos (src/scene.rs:7-9), in scene order */
typedef struct {
  uint32_t kind;      /* RTMI_OBJ_* */
  uint32_t first;     /* first sphere (kind SPHERE) or first triangle (TRIANGLE, MODEL) */
  uint32_t count;     /* 1, or the number of triangles of a model (file order) */
  uint32_t material;  /* index into materials */
} rtmi_object;

/* Sphere (src/shapes/sphere.rs:8-14): radius at the local origin under a row-major 3x4 local->world map */
typedef struct { double l2w[12]; double radius; } rtmi_sphere;

/* Triangle (src/shapes/triangle.rs:9-15) with the identity local->world map every loader uses
 * (src/scene_config.rs:189, src/shapes/model_obj.rs:54): world-space vertices, winding gives the normal */
typedef struct { double v[9]; } rtmi_triangle;

/* Material (src/material.rs:12-23); specular_coefficient / optical_density are read by the two extension types only */
typedef struct {
  uint32_t type;     /* RTMI_MAT_* */
  uint32_t present;  /* RTMI_HAS_* bits */
  float ambient[RTMI_NBINS], emitted[RTMI_NBINS], diffused[RTMI_NBINS], specular[RTMI_NBINS];
  float specular_coefficient;  /* Ns (src/material.rs:20) */
  float optical_density;       /* Ni (src/material.rs:22) */
} rtmi_material;

typedef struct {
  const rtmi_object* objects; uint32_t n_objects;
  const rtmi_sphere* spheres; uint32_t n_spheres;
  const rtmi_triangle* triangles; uint32_t n_triangles;
  const rtmi_material* materials; uint32_t n_materials;
  uint32_t hit_rule;  /* RTMI_HIT_* */
  uint32_t flags;     /* RTMI_SCENE_* */
} rtmi_scene_desc;

/* Camera (src/camera.rs:4-9): focus and the screen rectangle (p1 top-left, p2 top-right,
 * p3 bottom-left, p4 bottom-right) */
typedef struct { double focus[3], p1[3], p2[3], p3[3], p4[3]; uint32_t width, height; } rtmi_camera;

/* What SamplerIntegrator::render is asked to do (src/integrators/integrator.rs:17-118) plus the
 * partition that replaces the tile thread pool (src/parallel.rs). */
typedef struct {
  uint32_t integrator;  /* RTMI_INT_* */
  uint32_t max_depth;   /* RandomWalkIntegrator depth (src/main.rs:93); ignored by the AOV integrators */
  uint32_t spp;         /* total_samples(): the divisor of the Monte-Carlo mean */
  uint32_t flags;       /* RTMI_FLAG_* */
  uint64_t seed;        /* Philox key */
  uint32_t sample_begin, sample_end;  /* sample-range partition: this call renders samples [begin,end); end==0 -> [0,spp) */
  uint32_t tile_rank, tile_world;     /* tile partition: tile k (x-major as src/integrators/integrator.rs:50) is rendered iff k % world == rank; world==0 -> all */
  uint32_t tile_size;                 /* 0 -> 16 (src/integrators/integrator.rs:39) */
  uint32_t reserved;                  /* 0 */
} rtmi_render_params;

typedef struct {
  uint64_t segments;       /* closest-hit queries issued by the integrator (primary + bounce rays) */
  uint64_t paths;          /* (pixel, sample) pairs evaluated */
  uint64_t nodes_visited;  /* with RTMI_FLAG_COUNT_TRAVERSAL */
  uint64_t prims_tested;   /* with RTMI_FLAG_COUNT_TRAVERSAL: mesh triangles put through Moller-Trumbore */
  double ms_total;         /* host wall time of the call */
  double ms_kernels;       /* with RTMI_FLAG_TIME_KERNELS: device time summed over closest-hit kernel launches */
  uint32_t launches;       /* closest-hit kernel launches (bounce, or shade + traverse) */
  uint32_t batches;
  uint64_t mesh_rays;      /* rays handed to the BVH traversal kernel (two-kernel pipeline only) */
  double ms_traverse;      /* with RTMI_FLAG_TIME_KERNELS: the part of ms_kernels spent in traversal launches */
  uint32_t traverse_launches;
  uint32_t reserved;
} rtmi_stats;

typedef struct {
  uint32_t n_nodes, max_depth, n_mesh_triangles, n_spectra;
  uint64_t node_bytes, triangle_bytes;
  double bvh_build_ms;   /* host wall time rtmi_scene_create spent building the BVHs of the scene's models */
  double bvh_device_ms;  /* the part of it inside the device builder (RTMI_SCENE_GPU_BVH), copies to and from the device included */
  double upload_ms;      /* ... and flattening the rest and copying everything to the device */
} rtmi_scene_info;

typedef struct rtmi_ctx rtmi_ctx;
typedef struct rtmi_scene rtmi_scene;

/* ---- context ---------------------------------------------------------------------------- */

/* Binds a context to GPU `device_id` (-1 = the current HIP device). Replaces nothing in the
 * reference: the reference has no device. max_batch_paths caps the path slots one render batch uses;
 * 0 = default, 2^28 slots. A batch needs about 210 bytes of workspace per slot (two path queues, the
 * traversal list, the emit queue, the per-slot contributions), allocated only up to what a render
 * actually needs. A render overlaps its batches on up to three internal lanes, each with a workspace
 * of its own, so a context holds up to 3 x (one batch) of device memory; when that does not fit, the
 * library first gives up lanes (down to one, with the batch size asked for) and only then, if
 * max_batch_paths was 0, halves its own default batch. RTMI_LANES=1..4 fixes the lane count. */
int rtmi_ctx_create(int device_id, uint64_t max_batch_paths, rtmi_ctx** out);
void rtmi_ctx_destroy(rtmi_ctx* ctx);
/* Message of the calling thread's most recent failed call ("" if none). */
const char* rtmi_last_error(void);
/* Library version string, "rtmi <version> gfx950". */
const char* rtmi_version(void);

/* ---- spectra: SampledSpectrum::{init, from_RGB, to_XYZ} and XYZ_to_RGB
 *      (src/spectrum/mod.rs:386-522, 525-581, 590-609, 630-636), host side, used at load time */
int rtmi_spectrum_from_rgb(float r, float g, float b, int type, float out[RTMI_NBINS]);
int rtmi_spectrum_to_xyz(const float spectrum[RTMI_NBINS], float xyz[3]);
int rtmi_xyz_to_rgb(const float xyz[3], float rgb[3]);
/* The 60-bin CIE matching curves the kernels use (src/spectrum/mod.rs:268-270 after init()). */
int rtmi_spectrum_cie_tables(float x[RTMI_NBINS], float y[RTMI_NBINS], float z[RTMI_NBINS]);

/* ---- camera: Camera::new (src/camera.rs:19-51) ------------------------------------------- */
int rtmi_camera_new(const double focus[3], double focal_distance, double field_of_view_deg,
                    const double rotation_rpy[3], uint32_t width, uint32_t height, rtmi_camera* out);
/* Camera::get_ray (src/camera.rs:71-85), host side, for callers that build their own ray sets. */
int rtmi_camera_get_ray(const rtmi_camera* cam, double i, double j, rtmi_ray* out);

/* ---- scene: what Scene::add_shape builds (src/scene.rs:18-21), flattened and uploaded ----- */
int rtmi_scene_create(rtmi_ctx* ctx, const rtmi_scene_desc* desc, rtmi_scene** out);
void rtmi_scene_destroy(rtmi_scene* scene);
int rtmi_scene_get_info(const rtmi_scene* scene, rtmi_scene_info* out);

/* ---- Scene::intersect_ray (src/scene.rs:135-150) over a batch of rays --------------------- */
int rtmi_intersect_batch(rtmi_scene* scene, const rtmi_ray* host_rays, size_t n, rtmi_hit* host_hits, rtmi_stats* stats);
/* Same, rays and hits already in device memory; asynchronous on `stream`. */
int rtmi_intersect_batch_device(rtmi_scene* scene, const rtmi_ray* dev_rays, size_t n, rtmi_hit* dev_hits, void* stream);

/* ---- SamplerIntegrator::incident_light (src/integrators/random_walk.rs:49-75, albedo.rs:46-71,
 *      normal.rs:45-63) along caller-supplied rays. Ray i draws its random numbers from the
 *      Philox counter (pixel_base + i, sample_index, bounce). Output: n x 60 spectra (host). */
int rtmi_trace_batch(rtmi_scene* scene, const rtmi_ray* host_rays, size_t n, uint32_t integrator, uint32_t max_depth,
                     uint64_t seed, uint32_t pixel_base, uint32_t sample_index, float* host_spectra, rtmi_stats* stats);

/* ---- SamplerIntegrator::render (src/integrators/integrator.rs:30-118) --------------------- */
/* Whole render on one GPU. host_rgb8: W*H*3 bytes (gamma 1/2.2, `(256*v) as u8`, :80-90).
 * host_xyz (optional): W*H*3 floats, the mean CIE XYZ per pixel before tone mapping. */
int rtmi_render(rtmi_scene* scene, const rtmi_camera* cam, const rtmi_render_params* params,
                uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats);
/* The same work for this call's share of the samples/tiles only: adds the per-pixel SUM of the
 * samples' CIE XYZ into dev_xyz_sum (W*H*3 floats, device). Ranks of a multi-GPU job each call
 * this, sum (sample partition) or merge (tile partition) their buffers with one collective, and
 * finish with rtmi_tonemap. Asynchronous on `stream` unless RTMI_FLAG_TIME_KERNELS is set. */
int rtmi_render_accumulate(rtmi_scene* scene, const rtmi_camera* cam, const rtmi_render_params* params,
                           float* dev_xyz_sum, void* stream, rtmi_stats* stats);
/* rtmi_render_accumulate for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. It has no counterpart in the reference. The work, the arguments and the bits
 * added to dev_xyz_sum are those of rtmi_render_accumulate, but the path queues, traversal list, emit queue, per-slot
 * contributions, fill counters, device statistics and tile list live in a static 64 MiB device array that every GPU
 * the library runs on holds from the moment the library is loaded, and the call makes no hipMalloc, hipFree or other
 * allocating call. The render is cut into batches of at most 2^17 path slots (many more launches than the allocating
 * entry point needs, so it is slower). Limits, reported as RTMI_UNSUPPORTED: tile_size * tile_size <= 2^17 (tile_size
 * <= 360), at most 2^18 tiles in the call's share, no RTMI_FLAG_TIME_KERNELS. Every call on a GPU uses that GPU's one
 * array, whatever the context or scene: calls that overlap in time on one GPU race; calls on one stream, or otherwise
 * ordered, are safe. rtmi_render_collect_stats works after it as after rtmi_render_accumulate. */
int rtmi_render_accumulate_noalloc(rtmi_scene* scene, const rtmi_camera* cam, const rtmi_render_params* params,
                                   float* dev_xyz_sum, void* stream);
/* Fills *stats with the counters of the last render_accumulate on `scene` once its stream work
 * has completed (synchronises the stream). This is also where a render that had to drop paths (a path-queue region
 * overflowed: an internal sizing error, never expected) is reported, as RTMI_INTERNAL: rtmi_render_accumulate and
 * rtmi_render_accumulate_noalloc are asynchronous and return before that can be known. The condition is sticky on the
 * device until it has been reported once, so collecting the statistics of any later render on the same context still
 * reports it; rtmi_render and rtmi_multi_render collect them themselves. */
int rtmi_render_collect_stats(rtmi_scene* scene, void* stream, rtmi_stats* stats);
/* mean = sum / spp, XYZ -> RGB, clamp, gamma, u8 (src/integrators/integrator.rs:79-90). */
int rtmi_tonemap(rtmi_ctx* ctx, const float* dev_xyz_sum, uint32_t spp, uint32_t width, uint32_t height,
                 uint8_t* dev_rgb8, void* stream);

/* ---- one render over several GPUs of a node: what replaces the tile thread pool (src/parallel.rs:37-61) and the
 *      render loop that feeds it (src/integrators/integrator.rs:30-118) when more than one device is used.
 *      The scene is replicated on every device, each device renders its share on its own host thread and stream, and
 *      ONE collective merges the per-pixel XYZ sums (W*H*3 floats) on the first device: an RCCL reduce over xGMI.
 *        RTMI_PARTITION_SAMPLES  device r renders samples [r*spp/n, (r+1)*spp/n) of every pixel with the same Philox
 *                                key; perfectly balanced; differs from one device only by float summation order
 *        RTMI_PARTITION_TILES    device r renders the tiles k (x-major, src/integrators/integrator.rs:50) with
 *                                k % n == r; the buffers are disjoint, so the image is bit-identical to one device's
 *      device_ids may repeat (more shares than GPUs); shares that live on the same GPU cannot form an RCCL
 *      communicator, so the merge is then done with copies to the first device and adds. RCCL is bound at run time
 *      (a librccl already in the process, RTMI_RCCL_LIB, the directory of the loaded HIP runtime, the loader path). */
enum { RTMI_PARTITION_SAMPLES = 0, RTMI_PARTITION_TILES = 1 };
typedef struct rtmi_multi rtmi_multi;
typedef struct {
  uint32_t n_ranks;
  uint32_t uses_rccl;        /* 1: the merge is an ncclReduce; 0: one rank, or copies and adds */
  char rccl_library[240];    /* the library the RCCL entry points were taken from */
} rtmi_multi_info_t;
/* Contexts and scene replicas on device_ids[0..n); max_batch_paths as in rtmi_ctx_create, per device. */
int rtmi_multi_create(const int* device_ids, int n, uint64_t max_batch_paths, const rtmi_scene_desc* desc, rtmi_multi** out);
void rtmi_multi_destroy(rtmi_multi* multi);
int rtmi_multi_info(const rtmi_multi* multi, rtmi_multi_info_t* out);
/* SamplerIntegrator::render over all devices of the handle. The partition fields of `params` (sample_begin/end,
 * tile_rank/world) must be 0. Outputs as in rtmi_render. stats: totals over the devices (ms_total = wall time of the
 * call); per_rank_stats (optional): n entries, one per device. Blocks until the image is on the host. */
int rtmi_multi_render(rtmi_multi* multi, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition,
                      uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats, rtmi_stats* per_rank_stats);

/* Progressive form of rtmi_multi_render: the spp samples are taken in n_passes calls (pass = 0 .. n_passes-1, in order;
 * pass p renders the p-th of n_passes balanced sample ranges, shared out over the devices as above). Every call ends
 * with the reduce and the tone map, so it returns the image of all samples taken so far (their mean); after the last
 * pass that is the image rtmi_multi_render gives -- bit for bit under RTMI_PARTITION_TILES. The devices keep their own
 * sums between the calls; pass 0 clears them. The reference renders all-or-nothing (src/main.rs:79,88,97). */
int rtmi_multi_render_pass(rtmi_multi* multi, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition,
                           uint32_t pass, uint32_t n_passes, uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
