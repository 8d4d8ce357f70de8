// rt_types.h -- plain-data layouts shared by the host scene builder and the HIP kernels.
//
// Everything here is what actually sits in HBM:
//   objects   : the scene's object list in scene order (the closest-hit loop walks it like
//               Scene::intersect_ray does, src/scene.rs:135-150)
//   spheres   : local->world and world->local maps + radius (the inverse the reference recomputes
//               per test, src/shapes/sphere.rs:39,52, is stored once)
//   top_tris  : lone triangles with their flat normal
//   nodes     : one binary BVH per mesh, 64-byte nodes, all meshes concatenated
//   mesh_tris : mesh triangles in BVH leaf order, f64 vertices + their file-order index
//   mesh_frames : per mesh triangle, the normal and the divisor of the reported t (functions of the vertices alone)
//   materials : type tag + indices into a table of 60-bin spectra
#pragma once
#include <stdint.h>

#define RT_NBINS 60
#define RT_MISS 0xFFFFFFFFu
#define RT_MAX_DEPTH 32
#define RT_LEAF_FLAG 0x80000000u
#define RT_REF_NONE 0xFFFFFFFFu     // empty child slot
#define RT_LEAF_MAX_TRIS 15
#define RT_NO_SPECTRUM 0xFFu        // chain entry meaning "multiply by 1.0" / absent spectrum

enum { RT_OBJ_SPHERE = 0, RT_OBJ_TRIANGLE = 1, RT_OBJ_MODEL = 2 };
enum { RT_MAT_EMITTER = 0, RT_MAT_LAMBERTIAN = 1, RT_MAT_SPECULAR = 2, RT_MAT_GLOSSY = 3, RT_MAT_DIELECTRIC = 4 };
enum { RT_HAS_AMBIENT = 1, RT_HAS_EMITTED = 2, RT_HAS_DIFFUSED = 4, RT_HAS_SPECULAR = 8 };
enum { RT_HIT_FIRST_INDEX = 0, RT_HIT_NEAREST = 1 };
enum { RT_INT_RANDOM_WALK = 0, RT_INT_ALBEDO = 1, RT_INT_NORMAL = 2 };

struct DevObject {
  uint32_t kind;
  uint32_t first;      // sphere index / top_tris index / first mesh triangle (leaf order base)
  uint32_t count;
  uint32_t material;
  uint32_t root;       // models: index of the BVH root node
  uint32_t pad[3];
  double center[3];    // the point box_lo/box_hi (and a model's node boxes) are relative to
  double radius;       // the whole object lies within this distance of center
  float box_lo[3];     // binary32 box around the whole object, relative to center, padded and rounded outward
  float box_hi[3];
  // Models only: the box ModelObj::intersect tests before it looks at any triangle (src/shapes/model_obj.rs:62-66,78):
  // the union of the triangles' boxes grown from AABB::empty(), which is the origin point (src/geometry/aabb.rs:15-20).
  // gate_lo/gate_hi is the exact binary64 box; inner_lo/inner_hi is a binary32 box relative to center that lies strictly
  // inside it (shrunk and rounded inward): a ray that crosses the inner box certainly passes the exact test, so only the
  // rays in the thin shell between the two run the binary64 slab test.
  float inner_lo[3];
  float inner_hi[3];
  double gate_lo[3];
  double gate_hi[3];
  // Models only: the grid the node boxes are quantised on, relative to center: plane = grid_lo + q * grid_step
  float grid_lo[3];
  float grid_step[3];
  uint32_t pad2[2];
};
static_assert(sizeof(DevObject) == 192, "DevObject must be 192 bytes");

struct DevSphere {
  double l2w[12];
  double w2l[12];
  double radius;
  uint32_t translation_only;   // the linear parts of both maps are exactly the identity
  uint32_t pad;
};

// What the reference derives from a triangle's vertices alone, formed once on the host with the same IEEE
// operations (src/shapes/triangle.rs:30,83,91-92): n1 = normalize((p2-p1) x (p3-p1)), mapped through the identity
// local->world matrix, is the normal of a hit before normalisation; a hit reports t / |n1| and normalize(n1).
struct DevTriFrame {
  double n[3];         // normalize(n1)
  double len;          // sqrt(n1 . n1)
};

struct DevTopTri {
  double v[9];
  DevTriFrame frame;
};

// Child reference: RT_LEAF_FLAG | first << 4 | count for a leaf (first is relative to the
// model's first mesh triangle), a node index (relative to the model's root) otherwise,
// RT_REF_NONE for an empty slot. min_index = smallest file-order triangle index below the child.
// This is the builder's working form; the device walks the quantised form below.
struct BvhNode {
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  uint32_t ref0, ref1;
  uint32_t min_index0, min_index1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// What the kernels traverse: both children's boxes as 16-bit coordinates on a per-mesh grid (DevObject::grid_lo,
// grid_step), rounded outward by a further grid step, so a quantised box always contains the binary32 box it came
// from. A box test fetches 24 bytes of planes instead of 48, and a node 32 or 48 bytes instead of 64: the traversal is
// bound by the vector memory path (address and data-return units busy 75-90 % of its time), not by arithmetic.
//   words 0-5  lo0.x|lo0.y<<16, lo0.z|hi0.x<<16, hi0.y|hi0.z<<16, lo1.x|lo1.y<<16, lo1.z|hi1.x<<16, hi1.y|hi1.z<<16
//   RT_NODES_SMALL   (32 bytes; every mesh of the scene has at most RT_SMALL_MAX_TRIS triangles)
//              word 6 = ref0 | ref1 << 16, word 7 = min_index0 | min_index1 << 16, with 16-bit references:
//              0xFFFF none, 0x8000 | first << 2 | (count - 1) a leaf of 1..4 triangles, else a node index
//   RT_NODES_NEAREST (32 bytes; nearest-hit rule only, which never reads the minimum indices)
//              word 6 = ref0, word 7 = ref1, as in BvhNode
//   RT_NODES_WIDE    (48 bytes) words 6-9 = ref0, ref1, min_index0, min_index1, as in BvhNode; words 10-11 unused
//   RT_NODES_QUAD    (64 bytes; nearest-hit rule, large meshes) four children per node, made by pulling grandchildren of
//              the binary tree up into their grandparent: words 0-11 = the children's boxes, six 16-bit levels each
//              (lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of child 0, then child 1, 2, 3), words 12-15 = their references as in
//              BvhNode (RT_REF_NONE for an unused slot). A ray's walk through a large tree waits on a chain of
//              dependent node fetches; four-wide nodes halve the chain.
//   With RT_SMALL_F16 the twelve planes of the RT_NODES_SMALL format are binary16 values (coordinate / grid_step, grid_lo = 0,
//   rounded outward; csrc/bvh_quant.h) in the same positions instead of integer levels.
#ifndef RT_SMALL_F16
#define RT_SMALL_F16 1
#endif
struct alignas(16) RtWord4 { uint32_t x, y, z, w; };
enum { RT_NODES_SMALL = 0, RT_NODES_NEAREST = 1, RT_NODES_WIDE = 2, RT_NODES_QUAD = 3 };
#define RT_SMALL_MAX_TRIS 8188u
#define RT_SMALL_LEAF_FLAG 0x8000u
#define RT_SMALL_REF_NONE 0xFFFFu

struct DevMeshTri {
  double v[9];
  uint32_t index;      // file-order index inside its model
  uint32_t pad;
};
static_assert(sizeof(DevMeshTri) == 80, "DevMeshTri must be 80 bytes");

// 32 bytes on a 16-byte boundary: a record read at a wave-uniform index then comes with two aligned scalar loads
struct alignas(16) DevMaterial {
  uint32_t type;
  uint32_t present;
  uint8_t ambient, emitted, diffused, specular;  // rows of the spectrum table, RT_NO_SPECTRUM if absent
  float ns;            // exponent of the glossy lobe
  float ni;            // refractive index of a dielectric
  uint32_t pad[3];
};
static_assert(sizeof(DevMaterial) == 32, "DevMaterial must be 32 bytes");

struct DevScene {
  const DevObject* objects;
  const DevSphere* spheres;
  const DevTopTri* top_tris;
  const RtWord4* nodes;        // quantised nodes of every mesh, 2 or 3 x 16 bytes each (node_format)
  const DevMeshTri* mesh_tris;
  const DevTriFrame* mesh_frames;   // parallel to mesh_tris
  const DevMaterial* materials;
  const float* spectra;       // n_spectra x 60
  const uint32_t* mesh_objects;  // positions in `objects` of the models, ascending
  uint32_t n_meshes;
  uint32_t n_objects;
  uint32_t hit_rule;
  uint32_t stack_entries;     // LDS traversal stack depth per lane
  uint32_t n_spectra;
  uint32_t lds_nodes;         // > 0: all BVH nodes of the scene (this many 16-byte words) are staged into LDS by every workgroup
  uint32_t node_format;       // RT_NODES_*
  uint32_t extension_materials;  // some material is glossy or dielectric: kernels with that shading code are used
  uint32_t n_nodes;              // nodes of all meshes together, in nodes of node_format (bounds every node index)
  uint32_t pad0;
};

struct DevCamera {
  double focus[3], p1[3], p2[3], p3[3];
  uint32_t width, height;
};
