// rtmi_api.cpp -- host runtime behind the C ABI in include/rtmi.h: argument checking, scene
// flattening and upload, batch scheduling of the per-bounce kernels, statistics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtmi.h"
#include "bvh.h"
#include "lbvh.h"
#include "rt_kernels.h"
#include "rt_types.h"
#include "spectrum.h"

using namespace rtmi;

// the public structs and the device-side ones describe the same bytes
static_assert(RTMI_NBINS == RT_NBINS && RTMI_MAX_DEPTH == RT_MAX_DEPTH, "constants out of sync");
static_assert(sizeof(rtmi_hit) == 64 && sizeof(rtmi_ray) == 48, "ABI layout");

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

}  // namespace

namespace rtmi {
// for the other translation units of the library: sets the calling thread's rtmi_last_error() and returns `code`
int set_last_error(int code, const char* message) { g_error = message; return code; }
}

namespace {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? RTMI_OOM : RTMI_HIP_ERROR, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Grow-only device buffer.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return RTMI_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return RTMI_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Per-render counters, cleared when a render starts; `overflow` (last) is sticky: it is cleared only when the block is
// allocated, so a render that dropped paths is still reported when the statistics of a later render are collected.
struct DeviceStats { unsigned long long bounce_segments; TraversalCounters trav; unsigned long long mesh_rays; uint32_t overflow, pad; };
constexpr size_t kPerRenderStatsBytes = offsetof(DeviceStats, overflow);

}  // namespace

// Scratch of rtmi_render_accumulate_noalloc: the workspace of one batch of at most kStaticBatchPaths path slots (both path
// queues with the room for paths longer than 16 hits, the level-0 traversal list, the emit queue, the per-slot
// contributions, the fill counters, the device statistics) and the tile list of one render. Every GPU the library runs
// on holds its own copy of this array, at its own address, for the life of the process.
constexpr size_t kStaticBatchPaths = size_t(1) << 17;
constexpr uint32_t kStaticMaxSamplesPerBatch = 2 * kShards;     // keeps the queues' chunk slack at its default (region_chunks)
constexpr size_t kStaticMaxTiles = size_t(1) << 18;             // tiles of one call's share (4K at 16 x 16: 32 400)
constexpr size_t kStaticScratchBytes = size_t(64) << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

// This device's address of the scratch array (the array's name is only a host-side handle for these lookups).
int static_scratch_address(unsigned char** out) {
  void* p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!p) return fail(RTMI_HIP_ERROR, "hipGetSymbolAddress returned a null address for the static scratch array");
  *out = static_cast<unsigned char*>(p);
  return RTMI_OK;
}

}  // namespace

// Device memory one batch of launches works in, beyond the caller's inputs and outputs. A context owns one that grows
// on demand (ensure_workspace); rtmi_render_accumulate_noalloc uses one carved out of a static device array.
struct Workspace {
  PathQueue queue[2]{};
  size_t queue_entries = 0;       // entries of each path queue (kShards regions of queue_entries / kShards)
  bool queue_deep = false;
  uint32_t* mesh_list = nullptr;  // queue_entries positions
  EmitQueue emit{};
  uint32_t emit_region_entries = 0;
  size_t emit_paths = 0;
  bool emit_deep = false;
  float* contrib = nullptr;       // 3 floats per path slot
  uint32_t* counts = nullptr;     // kCountBytes
  DeviceStats* stats = nullptr;
  uint32_t* tiles = nullptr;      // tile ids of the current render
};

constexpr int kMaxLanes = 4;

// A workspace with the buffers it is carved from.
struct WorkspaceMem {
  Workspace ws;
  DevBuf queue_mem[2], contrib, counts, stats, emit_mem, mesh_list;
  void release_queues() {
    for (DevBuf* b : {&queue_mem[0], &queue_mem[1], &contrib, &emit_mem, &mesh_list}) b->release();
    ws.queue_entries = 0; ws.queue_deep = false; ws.emit_paths = 0; ws.emit_deep = false;
    ws.queue[0] = ws.queue[1] = PathQueue{}; ws.emit = EmitQueue{}; ws.emit_region_entries = 0;
    ws.mesh_list = nullptr; ws.contrib = nullptr;
  }
  void release_all() { release_queues(); counts.release(); stats.release(); ws.counts = nullptr; ws.stats = nullptr; }
};

struct rtmi_ctx {
  int device = 0;
  uint64_t max_batch_paths = 0;
  // Several workspaces (one call at a time per context): the batches of a render take turns in them, each on an internal
  // stream of its own, so that the thin tail of one batch's launches -- a handful of long traversals keeping a few
  // waves busy -- overlaps with the bulk of another batch's. Only the workspaces a render uses are allocated.
  WorkspaceMem wsm[kMaxLanes];
  hipStream_t lane[kMaxLanes] = {};
  hipEvent_t ev_fork = nullptr, ev_done[kMaxLanes] = {}, ev_resolved[kMaxLanes] = {};
  // Batches of one render overlap on this many internal lanes. RTMI_LANES (read once, 1..kMaxLanes) sets both; 1 = never overlap.
  int n_lanes = 2;                // a render that fits one batch is cut into this many (measured: 2 beats 3 by 1.5 %)
  int n_lanes_many = 3;           // a render of several batches alternates them over this many (3 beats 2 by 4.5 % on 32 batches)
  DevBuf tiles, rays, hits, spectra_out, image;
  bool batch_is_default = true;   // max_batch_paths was not chosen by the caller: it may shrink if memory is short
  bool inline_emit = false;       // RTMI_INLINE_EMIT (read once, when the context is created): no emit queue, for A/B measurements
  // Fault injection for the tests of the error paths (read once, when the context is created): RTMI_TEST_FAIL_BATCH=k makes
  // the k-th batch (from 0) of every render fail after the batches before it were launched; RTMI_TEST_OOM_LANE=w makes the
  // workspace of internal lane w (and above) report out-of-memory.
  int test_fail_batch = -1, test_oom_lane = -1;
  unsigned char* scratch = nullptr;  // this device's copy of the static scratch array (rtmi_render_accumulate_noalloc)
  void release_workspace() { for (auto& w : wsm) w.release_queues(); }
};

struct rtmi_scene {
  rtmi_ctx* ctx = nullptr;
  DevScene dev{};
  DevBuf objects, spheres, top_tris, nodes, mesh_tris, mesh_frames, materials, spectra, mesh_objects;
  rtmi_scene_info info{};
  // bookkeeping of the last render_accumulate, completed by rtmi_render_collect_stats
  rtmi_stats pending{};
  bool counting = false;
  DeviceStats* last_stats[kMaxLanes] = {};   // device counters of the last render_accumulate (one per workspace it used)
  bool fused_only = false;   // RTMI_PIPELINE=fused: one kernel per bounce even with meshes (for A/B measurements)
};

namespace {

// Default number of path slots per batch of launches: 256 Mi, about 56 GB of workspace per lane (three lanes when a
// render has several batches: 170 GB of the 288 GB). A launch over a deep queue level is bound by the longest traversals
// in it, not by how many paths it holds, so the fewer and larger the batches, the fewer of those tails a render pays for:
// the 2.1 G paths of the million-triangle 4K render take 180 ms in 16 batches of 2^27, 167 ms in 8 of 2^28 and 164 ms in
// 4 of 2^29 on two lanes (1080p x 64 spp fits one batch either way). Short of memory the lanes go first, then this halves.
constexpr uint64_t kDefaultBatchPaths = 1ull << 28;

int bind(const rtmi_ctx* ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return RTMI_OK;
}

PathQueue carve_queue(void* base, size_t entries, bool deep) {
  PathQueue q;
  q.base = static_cast<char*>(base); q.entries = entries; q.deep = deep ? 1u : 0u; q.pad = 0;
  return q;
}

size_t queue_bytes(size_t entries, bool deep) { return entries * PathQueue::bytes_per_entry(deep) + 256; }
constexpr size_t kCountLevelWords = size_t(kShards) * kCountStride;            // uint32 words per queue level
// fill counters: path-queue levels 0..depth+1, the emit queue, then one mesh list per level
constexpr size_t kCountLevels = 2 * RT_MAX_DEPTH + 5;
constexpr size_t kCountBytes = kCountLevels * kCountLevelWords * sizeof(uint32_t);

size_t emit_entries_for(size_t paths) {
  // emit queue: kShards regions, each twice its fair share of the batch (a wave that finds its region full
  // evaluates in place, so this is a performance knob, not a correctness bound)
  return size_t(((2 * paths / kShards + kBlock) + kBlock - 1) / kBlock * kBlock);
}
size_t emit_bytes(size_t region_entries, bool deep) { return region_entries * kShards * (4 + 4 + 16 + (deep ? 16 : 0)) + 4 * 256; }

EmitQueue carve_emit(void* mem, size_t region_entries, bool deep) {
  const size_t entries = region_entries * kShards;
  char* base = static_cast<char*>(mem);
  auto take = [&](size_t bytes) { void* r = base; base += (bytes + 255) & ~size_t(255); return r; };
  EmitQueue e;
  e.slot = (uint32_t*)take(entries * 4); e.info = (uint32_t*)take(entries * 4);
  e.chain0 = (uint4*)take(entries * 16); e.chain1 = deep ? (uint4*)take(entries * 16) : nullptr;
  return e;
}

// slack_chunks: see region_chunks(). A failed allocation leaves the context as release_workspace() would: nothing
// that a later call could mistake for a usable buffer.
int ensure_workspace(WorkspaceMem& m, size_t paths, size_t slack_chunks, bool deep, bool want_queues) {
  Workspace& ws = m.ws;
  const size_t entries = queue_entries(paths, slack_chunks);
  if (want_queues && (entries > ws.queue_entries || (deep && !ws.queue_deep))) {
    const size_t e = std::max(entries, ws.queue_entries);
    const bool d = deep || ws.queue_deep;
    ws.queue_entries = 0; ws.queue_deep = false; ws.mesh_list = nullptr;
    ws.queue[0] = ws.queue[1] = PathQueue{};
    for (DevBuf* b : {&m.queue_mem[0], &m.queue_mem[1], &m.mesh_list}) b->release();
    for (int i = 0; i < 2; ++i) {
      if (int rc = m.queue_mem[i].reserve(queue_bytes(e, d))) return rc;
      ws.queue[i] = carve_queue(m.queue_mem[i].p, e, d);
    }
    if (int rc = m.mesh_list.reserve(e * sizeof(uint32_t))) return rc;
    ws.mesh_list = m.mesh_list.as<uint32_t>();
    ws.queue_entries = e; ws.queue_deep = d;
  }
  if (want_queues && (paths > ws.emit_paths || (deep && !ws.emit_deep))) {
    const size_t p = std::max(paths, ws.emit_paths);
    const bool d = deep || ws.emit_deep;
    const size_t region = emit_entries_for(p);
    ws.emit_paths = 0; ws.emit_deep = false; ws.emit = EmitQueue{}; ws.emit_region_entries = 0;
    m.emit_mem.release();
    if (int rc = m.emit_mem.reserve(emit_bytes(region, d))) return rc;
    ws.emit = carve_emit(m.emit_mem.p, region, d);
    ws.emit_region_entries = uint32_t(region); ws.emit_paths = p; ws.emit_deep = d;
  }
  ws.contrib = nullptr;
  if (int rc = m.contrib.reserve(paths * 3 * sizeof(float))) return rc;
  ws.contrib = m.contrib.as<float>();
  if (int rc = m.counts.reserve(kCountBytes)) return rc;
  ws.counts = m.counts.as<uint32_t>();
  if (!m.stats.p) {
    if (int rc = m.stats.reserve(sizeof(DeviceStats))) return rc;
    HIP_TRY(hipMemset(m.stats.p, 0, sizeof(DeviceStats)));
  }
  ws.stats = m.stats.as<DeviceStats>();
  return RTMI_OK;
}

bool affine_inverse(const double* m, double* out) {
  const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = (a00 * c00 + a01 * c01) + a02 * c02;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = (a02 * a21 - a01 * a22) * id; out[2] = (a01 * a12 - a02 * a11) * id;
  out[4] = c01 * id; out[5] = (a00 * a22 - a02 * a20) * id; out[6] = (a02 * a10 - a00 * a12) * id;
  out[8] = c02 * id; out[9] = (a01 * a20 - a00 * a21) * id; out[10] = (a00 * a11 - a01 * a10) * id;
  for (int i = 0; i < 3; ++i) out[i * 4 + 3] = -((out[i * 4] * m[3] + out[i * 4 + 1] * m[7]) + out[i * 4 + 2] * m[11]);
  return true;
}

// Centre, bounding radius and padded binary32 box (relative to the centre, rounded outward) of an object whose
// extent is [lo, hi]. The pad covers the rounding of the exact binary64 tests, which can accept points a few
// ulps outside the mathematical surface.
void set_object_box(DevObject& ob, const double* lo, const double* hi) {
  double extent = 0.0, r2 = 0.0;
  for (int k = 0; k < 3; ++k) {
    ob.center[k] = 0.5 * (lo[k] + hi[k]);
    const double half = 0.5 * (hi[k] - lo[k]);
    extent = std::max(extent, half);
    r2 += half * half;
  }
  const double pad = extent * 0x1.0p-20 + 1e-30;
  ob.radius = (std::sqrt(r2) + pad) * (1.0 + 0x1.0p-20);
  for (int k = 0; k < 3; ++k) {
    const double dl = (lo[k] - ob.center[k]) - pad, dh = (hi[k] - ob.center[k]) + pad;
    float l = float(dl), h = float(dh);
    if (double(l) > dl) l = std::nextafterf(l, -FLT_MAX);
    if (double(h) < dh) h = std::nextafterf(h, FLT_MAX);
    ob.box_lo[k] = l; ob.box_hi[k] = h;
  }
}

// The box ModelObj::new grows for its triangles (src/shapes/model_obj.rs:62-66): AABB::empty() is the origin point
// (src/geometry/aabb.rs:15-20), then the union with every triangle's box. Stored exactly (binary64), plus a binary32
// box relative to the object's centre that lies strictly inside it: shrunk by 2^-19 of the farthest corner's distance
// and of the mesh radius, then rounded inward.
void set_model_gate(DevObject& ob, const double* verts, uint32_t n_tris, double radius) {
  double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {0.0, 0.0, 0.0};
  for (size_t i = 0; i < size_t(n_tris) * 3; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::fmin(lo[k], verts[i * 3 + k]);
      hi[k] = std::fmax(hi[k], verts[i * 3 + k]);
    }
  double reach = radius;
  for (int k = 0; k < 3; ++k) reach = std::max(reach, std::max(std::fabs(lo[k] - ob.center[k]), std::fabs(hi[k] - ob.center[k])));
  const double shrink = reach * 0x1.0p-19 + 1e-30;
  for (int k = 0; k < 3; ++k) {
    ob.gate_lo[k] = lo[k]; ob.gate_hi[k] = hi[k];
    const double dl = (lo[k] - ob.center[k]) + shrink, dh = (hi[k] - ob.center[k]) - shrink;
    float l = float(dl), h = float(dh);
    if (double(l) < dl) l = std::nextafterf(l, FLT_MAX);
    if (double(h) > dh) h = std::nextafterf(h, -FLT_MAX);
    ob.inner_lo[k] = l; ob.inner_hi[k] = h;
  }
}

// normalize((v1-v0) x (v2-v0)), its length and its normalisation, in the reference's operation order (this file is
// compiled without contraction, so every product and sum rounds separately, as on the device)
DevTriFrame tri_frame(const double* v) {
  const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
  const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double cl = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
  const double n0[3] = {c[0] / cl, c[1] / cl, c[2] / cl};
  // the reference then maps the normal through the triangle's (identity) local->world matrix
  // (src/shapes/triangle.rs:83); the products with 1 and 0 are exact but turn a -0 component into +0
  const double n1[3] = {(1.0 * n0[0] + 0.0 * n0[1]) + 0.0 * n0[2], (0.0 * n0[0] + 1.0 * n0[1]) + 0.0 * n0[2],
                        (0.0 * n0[0] + 0.0 * n0[1]) + 1.0 * n0[2]};
  DevTriFrame f;
  f.len = std::sqrt((n1[0] * n1[0] + n1[1] * n1[1]) + n1[2] * n1[2]);
  for (int k = 0; k < 3; ++k) f.n[k] = n1[k] / f.len;
  return f;
}

template <class T>
int upload(DevBuf& buf, const std::vector<T>& v) {
  if (int rc = buf.reserve(std::max<size_t>(v.size() * sizeof(T), 256))) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RTMI_OK;
}

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// HIP-event bracket around closest-hit launches (RTMI_FLAG_TIME_KERNELS)
struct KernelClock {
  struct Span { hipEvent_t a, b; bool traverse; };
  std::vector<Span> spans;
  bool on = false;
  hipError_t error = hipSuccess; // the first failure of an event call; the render reports it
  hipStream_t stream = nullptr;
  double traverse_ms = 0.0;      // of the spans drained so far
  uint32_t traverse_launches = 0;
  bool open = false;             // begin() recorded a span that end() has to close
  void note(hipError_t e) { if (e != hipSuccess && error == hipSuccess) error = e; }
  void begin(bool traverse = false) {
    if (traverse) ++traverse_launches;
    open = false;
    if (!on || error != hipSuccess) return;
    hipEvent_t a = nullptr, b = nullptr;
    note(hipEventCreate(&a));
    if (error == hipSuccess) note(hipEventCreate(&b));
    if (error == hipSuccess) note(hipEventRecord(a, stream));
    if (error != hipSuccess) {     // no half-made span is kept
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      return;
    }
    spans.push_back(Span{a, b, traverse});
    open = true;
  }
  void end() { if (open) note(hipEventRecord(spans.back().b, stream)); open = false; }
  double drain() {  // waits for the recorded work
    double total = 0.0;
    for (auto& s : spans) {
      note(hipEventSynchronize(s.b));
      float ms = 0.f;
      note(hipEventElapsedTime(&ms, s.a, s.b));
      total += ms;
      if (s.traverse) traverse_ms += ms;
      (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
    }
    spans.clear();
    return total;
  }
};

DevCamera to_dev_camera(const rtmi_camera& c) {
  DevCamera d;
  std::memcpy(d.focus, c.focus, sizeof d.focus); std::memcpy(d.p1, c.p1, sizeof d.p1);
  std::memcpy(d.p2, c.p2, sizeof d.p2); std::memcpy(d.p3, c.p3, sizeof d.p3);
  d.width = c.width; d.height = c.height;
  return d;
}

// Runs the launches of one batch in workspace `ws` and leaves the per-slot contributions in ws.contrib.
int run_batch(rtmi_scene* sc, const Workspace& ws, const BatchDesc& b, size_t n_paths, float* spectra_out, hipStream_t stream,
              KernelClock& clock, uint32_t& launches) {
  uint32_t* counts = ws.counts;
  float* contrib = ws.contrib;
  TraceOutputs o{contrib, contrib + n_paths, contrib + 2 * n_paths, spectra_out, EmitQueue{}, nullptr, 0};
  DeviceStats* ds = ws.stats;
  TraversalCounters* counters = sc->counting ? &ds->trav : nullptr;
  const uint32_t depth = b.integrator == RT_INT_RANDOM_WALK ? b.max_depth : 0;
  // queue level k holds the paths that enter bounce k; its kShards fill counters sit at counts + k * kCountLevelWords;
  // the emit queue's counters follow the last level, the per-level mesh lists' counters follow those
  const size_t emit_level = size_t(depth) + 2, mesh_level0 = emit_level + 1;
  const bool split = sc->dev.n_meshes > 0 && !sc->fused_only;
  const size_t levels_used = split ? mesh_level0 + std::max<uint32_t>(depth, 1) + 1 : mesh_level0;
  HIP_TRY(hipMemsetAsync(counts, 0, levels_used * kCountLevelWords * sizeof(uint32_t), stream));
  const bool deep = depth > 16;
  if (depth >= 1 && ws.emit_paths >= n_paths && (!deep || ws.emit_deep) && !sc->ctx->inline_emit) {
    o.emit = ws.emit; o.emit_counts = counts + emit_level * kCountLevelWords; o.emit_region_entries = ws.emit_region_entries;
  }
  HIP_TRY(hipMemsetAsync(contrib, 0, n_paths * 3 * sizeof(float), stream));
  const uint32_t region_entries = uint32_t(ws.queue_entries / kShards);
  auto level_counts = [&](size_t k) { return counts + k * kCountLevelWords; };
  if (!split) {
    clock.begin();
    launch_primary(sc->dev, b, ws.queue[0], level_counts(1), region_entries, o, counters, stream);
    clock.end();
    ++launches;
    for (uint32_t k = 1; k < depth; ++k) {
      clock.begin();
      launch_bounce(sc->dev, b, k, ws.queue[(k - 1) & 1], level_counts(k), uint32_t(n_paths), ws.queue[k & 1],
                    level_counts(k + 1), region_entries, o, counters, stream);
      clock.end();
      ++launches;
    }
  } else {
    // level k >= 1 lives in queue[k & 1]; level 0 has no queue (queue[0] only carries the listed primary rays' directions and
    // the hit records, at the paths' own indices). Per level: traverse, then shade (which fills level k + 1).
    // level k's fill counters: mesh class at level_counts(mesh_level0 + k), the rest at level_counts(k)
    auto mesh_list = [&](size_t k) { return MeshList{k == 0 ? ws.mesh_list : nullptr, level_counts(mesh_level0 + k)}; };
    auto both = [&](size_t k) { return LevelCounts{level_counts(mesh_level0 + k), level_counts(k)}; };
    clock.begin();
    launch_generate(sc->dev, b, ws.queue[0], region_entries, mesh_list(0), stream);
    clock.end();
    ++launches;
    const uint32_t levels = std::max<uint32_t>(depth, 1);   // albedo / normal integrators: the primary hit only
    for (uint32_t k = 0; k < levels; ++k) {
      const FixedOrigin origin{b.cam.focus[0], b.cam.focus[1], b.cam.focus[2], k == 0 && !b.rays};
      clock.begin(true);
      launch_traverse(sc->dev, ws.queue[k & 1], mesh_list(k), uint32_t(n_paths), region_entries, origin, counters, stream);
      clock.end();
      clock.begin();
      launch_shade(sc->dev, b, k, ws.queue[k & 1], both(k), uint32_t(n_paths), ws.queue[(k + 1) & 1], both(k + 1), region_entries, o,
                   stream);
      clock.end();
      launches += 2;
    }
  }
  launch_emit(sc->dev, o, deep, uint32_t(n_paths), stream);
  if (depth > 1) launch_sum_counts(level_counts(1), depth - 1, &ds->bounce_segments, stream);
  if (split) {
    launch_sum_counts(level_counts(mesh_level0), std::max<uint32_t>(depth, 1), &ds->mesh_rays, stream);
    // bounce segments of the two-kernel pipeline: the mesh class of levels 1.. (the other class was summed above)
    if (depth > 1) launch_sum_counts(level_counts(mesh_level0 + 1), depth - 1, &ds->bounce_segments, stream);
  }
  if (ws.queue_entries) launch_check_overflow(counts, uint32_t(levels_used), uint32_t(mesh_level0), split ? std::max<uint32_t>(depth, 1) + 1 : 0,
                                              region_entries, &ds->overflow, stream);
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

int check_scene(const rtmi_scene* s) {
  if (!s || !s->ctx) return fail(RTMI_BAD_ARG, "scene is NULL");
  return bind(s->ctx);
}

}  // namespace

// ================================================================================ context

extern "C" const char* rtmi_last_error(void) { return g_error.c_str(); }
extern "C" const char* rtmi_version(void) { return "rtmi 0.1.0 gfx950"; }

extern "C" int rtmi_ctx_create(int device_id, uint64_t max_batch_paths, rtmi_ctx** out) {
  if (!out) return fail(RTMI_BAD_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(RTMI_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= n) return fail(RTMI_NO_DEVICE, "device %d requested, %d visible", dev, n);
  HIP_TRY(hipSetDevice(dev));
  // a failure below destroys what was created so far (streams, events), not just the struct
  std::unique_ptr<rtmi_ctx, void (*)(rtmi_ctx*)> ctx(new rtmi_ctx, rtmi_ctx_destroy);
  ctx->device = dev;
  ctx->max_batch_paths = max_batch_paths ? max_batch_paths : kDefaultBatchPaths;
  ctx->batch_is_default = max_batch_paths == 0;
  if (ctx->max_batch_paths < 256) ctx->max_batch_paths = 256;
  if (ctx->max_batch_paths > (1ull << 30)) ctx->max_batch_paths = 1ull << 30;
  ctx->inline_emit = getenv("RTMI_INLINE_EMIT") != nullptr;
  if (const char* e = getenv("RTMI_TEST_FAIL_BATCH")) ctx->test_fail_batch = atoi(e);
  if (const char* e = getenv("RTMI_TEST_OOM_LANE")) ctx->test_oom_lane = atoi(e);
  HIP_TRY(upload_constants(spectrum_tables()));
  if (int rc = static_scratch_address(&ctx->scratch)) return rc;
  if (const char* e = getenv("RTMI_LANES")) ctx->n_lanes = ctx->n_lanes_many = std::min(std::max(atoi(e), 1), kMaxLanes);
  for (int i = 0; i < std::max(ctx->n_lanes, ctx->n_lanes_many); ++i) {
    HIP_TRY(hipStreamCreateWithFlags(&ctx->lane[i], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_resolved[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  *out = ctx.release();
  return RTMI_OK;
}

extern "C" void rtmi_ctx_destroy(rtmi_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& w : ctx->wsm) w.release_all();
  for (DevBuf* b : {&ctx->tiles, &ctx->rays, &ctx->hits, &ctx->spectra_out, &ctx->image}) b->release();
  for (hipStream_t st : ctx->lane) if (st) (void)hipStreamDestroy(st);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  for (hipEvent_t e : ctx->ev_done) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_resolved) if (e) (void)hipEventDestroy(e);
  delete ctx;
}

// ================================================================================ spectra, camera

extern "C" int rtmi_spectrum_from_rgb(float r, float g, float b, int type, float out[RTMI_NBINS]) {
  if (!out) return fail(RTMI_BAD_ARG, "out is NULL");
  if (type != RTMI_SPEC_REFLECTANCE && type != RTMI_SPEC_ILLUMINANT) return fail(RTMI_BAD_ARG, "unknown spectrum type %d", type);
  spectrum_from_rgb(r, g, b, type, out);
  return RTMI_OK;
}

extern "C" int rtmi_spectrum_to_xyz(const float spectrum[RTMI_NBINS], float xyz[3]) {
  if (!spectrum || !xyz) return fail(RTMI_BAD_ARG, "NULL argument");
  spectrum_to_xyz(spectrum, xyz);
  return RTMI_OK;
}

extern "C" int rtmi_xyz_to_rgb(const float xyz[3], float rgb[3]) {
  if (!xyz || !rgb) return fail(RTMI_BAD_ARG, "NULL argument");
  xyz_to_rgb(xyz, rgb);
  return RTMI_OK;
}

extern "C" int rtmi_spectrum_cie_tables(float x[RTMI_NBINS], float y[RTMI_NBINS], float z[RTMI_NBINS]) {
  if (!x || !y || !z) return fail(RTMI_BAD_ARG, "NULL argument");
  const SpectrumTables& t = spectrum_tables();
  std::memcpy(x, t.cie[0], sizeof t.cie[0]); std::memcpy(y, t.cie[1], sizeof t.cie[1]); std::memcpy(z, t.cie[2], sizeof t.cie[2]);
  return RTMI_OK;
}

// Camera::new (src/camera.rs:19-51): screen corners from focal distance, horizontal field of
// view in degrees and Rz(yaw) * Ry(pitch) * Rx(roll); the camera looks along local +X.
extern "C" int rtmi_camera_new(const double focus[3], double focal_distance, double field_of_view_deg,
                               const double rotation_rpy[3], uint32_t width, uint32_t height, rtmi_camera* out) {
  if (!focus || !rotation_rpy || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  if (width == 0 || height == 0) return fail(RTMI_BAD_ARG, "camera resolution %ux%u", width, height);
  const double dy = focal_distance * std::tan(field_of_view_deg / 2.0 * M_PI / 180.0);
  const double dz = dy * double(height) / double(width);
  const double sr = std::sin(rotation_rpy[0]), cr = std::cos(rotation_rpy[0]);
  const double sp = std::sin(rotation_rpy[1]), cp = std::cos(rotation_rpy[1]);
  const double sy = std::sin(rotation_rpy[2]), cy = std::cos(rotation_rpy[2]);
  const double rot[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                         sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                         -sp,     cp * sr,                cp * cr};
  auto corner = [&](double y, double z, double* dst) {
    const double v[3] = {focal_distance, y, z};
    for (int i = 0; i < 3; ++i) dst[i] = ((rot[i * 3] * v[0] + rot[i * 3 + 1] * v[1]) + rot[i * 3 + 2] * v[2]) + focus[i];
  };
  corner(-dy, dz, out->p1); corner(dy, dz, out->p2); corner(-dy, -dz, out->p3); corner(dy, -dz, out->p4);
  std::memcpy(out->focus, focus, 3 * sizeof(double));
  out->width = width; out->height = height;
  return RTMI_OK;
}

extern "C" int rtmi_camera_get_ray(const rtmi_camera* c, double i, double j, rtmi_ray* out) {
  if (!c || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  double d[3];
  for (int k = 0; k < 3; ++k) {
    const double right = (c->p2[k] - c->p1[k]) * i / double(c->width);
    const double down = (c->p3[k] - c->p1[k]) * j / double(c->height);
    d[k] = ((c->p1[k] + right) + down) - c->focus[k];
  }
  const double len = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  for (int k = 0; k < 3; ++k) { out->o[k] = c->focus[k]; out->d[k] = d[k] / len; }
  return RTMI_OK;
}

// ================================================================================ scene

extern "C" int rtmi_scene_create(rtmi_ctx* ctx, const rtmi_scene_desc* d, rtmi_scene** out) {
  if (!ctx || !d || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (int rc = bind(ctx)) return rc;
  if (d->hit_rule != RTMI_HIT_FIRST_INDEX && d->hit_rule != RTMI_HIT_NEAREST) return fail(RTMI_BAD_ARG, "unknown hit rule %u", d->hit_rule);
  if (d->flags & ~uint32_t(RTMI_SCENE_GPU_BVH)) return fail(RTMI_BAD_ARG, "unknown scene flags %#x", d->flags);
  if ((d->n_objects && !d->objects) || (d->n_spheres && !d->spheres) || (d->n_triangles && !d->triangles) ||
      (d->n_materials && !d->materials))
    return fail(RTMI_BAD_ARG, "a non-empty array is NULL");

  Timer create_timer;
  double bvh_build_ms = 0.0;
  std::vector<DevMaterial> mats(d->n_materials);
  bool extension_materials = false;
  std::vector<float> spectra;
  auto add_row = [&](const float* s) { spectra.insert(spectra.end(), s, s + RT_NBINS); return uint32_t(spectra.size() / RT_NBINS - 1); };
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    const rtmi_material& m = d->materials[i];
    if (m.type > RTMI_MAT_DIELECTRIC) return fail(RTMI_UNSUPPORTED, "material %u has unknown type %u", i, m.type);
    DevMaterial dm{};
    dm.type = m.type; dm.present = m.present & 15u;
    dm.ns = m.specular_coefficient; dm.ni = m.optical_density;
    if (m.type == RTMI_MAT_GLOSSY || m.type == RTMI_MAT_DIELECTRIC) extension_materials = true;
    dm.ambient = dm.emitted = dm.diffused = dm.specular = RT_NO_SPECTRUM;
    uint32_t rows[4] = {RT_NO_SPECTRUM, RT_NO_SPECTRUM, RT_NO_SPECTRUM, RT_NO_SPECTRUM};
    if (m.present & RTMI_HAS_AMBIENT) rows[0] = add_row(m.ambient);
    if (m.present & RTMI_HAS_EMITTED) rows[1] = add_row(m.emitted);
    if (m.present & RTMI_HAS_DIFFUSED) rows[2] = add_row(m.diffused);
    if (m.present & RTMI_HAS_SPECULAR) rows[3] = add_row(m.specular);
    if (spectra.size() / RT_NBINS >= RT_NO_SPECTRUM)
      return fail(RTMI_UNSUPPORTED, "more than %d distinct material spectra", int(RT_NO_SPECTRUM) - 1);
    dm.ambient = uint8_t(rows[0]); dm.emitted = uint8_t(rows[1]); dm.diffused = uint8_t(rows[2]); dm.specular = uint8_t(rows[3]);
    mats[i] = dm;
  }

  std::vector<DevObject> objects(d->n_objects);
  std::vector<DevSphere> spheres;
  std::vector<DevTopTri> top_tris;
  std::vector<RtWord4> nodes;        // the quantised nodes of every model
  uint32_t n_nodes = 0, largest_mesh = 0;
  constexpr uint32_t kLeafSize = 4;  // the 16-bit leaf references of RT_NODES_SMALL count up to 4
  for (uint32_t i = 0; i < d->n_objects; ++i)
    if (d->objects[i].kind == RTMI_OBJ_MODEL) largest_mesh = std::max(largest_mesh, d->objects[i].count);
  // One node format per scene (the kernels are instantiated per format): the 32-byte form with 16-bit references when
  // every mesh is small enough, else 32 bytes without minimum indices under the nearest rule, else 48 bytes.
  const bool gpu_bvh = (d->flags & RTMI_SCENE_GPU_BVH) != 0;
  int node_format = largest_mesh <= RT_SMALL_MAX_TRIS ? RT_NODES_SMALL
                  : d->hit_rule == RTMI_HIT_NEAREST ? RT_NODES_NEAREST : RT_NODES_WIDE;
  // RTMI_NODE_FORMAT=quad (nearest rule, host-built trees): four-wide 64-byte nodes whatever the mesh size. Measured on the
  // million-triangle mesh (4K, 16 spp): half the box-test steps per ray (12.8 against 24.0) but the traversal launches get
  // no shorter -- 0.64 ms against 0.60, because the deeper stacks (48 references per lane) leave room for three workgroups
  // per CU instead of five, and 0.57 ms at best with the stacks cut to 22 entries (a timing probe with wrong images). The
  // walk does not wait on the length of its chain of node fetches alone, so the format stays an option, not the default.
  if (const char* e = getenv("RTMI_NODE_FORMAT"))
    if (d->hit_rule == RTMI_HIT_NEAREST && !gpu_bvh && std::strcmp(e, "quad") == 0) node_format = RT_NODES_QUAD;
  double bvh_device_ms = 0.0;
  std::vector<DevMeshTri> mesh_tris;
  std::vector<DevTriFrame> mesh_frames;
  std::vector<uint32_t> mesh_objects;
  uint32_t max_depth = 0, max_stack = 0;
  for (uint32_t i = 0; i < d->n_objects; ++i) {
    const rtmi_object& o = d->objects[i];
    if (o.material >= d->n_materials) return fail(RTMI_BAD_ARG, "object %u: material %u out of range", i, o.material);
    DevObject dobj{};
    dobj.kind = o.kind; dobj.material = o.material; dobj.count = o.count;
    if (o.kind == RTMI_OBJ_SPHERE) {
      if (o.first >= d->n_spheres) return fail(RTMI_BAD_ARG, "object %u: sphere %u out of range", i, o.first);
      DevSphere s{};
      std::memcpy(s.l2w, d->spheres[o.first].l2w, sizeof s.l2w);
      s.radius = d->spheres[o.first].radius;
      if (!affine_inverse(s.l2w, s.w2l)) return fail(RTMI_BAD_ARG, "object %u: sphere transform is singular", i);
      s.translation_only = 1;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          const double want = r == c ? 1.0 : 0.0;
          // +0.0 exactly: a -0.0 entry would not add like a +0.0 one
          if (s.l2w[r * 4 + c] != want || s.w2l[r * 4 + c] != want || std::signbit(s.l2w[r * 4 + c]) || std::signbit(s.w2l[r * 4 + c]))
            s.translation_only = 0;
        }
      dobj.first = uint32_t(spheres.size());
      spheres.push_back(s);
      // image of the radius-r sphere under the linear part: half extent along axis k = r * |row k|
      double lo[3], hi[3];
      for (int k = 0; k < 3; ++k) {
        const double* row = s.l2w + 4 * k;
        const double h = std::fabs(s.radius) * std::sqrt((row[0] * row[0] + row[1] * row[1]) + row[2] * row[2]) * (1.0 + 0x1.0p-40);
        lo[k] = row[3] - h; hi[k] = row[3] + h;
      }
      set_object_box(dobj, lo, hi);
    } else if (o.kind == RTMI_OBJ_TRIANGLE) {
      if (o.first >= d->n_triangles) return fail(RTMI_BAD_ARG, "object %u: triangle %u out of range", i, o.first);
      DevTopTri t{};
      std::memcpy(t.v, d->triangles[o.first].v, sizeof t.v);
      const double* v = t.v;
      t.frame = tri_frame(v);
      dobj.first = uint32_t(top_tris.size());
      top_tris.push_back(t);
      double lo[3], hi[3];
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(v[k], std::min(v[3 + k], v[6 + k]));
        hi[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
      }
      set_object_box(dobj, lo, hi);
    } else if (o.kind == RTMI_OBJ_MODEL) {
      if (uint64_t(o.first) + o.count > d->n_triangles) return fail(RTMI_BAD_ARG, "object %u: triangles [%u,+%u) out of range", i, o.first, o.count);
      if (o.count >= (1u << 27)) return fail(RTMI_UNSUPPORTED, "object %u: %u triangles in one mesh", i, o.count);
      Timer bvh_timer;
      dobj.root = n_nodes;          // in nodes of the scene's format
      dobj.first = uint32_t(mesh_tris.size());
      const double* verts = o.count ? d->triangles[o.first].v : nullptr;
      if (gpu_bvh && o.count >= 2) {
        // Morton order + radix tree, built on the device (lbvh.hip); the same records come back as the host builder makes
        LbvhOut lb;
        std::string why;
        const hipError_t e = build_lbvh(verts, o.count, node_format, &lb, &why);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RTMI_OOM : RTMI_HIP_ERROR, "object %u: device BVH build: %s", i, why.c_str());
        std::memcpy(dobj.center, lb.center, sizeof dobj.center);
        dobj.radius = lb.radius;
        for (int k = 0; k < 3; ++k) {
          dobj.box_lo[k] = lb.box_lo[k]; dobj.box_hi[k] = lb.box_hi[k];
          dobj.grid_lo[k] = lb.grid_lo[k]; dobj.grid_step[k] = lb.grid_step[k];
        }
        n_nodes += lb.n_nodes;
        nodes.insert(nodes.end(), lb.words.begin(), lb.words.end());
        mesh_tris.insert(mesh_tris.end(), lb.tris.begin(), lb.tris.end());
        mesh_frames.insert(mesh_frames.end(), lb.frames.begin(), lb.frames.end());
        max_depth = std::max(max_depth, lb.max_depth);
        bvh_device_ms += lb.device_ms;
      } else {
        BvhBuild bb = build_bvh(verts, o.count, kLeafSize);
        std::memcpy(dobj.center, bb.center, sizeof dobj.center);
        dobj.radius = bb.radius;
        for (int k = 0; k < 3; ++k) {   // union of the root's two child boxes, which are already padded and rounded outward
          const BvhNode& r = bb.nodes[0];
          dobj.box_lo[k] = std::min(r.lo0[k], r.lo1[k]);
          dobj.box_hi[k] = std::max(r.hi0[k], r.hi1[k]);
        }
        if (node_format == RT_NODES_QUAD) {
          const QuadNodes q = quantize_nodes_quad(bb);
          if (!verify_quad_nodes(bb, q)) return fail(RTMI_INTERNAL, "object %u: the four-wide tree failed its consistency check", i);
          for (int k = 0; k < 3; ++k) { dobj.grid_lo[k] = q.grid_lo[k]; dobj.grid_step[k] = q.grid_step[k]; }
          n_nodes += q.n_nodes;
          nodes.insert(nodes.end(), q.words.begin(), q.words.end());
          max_stack = std::max(max_stack, q.max_stack + 1);       // the most references a walk can have waiting, plus one
          max_depth = std::max(max_depth, q.max_depth);
        } else {
          const QuantizedNodes q = quantize_nodes(bb, node_format);
          for (int k = 0; k < 3; ++k) { dobj.grid_lo[k] = q.grid_lo[k]; dobj.grid_step[k] = q.grid_step[k]; }
          n_nodes += uint32_t(bb.nodes.size());
          nodes.insert(nodes.end(), q.words.begin(), q.words.end());
          max_depth = std::max(max_depth, bb.max_depth);
        }
        mesh_tris.insert(mesh_tris.end(), bb.tris.begin(), bb.tris.end());
        for (const DevMeshTri& t : bb.tris) mesh_frames.push_back(tri_frame(t.v));
      }
      set_model_gate(dobj, verts, o.count, dobj.radius);
      mesh_objects.push_back(i);
      bvh_build_ms += bvh_timer.ms();
    } else {
      return fail(RTMI_UNSUPPORTED, "object %u has unknown kind %u", i, o.kind);
    }
    objects[i] = dobj;
  }
  const uint32_t stack_entries = n_nodes == 0 ? 0 : std::max(max_depth + 1, max_stack);
  const size_t stack_lds = size_t(stack_entries) * kBlock * sizeof(uint32_t);
  if (stack_lds > 150 * 1024) return fail(RTMI_UNSUPPORTED, "BVH depth %u needs more LDS than a CU has", max_depth);
  // Stage the whole tree into LDS only while three workgroups (nodes + stacks each) still fit in a CU's 160 KB.
  const size_t node_bytes = nodes.size() * sizeof(RtWord4);
  const bool stage = !nodes.empty() && 3 * (node_bytes + stack_lds) + 3 * 1024 <= 160 * 1024 && !getenv("RTMI_NO_LDS_NODES");

  std::unique_ptr<rtmi_scene> s(new rtmi_scene);
  s->ctx = ctx;
  int rc;
  if ((rc = upload(s->objects, objects)) || (rc = upload(s->spheres, spheres)) || (rc = upload(s->top_tris, top_tris)) ||
      (rc = upload(s->nodes, nodes)) || (rc = upload(s->mesh_tris, mesh_tris)) || (rc = upload(s->mesh_frames, mesh_frames)) ||
      (rc = upload(s->materials, mats)) || (rc = upload(s->mesh_objects, mesh_objects)) ||
      (rc = upload(s->spectra, spectra))) {
    rtmi_scene_destroy(s.release());
    return rc;
  }
  s->dev.objects = s->objects.as<DevObject>(); s->dev.spheres = s->spheres.as<DevSphere>();
  s->dev.top_tris = s->top_tris.as<DevTopTri>(); s->dev.nodes = s->nodes.as<RtWord4>();
  s->dev.mesh_tris = s->mesh_tris.as<DevMeshTri>(); s->dev.mesh_frames = s->mesh_frames.as<DevTriFrame>();
  s->dev.materials = s->materials.as<DevMaterial>();
  s->dev.spectra = s->spectra.as<float>();
  s->dev.mesh_objects = s->mesh_objects.as<uint32_t>(); s->dev.n_meshes = uint32_t(mesh_objects.size());
  s->dev.extension_materials = extension_materials ? 1u : 0u;
  if (const char* p = getenv("RTMI_PIPELINE")) s->fused_only = std::strcmp(p, "fused") == 0;
  s->dev.n_objects = d->n_objects; s->dev.hit_rule = d->hit_rule; s->dev.stack_entries = stack_entries;
  s->dev.n_spectra = uint32_t(spectra.size() / RT_NBINS);
  s->dev.lds_nodes = stage ? uint32_t(nodes.size()) : 0;
  s->dev.node_format = uint32_t(node_format);
  s->dev.n_nodes = n_nodes;
  s->info.n_nodes = n_nodes; s->info.max_depth = max_depth;
  s->info.n_mesh_triangles = uint32_t(mesh_tris.size()); s->info.n_spectra = s->dev.n_spectra;
  s->info.node_bytes = nodes.size() * sizeof(RtWord4); s->info.triangle_bytes = mesh_tris.size() * (sizeof(DevMeshTri) + sizeof(DevTriFrame));
  s->info.bvh_device_ms = bvh_device_ms;
  s->info.bvh_build_ms = bvh_build_ms; s->info.upload_ms = create_timer.ms() - bvh_build_ms;
  *out = s.release();
  return RTMI_OK;
}

extern "C" void rtmi_scene_destroy(rtmi_scene* s) {
  if (!s) return;
  if (s->ctx) { (void)hipSetDevice(s->ctx->device); (void)hipDeviceSynchronize(); }
  for (DevBuf* b : {&s->objects, &s->spheres, &s->top_tris, &s->nodes, &s->mesh_tris, &s->mesh_frames, &s->materials, &s->spectra, &s->mesh_objects})
    b->release();
  delete s;
}

extern "C" int rtmi_scene_get_info(const rtmi_scene* s, rtmi_scene_info* out) {
  if (!s || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  *out = s->info;
  return RTMI_OK;
}

// ================================================================================ intersect

extern "C" int rtmi_intersect_batch_device(rtmi_scene* s, const rtmi_ray* dev_rays, size_t n, rtmi_hit* dev_hits, void* stream) {
  if (int rc = check_scene(s)) return rc;
  if (n && (!dev_rays || !dev_hits)) return fail(RTMI_BAD_ARG, "NULL ray or hit buffer");
  launch_intersect(s->dev, dev_rays, n, dev_hits, nullptr, static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

extern "C" int rtmi_intersect_batch(rtmi_scene* s, const rtmi_ray* rays, size_t n, rtmi_hit* hits, rtmi_stats* stats) {
  if (int rc = check_scene(s)) return rc;
  if (n && (!rays || !hits)) return fail(RTMI_BAD_ARG, "NULL ray or hit buffer");
  rtmi_ctx* ctx = s->ctx;
  Timer timer;
  const size_t chunk = std::min<size_t>(std::max<size_t>(n, 1), size_t(1) << 24);
  if (int rc = ctx->rays.reserve(chunk * sizeof(rtmi_ray))) return rc;
  if (int rc = ctx->hits.reserve(chunk * sizeof(rtmi_hit))) return rc;
  if (int rc = ensure_workspace(ctx->wsm[0], 1, 0, false, false)) return rc;
  DeviceStats* ds = ctx->wsm[0].ws.stats;
  const bool count = stats != nullptr;
  if (count) HIP_TRY(hipMemset(ds, 0, kPerRenderStatsBytes));
  for (size_t at = 0; at < n; at += chunk) {
    const size_t m = std::min(chunk, n - at);
    HIP_TRY(hipMemcpy(ctx->rays.p, rays + at, m * sizeof(rtmi_ray), hipMemcpyHostToDevice));
    launch_intersect(s->dev, ctx->rays.p, m, ctx->hits.p, count ? &ds->trav : nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits + at, ctx->hits.p, m * sizeof(rtmi_hit), hipMemcpyDeviceToHost));
  }
  if (stats) {
    DeviceStats h;
    HIP_TRY(hipMemcpy(&h, ds, sizeof h, hipMemcpyDeviceToHost));
    std::memset(stats, 0, sizeof *stats);
    stats->segments = n; stats->paths = n;
    stats->nodes_visited = h.trav.nodes; stats->prims_tested = h.trav.prims; stats->mesh_rays = h.mesh_rays;
    if (getenv("RTMI_VERBOSE") && h.trav.node_slots)
      std::fprintf(stderr, "rtmi: traversal lane occupancy: node loop %.1f %% (%llu steps), triangle loop %.1f %% (%llu tests), "
                   "service block %.1f %% (%llu lanes)\n", 100.0 * h.trav.nodes / h.trav.node_slots, h.trav.nodes,
                   100.0 * h.trav.prims / std::max(1ull, h.trav.prim_slots), h.trav.prims,
                   100.0 * h.trav.service_lanes / std::max(1ull, h.trav.service_slots), h.trav.service_lanes);
    stats->ms_total = timer.ms();
  }
  return RTMI_OK;
}

// ================================================================================ trace

extern "C" int rtmi_trace_batch(rtmi_scene* s, const rtmi_ray* rays, size_t n, uint32_t integrator, uint32_t max_depth,
                                uint64_t seed, uint32_t pixel_base, uint32_t sample_index, float* spectra, rtmi_stats* stats) {
  if (int rc = check_scene(s)) return rc;
  if (n && (!rays || !spectra)) return fail(RTMI_BAD_ARG, "NULL ray or output buffer");
  if (integrator > RTMI_INT_NORMAL) return fail(RTMI_BAD_ARG, "unknown integrator %u", integrator);
  if (max_depth > RTMI_MAX_DEPTH) return fail(RTMI_UNSUPPORTED, "max_depth %u > %d", max_depth, RTMI_MAX_DEPTH);
  rtmi_ctx* ctx = s->ctx;
  Timer timer;
  const size_t chunk = std::min<size_t>(std::max<size_t>(n, 1), std::min<uint64_t>(ctx->max_batch_paths, 1u << 22));
  const bool deep = max_depth > 16;
  if (int rc = ensure_workspace(ctx->wsm[0], chunk, 0, deep, true)) return rc;
  if (int rc = ctx->rays.reserve(chunk * sizeof(rtmi_ray))) return rc;
  if (int rc = ctx->spectra_out.reserve(chunk * RT_NBINS * sizeof(float))) return rc;
  DeviceStats* ds = ctx->wsm[0].ws.stats;
  HIP_TRY(hipMemset(ds, 0, kPerRenderStatsBytes));
  s->counting = false;
  for (auto& b : s->last_stats) b = nullptr;
  s->last_stats[0] = ds;
  KernelClock clock;
  uint32_t launches = 0, batches = 0;
  for (size_t at = 0; at < n; at += chunk) {
    const size_t m = std::min(chunk, n - at);
    HIP_TRY(hipMemcpy(ctx->rays.p, rays + at, m * sizeof(rtmi_ray), hipMemcpyHostToDevice));
    BatchDesc b{};
    b.integrator = integrator; b.max_depth = max_depth; b.seed = seed;
    b.sample0 = sample_index; b.n_samples = 1;
    b.rays = ctx->rays.as<double>(); b.n_rays = uint32_t(m); b.pixel_base = pixel_base + uint32_t(at);
    if (int rc = run_batch(s, ctx->wsm[0].ws, b, m, ctx->spectra_out.as<float>(), nullptr, clock, launches)) return rc;
    HIP_TRY(hipMemcpy(spectra + at * RT_NBINS, ctx->spectra_out.p, m * RT_NBINS * sizeof(float), hipMemcpyDeviceToHost));
    ++batches;
  }
  if (stats) {
    DeviceStats h;
    HIP_TRY(hipMemcpy(&h, ds, sizeof h, hipMemcpyDeviceToHost));
    std::memset(stats, 0, sizeof *stats);
    stats->segments = n + h.bounce_segments; stats->paths = n;
    stats->ms_total = timer.ms(); stats->launches = launches; stats->batches = batches;
  }
  return RTMI_OK;
}

// ================================================================================ render

namespace {

// What a render call covers, after its arguments have been checked.
struct RenderPlan {
  uint32_t T = 16, tiles_x = 0, tiles_y = 0;
  uint32_t n_tiles = 0;          // tiles of this call's share: k = tile_rank + j * tile_world, j < n_tiles
  uint32_t first = 0, stride = 1;
  uint64_t pixels = 0;           // pixels of those tiles that lie inside the image
  uint32_t s_begin = 0, s_end = 0;
  bool deep = false, two_kernel = false, want_queues = false;
};

int plan_render(rtmi_scene* s, const rtmi_camera* cam, const rtmi_render_params* p, const float* dev_xyz_sum, RenderPlan& r) {
  if (!cam || !p || !dev_xyz_sum) return fail(RTMI_BAD_ARG, "NULL argument");
  if (cam->width == 0 || cam->height == 0) return fail(RTMI_BAD_ARG, "camera resolution %ux%u", cam->width, cam->height);
  if (uint64_t(cam->width) * cam->height >= (1ull << 32)) return fail(RTMI_UNSUPPORTED, "image too large");
  if (p->integrator > RTMI_INT_NORMAL) return fail(RTMI_BAD_ARG, "unknown integrator %u", p->integrator);
  if (p->max_depth > RTMI_MAX_DEPTH) return fail(RTMI_UNSUPPORTED, "max_depth %u > %d", p->max_depth, RTMI_MAX_DEPTH);
  if (p->spp == 0) return fail(RTMI_BAD_ARG, "spp is 0");
  r.s_begin = p->sample_end ? p->sample_begin : 0; r.s_end = p->sample_end ? p->sample_end : p->spp;
  if (r.s_begin > r.s_end || r.s_end > p->spp) return fail(RTMI_BAD_ARG, "sample range [%u,%u) outside [0,%u)", r.s_begin, r.s_end, p->spp);
  r.T = p->tile_size ? p->tile_size : 16;
  if (r.T % kPixelBlock != 0 || r.T > 1024) return fail(RTMI_BAD_ARG, "tile_size %u must be a multiple of %u", r.T, kPixelBlock);
  if (p->tile_world && p->tile_rank >= p->tile_world) return fail(RTMI_BAD_ARG, "tile_rank %u >= tile_world %u", p->tile_rank, p->tile_world);
  r.tiles_x = (cam->width + r.T - 1) / r.T; r.tiles_y = (cam->height + r.T - 1) / r.T;
  const uint64_t all = uint64_t(r.tiles_x) * r.tiles_y;
  if (all >= (1ull << 32)) return fail(RTMI_UNSUPPORTED, "too many tiles");
  r.first = p->tile_world ? p->tile_rank : 0; r.stride = p->tile_world ? p->tile_world : 1;
  r.n_tiles = r.first < all ? uint32_t((all - r.first + r.stride - 1) / r.stride) : 0;   // 64-bit: a large stride must not wrap
  if (p->flags & RTMI_FLAG_EMPTY_SHARE) r.n_tiles = 0;
  r.pixels = 0;
  for (uint32_t j = 0; j < r.n_tiles; ++j) {
    const uint32_t k = uint32_t(r.first + uint64_t(j) * r.stride);
    const uint32_t x0 = (k / r.tiles_y) * r.T, y0 = (k % r.tiles_y) * r.T;
    r.pixels += uint64_t(std::min(r.T, cam->width - x0)) * std::min(r.T, cam->height - y0);
  }
  r.deep = p->integrator == RTMI_INT_RANDOM_WALK && p->max_depth > 16;
  // the fused kernel needs queues only when paths can continue; the two-kernel pipeline keeps even primary rays in one
  r.two_kernel = s->dev.n_meshes > 0 && !s->fused_only;
  r.want_queues = r.two_kernel || (p->integrator == RTMI_INT_RANDOM_WALK && p->max_depth > 1);
  const uint32_t n_samples = r.s_end - r.s_begin;
  s->pending = rtmi_stats{};
  s->pending.paths = r.pixels * n_samples;
  s->pending.segments = r.pixels * n_samples;  // primary rays; bounce rays are added from the device counters
  s->counting = (p->flags & RTMI_FLAG_COUNT_TRAVERSAL) != 0;
  return RTMI_OK;
}

// The batches of one render: runs of tiles x runs of samples, each at most tiles_per_batch x samples_per_batch paths;
// every batch ends by adding its contributions to the per-pixel sums. With one workspace everything runs on `stream`.
// With two, batch i runs in workspace i % 2 on the context's internal stream i % 2, while the additions to the per-pixel
// sums stay on `stream`, in batch order (so the sums do not depend on how the batches overlapped); `stream` is forked
// into the two lanes at the start and has joined them again when the last addition has been issued.
int render_batches(rtmi_scene* s, const Workspace* const* wss, int n_ws, const rtmi_camera* cam, const rtmi_render_params* p,
                   const RenderPlan& r, uint32_t tiles_per_batch, uint32_t samples_per_batch, float* dev_xyz_sum, hipStream_t stream,
                   const Timer& timer) {
  rtmi_ctx* ctx = s->ctx;
  launch_fill_tiles(wss[0]->tiles, r.n_tiles, r.first, r.stride, stream);
  for (auto& b : s->last_stats) b = nullptr;
  for (int w = 0; w < n_ws; ++w) {
    HIP_TRY(hipMemsetAsync(wss[w]->stats, 0, kPerRenderStatsBytes, stream));   // the overflow word stays as it is
    s->last_stats[w] = wss[w]->stats;
  }
  bool forked = false;
  if (n_ws > 1) {
    HIP_TRY(hipEventRecord(ctx->ev_fork, stream));
    forked = true;
    for (int w = 0; w < n_ws; ++w)
      if (hipError_t e = hipStreamWaitEvent(ctx->lane[w], ctx->ev_fork, 0)) {
        // lanes that already wait on the fork event have nothing queued behind it: nothing to join
        return fail(RTMI_HIP_ERROR, "hipStreamWaitEvent on an internal lane: %s", hipGetErrorString(e));
      }
  }
  KernelClock clock;
  clock.on = (p->flags & RTMI_FLAG_TIME_KERNELS) != 0;
  clock.stream = stream;
  uint32_t launches = 0, batches = 0;
  double kernel_ms = 0.0;
  const uint64_t per_tile = uint64_t(r.T) * r.T;
  // One batch. A failure leaves through `rc`; the caller's stream is then joined with the lanes below, so that work the
  // caller queues on `stream` after a failed render cannot overlap what the lanes still have in flight.
  auto one_batch = [&](uint32_t t0, uint32_t nt, uint32_t s0) -> int {
    const int w = n_ws > 1 ? int(batches % uint32_t(n_ws)) : 0;
    const Workspace& ws = *wss[w];
    hipStream_t lane = n_ws > 1 ? ctx->lane[w] : stream;
    BatchDesc b{};
    b.cam = to_dev_camera(*cam);
    b.tile_list = ws.tiles + t0; b.n_tiles = nt; b.tile_size = r.T; b.tiles_y = r.tiles_y;
    b.n_local_pixels = uint32_t(nt * per_tile);
    b.by_local_pixels = make_fast_div(b.n_local_pixels); b.by_tile_pixels = make_fast_div(uint32_t(per_tile));
    b.by_tiles_y = make_fast_div(r.tiles_y); b.by_blocks_per_row = make_fast_div(r.T / kPixelBlock);
    b.sample0 = s0; b.n_samples = std::min(samples_per_batch, r.s_end - s0);
    b.integrator = p->integrator; b.max_depth = p->max_depth; b.seed = p->seed;
    const size_t n_paths = size_t(b.n_local_pixels) * b.n_samples;
    if (ctx->test_fail_batch >= 0 && batches == uint32_t(ctx->test_fail_batch))
      return fail(RTMI_INTERNAL, "injected failure at batch %u (RTMI_TEST_FAIL_BATCH)", batches);
    // the workspace is free again once the batch that last used it has been added to the sums
    if (n_ws > 1 && batches >= uint32_t(n_ws)) HIP_TRY(hipStreamWaitEvent(lane, ctx->ev_resolved[w], 0));
    if (int rc = run_batch(s, ws, b, n_paths, nullptr, lane, clock, launches)) return rc;
    if (n_ws > 1) {
      HIP_TRY(hipEventRecord(ctx->ev_done[w], lane));
      HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_done[w], 0));
    }
    launch_resolve(b, TraceOutputs{ws.contrib, ws.contrib + n_paths, ws.contrib + 2 * n_paths, nullptr, EmitQueue{}, nullptr, 0}, dev_xyz_sum, stream);
    HIP_TRY(hipGetLastError());
    if (n_ws > 1) HIP_TRY(hipEventRecord(ctx->ev_resolved[w], stream));
    if (clock.on) kernel_ms += clock.drain();
    if (clock.error != hipSuccess) return fail(RTMI_HIP_ERROR, "HIP event of RTMI_FLAG_TIME_KERNELS: %s", hipGetErrorString(clock.error));
    ++batches;
    return RTMI_OK;
  };
  int rc = RTMI_OK;
  for (uint32_t t0 = 0; t0 < r.n_tiles && rc == RTMI_OK; t0 += tiles_per_batch) {
    const uint32_t nt = std::min(tiles_per_batch, r.n_tiles - t0);
    for (uint32_t s0 = r.s_begin; s0 < r.s_end && rc == RTMI_OK; s0 += samples_per_batch) rc = one_batch(t0, nt, s0);
  }
  if (rc != RTMI_OK) {
    const std::string message = g_error;     // the joins below must not replace the failure's own message
    (void)clock.drain();
    if (forked)
      for (int w = 0; w < n_ws; ++w)
        if (hipEventRecord(ctx->ev_done[w], ctx->lane[w]) == hipSuccess) (void)hipStreamWaitEvent(stream, ctx->ev_done[w], 0);
    (void)hipGetLastError();
    g_error = message;
    return rc;
  }
  s->pending.launches = launches; s->pending.batches = batches; s->pending.ms_kernels = kernel_ms;
  s->pending.ms_traverse = clock.traverse_ms; s->pending.traverse_launches = clock.traverse_launches;
  s->pending.ms_total = timer.ms();
  return RTMI_OK;
}

}  // namespace

extern "C" int rtmi_render_accumulate(rtmi_scene* s, const rtmi_camera* cam, const rtmi_render_params* p, float* dev_xyz_sum,
                                      void* stream_, rtmi_stats* stats) {
  if (int rc = check_scene(s)) return rc;
  RenderPlan r;
  if (int rc = plan_render(s, cam, p, dev_xyz_sum, r)) return rc;
  rtmi_ctx* ctx = s->ctx;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Timer timer;
  const uint32_t n_samples = r.s_end - r.s_begin;
  if (r.n_tiles == 0 || n_samples == 0) { for (auto& b : s->last_stats) b = nullptr; if (stats) *stats = s->pending; return RTMI_OK; }

  const uint64_t per_tile = uint64_t(r.T) * r.T;
  // Instrumented renders keep to one lane (their events and counters describe one launch at a time).
  const bool one_lane = (p->flags & (RTMI_FLAG_TIME_KERNELS | RTMI_FLAG_COUNT_TRAVERSAL)) != 0;
  uint32_t tiles_per_batch = 0, samples_per_batch = 0;
  int n_ws = 1;
  for (;;) {
    uint64_t cap = std::max<uint64_t>(ctx->max_batch_paths, per_tile);
    // A render that fits one batch is cut into one batch per lane when it is large enough for every part to fill the
    // chip many times over: the parts overlap where a single batch's launches would each drain to a thin tail.
    const uint64_t all_paths = uint64_t(r.n_tiles) * per_tile * n_samples;
    const int lanes = one_lane ? 1 : (all_paths > cap ? ctx->n_lanes_many : ctx->n_lanes);
    if (lanes > 1 && all_paths <= cap && all_paths >= (uint64_t(lanes) << 23)) {
      const uint32_t parts = uint32_t(lanes);
      if (n_samples >= parts) cap = uint64_t(r.n_tiles) * per_tile * ((n_samples + parts - 1) / parts);
      else if (r.n_tiles >= parts) cap = uint64_t((r.n_tiles + parts - 1) / parts) * per_tile;
    }
    tiles_per_batch = uint32_t(std::min<uint64_t>(r.n_tiles, cap / per_tile));
    samples_per_batch = uint32_t(std::min<uint64_t>(n_samples, std::max<uint64_t>(1, cap / (tiles_per_batch * per_tile))));
    const size_t batch_paths = size_t(tiles_per_batch) * per_tile * samples_per_batch;
    const uint64_t n_batches = ((r.n_tiles + tiles_per_batch - 1) / tiles_per_batch) * uint64_t((n_samples + samples_per_batch - 1) / samples_per_batch);
    n_ws = int(std::min<uint64_t>(uint64_t(lanes), n_batches));
    int rc = RTMI_OK;
    for (int w = 0; w < n_ws && rc == RTMI_OK; ++w) {
      if (ctx->test_oom_lane >= 0 && w >= ctx->test_oom_lane) rc = fail(RTMI_OOM, "injected out-of-memory for lane %d (RTMI_TEST_OOM_LANE)", w);
      else rc = ensure_workspace(ctx->wsm[w], batch_paths, samples_per_batch, r.deep, r.want_queues);
    }
    if (rc == RTMI_OK) break;
    if (rc != RTMI_OOM) return rc;
    (void)hipGetLastError();
    ctx->release_workspace();
    // Every lane holds a workspace of its own, so a context's memory is (lanes in use) x (one batch). Short of memory, the
    // overlap goes first: fewer lanes, down to one, with the batch the caller asked for ...
    if (lanes > 1) {
      ctx->n_lanes = std::min(ctx->n_lanes, lanes - 1);
      ctx->n_lanes_many = std::min(ctx->n_lanes_many, lanes - 1);
      continue;
    }
    // ... and only the library's own default batch (sized for an otherwise empty 288 GB device) is then halved: the image
    // does not depend on it. A cap the caller chose is never second-guessed.
    if (!ctx->batch_is_default || ctx->max_batch_paths <= (1ull << 20)) return rc;
    ctx->max_batch_paths >>= 1;
  }
  if (int rc = ctx->tiles.reserve(size_t(r.n_tiles) * sizeof(uint32_t))) return rc;
  const Workspace* wss[kMaxLanes];
  for (int w = 0; w < kMaxLanes; ++w) { ctx->wsm[w].ws.tiles = ctx->tiles.as<uint32_t>(); wss[w] = &ctx->wsm[w].ws; }
  if (int rc = render_batches(s, wss, n_ws, cam, p, r, tiles_per_batch, samples_per_batch, dev_xyz_sum, stream, timer)) return rc;
  if (stats) {
    if ((p->flags & RTMI_FLAG_TIME_KERNELS) || s->counting) return rtmi_render_collect_stats(s, stream_, stats);
    *stats = s->pending;
  }
  return RTMI_OK;
}

// ================================================================================ render without allocating

namespace {

// Lays the workspace out in `base` (kStaticScratchBytes, 256-byte aligned). Returns false if it does not fit.
bool carve_static_workspace(unsigned char* base, Workspace& ws) {
  size_t at = 0;
  auto take = [&](size_t bytes) { void* r = base + at; at += (bytes + 255) & ~size_t(255); return r; };
  const size_t entries = queue_entries(kStaticBatchPaths, 0);
  for (int i = 0; i < 2; ++i) ws.queue[i] = carve_queue(take(queue_bytes(entries, true)), entries, true);
  ws.queue_entries = entries; ws.queue_deep = true;
  ws.mesh_list = static_cast<uint32_t*>(take(entries * sizeof(uint32_t)));
  const size_t region = emit_entries_for(kStaticBatchPaths);
  ws.emit = carve_emit(take(emit_bytes(region, true)), region, true);
  ws.emit_region_entries = uint32_t(region); ws.emit_paths = kStaticBatchPaths; ws.emit_deep = true;
  ws.contrib = static_cast<float*>(take(kStaticBatchPaths * 3 * sizeof(float)));
  ws.counts = static_cast<uint32_t*>(take(kCountBytes));
  ws.stats = static_cast<DeviceStats*>(take(sizeof(DeviceStats)));
  ws.tiles = static_cast<uint32_t*>(take(kStaticMaxTiles * sizeof(uint32_t)));
  return at <= kStaticScratchBytes;
}

}  // namespace

extern "C" int rtmi_render_accumulate_noalloc(rtmi_scene* s, const rtmi_camera* cam, const rtmi_render_params* p, float* dev_xyz_sum,
                                              void* stream_) {
  if (int rc = check_scene(s)) return rc;
  RenderPlan r;
  if (int rc = plan_render(s, cam, p, dev_xyz_sum, r)) return rc;
  if (p->flags & RTMI_FLAG_TIME_KERNELS) return fail(RTMI_UNSUPPORTED, "RTMI_FLAG_TIME_KERNELS creates HIP events, which allocates; not available here");
  const uint64_t per_tile = uint64_t(r.T) * r.T;
  if (per_tile > kStaticBatchPaths)
    return fail(RTMI_UNSUPPORTED, "tile_size %u: one tile (%llu paths) exceeds the static workspace's batch of %zu path slots",
                r.T, (unsigned long long)per_tile, kStaticBatchPaths);
  if (r.n_tiles > kStaticMaxTiles)
    return fail(RTMI_UNSUPPORTED, "%u tiles in this call's share; the static workspace lists at most %zu (use larger tiles or a tile partition)",
                r.n_tiles, kStaticMaxTiles);
  rtmi_ctx* ctx = s->ctx;
  if (!ctx->scratch) return fail(RTMI_INTERNAL, "the context has no static scratch address");
  Workspace ws;
  if (!carve_static_workspace(ctx->scratch, ws)) return fail(RTMI_INTERNAL, "the static workspace layout exceeds the scratch array");
  Timer timer;
  const uint32_t n_samples = r.s_end - r.s_begin;
  if (r.n_tiles == 0 || n_samples == 0) { for (auto& b : s->last_stats) b = nullptr; return RTMI_OK; }
  const uint32_t tiles_per_batch = uint32_t(std::min<uint64_t>(r.n_tiles, kStaticBatchPaths / per_tile));
  const uint32_t samples_per_batch = uint32_t(std::min<uint64_t>(std::min<uint32_t>(n_samples, kStaticMaxSamplesPerBatch),
                                                                 std::max<uint64_t>(1, kStaticBatchPaths / (tiles_per_batch * per_tile))));
  const Workspace* wss[1] = {&ws};
  return render_batches(s, wss, 1, cam, p, r, tiles_per_batch, samples_per_batch, dev_xyz_sum, static_cast<hipStream_t>(stream_), timer);
}

extern "C" int rtmi_render_collect_stats(rtmi_scene* s, void* stream_, rtmi_stats* stats) {
  if (int rc = check_scene(s)) return rc;
  if (!stats) return fail(RTMI_BAD_ARG, "stats is NULL");
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
  *stats = s->pending;
  TraversalCounters trav{};
  for (DeviceStats* block : s->last_stats) {
    if (!block) continue;
    DeviceStats h;
    HIP_TRY(hipMemcpy(&h, block, sizeof h, hipMemcpyDeviceToHost));
    if (h.overflow) {
      // reported once: the word is sticky on the device (a render does not clear it), so it is cleared here
      (void)hipMemset(reinterpret_cast<char*>(block) + offsetof(DeviceStats, overflow), 0, sizeof(uint32_t));
      return fail(RTMI_INTERNAL, "a path-queue region overflowed and paths were dropped: the image of this render, or of a render "
                                 "before it on this context whose statistics were not collected, is incomplete "
                                 "(internal sizing error; please report the scene and the batch size)");
    }
    stats->segments += h.bounce_segments;
    stats->nodes_visited += h.trav.nodes; stats->prims_tested += h.trav.prims; stats->mesh_rays += h.mesh_rays;
    trav.nodes += h.trav.nodes; trav.prims += h.trav.prims; trav.node_slots += h.trav.node_slots; trav.prim_slots += h.trav.prim_slots;
    trav.service_slots += h.trav.service_slots; trav.service_lanes += h.trav.service_lanes;
    for (int k = 0; k < 8; ++k) trav.idle[k] += h.trav.idle[k];
  }
  if (getenv("RTMI_VERBOSE") && trav.node_slots)
    std::fprintf(stderr, "rtmi: traversal lane occupancy: node loop %.1f %% (%llu steps), triangle loop %.1f %% (%llu tests), "
                 "service block %.1f %% (%llu lanes)\n", 100.0 * trav.nodes / trav.node_slots, trav.nodes,
                 100.0 * trav.prims / std::max(1ull, trav.prim_slots), trav.prims,
                 100.0 * trav.service_lanes / std::max(1ull, trav.service_slots), trav.service_lanes);
  if (getenv("RTMI_VERBOSE") && trav.node_slots)
    std::fprintf(stderr, "rtmi: lanes sitting out, %% of the phase's lane slots: node steps -- finished %.1f, no ray %.1f, second leaf %.1f, "
                 "walk done with a leaf parked %.1f, leaf reached inside the group %.1f; triangle steps -- walking %.1f, finished %.1f, no ray %.1f\n",
                 100.0 * trav.idle[0] / trav.node_slots, 100.0 * trav.idle[1] / trav.node_slots, 100.0 * trav.idle[2] / trav.node_slots,
                 100.0 * trav.idle[3] / trav.node_slots, 100.0 * trav.idle[7] / trav.node_slots, 100.0 * trav.idle[4] / std::max(1ull, trav.prim_slots),
                 100.0 * trav.idle[5] / std::max(1ull, trav.prim_slots), 100.0 * trav.idle[6] / std::max(1ull, trav.prim_slots));
  return RTMI_OK;
}

extern "C" int rtmi_tonemap(rtmi_ctx* ctx, const float* dev_xyz_sum, uint32_t spp, uint32_t width, uint32_t height,
                            uint8_t* dev_rgb8, void* stream) {
  if (!ctx || !dev_xyz_sum || !dev_rgb8) return fail(RTMI_BAD_ARG, "NULL argument");
  if (spp == 0) return fail(RTMI_BAD_ARG, "spp is 0");
  if (int rc = bind(ctx)) return rc;
  launch_tonemap(dev_xyz_sum, spp, width * height, dev_rgb8, static_cast<hipStream_t>(stream));
  HIP_TRY(hipGetLastError());
  return RTMI_OK;
}

extern "C" int rtmi_render(rtmi_scene* s, const rtmi_camera* cam, const rtmi_render_params* p, uint8_t* host_rgb8,
                           float* host_xyz, rtmi_stats* stats) {
  if (int rc = check_scene(s)) return rc;
  if (!cam || !p) return fail(RTMI_BAD_ARG, "NULL argument");
  if (!host_rgb8 && !host_xyz) return fail(RTMI_BAD_ARG, "no output buffer");
  rtmi_ctx* ctx = s->ctx;
  Timer timer;
  const size_t n_pix = size_t(cam->width) * cam->height;
  if (int rc = ctx->image.reserve(n_pix * 3 * sizeof(float) + n_pix * 3)) return rc;
  float* xyz = ctx->image.as<float>();
  uint8_t* rgb8 = reinterpret_cast<uint8_t*>(xyz + n_pix * 3);
  HIP_TRY(hipMemset(xyz, 0, n_pix * 3 * sizeof(float)));
  rtmi_stats local{};
  if (int rc = rtmi_render_accumulate(s, cam, p, xyz, nullptr, &local)) return rc;
  if (int rc = rtmi_render_collect_stats(s, nullptr, &local)) return rc;
  if (host_rgb8) {
    if (int rc = rtmi_tonemap(ctx, xyz, p->spp, cam->width, cam->height, rgb8, nullptr)) return rc;
    HIP_TRY(hipMemcpy(host_rgb8, rgb8, n_pix * 3, hipMemcpyDeviceToHost));
  }
  if (host_xyz) {
    HIP_TRY(hipMemcpy(host_xyz, xyz, n_pix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    const float inv = float(p->spp);
    for (size_t i = 0; i < n_pix * 3; ++i) host_xyz[i] /= inv;
  }
  local.ms_total = timer.ms();
  if (stats) *stats = local;
  return RTMI_OK;
}
