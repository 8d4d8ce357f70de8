// rtmi_multi.cpp -- one render over several GPUs behind the C ABI: the scene replicated on every device, the paths
// partitioned by sample range or by interleaved tiles, one host thread per device driving rtmi_render_accumulate, and
// one collective (RCCL reduce to the first device) that merges the per-pixel XYZ sums. Replaces the reference's tile
// thread pool (src/parallel.rs:37-61) and the render loop around it (src/integrators/integrator.rs:30-118).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"
#include "rt_kernels.h"

namespace rtmi {
int set_last_error(int code, const char* message);   // rtmi_api.cpp: sets the calling thread's rtmi_last_error()
}

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return rtmi::set_last_error(code, buf);
}

// ---------------------------------------------------------------------------------------------- RCCL, bound at run time
//
// The collective library is looked up when a multi-GPU handle is first created, not linked: it must be the build that
// belongs to the HIP runtime the process already uses (PyTorch ships its own pair), and single-GPU users of the library
// should not need it at all. Order: a librccl already loaded in the process, RTMI_RCCL_LIB, the directory of the loaded
// HIP runtime, then the loader's search path.
struct Rccl {
  typedef void* comm_t;
  void* lib = nullptr;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string path, error;
  static constexpr int kFloat32 = 7, kSum = 0;   // ncclFloat32, ncclSum (rccl.h)
};

Rccl* load_rccl(std::string* why_not = nullptr) {
  static Rccl r;
  static bool tried = false;
  if (why_not) *why_not = r.error;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  std::vector<std::pair<std::string, int>> candidates;
  for (const char* name : {"librccl.so.1", "librccl.so"}) candidates.push_back({name, RTLD_NOW | RTLD_NOLOAD});
  if (const char* env = getenv("RTMI_RCCL_LIB")) candidates.push_back({env, RTLD_NOW});
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      for (const char* name : {"librccl.so.1", "librccl.so"}) candidates.push_back({dir + name, RTLD_NOW});
    }
  }
  for (const char* name : {"librccl.so.1", "librccl.so"}) candidates.push_back({name, RTLD_NOW});
  for (const auto& c : candidates) {
    void* h = dlopen(c.first.c_str(), c.second);
    if (!h) { if (!(c.second & RTLD_NOLOAD)) r.error = dlerror(); continue; }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(h, "ncclReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString) {
      r.lib = h; r.path = c.first;
      return &r;
    }
    r.error = c.first + " lacks an RCCL entry point";
    dlclose(h);
  }
  if (r.error.empty()) r.error = "no librccl.so next to the HIP runtime or on the loader path";
  if (why_not) *why_not = r.error;
  return nullptr;
}

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Rank {
  int device = 0;
  rtmi_ctx* ctx = nullptr;
  rtmi_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  float* sums = nullptr;          // W * H * 3 floats on this rank's device
  size_t sums_floats = 0;
  Rccl::comm_t comm = nullptr;
  int rc = RTMI_OK;               // of the last render leg, with its message
  std::string error;
  rtmi_stats stats{};
};

}  // namespace

struct rtmi_multi {
  std::vector<Rank> ranks;
  Rccl* rccl = nullptr;           // non-null: the merge is one ncclReduce per rank; null: copies to the first device and adds
  float* total = nullptr;         // first device: the merged sums of a pass (the ranks' own sums stay apart for the next pass)
  size_t total_floats = 0;
  float* staging = nullptr;       // first device: a peer's sums on their way into the total (merge without RCCL)
  size_t staging_floats = 0;
  uint8_t* rgb8 = nullptr;        // first device
  size_t rgb8_bytes = 0;
  // the progressive render in flight (rtmi_multi_render_pass): which pass comes next, of how many, and what it renders
  uint32_t next_pass = 0, n_passes = 0, pass_partition = 0;
  rtmi_camera pass_cam{};
  rtmi_render_params pass_params{};
};

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return fail(e_ == hipErrorOutOfMemory ? RTMI_OOM : RTMI_HIP_ERROR, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" void rtmi_multi_destroy(rtmi_multi* m) {
  if (!m) return;
  for (Rank& r : m->ranks) {
    (void)hipSetDevice(r.device);
    if (r.comm && m->rccl) (void)m->rccl->CommDestroy(r.comm);
    if (r.scene) rtmi_scene_destroy(r.scene);
    if (r.sums) (void)hipFree(r.sums);
    if (r.stream) (void)hipStreamDestroy(r.stream);
    if (r.ctx) rtmi_ctx_destroy(r.ctx);
  }
  if (!m->ranks.empty()) {
    (void)hipSetDevice(m->ranks[0].device);
    if (m->staging) (void)hipFree(m->staging);
    if (m->total) (void)hipFree(m->total);
    if (m->rgb8) (void)hipFree(m->rgb8);
  }
  delete m;
}

extern "C" int rtmi_multi_create(const int* device_ids, int n, uint64_t max_batch_paths, const rtmi_scene_desc* desc, rtmi_multi** out) {
  if (!device_ids || !desc || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (n < 1 || n > 64) return fail(RTMI_BAD_ARG, "%d devices (1..64)", n);
  std::unique_ptr<rtmi_multi, void (*)(rtmi_multi*)> m(new rtmi_multi, rtmi_multi_destroy);
  m->ranks.resize(size_t(n));
  for (int i = 0; i < n; ++i) {
    Rank& r = m->ranks[size_t(i)];
    if (int rc = rtmi_ctx_create(device_ids[i], max_batch_paths, &r.ctx)) return rc;
    HIP_TRY(hipGetDevice(&r.device));    // rtmi_ctx_create left its device current (-1 resolves to a real id here)
    HIP_TRY(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    if (int rc = rtmi_scene_create(r.ctx, desc, &r.scene)) return rc;   // the scene is replicated: it is read-only and small
  }
  std::set<int> distinct;
  for (const Rank& r : m->ranks) distinct.insert(r.device);
  const bool all_distinct = distinct.size() == m->ranks.size();
  // One rank needs no collective (RTMI_RCCL_SINGLE=1 runs it anyway, to exercise the binding on a one-GPU machine).
  // Ranks that share a GPU (more shares than devices) cannot form an RCCL communicator: their sums are merged by copies
  // to the first device and adds.
  if (all_distinct && (n > 1 || getenv("RTMI_RCCL_SINGLE"))) {
    std::string why_not;
    m->rccl = load_rccl(&why_not);
    if (!m->rccl) return fail(RTMI_RCCL_ERROR, "RCCL could not be loaded (%s); set RTMI_RCCL_LIB to the librccl.so of this process's HIP runtime",
                              why_not.c_str());
    std::vector<Rccl::comm_t> comms(size_t(n), nullptr);
    std::vector<int> devs;
    for (const Rank& r : m->ranks) devs.push_back(r.device);
    const int rc = m->rccl->CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) return fail(RTMI_RCCL_ERROR, "ncclCommInitAll over %d devices: %s", n, m->rccl->GetErrorString(rc));
    for (int i = 0; i < n; ++i) m->ranks[size_t(i)].comm = comms[size_t(i)];
  }
  *out = m.release();
  return RTMI_OK;
}

extern "C" int rtmi_multi_info(const rtmi_multi* m, rtmi_multi_info_t* out) {
  if (!m || !out) return fail(RTMI_BAD_ARG, "NULL argument");
  std::memset(out, 0, sizeof *out);
  out->n_ranks = uint32_t(m->ranks.size());
  out->uses_rccl = m->rccl ? 1u : 0u;
  if (m->rccl) std::snprintf(out->rccl_library, sizeof out->rccl_library, "%s", m->rccl->path.c_str());
  return RTMI_OK;
}

namespace {

int ensure_rank_buffers(rtmi_multi* m, size_t n_pix) {
  for (Rank& r : m->ranks) {
    HIP_TRY(hipSetDevice(r.device));
    if (r.sums_floats < n_pix * 3) {
      if (r.sums) { (void)hipFree(r.sums); r.sums = nullptr; r.sums_floats = 0; }
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&r.sums), n_pix * 3 * sizeof(float)));
      HIP_TRY(hipMemset(r.sums, 0, n_pix * 3 * sizeof(float)));   // never handed out uninitialised
      r.sums_floats = n_pix * 3;
    }
  }
  HIP_TRY(hipSetDevice(m->ranks[0].device));
  if (m->rgb8_bytes < n_pix * 3) {
    if (m->rgb8) { (void)hipFree(m->rgb8); m->rgb8 = nullptr; m->rgb8_bytes = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->rgb8), n_pix * 3));
    m->rgb8_bytes = n_pix * 3;
  }
  if (m->total_floats < n_pix * 3) {
    if (m->total) { (void)hipFree(m->total); m->total = nullptr; m->total_floats = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->total), n_pix * 3 * sizeof(float)));
    m->total_floats = n_pix * 3;
  }
  if (!m->rccl && m->ranks.size() > 1 && m->staging_floats < n_pix * 3) {
    if (m->staging) { (void)hipFree(m->staging); m->staging = nullptr; m->staging_floats = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->staging), n_pix * 3 * sizeof(float)));
    m->staging_floats = n_pix * 3;
  }
  return RTMI_OK;
}

// One rank's leg of a pass, on its own host thread: its share of the samples or tiles added into its own buffer.
void render_leg(Rank* r, const rtmi_camera* cam, rtmi_render_params p, size_t n_floats, bool clear) {
  r->rc = RTMI_OK; r->error.clear();
  auto check = [&](int rc) { if (rc != RTMI_OK && r->rc == RTMI_OK) { r->rc = rc; r->error = rtmi_last_error(); } return rc == RTMI_OK; };
  hipError_t e = hipSetDevice(r->device);
  if (e == hipSuccess && clear) e = hipMemsetAsync(r->sums, 0, n_floats * sizeof(float), r->stream);
  if (e != hipSuccess) { r->rc = RTMI_HIP_ERROR; r->error = hipGetErrorString(e); return; }
  if (!check(rtmi_render_accumulate(r->scene, cam, &p, r->sums, r->stream, nullptr))) return;
  check(rtmi_render_collect_stats(r->scene, r->stream, &r->stats));   // waits for the stream
}

// Samples [s_begin, s_end) of the render described by `params`, shared out over the ranks and added to their sums
// (cleared first when `clear`); then the one exchange: every rank's W * H * 3 sums reduced into `total` on the first
// device, which is tone-mapped as the mean over `divisor` samples.
int render_samples(rtmi_multi* m, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition, uint32_t s_begin,
                   uint32_t s_end, bool clear, uint32_t divisor, uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats,
                   rtmi_stats* per_rank_stats) {
  Timer timer;
  const size_t n = m->ranks.size(), n_pix = size_t(cam->width) * cam->height;
  if (int rc = ensure_rank_buffers(m, n_pix)) return rc;

  // shares: contiguous balanced sample ranges (sizes differ by at most one), or tiles k with k % n == rank
  std::vector<std::thread> threads;
  const uint32_t n_samples = s_end - s_begin;
  for (size_t i = 0; i < n; ++i) {
    rtmi_render_params p = *params;
    if (partition == RTMI_PARTITION_SAMPLES) {
      const uint32_t base = n_samples / uint32_t(n), extra = n_samples % uint32_t(n);
      const uint32_t begin = s_begin + uint32_t(i) * base + std::min<uint32_t>(uint32_t(i), extra);
      const uint32_t end = begin + base + (i < extra ? 1u : 0u);
      p.sample_begin = begin; p.sample_end = end;
      // more ranks than samples: this rank gets an empty tile share (it still clears its buffer for the merge)
      if (end == begin) { p.sample_begin = p.sample_end = 0; p.flags |= RTMI_FLAG_EMPTY_SHARE; }
    } else {
      p.tile_rank = uint32_t(i); p.tile_world = uint32_t(n);
      p.sample_begin = s_begin; p.sample_end = s_end;
      if (s_begin == 0 && s_end == params->spp) p.sample_begin = p.sample_end = 0;
    }
    threads.emplace_back(render_leg, &m->ranks[i], cam, p, n_pix * 3, clear);
  }
  for (auto& t : threads) t.join();
  for (size_t i = 0; i < n; ++i)
    if (m->ranks[i].rc != RTMI_OK) return fail(m->ranks[i].rc, "device %d: %s", m->ranks[i].device, m->ranks[i].error.c_str());

  Rank& root = m->ranks[0];
  if (m->rccl) {
    int rc = m->rccl->GroupStart();
    hipError_t hip_rc = hipSuccess;
    for (size_t i = 0; i < n && rc == 0 && hip_rc == hipSuccess; ++i) {
      Rank& r = m->ranks[i];
      hip_rc = hipSetDevice(r.device);
      // ncclReduce only uses the receive buffer on the root; the other ranks pass none, so that nothing can touch their own
      // sums, which the next pass of a progressive render adds to
      if (hip_rc == hipSuccess)
        rc = m->rccl->Reduce(r.sums, i == 0 ? m->total : nullptr, n_pix * 3, Rccl::kFloat32, Rccl::kSum, 0, r.comm, r.stream);
    }
    const int rc_end = m->rccl->GroupEnd();   // the group is closed on every path, also after a failed call inside it
    if (rc == 0) rc = rc_end;
    if (hip_rc != hipSuccess) return fail(RTMI_HIP_ERROR, "hipSetDevice before ncclReduce: %s", hipGetErrorString(hip_rc));
    if (rc != 0) return fail(RTMI_RCCL_ERROR, "ncclReduce of %zu floats over %zu ranks: %s", n_pix * 3, n, m->rccl->GetErrorString(rc));
    for (Rank& r : m->ranks) { HIP_TRY(hipSetDevice(r.device)); HIP_TRY(hipStreamSynchronize(r.stream)); }
  } else {
    HIP_TRY(hipSetDevice(root.device));
    HIP_TRY(hipMemcpyAsync(m->total, root.sums, n_pix * 3 * sizeof(float), hipMemcpyDeviceToDevice, root.stream));
    for (size_t i = 1; i < n; ++i) {
      Rank& r = m->ranks[i];
      const float* src = r.sums;
      if (r.device != root.device) {
        HIP_TRY(hipMemcpyPeerAsync(m->staging, root.device, r.sums, r.device, n_pix * 3 * sizeof(float), root.stream));
        src = m->staging;
      }
      rtmi::launch_add(m->total, src, n_pix * 3, root.stream);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(root.stream));
  }

  HIP_TRY(hipSetDevice(root.device));
  if (host_rgb8) {
    if (int rc = rtmi_tonemap(root.ctx, m->total, divisor, cam->width, cam->height, m->rgb8, root.stream)) return rc;
    HIP_TRY(hipMemcpyAsync(host_rgb8, m->rgb8, n_pix * 3, hipMemcpyDeviceToHost, root.stream));
  }
  if (host_xyz) HIP_TRY(hipMemcpyAsync(host_xyz, m->total, n_pix * 3 * sizeof(float), hipMemcpyDeviceToHost, root.stream));
  HIP_TRY(hipStreamSynchronize(root.stream));
  if (host_xyz) {
    const float spp = float(divisor);
    for (size_t i = 0; i < n_pix * 3; ++i) host_xyz[i] /= spp;
  }
  if (per_rank_stats) for (size_t i = 0; i < n; ++i) per_rank_stats[i] = m->ranks[i].stats;
  if (stats) {
    rtmi_stats total{};
    for (const Rank& r : m->ranks) {
      total.segments += r.stats.segments; total.paths += r.stats.paths; total.nodes_visited += r.stats.nodes_visited;
      total.prims_tested += r.stats.prims_tested; total.mesh_rays += r.stats.mesh_rays; total.launches += r.stats.launches;
      total.batches += r.stats.batches; total.traverse_launches += r.stats.traverse_launches;
      total.ms_kernels = std::max(total.ms_kernels, r.stats.ms_kernels); total.ms_traverse = std::max(total.ms_traverse, r.stats.ms_traverse);
    }
    total.ms_total = timer.ms();
    *stats = total;
  }
  return RTMI_OK;
}

int check_multi_args(const rtmi_multi* m, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition, const void* rgb8,
                     const void* xyz) {
  if (!m || !cam || !params) return fail(RTMI_BAD_ARG, "NULL argument");
  if (!rgb8 && !xyz) return fail(RTMI_BAD_ARG, "no output buffer");
  if (partition != RTMI_PARTITION_SAMPLES && partition != RTMI_PARTITION_TILES) return fail(RTMI_BAD_ARG, "unknown partition %u", partition);
  if (params->sample_end || params->sample_begin || params->tile_world || params->tile_rank)
    return fail(RTMI_BAD_ARG, "the partition fields of rtmi_render_params must be 0: the multi-device entry points set them per device");
  if (cam->width == 0 || cam->height == 0) return fail(RTMI_BAD_ARG, "camera resolution %ux%u", cam->width, cam->height);
  if (params->spp == 0) return fail(RTMI_BAD_ARG, "spp is 0");
  return RTMI_OK;
}

}  // namespace

extern "C" int rtmi_multi_render(rtmi_multi* m, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition,
                                 uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats, rtmi_stats* per_rank_stats) {
  if (int rc = check_multi_args(m, cam, params, partition, host_rgb8, host_xyz)) return rc;
  m->n_passes = m->next_pass = 0;   // a whole render clears the ranks' sums: a progressive render in flight is over
  return render_samples(m, cam, params, partition, 0, params->spp, true, params->spp, host_rgb8, host_xyz, stats, per_rank_stats);
}

extern "C" int rtmi_multi_render_pass(rtmi_multi* m, const rtmi_camera* cam, const rtmi_render_params* params, uint32_t partition,
                                      uint32_t pass, uint32_t n_passes, uint8_t* host_rgb8, float* host_xyz, rtmi_stats* stats) {
  if (int rc = check_multi_args(m, cam, params, partition, host_rgb8, host_xyz)) return rc;
  if (n_passes == 0 || pass >= n_passes || n_passes > params->spp)
    return fail(RTMI_BAD_ARG, "pass %u of %u over %u samples", pass, n_passes, params->spp);
  // The ranks' sums carry over from pass to pass: pass 0 starts a render (and clears them), every later pass must be the
  // next one of that same render -- same pass count, partition, camera and parameters -- or it would add to sums that
  // belong to another image.
  if (pass == 0) {
    m->n_passes = n_passes; m->pass_partition = partition; m->pass_cam = *cam; m->pass_params = *params;
  } else if (m->n_passes != n_passes || m->next_pass != pass || m->pass_partition != partition ||
             std::memcmp(&m->pass_cam, cam, sizeof *cam) != 0 || std::memcmp(&m->pass_params, params, sizeof *params) != 0) {
    return fail(RTMI_BAD_ARG, "pass %u of %u does not continue the progressive render in flight (next pass %u of %u): passes run "
                              "in order from 0 with the same camera, parameters and partition", pass, n_passes, m->next_pass, m->n_passes);
  }
  // balanced contiguous sample ranges, like the ranks' own
  const uint32_t base = params->spp / n_passes, extra = params->spp % n_passes;
  const uint32_t begin = pass * base + std::min(pass, extra), end = begin + base + (pass < extra ? 1u : 0u);
  const int rc = render_samples(m, cam, params, partition, begin, end, pass == 0, end, host_rgb8, host_xyz, stats, nullptr);
  if (rc == RTMI_OK) m->next_pass = pass + 1;
  else m->n_passes = m->next_pass = 0;      // the sums are in an unknown state: the render has to start again at pass 0
  return rc;
}
