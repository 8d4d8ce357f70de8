// rt_kernels.h -- launch interface between the host runtime (rtmi_api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rt_types.h"
#include "spectrum.h"

namespace rtmi {

// Path-state queue, structure of arrays (one element per path slot, coalesced per wave). The arrays sit one
// after another in a single allocation, each `entries` elements long, so the struct is a base pointer and a
// count instead of seventeen pointers: with two queues among a kernel's arguments the pointers alone exceeded
// the scalar register file and the compiler spilled them into vector registers.
struct PathQueue {
  char* base;
  uint64_t entries;     // multiple of 64 (keeps every array 256-byte aligned)
  uint32_t deep;        // chain1 is present (paths longer than 16 hits)
  uint32_t pad;
#if defined(__HIPCC__)
#define RTMI_Q_ARRAY(T, name, offset_bytes_per_entry) \
  __host__ __device__ __forceinline__ T* name() const { return reinterpret_cast<T*>(base + entries * uint64_t(offset_bytes_per_entry)); }
#else
#define RTMI_Q_ARRAY(T, name, offset_bytes_per_entry) \
  T* name() const { return reinterpret_cast<T*>(base + entries * uint64_t(offset_bytes_per_entry)); }
#endif
  // fields that are read and written together share 16-byte elements: one address and one memory instruction each
  RTMI_Q_ARRAY(double2, o_xy, 0)          // origin x, y
  RTMI_Q_ARRAY(double2, oz_dx, 16)        // origin z, direction x
  RTMI_Q_ARRAY(double2, d_yz, 32)         // direction y, z
  // closest mesh hit of the entry's ray, used by the two-kernel pipeline: hit_obj is cleared when the entry is
  // created (RT_MISS: no mesh was hit), the rest is filled in by traverse_kernel.
  // x = t as the reference reports it (what objects are compared by), y = Moller-Trumbore's t along the re-normalised
  // ray (places the hit point)
  RTMI_Q_ARRAY(double2, hit_t_raw, 48)
  RTMI_Q_ARRAY(uint2, hit_prim_slot, 64)  // file-order triangle index, position of that triangle in leaf order
  // Where the path's contribution goes. It also names the path: slot = sample_local * n_local_pixels + local pixel, from
  // which the pixel and sample (the Philox counter words) are recomputed where a path draws random numbers. How many ray
  // casts a path may still make is not stored either: every path of queue level k has made k.
  RTMI_Q_ARRAY(uint32_t, slot, 72)
  RTMI_Q_ARRAY(uint32_t, hit_obj, 76)
  // Spectrum-table rows picked up at the path's hits, one byte each, as eight arrays of one word (hits 4w .. 4w + 3 in
  // word w): a path of level k has k hits behind it, so a level reads and writes only its first ceil(k / 4) words -- one
  // word where most of the paths are (levels 1 to 4). Words 4..7 exist only when `deep`.
#if defined(__HIPCC__)
  __host__ __device__ __forceinline__ uint32_t* chain_word(uint32_t w) const { return reinterpret_cast<uint32_t*>(base + entries * uint64_t(80 + 4 * w)); }
#else
  uint32_t* chain_word(uint32_t w) const { return reinterpret_cast<uint32_t*>(base + entries * uint64_t(80 + 4 * w)); }
#endif
#undef RTMI_Q_ARRAY
  static constexpr uint64_t bytes_per_entry(bool deep_) { return deep_ ? 112 : 96; }
};

constexpr int kBlock = 256;            // 4 waves of 64
constexpr uint32_t kPixelBlock = 8;    // a wave covers an 8x8 pixel block of its tile

// A queue is split into kShards regions, each with its own fill counter on its own 128-byte
// line: one returning atomic per wave on a single word saturates at about 90 atomics per
// microsecond chip-wide, which capped the whole kernel. Work is dealt in chunks of kBlock
// paths; the survivors of chunk c go to region c % kShards, so a region receives at most
// ceil(C / kShards) chunks of at most kBlock survivors each, where C bounds the chunk numbers a
// kernel can deal. C exceeds ceil(paths / kBlock) by the partial chunks of the producer's input:
//   - one per region of a fused-pipeline level (kShards),
//   - one per region and class of a two-kernel-pipeline level (2 * kShards),
//   - one per sample of a camera batch for the level-0 mesh list, whose chunks are numbered
//     (pixel chunk) * n_samples + sample (`slack_chunks` carries that term).
// The second word of every counter line is an overflow flag: a wave that is handed a position past
// its region's end sets it and drops the path instead of writing outside the region.
constexpr uint32_t kShards = 64;
constexpr uint32_t kCountStride = 32;  // in uint32 units: one counter per 128-byte line
constexpr uint32_t kOverflowWord = 1;  // word of a counter line that flags an overflow of its region
inline uint32_t region_chunks(size_t max_paths, size_t slack_chunks = 0) {
  const size_t chunks = (max_paths + kBlock - 1) / kBlock + std::max<size_t>(2 * kShards, slack_chunks);
  return uint32_t((chunks + kShards - 1) / kShards);
}
inline size_t queue_entries(size_t max_paths, size_t slack_chunks = 0) {
  return size_t(region_chunks(max_paths, slack_chunks)) * kBlock * kShards;
}

// One batch of primary paths: samples [sample0, sample0+n_samples) of the pixels of a run of
// tiles. Path slot = sample_local * n_local_pixels + local_pixel.
// n / d for a divisor fixed at launch: one 32 x 32 -> 64 multiply and a shift instead of the two dozen instructions
// of a division by a run-time value. Exact for n < 2^31 and 1 <= d <= 2^30 (path and pixel indices stay below 2^30):
// m = floor(2^shift / d) + 1 with shift = 31 + ceil(log2 d), which always fits 32 bits.
struct FastDiv { uint32_t d, m, shift, pad; };
inline FastDiv make_fast_div(uint32_t d) {
  if (d == 0) d = 1;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  FastDiv f{d, 0, 31 + s, 0};
  f.m = uint32_t((1ull << f.shift) / d + 1);
  return f;
}

struct BatchDesc {
  DevCamera cam;
  const uint32_t* tile_list;   // tile ids (x-major) this batch covers, device memory
  uint32_t n_tiles;            // entries of tile_list used
  uint32_t tile_size;          // multiple of 8
  uint32_t tiles_y;            // tiles per image column
  uint32_t n_local_pixels;     // n_tiles * tile_size^2 (includes pixels clipped off the image)
  FastDiv by_local_pixels, by_tile_pixels, by_tiles_y, by_blocks_per_row;   // n_local_pixels, tile_size^2, tiles_y, tile_size / 8
  uint32_t sample0, n_samples;
  uint32_t integrator, max_depth;
  uint64_t seed;
  // caller-supplied rays instead of camera rays (rtmi_trace_batch): 6 doubles each, device memory
  const double* rays;
  uint32_t n_rays, pixel_base;
};

// Paths that ended on an emitter, waiting for their 60-bin product to be formed. Evaluating the
// product inside the bounce kernel made every wave pay ~1000 instructions whenever one of its 64
// lanes reached a light; queued, the work runs dense in its own kernel. Same sharded layout as a
// path queue (kShards regions of emit_region_entries, one fill counter per 128-byte line).
struct EmitQueue {
  uint32_t* slot;     // contribution slot, RT_MISS for a reserved-but-unused entry
  uint32_t* info;     // emitted spectrum row | number of chain entries << 8
  uint4* chain0;
  uint4* chain1;      // only when depth > 16
};

struct TraceOutputs {
  float* contrib_x; float* contrib_y; float* contrib_z;  // per path slot, CIE XYZ of the sample
  float* spectra;                                        // optional: per path slot, 60 bins
  EmitQueue emit;                                        // used when emit_counts != nullptr
  uint32_t* emit_counts;
  uint32_t emit_region_entries;
};

// nodes, prims: lane-level work. The *_slots fields count 64 per wave-level iteration of the traversal kernel's node
// loop, triangle loop and service block (and service_lanes the lanes actually serviced), so that
// nodes / node_slots etc. give the lane occupancy of each phase.
// idle[]: what the lanes that sat a step out were doing -- node steps: [0] finished, waiting to be serviced, [1] no ray
// (the wave's share is used up), [2] holding a second leaf while the first is still parked, [3] walk finished, parked leaf
// pending; triangle steps: [4] walking (no leaf yet), [5] finished, [6] no ray; [7] node steps again: reached its first leaf earlier in this
// group of steps.
struct TraversalCounters { unsigned long long nodes, prims, node_slots, prim_slots, service_slots, service_lanes, idle[8]; };

// copies the spectral tables into the current device's constant memory
hipError_t upload_constants(const SpectrumTables& t);

// Bounce 0: generates the camera rays of the batch (or reads caller rays), finds the closest
// hit, shades, and appends surviving paths to `out`. out_counts: kShards counters, kCountStride
// apart, zeroed by the caller. region_entries = region_chunks(max_paths) * kBlock for both queues.
void launch_primary(const DevScene& sc, const BatchDesc& b, PathQueue out, uint32_t* out_counts, uint32_t region_entries,
                    TraceOutputs o, TraversalCounters* counters, hipStream_t s);
// Bounce k > 0: consumes the paths of `in` (fill counters in_counts), at most max_in in total.
// level: the queue level of `in` (its paths have made `level` ray casts).
void launch_bounce(const DevScene& sc, const BatchDesc& b, uint32_t level, PathQueue in, const uint32_t* in_counts, uint32_t max_in,
                   PathQueue out, uint32_t* out_counts, uint32_t region_entries, TraceOutputs o,
                   TraversalCounters* counters, hipStream_t s);
// Two-kernel pipeline, used when the scene has meshes. BVH traversal lengths differ by an order of
// magnitude between the rays of one wave (a ray caught inside a closed mesh under the first-index
// rule visits every box on its line), which left four lanes in five idle in the fused kernel. Here
//   generate_kernel (dense)            level 0 only: lists the primary rays that meet a mesh's box;
//   traverse_kernel (persistent waves) walks the BVHs for the listed entries only, each wave refilling
//                   its idle lanes from its share of the list, and records the mesh hit per entry;
//   shade_kernel    (dense, coherent)  for every path of level k: exact sphere / lone-triangle tests,
//                   folded with the recorded mesh hit, shading, append of the continuing paths to level
//                   k+1, those whose new ray meets a mesh's box at the front of their region, the rest at its back.
// Level 0 has no queue: a path is recreated from its index (camera jitter or caller ray).
// What traverse_kernel works through. Level 0: `entries` lists the positions of the primary rays that meet a mesh's
// box (kShards regions of region_entries positions, region r fed by chunks c % kShards == r). Levels >= 1:
// entries == nullptr, the work is the mesh class of the queue level itself (the front of each region).
struct MeshList {
  uint32_t* entries;
  uint32_t* counts;    // kShards fill counters, kCountStride apart, zeroed by the caller
};
// Fill counters of one queue level: per region, the paths whose ray meets a mesh's box are stored from the region's
// front (`mesh`), the others from its back (`other`).
struct LevelCounts { uint32_t* mesh; uint32_t* other; };
// Rays of level 0 of a camera batch all start at the camera's focus; their entries hold only a direction.
struct FixedOrigin { double x, y, z; bool fixed; };
void launch_generate(const DevScene& sc, const BatchDesc& b, PathQueue q, uint32_t region_entries, MeshList mesh, hipStream_t s);
// level 0: the paths (max_in of them) are recreated from their index; otherwise they are read from `in`, queue level `level`
void launch_shade(const DevScene& sc, const BatchDesc& b, uint32_t level, PathQueue in, LevelCounts in_counts, uint32_t max_in,
                  PathQueue out, LevelCounts out_counts, uint32_t region_entries, TraceOutputs o, hipStream_t s);
void launch_traverse(const DevScene& sc, PathQueue q, MeshList mesh, uint32_t max_in, uint32_t region_entries, FixedOrigin origin,
                     TraversalCounters* counters, hipStream_t s);
// Forms the spectrum / XYZ of every queued emitter hit and writes the path's contribution.
void launch_emit(const DevScene& sc, TraceOutputs o, bool deep, uint32_t max_entries, hipStream_t s);
// rays: n x {o[3], d[3]} doubles; hits: n x rtmi_hit (64 bytes); both device memory.
void launch_intersect(const DevScene& sc, const void* rays, size_t n, void* hits, TraversalCounters* counters, hipStream_t s);
// tiles[j] = first + j * stride for j < n: the tile ids (x-major) of one render's share
void launch_fill_tiles(uint32_t* tiles, uint32_t n, uint32_t first, uint32_t stride, hipStream_t s);
// Adds the batch's per-slot contributions into the per-pixel sums, in sample order.
void launch_resolve(const BatchDesc& b, TraceOutputs o, float* xyz_sum, hipStream_t s);
// *total += every fill counter of n_levels consecutive queue levels
void launch_sum_counts(const uint32_t* counts, uint32_t n_levels, unsigned long long* total, hipStream_t s);
// *flag = 1 when a region of the first n_levels queue levels flagged an overflow, or when the two classes of a region
// (level k and level pair_level0 + k, k < n_pairs) together hold more than region_entries paths
void launch_check_overflow(const uint32_t* counts, uint32_t n_levels, uint32_t pair_level0, uint32_t n_pairs,
                           uint32_t region_entries, uint32_t* flag, hipStream_t s);
// dst[i] += src[i]
void launch_add(float* dst, const float* src, size_t n, hipStream_t s);
void launch_tonemap(const float* xyz_sum, uint32_t spp, uint32_t n_pixels, uint8_t* rgb8, hipStream_t s);

}  // namespace rtmi
