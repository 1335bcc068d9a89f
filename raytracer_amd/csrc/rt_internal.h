// Host-side plumbing shared by the translation units of librt355.so: context, error capture,
// device buffers, host/device pointer staging and per-kernel HIP-event timing.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "../../include/rt355.h"
#include "rt_geom.h"
#include "rt_kdq.h"

// RAII device allocation
struct DevMem {
  void *p = nullptr; size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
  ~DevMem() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  hipError_t alloc(size_t n) { release(); if (n == 0) n = 16; hipError_t e = hipMalloc(&p, n); if (e == hipSuccess) bytes = n; return e; }
  hipError_t reserve(size_t n) { return (n <= bytes) ? hipSuccess : alloc(n); }    // grow-only, contents dropped
  template <typename T> T *as() const { return (T *)p; }
};

#define RT_MAX_IN_FLIGHT 8
// Experiment / test switches of a context.  Their defaults are the measured best; rt_init reads the RT355_* environment
// variables ONCE into this struct and rt_set_option changes single entries afterwards, so no frame pays a getenv.
struct RtKnobs {
  long tail_threshold = -1;      // paths left at which the tail kernel takes over (< 0: the built-in default)
  int whole_frame = 0;           // 1: the whole frame in one launch
  int bounce_passes = 0;         // single-bounce passes between the first pass and the tail kernel
  int tail_handoff = 0;          // > 0: tail workgroups down to this many paths (<= 64) hand them on to a further launch
  int tail_rounds = 1;           // how many such further launches (1 .. 3)
  int tail_persistent_handoff = 0;    // persistent tail: waves down to this many paths (<= 32) hand them to a second launch; 0: they keep them
  int tail_persistent = -1;      // tail kernel with persistent lanes, this many queued paths per lane and workgroup; 0: the compacting
                                 // tail kernel; < 0: by the number of frames in flight (persistent lanes pay from three on)
  int no_lds = 0;                // 1: scans read the primitives through scalar loads instead of the LDS copy
  int no_cull = 0;               // 1: the first pass scans every primitive (no per-workgroup culling)
  int scan_groups = 1;           // 0: scenes are uploaded without scan groups (every fp32 all-primitives scan tests every primitive)
  int primary_tiles = -1;        // the first pass with persistent workgroups over several 256-path tiles each (1), one workgroup per tile (0),
                                 // or (-1) by the number of frames in flight (persistent workgroups pay from three on)
  int cull_lists = 1;            // 0: every workgroup of the first pass works out its own list of reachable primitives, every frame
  int raster_order = 0;          // 1: paths numbered row by row even where 32x8 tiles would fit
  int bvh = -1;                  // hierarchy at upload: -1 by scene size, 0 never, 1 always
  int bvh_one_order = 0;         // 1: the hierarchy is walked in one child order for every ray (no near-child-first copies)
  double wave_gather = 8;        // unsorted request queue (gather_sort = 0): a wave serves its 64 requests together while the box of their search
                                 // balls is within this many ball areas (0: every lane gathers on its own)
  int defer_gathers = -1;        // fp32 frames with photon maps: the gathers go through a request queue and kernels of their own (1), are done where
                                 // the hit is shaded (0), or (-1) queued when the maps hold 200 000 photons or more (below that the extra launches cost
                                 // more than the walks)
  int gather_sort = 1;           // the requests are served in the order of their positions
  int gather_split = 4096;       // ... and subtrees of the maps of at most this many photons are work items of their own (0: one wave per 64 requests does it all)
  int knn_heap = 0;              // 1: k-nearest gathers use the per-lane heap kernel only
  int knn_sort = -1;             // rt_gather_radiance, k > 0: the queries are first put in the order of their positions (1), taken as they come (0),
                                 // or (-1) sorted when 64 consecutive ones lie far apart compared with the distance to a k-th neighbour
  int knn_verbose = 0;
  int probe = 0;
  double knn_radius_scale = 0;   // > 1: the wave-cooperative k-nearest gather searches balls this much larger than needed (tests)
};
struct rt_ctx {
  int device = 0;
  int n_cu = 256;                        // compute units of the device (persistent kernels size their grids by it)
  hipStream_t stream = nullptr;          // the stream launches go to: streams[0], except while a queued frame is put on another
  hipStream_t streams[RT_MAX_IN_FLIGHT] = {};
  int frames_in_flight = 1;              // n > 1: consecutive queued frames rotate through n streams (rt_set_frames_in_flight)
  bool alt_used = false;                 // something was queued on streams[1..] since they were last synchronised
  std::string err;
  bool profiling = false;
  int profile_every = 1;         // time every n-th launch of a selected kernel name (rt_profile_enable(ctx, n))
  std::map<std::string, uint64_t> profile_seen;
  std::string profile_only;      // when non-empty: ",name1,name2," -- only launches under these names are timed
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> events;
  std::vector<hipEvent_t> event_pool;   // events free for reuse (creating one per launch costs more than recording it)
  hipEvent_t take_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
  }
  uint32_t *pinned = nullptr;    // small pinned scratch for counters read back each bounce
  std::vector<uint32_t> kd_work; // the lower-level segments of the kd build in flight (read by a queued copy)
  RtKnobs knobs;
  DevMem ws_kd;                  // kd-tree build workspace (grow-only)
  // Device blocks of freed photon maps, kept for the next map of the same size (a map is six allocations of up to 88 bytes a
  // photon; building a map per frame of an animation, or per benchmark repetition, otherwise spends a fifth of its time in
  // hipMalloc / hipFree).  At most RT_MAP_POOL_BLOCKS blocks and RT_MAP_POOL_BYTES bytes; rt_shutdown frees them.
  std::vector<std::pair<size_t, void *>> map_pool;
  size_t map_pool_bytes = 0;
  hipError_t pool_take(DevMem &m, size_t bytes) {
    if (bytes == 0) bytes = 16;
    for (size_t i = 0; i < map_pool.size(); i++)
      if (map_pool[i].first == bytes) {
        m.release(); m.p = map_pool[i].second; m.bytes = bytes;
        map_pool_bytes -= bytes; map_pool[i] = map_pool.back(); map_pool.pop_back();
        return hipSuccess;
      }
    return m.alloc(bytes);
  }
  void pool_give(DevMem &m) {
    if (!m.p) return;
    if (map_pool.size() < 24 && map_pool_bytes + m.bytes <= ((size_t)2 << 30)) { map_pool.emplace_back(m.bytes, m.p); map_pool_bytes += m.bytes; m.p = nullptr; m.bytes = 0; }
    else m.release();
  }
  hipEvent_t frame_done[RT_MAX_IN_FLIGHT] = {};   // recorded behind the last frame queued on streams[i], i >= 1
  bool frame_pending[RT_MAX_IN_FLIGHT] = {};      // ... and not yet joined into streams[0] (rt_join) or synchronised
};

int rt_fail(rt_ctx *ctx, int code, const char *fmt, ...);

#define RT_HIP(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return rt_fail((ctx), RT_ERR_HIP, "%s failed: %s (%s:%d)", #call,    \
                                         hipGetErrorString(e_), __FILE__, __LINE__);           \
  } while (0)

#define RT_TRY(expr) do { int rc_ = (expr); if (rc_ != RT_OK) return rc_; } while (0)

bool rt_is_device_ptr(const void *p);

// Returns a device pointer holding `bytes` from `src` (device pointers pass through; host
// pointers are copied into `tmp` on the context's stream).
int rt_stage_in(rt_ctx *ctx, const void *src, size_t bytes, DevMem &tmp, const void **out);
// Device buffer to write results into: `dst` itself if it is a device pointer, else `tmp`.
int rt_stage_out_begin(rt_ctx *ctx, void *dst, size_t bytes, DevMem &tmp, void **out);
// Copies `tmp` back to a host `dst` (no-op for device `dst`); synchronises the stream.
int rt_stage_out_end(rt_ctx *ctx, void *dst, size_t bytes, DevMem &tmp);

static inline bool rt_timing(rt_ctx *ctx, const char *name);
struct KernelTimer {
  rt_ctx *ctx; hipEvent_t a = nullptr, b = nullptr; const char *name;
  KernelTimer(rt_ctx *c, const char *n) : ctx(c), name(n) {
    if (rt_timing(ctx, n)) { a = ctx->take_event(); b = ctx->take_event(); (void)hipEventRecord(a, ctx->stream); }   // (library routines: events around the call)
  }
  ~KernelTimer() {
    if (a) { (void)hipEventRecord(b, ctx->stream); ctx->events[name].push_back({ a, b }); }
  }
};

static inline bool rt_timing(rt_ctx *ctx, const char *name) {
  if (!ctx->profiling || !(ctx->profile_only.empty() || ctx->profile_only.find("," + std::string(name) + ",") != std::string::npos)) return false;
  return ctx->profile_every <= 1 || ctx->profile_seen[name]++ % (uint64_t)ctx->profile_every == 0;
}

// launch on the context's stream; when profiling is on for `name`, the dispatch itself carries the start/stop events
// (hipExtLaunchKernelGGL), which does not put extra barrier packets between the kernels of a frame
#define RT_LAUNCH_SMEM(ctx, name, kernel, grid, block, smem, ...)                              \
  do {                                                                                         \
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;                                               \
    if (rt_timing((ctx), (name))) { ev_a_ = (ctx)->take_event(); ev_b_ = (ctx)->take_event(); } \
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), (smem), (ctx)->stream, ev_a_, ev_b_, 0, __VA_ARGS__); \
    if (ev_a_) (ctx)->events[(name)].push_back({ ev_a_, ev_b_ });                              \
  } while (0)
#define RT_LAUNCH(ctx, name, kernel, grid, block, ...) RT_LAUNCH_SMEM(ctx, name, kernel, grid, block, 0, __VA_ARGS__)

static inline uint32_t div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// ---- scene ----
struct rt_scene {
  rt_ctx *ctx = nullptr;
  DevMem meta, geom32, geom64, colour32, colour64, sgeom32, sgeom64, sorig, sflags, surface32, surface64, spec_gt, hit32, hit64;
  DevMem bvh_nodes, bvh_prims, bvh_unbounded, sbound, sot_cam, gsphere, gmask;
  DevScene<float> d32; DevScene<double> d64;
  // per-scene workspaces (grow-only) so that steady-state calls do not allocate
  // rt_trace_tile's workspaces, one set per frame in flight
  // what a workspace's lists of reachable primitives (first pass, fp32) were made for
  struct CullKey { int32_t x0, y0, w, h, block, sub_sample; uint32_t n_paths, tiles_x, band_shift, band_period, band_first; };
  struct TraceWs { DevMem paths, queue, ctr, masks, blocks, levels, gq, gacc, gsort, gtemp, gitems, cull; CullKey cull_key = {}; bool cull_valid = false; } tws[RT_MAX_IN_FLIGHT];
  int next_slot = 0;
  DevMem ws_emit_kind, ws_emit_rec, ws_emit_seg, ws_emit_blocks, ws_emit_shard, ws_misc;
};

int rt_sort_pairs_u64(rt_ctx *ctx, DevMem &temp, const unsigned long long *keys_in, unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n);

// ---- photon map ----
struct rt_photon_map {
  rt_ctx *ctx = nullptr;
  uint32_t n = 0;
  DevMem node, pay, rec64, perm, dim;
  DevMem ws_flags;                 // per-workgroup fallback flags of the k-nearest gather (grow-only)
  DevMem ws_queue, ws_sort;        // request queue, accumulators and sort buffers of the queued radius gather (grow-only)
  DevMem ws_knn_sort;              // keys, order, reordered queries and results of a k-nearest gather that sorts its queries (grow-only)
  float bbox[6] = { 0, 0, 0, 0, 0, 0 };   // min xyz, max xyz of the photon positions
  bool core_valid = false;
  float core[6] = { 0, 0, 0, 0, 0, 0 };   // the same without the outermost 2 % on either side of every axis (from the nodes of the tree's top levels)
  DevMap view() const {
    DevMap m; m.n = n; m.node = node.as<float4>(); m.pay = pay.as<float>(); m.rec64 = rec64.as<double>(); m.perm = perm.as<int32_t>();
    return m;
  }
};

// looks up this device's copy of the static scratch array (rt_kd.hip); called by rt_init with the device current
int rt_static_scratch_setup(rt_ctx *ctx);
rt_ctx *rt_lists_ctx(rt_photon_lists *lists);
int rt_kd_build_device(rt_ctx *ctx, const float4 *d_pos, uint32_t n, int32_t *d_perm, uint8_t *d_dim, bool want_box);
