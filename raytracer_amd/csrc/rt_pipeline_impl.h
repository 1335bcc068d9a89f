// Definitions of the pl_* drivers; included by rt_pipeline_f32.hip / rt_pipeline_f64.hip, which
// instantiate them for one precision each.
#pragma once
#include <string.h>
#include "rt_emit.h"
#include "rt_pipeline.h"
#include "rt_trace.h"
#include <stdlib.h>


// working epsilon of a precision: the reference's EPSILON for fp64, the coarser epsilon_f32 for fp32
template <typename T> static double working_eps(const rt_params *p) {
  if (sizeof(T) == 8) return p->epsilon;
  return p->epsilon_f32 > 0 ? p->epsilon_f32 : 1e-4;
}

// Scenes of up to this many primitives are staged whole in LDS by every workgroup (<= 56 KB); larger
// scenes read the primitives through wave-uniform (scalar) loads from global memory.
template <typename T> static bool scene_fits_lds(uint32_t n) { return scan_lds_bytes<T>(n) <= 56 * 1024; }

template <typename T> struct SceneOf;
template <> struct SceneOf<float> { static DevScene<float> &get(rt_scene *s) { return s->d32; } };
template <> struct SceneOf<double> { static DevScene<double> &get(rt_scene *s) { return s->d64; } };

// ------------------------------------------------------------------ unit-test surfaces
template <typename T>
__global__ void __launch_bounds__(256) k_intersect_pairs(uint32_t n, const double *o, const double *d, int normalize_dir,
                                                         const uint8_t *meta, const double *geom, T eps, double *t_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3<T> ro = mk<T>(T(o[3 * i]), T(o[3 * i + 1]), T(o[3 * i + 2]));
  V3<T> rd = mk<T>(T(d[3 * i]), T(d[3 * i + 1]), T(d[3 * i + 2]));
  if (normalize_dir) rd = normalize<T>(rd, eps);                        // Ray ctor, rayphoton.js:484-487
  T g[RT_GEOM_STRIDE];
  for (int k = 0; k < RT_GEOM_STRIDE; k++) g[k] = T(geom[(size_t)i * RT_GEOM_STRIDE + k]);
  T t = isect_any_type<T>(ro, rd, meta[4 * i], meta[4 * i + 2], g, eps);
  t_out[i] = (t > T(0)) ? (double)t : __builtin_nan("");
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_scan_rays(DevScene<T> S, uint32_t n, const double *o, const double *d,
                                                   const double *source_dist, T eps, double *t_out, int32_t *idx_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3<T> ro = mk<T>(T(o[3 * i]), T(o[3 * i + 1]), T(o[3 * i + 2]));
  V3<T> rd = mk<T>(T(d[3 * i]), T(d[3 * i + 1]), T(d[3 * i + 2]));
  T t = Limits<T>::inf();
  int idx = scan_scene<T, MODE>(S, global_scan_src<T>(S), ro, rd, eps, MODE == 2 ? T(source_dist[i]) : T(0), t);
  idx_out[i] = idx;
  if (t_out) t_out[i] = (MODE == 2) ? 0.0 : (double)t;
}

template <typename T>
int pl_intersect_pairs(rt_ctx *ctx, const rt_params *p, uint32_t n, const double *o, const double *d, int normalize_dir,
                       const uint8_t *meta, const double *geom, double *t_out) {
  RT_LAUNCH(ctx, "intersect_pairs", k_intersect_pairs<T>, div_up(n, 256), 256, n, o, d, normalize_dir, meta, geom, T(working_eps<T>(p)), t_out);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

template <typename T>
int pl_intersect_scene(rt_scene *sc, const rt_params *p, int mode, uint32_t n, const double *o, const double *d,
                       const double *source_dist, double *t_out, int32_t *idx_out) {
  rt_ctx *ctx = sc->ctx;
  DevScene<T> &S = SceneOf<T>::get(sc);
  const uint32_t grid = div_up(n, 256);
  const T eps = T(working_eps<T>(p));
  if (mode == RT_SCAN_CLOSEST) RT_LAUNCH(ctx, "scan_closest", (k_scan_rays<T, 0>), grid, 256, S, n, o, d, source_dist, eps, t_out, idx_out);
  else if (mode == RT_SCAN_CLOSEST_NOLIGHT) RT_LAUNCH(ctx, "scan_closest_nolight", (k_scan_rays<T, 1>), grid, 256, S, n, o, d, source_dist, eps, t_out, idx_out);
  else RT_LAUNCH(ctx, "scan_any", (k_scan_rays<T, 2>), grid, 256, S, n, o, d, source_dist, eps, t_out, idx_out);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

// ------------------------------------------------------------------ renderer
// 256-byte aligned sub-arrays out of one allocation; with base == nullptr only the total size is computed
template <typename T> static PathSoA<T> carve_paths(void *base, size_t n, size_t *total = nullptr) {
  char *p = (char *)base;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 255) & ~(size_t)255; return (void *)r; };
  PathSoA<T> P;
  P.ox = (T *)take(n * sizeof(T)); P.oy = (T *)take(n * sizeof(T)); P.oz = (T *)take(n * sizeof(T));
  P.dx = (T *)take(n * sizeof(T)); P.dy = (T *)take(n * sizeof(T)); P.dz = (T *)take(n * sizeof(T));
  P.max_dist = (T *)take(n * sizeof(T)); P.importance = (T *)take(n * sizeof(T));
  P.acc_r = (T *)take(n * sizeof(T)); P.acc_g = (T *)take(n * sizeof(T)); P.acc_b = (T *)take(n * sizeof(T));
  P.hit_t = (T *)take(n * sizeof(T)); P.hit_idx = (int32_t *)take(n * 4);
  P.draw = (uint32_t *)take(n * 4); P.depth = (int32_t *)take(n * 4); P.stack = (uint64_t *)take(n * 8);
  P.key = (uint32_t *)take(n * 4);
  P.out = (T *)take(n * 4 * sizeof(T));
  if (total) *total = (size_t)(p - (char *)base);
  return P;
}

// WHITTED: the older tracer (ray.js).  FULL: direct light and/or photon gathers are on (otherwise kernels with those
// terms compiled out run: fewer registers, more waves in flight).
template <typename T, bool WHITTED, bool FULL>
static int trace_tile_mode(rt_scene *sc, int slot, const rt_params *p, DevMap md, DevMap mc, int x0, int y0, int w, int h,
                           double *pre, double *post, uint8_t *rgba8, rt_trace_stats *stats) {
  rt_ctx *ctx = sc->ctx;
  rt_scene::TraceWs &W = sc->tws[(unsigned)slot % RT_MAX_IN_FLIGHT];
  const RtKnobs &K = ctx->knobs;
  DevScene<T> S = SceneOf<T>::get(sc);                        // (a copy: the kernels take it by value)
  if constexpr (sizeof(T) == 4) { if (K.no_cull) S.sbound = nullptr; }
  const int blk = p->super_sample_block > 1 ? p->super_sample_block : 1;       // one traced sample per blk x blk pixels
  const int wb = (w + blk - 1) / blk;
  int hb = (h + blk - 1) / blk;
  int band_shift = 0;
  if (p->band_count > 1) {                                    // this call traces one rank's row bands of the tile
    while ((1 << band_shift) < p->band_rows) band_shift++;
    hb = (int)rt_band_rows((uint32_t)h, p->band_rows, p->band_count, p->band_index);
  }
  const size_t n_pixels = (size_t)wb * hb;
  const size_t n_paths = n_pixels * p->sub_sample * p->sub_sample;
  if (n_paths == 0) return RT_OK;
  if (n_paths > 0x7FFFFFFFull) return rt_fail(ctx, RT_ERR_TOO_LARGE, "tile has too many paths (%zu)", n_paths);
  size_t path_bytes = 0;
  carve_paths<T>(nullptr, n_paths, &path_bytes);
  RT_HIP(ctx, W.paths.reserve(path_bytes));
  // two queues of n_paths entries, or the sub-queues between the first pass and the tail kernel
  const uint32_t subq_cap_all = div_up(div_up((uint32_t)n_paths, 256), RT_SUBQ) * 256;
  const size_t queue_len = std::max<size_t>(n_paths, (size_t)RT_SUBQ * subq_cap_all);
  RT_HIP(ctx, W.queue.reserve(queue_len * 8));
  RT_HIP(ctx, W.ctr.reserve(sizeof(TraceCounters)));
  PathSoA<T> P = carve_paths<T>(W.paths.p, n_paths);
  uint32_t *q_cur = W.queue.as<uint32_t>(), *q_nxt = q_cur + queue_len;
  TraceCounters *ctr = W.ctr.as<TraceCounters>();
  TraceCfg C;
  C.epsilon = working_eps<T>(p); C.max_trace_dist = p->max_trace_dist; C.diffuse_radius = p->diffuse_radius;
  C.caustic_radius = p->caustic_radius; C.standard_lamp_area = p->standard_lamp_area;
  C.shadow_bias = p->shadow_bias > 0 ? p->shadow_bias : C.epsilon;
  C.max_depth = p->max_depth; C.sub_sample = p->sub_sample; C.use_direct_light = p->use_direct_light;
  C.use_photon_maps = p->use_photon_maps; C.max_bounces = p->max_bounces > 0 ? p->max_bounces : 4096;
  C.wave_gather_ratio = (float)ctx->knobs.wave_gather;
  C.seed = p->seed; C.x0 = x0; C.y0 = y0; C.w = wb; C.h = hb; C.block = blk; C.n_paths = (uint32_t)n_paths;
  C.magic_w = (uint32_t)(0x100000000ull / (uint64_t)wb); C.magic_ss = (uint32_t)(0x100000000ull / (uint64_t)(p->sub_sample * p->sub_sample));
  C.magic_s = (uint32_t)(0x100000000ull / (uint64_t)p->sub_sample);
  C.tiles_x = (wb % 32 == 0 && hb % 8 == 0 && !K.raster_order) ? (uint32_t)(wb / 32) : 0u;
  C.magic_tx = C.tiles_x ? (uint32_t)(0x100000000ull / (uint64_t)C.tiles_x) : 0u;
  C.band_shift = (uint32_t)band_shift; C.band_period = band_shift ? (uint32_t)(p->band_rows * p->band_count) : 0u;
  C.band_first = band_shift ? (uint32_t)(p->band_rows * p->band_index) : 0u;
  C.fuse_resolve = (!WHITTED && p->sub_sample == 1) ? 1 : 0; C.pre = pre; C.post = post; C.rgba8 = rgba8;
  if (!p->use_photon_maps) { md.n = 0; mc.n = 0; }
  C.gq = nullptr; C.gq_count = nullptr; C.gq_cap = 0; C.gacc = nullptr; C.gq_order = nullptr; C.gq_valid = nullptr; C.gq_first = 0; C.gsat = &ctr->gather_saturations;
  bool defer = false;
  if constexpr (sizeof(T) == 4 && FULL && !WHITTED) {
    defer = p->use_photon_maps && (md.n || mc.n) && (K.defer_gathers < 0 ? (uint64_t)md.n + mc.n >= 200000u : K.defer_gathers != 0);
    if (defer) {                                              // request queue (a counter in its first 256 bytes) and per-path accumulators
      const uint32_t gq_cap = (uint32_t)std::min<size_t>(std::max<size_t>(2 * n_paths, 4096), (size_t)1 << 26);      // (requests beyond it are gathered where they are shaded)
      RT_HIP(ctx, W.gq.reserve(RT_GQ_HEADER_BYTES + (size_t)gq_cap * 52));
      RT_HIP(ctx, W.gacc.reserve(n_paths * 3 * 8));
      C.gq_count = W.gq.as<uint32_t>(); C.gq = (float4 *)((char *)W.gq.p + RT_GQ_HEADER_BYTES); C.gq_cap = gq_cap; C.gacc = W.gacc.as<long long>();
      C.gq_valid = (uint32_t *)((char *)W.gq.p + RT_GQ_HEADER_BYTES + (size_t)gq_cap * 48); C.gq_first = (uint32_t)std::min<size_t>(n_paths, gq_cap);
      RT_HIP(ctx, hipMemsetAsync(C.gq_count, 0, RT_GQ_HEADER_BYTES, ctx->stream));      // later requests; the item shards' lengths and cursors
      RT_HIP(ctx, hipMemsetAsync(C.gq_valid, 0, (size_t)gq_cap * 4, ctx->stream));
      RT_HIP(ctx, hipMemsetAsync(C.gacc, 0, n_paths * 3 * 8, ctx->stream));
      C.fuse_resolve = 0;                                     // the resolve pass adds the accumulators
    }
  }
  T *levels = nullptr;
  if (WHITTED) {      // one record of 4 values per path and recursion level (levels 0 .. max_depth)
    if (p->max_depth < 0 || p->max_depth > 255) return rt_fail(ctx, RT_ERR_INVALID, "max_depth %d outside [0, 255]", p->max_depth);
    RT_HIP(ctx, W.levels.reserve((size_t)(p->max_depth + 1) * n_paths * 4 * sizeof(T)));
    levels = W.levels.as<T>();
    C.max_bounces = p->max_depth + 2;                       // every path ends within max_depth + 1 levels
    md.n = 0; mc.n = 0;
  }

  RT_HIP(ctx, hipMemsetAsync(ctr, 0, sizeof(TraceCounters), ctx->stream));
  const uint32_t max_blocks = div_up(n_paths, 256);
  RT_HIP(ctx, W.masks.reserve((size_t)max_blocks * 4 * 8));
  RT_HIP(ctx, W.blocks.reserve(((size_t)max_blocks + 1) * 4));
  unsigned long long *wave_mask = W.masks.as<unsigned long long>();
  uint32_t *block_count = W.blocks.as<uint32_t>();
  uint32_t *counts = ctr->counts;                           // counts[b] = queue length of pass b (device side)
  bool lds = !S.bvh && scene_fits_lds<T>(S.n_shapes);       // hierarchy scans read the primitives they reach from global memory
  if (K.no_lds) lds = false;                                // (experiments: wave-uniform scalar loads instead)
  const size_t smem = lds ? scan_lds_bytes<T>(S.n_shapes) : 0;
  // k_tail (fp32) also keeps its workgroup's path states in LDS: it stages the scene only while both fit in 64 KB
  const bool lds_tail = lds && (sizeof(T) == 8 || smem + (size_t)(RT_LSTATE_WORDS + 3) * 4 * RT_TAIL_THREADS + 256 <= 64 * 1024);
  constexpr bool SHADE_SCANS = FULL && !WHITTED;            // only directLight's shadow rays scan primitives while shading
  // The first pass (k_primary) is fused.  Wide later bounces run as wavefront passes (intersect_closest, shade,
  // ordered compaction of the survivors); once at most tail_threshold paths are left, one k_tail launch follows
  // each of them to its end.  Queue lengths live on the device; a launch's grid is only an upper bound, so the
  // host reads a length back only when it needs one to choose the next step.
  // (measured on 3840x2160 frames: with the all-primitives scan the direct route wins, 0.52 vs 0.66 ms; with the
  // hierarchy the wavefront passes win, 4.9 vs 5.3 ms)
  uint32_t tail_threshold = S.bvh ? 1u << 21 : 1u << 24;
  if (K.tail_threshold >= 0) tail_threshold = (uint32_t)K.tail_threshold;
  // option whole_frame = 1: the whole frame in ONE launch (k_tail<RT_TAIL_PRIMARY>), every workgroup carrying its paths
  // from the camera ray to their end -- fewest launches, but the fused first pass then runs at the tail kernel's
  // lower occupancy; measured slower than the pipeline below on full frames.
  const bool whole_frame = K.whole_frame != 0;
  uint32_t ub = (uint32_t)n_paths;                          // upper bound of the current queue length
  uint32_t wave_bounces = 0;
  // when the tail kernel follows the first pass directly, the survivors go to it through sub-queues (their order
  // does not matter to it) and no scan / scatter launches are needed
  const bool direct = !whole_frame && ub <= tail_threshold && RT_MAX_WAVE_BOUNCES >= 1;
  C.subq_cap = direct ? subq_cap_all : 0;
  // ... optionally (option bounce_passes = 1|2) after single-bounce passes at the first pass's occupancy (k_bounce).
  // Off by default: on the 1080p frames measured, a pass costs more (state through global memory, another append)
  // than the iteration of k_tail it replaces.
  uint32_t bounce_passes = direct ? std::min<uint32_t>((uint32_t)std::max(K.bounce_passes, 0), RT_MAX_BOUNCE_PASSES) : 0;
  if ((int)bounce_passes > C.max_bounces - 1) bounce_passes = (uint32_t)std::max(C.max_bounces - 1, 0);    // pass j traces segment j
  C.iter_base = bounce_passes;
  C.cull_lists = nullptr; C.cull_stride = 0;
  if constexpr (sizeof(T) == 4) {
    // The first pass's per-workgroup lists of reachable primitives depend on the scene (fixed for this workspace), its
    // camera and the tile's layout only: made by one small launch when the layout changes, read by every frame after.
    if (lds && S.sbound && !S.bvh && !whole_frame && K.cull_lists) {
      rt_scene::CullKey key = { x0, y0, wb, hb, blk, p->sub_sample, (uint32_t)n_paths, C.tiles_x, C.band_shift, C.band_period, C.band_first };
      const uint32_t stride = cull_list_words(S.n_shapes), n_wg = div_up(n_paths, 256);
      const size_t need = (size_t)n_wg * stride * 4;
      if (!W.cull_valid || W.cull.bytes < need || memcmp(&W.cull_key, &key, sizeof key) != 0) {
        W.cull_valid = false;
        RT_HIP(ctx, W.cull.reserve(need));
        RT_LAUNCH(ctx, "cull_lists", k_cull_lists, div_up(n_wg, 4), 256, S, C, n_wg, 256u, W.cull.as<uint32_t>(), stride);
        W.cull_key = key; W.cull_valid = true;
      }
      C.cull_lists = W.cull.as<uint32_t>(); C.cull_stride = stride;
    }
  }
  // Hierarchy scenes: the first pass scans, from LDS, the primitives its workgroup's pyramid can reach (k_primary_gathered);
  // the lists are made like the small scenes' (one launch per scene, camera and tile layout).  Option cull_lists = 0, or a
  // scene so large that testing every primitive against every workgroup would cost more than it saves: the walk per ray.
  const uint32_t *big_lists = nullptr; uint32_t big_stride = 0;
  constexpr uint32_t big_cap = 256;
  if constexpr (sizeof(T) == 4) {
    const uint32_t n_wg = div_up(n_paths, 256);
    if (S.bvh && S.sbound && S.sot_cam && !whole_frame && K.cull_lists && (uint64_t)S.n_shapes * n_wg <= (4ull << 30)) {
      rt_scene::CullKey key = { x0, y0, wb, hb, blk, p->sub_sample, (uint32_t)n_paths, C.tiles_x, C.band_shift, C.band_period, C.band_first };
      const uint32_t stride = 16u + big_cap;
      const size_t need = (size_t)n_wg * stride * 4;
      if (!W.cull_valid || W.cull.bytes < need || memcmp(&W.cull_key, &key, sizeof key) != 0) {
        W.cull_valid = false;
        RT_HIP(ctx, W.cull.reserve(need));
        RT_LAUNCH(ctx, "cull_lists", k_cull_lists_big, div_up(n_wg, 4), 256, S, C, n_wg, 256u, W.cull.as<uint32_t>(), stride, big_cap);
        W.cull_key = key; W.cull_valid = true;
      }
      big_lists = W.cull.as<uint32_t>(); big_stride = stride;
    }
  }
  if (whole_frame) {
    const uint32_t tblocks = div_up(n_paths, RT_TAIL_THREADS);
    if (lds_tail) RT_LAUNCH_SMEM(ctx, "trace", (k_tail<T, true, WHITTED, FULL, RT_TAIL_PRIMARY>), tblocks, RT_TAIL_THREADS, smem, S, P, C, md, mc, levels, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, 0u, ctr, 0, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, 0);
    else RT_LAUNCH_SMEM(ctx, "trace", (k_tail<T, false, WHITTED, FULL, RT_TAIL_PRIMARY>), tblocks, RT_TAIL_THREADS, 0, S, P, C, md, mc, levels, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0u, 0u, ctr, 0, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, 0);
  } else
  for (;;) {
    const uint32_t blocks = div_up(ub, 256);
    uint32_t *cnt = counts + wave_bounces;
    if ((wave_bounces > 0 && ub <= tail_threshold) || wave_bounces >= RT_MAX_WAVE_BOUNCES) {
      // (the grid is an upper bound on the chunks: every sub-queue may end with a partly filled one)
      const uint32_t n_subq = direct ? RT_SUBQ : 1u, cap = direct ? C.subq_cap : (uint32_t)n_paths;
      if (direct) {
        for (uint32_t j = 1; j <= bounce_passes; j++) {
          const uint32_t bblocks = div_up(ub, 256) + RT_SUBQ;
          if (lds) RT_LAUNCH_SMEM(ctx, "bounce", (k_bounce<T, true, WHITTED, FULL>), bblocks, 256, smem, S, P, C, md, mc, levels, (const uint32_t *)q_cur, (const uint32_t *)ctr->subq_count[j - 1], q_nxt, ctr->subq_count[j], ctr, (int)j);
          else RT_LAUNCH_SMEM(ctx, "bounce", (k_bounce<T, false, WHITTED, FULL>), bblocks, 256, 0, S, P, C, md, mc, levels, (const uint32_t *)q_cur, (const uint32_t *)ctr->subq_count[j - 1], q_nxt, ctr->subq_count[j], ctr, (int)j);
          std::swap(q_cur, q_nxt);
          wave_bounces++;
        }
      }
      const uint32_t *lens = direct ? ctr->subq_count[bounce_passes] : cnt;
      uint32_t tblocks = div_up(ub, RT_TAIL_THREADS) + (direct ? RT_SUBQ : 0);
      // Workgroups that are down to a thinly filled wave hand their paths on to a further, much smaller launch (see
      // k_tail): `rounds` such launches follow the first, the last of which keeps its paths to their end.
      const uint32_t handoff = (direct && K.tail_handoff > 0) ? std::min<uint32_t>((uint32_t)K.tail_handoff, 64u) : 0u;
      const uint32_t rounds = handoff ? std::min<uint32_t>((uint32_t)std::max(K.tail_rounds, 1), RT_MAX_TAIL_ROUNDS) : 0u;
      // Which tail kernel?  The compacting one finishes a frame sooner when the frame has the chip to itself (0.18 against
      // 0.23 ms on the 1080p bench frame); with persistent lanes a frame costs fewer instructions but its longest paths
      // start later, which pays once three or more frames are in flight to fill the gaps (0.118 -> 0.101 ms per frame).
      const int persistent = K.tail_persistent >= 0 ? K.tail_persistent : (ctx->frames_in_flight >= 3 ? 4 : 0);
      if (direct && persistent > 0) {
        const uint32_t per_wg = 256u * (uint32_t)std::min(persistent, 64);
        const uint32_t pblocks = div_up(ub, per_wg) + RT_SUBQ;
        // Waves that are down to a few paths hand them on to a second launch (see the kernel).  A sub-queue of that launch
        // takes the paths of every RT_SUBQ-th wave, far fewer than its capacity (>= n_paths / RT_SUBQ) on any frame of size;
        // a wave that finds no room keeps its paths.
        const uint32_t ho = (uint32_t)std::min(std::max(K.tail_persistent_handoff, 0), 32);
        uint32_t *c_out = ctr->subq_count[RT_MAX_BOUNCE_PASSES + 1];
        if (lds) RT_LAUNCH_SMEM(ctx, "tail", (k_tail_persistent<T, true, WHITTED, FULL>), pblocks, 256, smem, S, P, C, md, mc, levels, (const uint32_t *)q_cur, lens, n_subq, cap, ctr, (int)wave_bounces, per_wg, ho, q_nxt, c_out, C.subq_cap);
        else RT_LAUNCH_SMEM(ctx, "tail", (k_tail_persistent<T, false, WHITTED, FULL>), pblocks, 256, 0, S, P, C, md, mc, levels, (const uint32_t *)q_cur, lens, n_subq, cap, ctr, (int)wave_bounces, per_wg, ho, q_nxt, c_out, C.subq_cap);
        if (ho) {
          const uint32_t rblocks = div_up(pblocks * 4u * ho, 256u) + RT_SUBQ;
          if (lds) RT_LAUNCH_SMEM(ctx, "tail_rest", (k_tail_persistent<T, true, WHITTED, FULL>), rblocks, 256, smem, S, P, C, md, mc, levels, (const uint32_t *)q_nxt, (const uint32_t *)c_out, (uint32_t)RT_SUBQ, C.subq_cap, ctr, (int)wave_bounces, 256u, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u);
          else RT_LAUNCH_SMEM(ctx, "tail_rest", (k_tail_persistent<T, false, WHITTED, FULL>), rblocks, 256, 0, S, P, C, md, mc, levels, (const uint32_t *)q_nxt, (const uint32_t *)c_out, (uint32_t)RT_SUBQ, C.subq_cap, ctr, (int)wave_bounces, 256u, 0u, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u);
        }
        break;
      }
      for (uint32_t r = 0; r <= rounds; r++) {
        const uint32_t ho = r < rounds ? handoff : 0u;
        uint32_t *c_out = ho ? ctr->subq_count[RT_MAX_BOUNCE_PASSES + 1 + r] : nullptr;
        uint32_t *q_out = ho ? q_nxt : nullptr;
        const uint32_t ns = r == 0 ? n_subq : RT_SUBQ, cp = r == 0 ? cap : C.subq_cap;
        const char *name = r == 0 ? "tail" : "tail_rest";
        if (lds_tail) RT_LAUNCH_SMEM(ctx, name, (k_tail<T, true, WHITTED, FULL, RT_TAIL_QUEUE>), tblocks, RT_TAIL_THREADS, smem, S, P, C, md, mc, levels, (const uint32_t *)q_cur, lens, ns, cp, ctr, (int)wave_bounces, ho, q_out, c_out, r > 0 ? 1 : 0);
        else RT_LAUNCH_SMEM(ctx, name, (k_tail<T, false, WHITTED, FULL, RT_TAIL_QUEUE>), tblocks, RT_TAIL_THREADS, 0, S, P, C, md, mc, levels, (const uint32_t *)q_cur, lens, ns, cp, ctr, (int)wave_bounces, ho, q_out, c_out, r > 0 ? 1 : 0);
        if (!ho) break;
        lens = c_out; std::swap(q_cur, q_nxt);
        tblocks = div_up(tblocks * handoff, RT_TAIL_THREADS) + RT_SUBQ;     // (every workgroup hands on at most `handoff` paths)
      }
      break;
    }
    if (wave_bounces == 0) {
      bool tiles_done = false;
      if constexpr (sizeof(T) == 4) {
        // persistent workgroups where everything they rely on is there: the scene in LDS, ready-made lists, sub-queues
        if (lds && C.cull_lists && C.subq_cap && (K.primary_tiles >= 0 ? K.primary_tiles != 0 : ctx->frames_in_flight >= 3) && S.n_shapes <= RT_LDS_HIT_MAX) {
          const uint32_t resident = (uint32_t)ctx->n_cu * 8u;
          const uint32_t per_block = std::min<uint32_t>(div_up(blocks, resident), RT_PT_MAX_TILES);
          RT_LAUNCH_SMEM(ctx, "primary", (k_primary_tiles<WHITTED, FULL>), div_up(blocks, per_block), 256, smem, S, P, C, md, mc, levels, q_cur, cnt, ctr, blocks, per_block);
          tiles_done = true;
        }
      }
      if constexpr (sizeof(T) == 4) {
        if (!tiles_done && big_lists) {
          RT_LAUNCH_SMEM(ctx, "primary", (k_primary_gathered<WHITTED, FULL>), blocks, 256, gathered_lds_bytes(big_cap), S, P, C, md, mc, levels, q_cur, cnt, wave_mask, block_count, ctr,
                         big_lists, big_stride, big_cap);
          tiles_done = true;
        }
      }
      if (tiles_done) {}
      else if (lds) RT_LAUNCH_SMEM(ctx, "primary", (k_primary<T, true, WHITTED, FULL>), blocks, 256, smem, S, P, C, md, mc, levels, q_cur, cnt, wave_mask, block_count, ctr);
      else RT_LAUNCH_SMEM(ctx, "primary", (k_primary<T, false, WHITTED, FULL>), blocks, 256, 0, S, P, C, md, mc, levels, q_cur, cnt, wave_mask, block_count, ctr);
    } else {
      if (lds) RT_LAUNCH_SMEM(ctx, "intersect_closest", (k_intersect_closest<T, true, WHITTED>), blocks, 256, smem, S, P, C, q_cur, cnt, ctr);
      else RT_LAUNCH_SMEM(ctx, "intersect_closest", (k_intersect_closest<T, false, WHITTED>), blocks, 256, 0, S, P, C, q_cur, cnt, ctr);
      if (lds && SHADE_SCANS && C.use_direct_light)
        RT_LAUNCH_SMEM(ctx, "shade", (k_shade<T, true, WHITTED, FULL>), blocks, 256, smem, S, P, C, md, mc, levels, q_cur, cnt, wave_mask, block_count, ctr, (int)wave_bounces);
      else
        RT_LAUNCH_SMEM(ctx, "shade", (k_shade<T, false, WHITTED, FULL>), blocks, 256, 0, S, P, C, md, mc, levels, q_cur, cnt, wave_mask, block_count, ctr, (int)wave_bounces);
    }
    wave_bounces++;
    if (direct) continue;                                   // the survivors are in the sub-queues of q_cur already
    RT_LAUNCH(ctx, "queue_scan", k_scan_blocks, 1, 1024, block_count, blocks, counts + wave_bounces, cnt);
    RT_LAUNCH(ctx, "queue_scatter", k_queue_scatter, blocks, 256, q_cur, cnt, wave_mask, block_count, q_nxt);
    std::swap(q_cur, q_nxt);
    // the first pass of a large frame is followed blind; after that a length is worth a round trip
    if (ub > tail_threshold && wave_bounces >= 2) {
      RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, counts + wave_bounces, 4, hipMemcpyDeviceToHost, ctx->stream));
      RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ub = ctx->pinned[0];
      if (ub == 0) break;
    }
  }
  if (defer) {
    // the requests sorted by position, so that the 64 a wave serves are neighbours (hits after a bounce arrive in any order)
    if (K.gather_sort) {
      RT_HIP(ctx, W.gsort.reserve((size_t)C.gq_cap * 24));
      unsigned long long *kin = W.gsort.as<unsigned long long>(), *kout = kin + C.gq_cap;
      uint32_t *vin = (uint32_t *)(kout + C.gq_cap), *vout = vin + C.gq_cap;
      const double rmax = std::max(md.n ? p->diffuse_radius : 0.0, mc.n ? p->caustic_radius : 0.0);
      const float inv_cell = rmax > 0 ? (float)(RT_GATHER_CELLS_PER_RADIUS / rmax) : 1.0f;
      RT_LAUNCH(ctx, "gather_keys", k_gather_keys, div_up(C.gq_cap, 256), 256, C, inv_cell, kin, vin);
      { KernelTimer timer_(ctx, "gather_sort"); RT_TRY(rt_sort_pairs_u64(ctx, W.gtemp, kin, kout, vin, vout, C.gq_cap)); }
      C.gq_order = vout;
    }
    // subtrees of at most ~4096 photons become work items of their own (option gather_split = that size; 0: no splitting)
    GatherItems items; items.items = nullptr; items.count = C.gq_count + RT_ITEM_WORDS; items.cap = 0; items.tag = 0; items.stage = nullptr; items.stage_n = 0; items.full = false;
    int split_d = 0, split_c = 0;
    if (K.gather_sort && K.gather_split > 0) {
      auto depth_for = [&](uint32_t n) { int d = 0; while (((uint64_t)n >> d) > (uint64_t)K.gather_split && d < 24) d++; return d; };
      split_d = md.n ? depth_for(md.n) : 0; split_c = mc.n ? depth_for(mc.n) : 0;
      items.cap = (uint32_t)(std::min<size_t>(std::max<size_t>(n_paths, 1u << 16), 1u << 22) / RT_ITEM_SHARDS) * 2u;      // per shard (the shards fill unevenly)
      RT_HIP(ctx, W.gitems.reserve((size_t)items.cap * RT_ITEM_SHARDS * 8));
      items.items = W.gitems.as<uint2>();
    }
    RT_LAUNCH(ctx, "gather_deferred", k_gather_deferred, div_up(C.gq_cap, 256), 256, C, md, mc, items, split_d, split_c);
    if (items.cap) RT_LAUNCH(ctx, "gather_items", k_gather_items, 256 * RT_GATHER_ITEM_WGS, 256, C, md, mc, items, C.gq_count + RT_ITEM_WORDS * (1u + RT_ITEM_SHARDS));
  }
  if (WHITTED)
    RT_LAUNCH(ctx, "resolve", k_resolve_a<T>, div_up(n_pixels, 256), 256, P, C, levels, (uint32_t)n_paths, (int)p->max_depth, (int)n_pixels,
              (int)p->sub_sample, pre, post, rgba8);
  else if (!C.fuse_resolve)
    RT_LAUNCH(ctx, "resolve", k_resolve<T>, div_up(n_pixels, 256), 256, P, C, (int)n_pixels, (int)p->sub_sample, pre, post, rgba8);
  bool readback = stats != nullptr;
#ifdef RT_TAIL_PROBE
  readback = true;
#endif
  // without `stats` nothing has to come back: the frame stays queued on the stream and the caller synchronises when
  // it wants the pixels (frames of a sequence then run back to back, with no host round trip between them)
#ifdef RT_GATHER_PROBE
  if (K.probe && defer) {
    unsigned long long h[8];
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gather_probe), sizeof h));
    fprintf(stderr, "gather probe: walks %llu steps %llu candidates %llu in-ball lane tests %llu needing lanes %llu consumes %llu overflows %llu longest request wave %llu cycles\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    { unsigned long long hh[40]; RT_HIP(ctx, hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_gather_hist), sizeof hh)); for (int b = 0; b < 32; b++) if (hh[b]) fprintf(stderr, "   request waves of 2^%d cycles: %llu\n", b, hh[b]); memset(hh, 0, sizeof hh); RT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_gather_hist), hh, sizeof hh)); }
    memset(h, 0, sizeof h);
    RT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_gather_probe), h, sizeof h));
  }
#endif
  if (!readback) { RT_HIP(ctx, hipGetLastError()); return RT_OK; }
  TraceCounters hc;
  const uint32_t *hcounts = hc.counts;
  RT_HIP(ctx, hipMemcpyAsync(&hc, ctr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipGetLastError());
#ifdef RT_TAIL_PROBE
  if (K.probe) {
    static const char *ph[5] = { "load", "scan", "shade+store", "compact", "top" };
    for (int a = 0; a < 3; a++) {
      fprintf(stderr, "probe %s: block-iterations %llu;", a == 0 ? "iter 1" : (a == 1 ? "iter 2-3" : "iter 4+"), hc.probe[a][5]);
      for (int k = 0; k < 5; k++) fprintf(stderr, " %s %.0f", ph[k], hc.probe[a][5] ? (double)hc.probe[a][k] / (double)hc.probe[a][5] : 0.0);
      fprintf(stderr, "  (cycles per block-iteration, lane 0)\n");
    }
  }
#endif
  if (stats) {
    uint64_t rays = 0, tail = 0; uint32_t iters = 0;
    for (uint32_t b = 0; b < (direct ? 1u : wave_bounces); b++) rays += hcounts[b];
    for (int k = 0; k < RT_CTR_SLOTS; k++) { tail += hc.tail_rays[k]; iters = hc.max_iter[k] > iters ? hc.max_iter[k] : iters; }
    stats->rays = rays + tail; stats->bounces = (direct ? 1u : wave_bounces) + iters;
    for (int k = 0; k < RT_CTR_SLOTS; k++) { hc.shadow_rays += hc.prim_tests[k * 16 + 1]; hc.gathers += hc.prim_tests[k * 16 + 2]; hc.stack_overflows += hc.prim_tests[k * 16 + 3]; }
    stats->shadow_rays = hc.shadow_rays; stats->gathers = hc.gathers;
    stats->stack_overflows = hc.stack_overflows; stats->gather_saturations = hc.gather_saturations;
    stats->tail_rays = whole_frame ? tail - n_paths : tail;       // segments after the first pass that k_tail traced
    uint64_t tests = 0;
    for (int k = 0; k < RT_CTR_SLOTS; k++) tests += hc.prim_tests[k * 16];
    // without a hierarchy every segment is tested against every primitive (shadow rays stop at the first occluder)
    // the fp32 primary pass scans only the primitives each workgroup's rays can reach and counts them
    const bool culled = !whole_frame && lds && sizeof(T) == 4 && S.sbound != nullptr;
    // fp32 scans of a scene with scan groups (<= 64 primitives) count what they test after skipping the groups out of reach
    bool grouped = false;
    if constexpr (sizeof(T) == 4) grouped = S.n_groups > 0 && S.n_shapes <= 64;
    stats->prim_tests = (S.bvh || grouped) ? tests
                                           : (culled ? tests + (stats->rays - n_paths) * (uint64_t)S.n_shapes : stats->rays * (uint64_t)S.n_shapes);
    stats->prims_per_ray_brute_force = S.n_shapes;
  }
  return RT_OK;
}

template <typename T>
int pl_trace_tile(rt_scene *sc, int slot, const rt_params *p, DevMap md, DevMap mc, int x0, int y0, int w, int h,
                  double *pre, double *post, uint8_t *rgba8, rt_trace_stats *stats) {
  if (p->mode == RT_MODE_WHITTED) {
    if (!SceneOf<T>::get(sc).surface)
      return rt_fail(sc->ctx, RT_ERR_INVALID, "RT_MODE_WHITTED needs rt_scene_desc.shape_surface (per-shape reflectance, specular)");
    return trace_tile_mode<T, true, false>(sc, slot, p, md, mc, x0, y0, w, h, pre, post, rgba8, stats);
  }
  const bool full = p->use_direct_light || (p->use_photon_maps && (md.n || mc.n));
  if (full) return trace_tile_mode<T, false, true>(sc, slot, p, md, mc, x0, y0, w, h, pre, post, rgba8, stats);
  return trace_tile_mode<T, false, false>(sc, slot, p, md, mc, x0, y0, w, h, pre, post, rgba8, stats);
}

// ------------------------------------------------------------------ explicit rays, points and photons
// Scene.traceRay(ray, maxDist, depth, importance, materialStack) (rayphoton.js:1626-1838) for n explicit rays, one lane
// each from the ray's first hit to the end of its path.  Ray i draws from stream (seed, RENDER, C.x0 + i, 0) -- the
// stream of pixel (C.x0 + i) % width of row (C.x0 + i) / width in a frame of the same seed -- starting at draw
// first_draw[i].  The call sites of the reference hand it unit directions (Ray's constructor normalises).
template <typename T, bool FULL>
__global__ void __launch_bounds__(64) k_trace_rays(DevScene<T> S, TraceCfg C, DevMap Md, DevMap Mc, uint32_t n, const double *o, const double *d,
                                                   const double *max_dist, const int32_t *depth, const double *importance,
                                                   const unsigned long long *stacks, const uint32_t *first_draw, double *rgba, TraceCounters *ctr) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  ShadeTally tally;
  unsigned long long rays = 0;
  if (p < n) {
    const ScanSrc<T> G = global_scan_src<T>(S);
    PathState<T> st;
    st.o = mk<T>(T(o[3 * p]), T(o[3 * p + 1]), T(o[3 * p + 2])); st.d = mk<T>(T(d[3 * p]), T(d[3 * p + 1]), T(d[3 * p + 2]));
    st.max_dist = max_dist ? T(max_dist[p]) : T(C.max_trace_dist);
    st.importance = importance ? T(importance[p]) : T(1);
    st.acc = mk<T>(0, 0, 0);
    st.draw = first_draw ? first_draw[p] : 0u; st.depth = depth ? depth[p] & 0xFFFF : 0; st.blk_idx = 0xFFFFFFFFu;
    st.key = path_key(S.width, C, p);
    MatStack ms; ms.init();
    st.stack = stacks ? stacks[p] : ms.bits;
    T o4[4] = { T(0), T(0), T(0), T(0) };                    // COL_BLACK: what a ray of importance < 0.01 returns (:1627)
    if (!(st.importance < T(0.01))) {
      for (int bounce = 0; bounce < C.max_bounces; bounce++) {
        T t; rays++;
        const int idx = scan_scene<T, 0>(S, G, st.o, st.d, T(C.epsilon), T(0), t, &tally.tests);
        if (!shade_body<T, FULL>(S, G, C, Md, Mc, p, st, t, idx, bounce, o4, tally)) break;
      }
    }
    for (int c = 0; c < 4; c++) rgba[4 * (size_t)p + c] = (double)o4[c];
  }
  for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
  if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&ctr->tail_rays[blockIdx.x % RT_CTR_SLOTS], rays);
  flush_tally(tally, ctr);
}

template <typename T>
int pl_trace_rays(rt_scene *sc, const rt_params *p, DevMap md, DevMap mc, uint32_t n, const double *o, const double *d, const double *max_dist,
                  const int32_t *depth, const double *importance, const uint64_t *stacks, uint32_t first_stream, const uint32_t *first_draw,
                  double *rgba, rt_trace_stats *stats) {
  rt_ctx *ctx = sc->ctx;
  DevScene<T> &S = SceneOf<T>::get(sc);
  rt_scene::TraceWs &W = sc->tws[0];
  RT_HIP(ctx, W.ctr.reserve(sizeof(TraceCounters)));
  TraceCounters *ctr = W.ctr.as<TraceCounters>();
  RT_HIP(ctx, hipMemsetAsync(ctr, 0, sizeof(TraceCounters), ctx->stream));
  TraceCfg C = {};
  C.epsilon = working_eps<T>(p); C.max_trace_dist = p->max_trace_dist; C.diffuse_radius = p->diffuse_radius;
  C.caustic_radius = p->caustic_radius; C.standard_lamp_area = p->standard_lamp_area;
  C.shadow_bias = p->shadow_bias > 0 ? p->shadow_bias : C.epsilon;
  C.max_depth = p->max_depth; C.sub_sample = 1; C.use_direct_light = p->use_direct_light;
  C.use_photon_maps = p->use_photon_maps; C.max_bounces = p->max_bounces > 0 ? p->max_bounces : 4096;
  C.wave_gather_ratio = (float)ctx->knobs.wave_gather;
  C.seed = p->seed; C.block = 1; C.n_paths = n;
  // one row of n "pixels" starting at column first_stream: path i then keys its stream with first_stream + i
  C.x0 = (int32_t)first_stream; C.y0 = 0; C.w = (int32_t)n; C.h = 1;
  C.magic_w = (uint32_t)(0x100000000ull / (uint64_t)n); C.magic_ss = 0; C.magic_s = 0;
  if (!p->use_photon_maps) { md.n = 0; mc.n = 0; }
  const bool full = p->use_direct_light || (p->use_photon_maps && (md.n || mc.n));
  if (full) RT_LAUNCH(ctx, "trace_rays", (k_trace_rays<T, true>), div_up(n, 64), 64, S, C, md, mc, n, o, d, max_dist, depth, importance,
                      (const unsigned long long *)stacks, first_draw, rgba, ctr);
  else RT_LAUNCH(ctx, "trace_rays", (k_trace_rays<T, false>), div_up(n, 64), 64, S, C, md, mc, n, o, d, max_dist, depth, importance,
                 (const unsigned long long *)stacks, first_draw, rgba, ctr);
  RT_HIP(ctx, hipGetLastError());
  if (stats) {
    TraceCounters hc;
    RT_HIP(ctx, hipMemcpyAsync(&hc, ctr, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *stats = rt_trace_stats{};
    for (int k = 0; k < RT_CTR_SLOTS; k++) stats->rays += hc.tail_rays[k];
    for (int k = 0; k < RT_CTR_SLOTS; k++) { hc.shadow_rays += hc.prim_tests[k * 16 + 1]; hc.gathers += hc.prim_tests[k * 16 + 2]; hc.stack_overflows += hc.prim_tests[k * 16 + 3]; }
    stats->shadow_rays = hc.shadow_rays; stats->gathers = hc.gathers; stats->stack_overflows = hc.stack_overflows;
  }
  return RT_OK;
}

// Scene.directLight(point, normal, localColour) (rayphoton.js:1841-1880) for n points; point i draws from stream
// (seed, RENDER, first_stream + i, 0) starting at first_draw[i].
template <typename T>
__global__ void __launch_bounds__(64) k_direct_light_points(DevScene<T> S, TraceCfg C, uint32_t n, const double *point, const double *normal,
                                                            const double *local, unsigned long long first_stream, const uint32_t *first_draw,
                                                            double *rgb, uint32_t *draws) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ScanSrc<T> G = global_scan_src<T>(S);
  RngStream rng;
  rng.init(C.seed, RT_DOMAIN_RENDER, (uint32_t)(first_stream + i), 0u, first_draw ? first_draw[i] : 0u);
  unsigned long long shadow = 0; uint32_t tests = 0;
  const V3<T> c = direct_light<T>(S, G, C, mk<T>(T(point[3 * i]), T(point[3 * i + 1]), T(point[3 * i + 2])),
                                  mk<T>(T(normal[3 * i]), T(normal[3 * i + 1]), T(normal[3 * i + 2])),
                                  mk<T>(T(local[3 * i]), T(local[3 * i + 1]), T(local[3 * i + 2])), rng, shadow, &tests);
  rgb[3 * (size_t)i] = (double)c.x; rgb[3 * (size_t)i + 1] = (double)c.y; rgb[3 * (size_t)i + 2] = (double)c.z;
  if (draws) draws[i] = rng.draw;
}

template <typename T>
int pl_direct_light(rt_scene *sc, const rt_params *p, uint32_t n, const double *point, const double *normal, const double *local,
                    uint32_t first_stream, const uint32_t *first_draw, double *rgb, uint32_t *draws) {
  rt_ctx *ctx = sc->ctx;
  DevScene<T> &S = SceneOf<T>::get(sc);
  TraceCfg C = {};
  C.epsilon = working_eps<T>(p); C.standard_lamp_area = p->standard_lamp_area; C.seed = p->seed;
  RT_LAUNCH(ctx, "direct_light_points", k_direct_light_points<T>, div_up(n, 64), 64, S, C, n, point, normal, local, (unsigned long long)first_stream,
            first_draw, rgb, draws);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

// createPhoton / emitPhoton for explicit photons (k_photons_explicit, rt_emit.h)
template <typename T>
int pl_photons_explicit(rt_scene *sc, const rt_params *p, uint32_t n, int domain, uint64_t first_stream, const uint32_t *first_draw, int create,
                        const int32_t *source, const int32_t *depth, const int8_t *caustic, const uint64_t *stacks, int track_only_caustic,
                        double *io, uint8_t *kind, uint32_t *draws) {
  rt_ctx *ctx = sc->ctx;
  DevScene<T> &S = SceneOf<T>::get(sc);
  if (!sc->ws_misc.p) { RT_HIP(ctx, sc->ws_misc.alloc(sizeof(EmitCounters))); RT_HIP(ctx, hipMemsetAsync(sc->ws_misc.p, 0, sizeof(EmitCounters), ctx->stream)); }
  EmitCfg C; C.epsilon = working_eps<T>(p); C.max_depth = p->max_depth; C.num_diffuse = p->num_photons_diffuse;
  C.num_caustic = p->num_photons_caustic; C.seed = p->seed;
  RT_LAUNCH(ctx, "photons_explicit", k_photons_explicit<T>, div_up(n, 64), 64, S, C, n, domain, (unsigned long long)first_stream, first_draw, create,
            source, depth, caustic, (const unsigned long long *)stacks, track_only_caustic, io, kind, draws, sc->ws_misc.as<EmitCounters>());
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

// ------------------------------------------------------------------ emission
template <typename T>
int pl_emit_batch(rt_scene *sc, const rt_params *p, int phase, uint64_t a0, uint32_t count) {
  rt_ctx *ctx = sc->ctx;
  DevScene<T> &S = SceneOf<T>::get(sc);
  RT_HIP(ctx, sc->ws_emit_kind.reserve(count));
  RT_HIP(ctx, sc->ws_emit_seg.reserve(count));
  RT_HIP(ctx, sc->ws_emit_rec.reserve((size_t)count * RT_REC_STRIDE * sizeof(T)));
  if (!sc->ws_misc.p) { RT_HIP(ctx, sc->ws_misc.alloc(sizeof(EmitCounters))); RT_HIP(ctx, hipMemsetAsync(sc->ws_misc.p, 0, sizeof(EmitCounters), ctx->stream)); }
  EmitCfg C; C.epsilon = working_eps<T>(p); C.max_depth = p->max_depth; C.num_diffuse = p->num_photons_diffuse;
  C.num_caustic = p->num_photons_caustic; C.seed = p->seed;
  // persistent waves: as many workgroups as stay resident (RT_EMIT_MIN_BLOCKS per CU), fewer for a small batch; every
  // wave takes an equal contiguous share of the batch (a multiple of 64 attempts)
  const uint32_t wgs = std::max<uint32_t>(1u, std::min<uint32_t>(div_up(count, 256), (uint32_t)ctx->n_cu * RT_EMIT_MIN_BLOCKS));
  const uint32_t per_wave = div_up(div_up(count, wgs * 4), 64) * 64;
  if (!S.bvh && scene_fits_lds<T>(S.n_shapes))
    RT_LAUNCH_SMEM(ctx, "photon_emit_bounce", (k_emit<T, true>), wgs, 256, scan_lds_bytes<T>(S.n_shapes), S, C, phase,
                   (unsigned long long)a0, count, per_wave, sc->ws_emit_kind.as<uint8_t>(), sc->ws_emit_rec.as<T>(), sc->ws_emit_seg.as<uint8_t>(), sc->ws_misc.as<EmitCounters>());
  else
    RT_LAUNCH_SMEM(ctx, "photon_emit_bounce", (k_emit<T, false>), wgs, 256, 0, S, C, phase,
                   (unsigned long long)a0, count, per_wave, sc->ws_emit_kind.as<uint8_t>(), sc->ws_emit_rec.as<T>(), sc->ws_emit_seg.as<uint8_t>(), sc->ws_misc.as<EmitCounters>());
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

template <typename T>
int pl_emit_compact(rt_scene *sc, uint32_t count, int want, uint64_t a0, uint32_t out_base, uint32_t out_cap,
                    double *out_rec, int64_t *out_attempt, uint8_t *out_kind, uint32_t *n_match) {
  rt_ctx *ctx = sc->ctx;
  const uint32_t blocks = div_up(count, 256);
  RT_HIP(ctx, sc->ws_emit_blocks.reserve(((size_t)blocks + 1) * 4));
  uint32_t *bc = sc->ws_emit_blocks.as<uint32_t>();
  RT_LAUNCH(ctx, "compact_count", k_compact_count, blocks, 256, sc->ws_emit_kind.as<uint8_t>(), count, want, bc);
  RT_LAUNCH(ctx, "compact_scan", k_scan_blocks, 1, 1024, bc, blocks, bc + blocks, (const uint32_t *)nullptr);
  RT_LAUNCH(ctx, "compact_scatter", k_compact_scatter<T>, blocks, 256, sc->ws_emit_kind.as<uint8_t>(), sc->ws_emit_rec.as<T>(), count,
            want, bc, (unsigned long long)a0, out_base, out_cap, out_rec, (long long *)out_attempt, out_kind);
  RT_HIP(ctx, hipMemcpyAsync(ctx->pinned, bc + blocks, 4, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RT_HIP(ctx, hipGetLastError());
  *n_match = ctx->pinned[0];
  return RT_OK;
}

// The batch in the workspace, compacted in attempt order into a shard (rt_emit.h) of capacity `cap` records; the
// number of stored photons goes to *n_dev (device memory; it may exceed cap, the records beyond cap are dropped).
template <typename T>
int pl_emit_shard(rt_scene *sc, uint32_t count, int format, uint32_t tag_offset, uint32_t cap, uint32_t *shard, uint32_t *n_dev) {
  rt_ctx *ctx = sc->ctx;
  const uint32_t blocks = div_up(count, 256);
  RT_HIP(ctx, sc->ws_emit_blocks.reserve(((size_t)blocks + 1) * 4));
  uint32_t *bc = sc->ws_emit_blocks.as<uint32_t>();
  RT_LAUNCH(ctx, "compact_count", k_compact_count, blocks, 256, sc->ws_emit_kind.as<uint8_t>(), count, 0, bc);
  RT_LAUNCH(ctx, "compact_scan", k_scan_blocks, 1, 1024, bc, blocks, n_dev, (const uint32_t *)nullptr);
  if (format == RT_SHARD_PACKED)
    RT_LAUNCH(ctx, "compact_shard", (k_compact_shard<T, RT_SHARD_PACKED>), blocks, 256, sc->ws_emit_kind.as<uint8_t>(), sc->ws_emit_rec.as<T>(), count,
              (const uint32_t *)bc, tag_offset, cap, shard);
  else
    RT_LAUNCH(ctx, "compact_shard", (k_compact_shard<T, RT_SHARD_FULL>), blocks, 256, sc->ws_emit_kind.as<uint8_t>(), sc->ws_emit_rec.as<T>(), count,
              (const uint32_t *)bc, tag_offset, cap, shard);
  RT_HIP(ctx, hipGetLastError());
  return RT_OK;
}

#define RT_INSTANTIATE_PIPELINE(T)                                                                                     \
  template int pl_intersect_pairs<T>(rt_ctx *, const rt_params *, uint32_t, const double *, const double *, int,       \
                                     const uint8_t *, const double *, double *);                                       \
  template int pl_intersect_scene<T>(rt_scene *, const rt_params *, int, uint32_t, const double *, const double *,     \
                                     const double *, double *, int32_t *);                                             \
  template int pl_trace_tile<T>(rt_scene *, int, const rt_params *, DevMap, DevMap, int, int, int, int, double *, double *, \
                                uint8_t *, rt_trace_stats *);                                                          \
  template int pl_trace_rays<T>(rt_scene *, const rt_params *, DevMap, DevMap, uint32_t, const double *, const double *, const double *, \
                                const int32_t *, const double *, const uint64_t *, uint32_t, const uint32_t *, double *, rt_trace_stats *); \
  template int pl_direct_light<T>(rt_scene *, const rt_params *, uint32_t, const double *, const double *, const double *, uint32_t, \
                                  const uint32_t *, double *, uint32_t *);                                             \
  template int pl_photons_explicit<T>(rt_scene *, const rt_params *, uint32_t, int, uint64_t, const uint32_t *, int, const int32_t *, \
                                      const int32_t *, const int8_t *, const uint64_t *, int, double *, uint8_t *, uint32_t *); \
  template int pl_emit_batch<T>(rt_scene *, const rt_params *, int, uint64_t, uint32_t);                               \
  template int pl_emit_shard<T>(rt_scene *, uint32_t, int, uint32_t, uint32_t, uint32_t *, uint32_t *);                \
  template int pl_emit_compact<T>(rt_scene *, uint32_t, int, uint64_t, uint32_t, uint32_t, double *, int64_t *,        \
                                  uint8_t *, uint32_t *);
